"""cae_tools_amd — MI355X-native (gfx950) drop-in for the cae_tools ConvAEModel hot path.

The compute path is hand-written HIP behind the C ABI in include/cae_hip.h
(cae_tools_amd/csrc/libcae_hip.so); this package is the Python host that mirrors the
reference's model / CLI surface.  There is no CPU fallback: without the built library and a
GPU every compute entry point raises.
"""
__version__ = "0.1.0"
