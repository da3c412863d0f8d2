"""Encoder: parameter container + HIP-backed forward.

Mirrors the constructor / forward / state_dict surface of the reference Encoder
(src/cae_tools/models/encoder.py:36-64): per layer Conv2d(k, stride) -> BatchNorm2d -> ReLU,
flatten, Linear(C*y*x, fc) -> ReLU -> Linear(fc, latent); PyTorch default initialisation.
The arithmetic runs in libcae_hip (see ConvAEModel / HipEngine).  Training and scoring run the
encoder and decoder fused; calling this module on its own (encoder.py:60-64) runs the engine it is
attached to through cae_encode: EVAL mode (running BatchNorm statistics), there is no train-mode
module-level forward.  A module that is not attached to an engine raises.
"""
import torch

from ._params import ParamBag, add_batchnorm, add_conv


class Encoder(ParamBag):

    def __init__(self, layers, encoded_space_dim, fc_size):
        super().__init__()
        self.layers = list(layers)
        self.encoded_space_dim = encoded_space_dim
        self.fc_size = fc_size
        for i, layer in enumerate(self.layers):
            (cin, _, _) = layer.get_input_dimensions()
            (cout, _, _) = layer.get_output_dimensions()
            (kh, kw) = layer.kernel_hw()
            add_conv(self, f"encoder_cnn.{3 * i}", (cout, cin, kh, kw), cout)
            add_batchnorm(self, f"encoder_cnn.{3 * i + 1}", cout)
        (chan, y, x) = self.layers[-1].get_output_dimensions()
        add_conv(self, "encoder_lin.0", (fc_size, chan * y * x), fc_size)
        add_conv(self, "encoder_lin.2", (encoded_space_dim, fc_size), encoded_space_dim)
        self._engine = None

    def attach(self, engine):
        self._engine = engine

    def forward(self, x):
        """x (B, C, h, w) fp32 CUDA tensor -> z (B, encoded_space_dim); eval mode, on the attached engine
        (whose parameter arena holds the live weights: HipEngine.load_state / ConvAEModel put them there)"""
        if self._engine is None:
            raise RuntimeError("Encoder.forward needs an attached HipEngine (ConvAEModel attaches one when it builds or "
                               "loads a model): there is no CPU path")
        return self._engine.encode(x)

    __call__ = forward
