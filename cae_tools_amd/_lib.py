"""ctypes binding of libcae_hip.so (C ABI: include/cae_hip.h).  Fails loudly when the library
is missing: there is no other compute path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcae_hip.so")


class CaeError(RuntimeError):
    pass


class LayerSpecC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("in_c", "in_h", "in_w", "out_c", "out_h", "out_w", "k_h", "k_w",
                                          "stride", "output_padding")]


class TensorInfoC(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("arena", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
                ("offset", C.c_int64), ("numel", C.c_int64)]


class ProfileRecC(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("layer", C.c_int32), ("reserved", C.c_int32), ("micros", C.c_double),
                ("bytes", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)

# name -> (restype, argtypes); every symbol include/cae_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    "cae_last_error": (C.c_char_p, []),
    "cae_abi_version": (C.c_int, []),
    "cae_engine_create": (C.c_int, [C.POINTER(LayerSpecC), C.c_int, C.POINTER(LayerSpecC), C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.POINTER(_P)]),
    "cae_engine_destroy": (None, [_P]),
    "cae_param_count": (C.c_int64, [_P]),
    "cae_buffer_count": (C.c_int64, [_P]),
    "cae_tensor_count": (C.c_int, [_P]),
    "cae_tensor_info": (C.c_int, [_P, C.c_int, C.POINTER(TensorInfoC)]),
    "cae_workspace_bytes": (C.c_int64, [_P]),
    "cae_bind": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64]),
    "cae_set_stream": (C.c_int, [_P, _P]),
    "cae_set_graph_mode": (C.c_int, [_P, C.c_int]),
    "cae_set_kernel_mode": (C.c_int, [_P, C.c_int]),
    "cae_set_capture_only": (C.c_int, [_P, C.c_int]),
    "cae_set_hyper": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "cae_set_dataset": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64]),
    "cae_set_cursor": (C.c_int, [_P, C.c_int64, C.c_int]),
    "cae_set_adam_step": (C.c_int, [_P, C.c_int]),
    "cae_train_step": (C.c_int, [_P, C.c_int, _P, C.c_int]),
    "cae_train_steps": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int]),
    "cae_eval_steps": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int]),
    "cae_forward_backward": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int]),
    "cae_adam_step": (C.c_int, [_P]),
    "cae_forward_backward_sync": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, ALLREDUCE_FN, _P]),
    "cae_dp_unique_id": (C.c_int, [_P]),
    "cae_dp_init": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "cae_dp_shutdown": (C.c_int, [_P]),
    "cae_dp_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cae_dp_set_overlap": (C.c_int, [_P, C.c_int]),
    "cae_dp_broadcast_state": (C.c_int, [_P, C.c_int, C.c_int]),
    "cae_dp_train_step": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "cae_dp_train_steps": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "cae_dp_eval_steps": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "cae_dp_read_losses": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "cae_eval_step": (C.c_int, [_P, C.c_int, _P, C.c_int]),
    "cae_score": (C.c_int, [_P, _P, C.c_int, _P]),
    "cae_encode": (C.c_int, [_P, _P, C.c_int, _P]),
    "cae_decode": (C.c_int, [_P, _P, C.c_int, _P]),
    "cae_read_losses": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "cae_loss_slots": (C.c_int, [_P]),
    "cae_sync": (C.c_int, [_P]),
    "cae_graph_count": (C.c_int, [_P]),
    "cae_debug_read": (C.c_int64, [_P, C.c_char_p, C.c_int, _P, C.c_int64]),
    "cae_profile_begin": (C.c_int, [_P]),
    "cae_debug_launch_floor": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    "cae_profile_end": (C.c_int, [_P, C.POINTER(ProfileRecC), C.c_int]),
    "cae_trace_range_push": (C.c_int, [C.c_char_p]),
    "cae_trace_range_pop": (C.c_int, []),
    "cae_scan_f32": (C.c_int, [_P, C.c_int64, _P, C.POINTER(C.c_double)]),
    "cae_normalise_pack": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int64, _P, C.c_int, C.c_int, C.c_float,
                                     C.c_float, C.c_int, _P]),
    "cae_normalise_pack_rows": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int64, _P, C.c_int, C.c_int, C.c_float,
                                          C.c_float, C.c_int, _P, _P]),
    "cae_invert_permutation": (C.c_int, [_P, C.c_int64, _P, _P]),
    "cae_denormalise_f64": (C.c_int, [_P, C.c_int64, C.c_double, C.c_double, _P, _P]),
    "cae_bswap32": (C.c_int, [_P, C.c_int64, _P]),
    "cae_metric_sums": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.c_double, _P, _P]),
    # ---- include/cae_unet.h ----
    "unet_engine_create": (C.c_int, [C.POINTER(LayerSpecC), C.c_int, C.POINTER(LayerSpecC), C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.POINTER(C.c_void_p)]),
    "unet_engine_destroy": (None, [_P]),
    "unet_param_count": (C.c_int64, [_P]),
    "unet_buffer_count": (C.c_int64, [_P]),
    "unet_tensor_count": (C.c_int, [_P]),
    "unet_tensor_info": (C.c_int, [_P, C.c_int, _P]),
    "unet_workspace_bytes": (C.c_int64, [_P]),
    "unet_bind": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64]),
    "unet_set_stream": (C.c_int, [_P, _P]),
    "unet_set_kernel_mode": (C.c_int, [_P, C.c_int]),
    "unet_set_hyper": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_uint32]),
    "unet_set_step": (C.c_int, [_P, C.c_int64]),
    "unet_set_dataset": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, C.c_int64]),
    "unet_train_step": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.c_int]),
    "unet_forward_backward": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.c_int, _P, C.c_double]),
    "unet_apply_gradients": (C.c_int, [_P, _P]),
    "unet_eval_step": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.c_int]),
    "unet_score": (C.c_int, [_P, _P, C.c_int, _P]),
    "unet_loss_slots": (C.c_int, [_P]),
    "unet_read_losses": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "unet_sync": (C.c_int, [_P]),
    "unet_debug_read": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    # ---- include/cae_vae.h ----
    "vae_engine_create": (C.c_int, [C.POINTER(LayerSpecC), C.c_int, C.POINTER(LayerSpecC), C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.POINTER(C.c_void_p)]),
    "vae_engine_destroy": (None, [_P]),
    "vae_param_count": (C.c_int64, [_P]),
    "vae_buffer_count": (C.c_int64, [_P]),
    "vae_tensor_count": (C.c_int, [_P]),
    "vae_tensor_info": (C.c_int, [_P, C.c_int, _P]),
    "vae_workspace_bytes": (C.c_int64, [_P]),
    "vae_bind": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64]),
    "vae_set_stream": (C.c_int, [_P, _P]),
    "vae_set_hyper": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.c_double, C.c_uint32]),
    "vae_set_kernel_mode": (C.c_int, [_P, C.c_int]),
    "vae_set_step": (C.c_int, [_P, C.c_int64]),
    "vae_set_dataset": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64]),
    "vae_train_step": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.c_int]),
    "vae_forward_backward": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.c_int, _P, C.c_double]),
    "vae_apply_gradients": (C.c_int, [_P, _P]),
    "vae_eval_step": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.c_int]),
    "vae_score": (C.c_int, [_P, _P, C.c_int, _P]),
    "vae_loss_slots": (C.c_int, [_P]),
    "vae_read_losses": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "vae_sync": (C.c_int, [_P]),
    # ---- include/cae_linear.h ----
    "lin_engine_create": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    "lin_engine_destroy": (None, [_P]),
    "lin_param_count": (C.c_int64, [_P]),
    "lin_workspace_bytes": (C.c_int64, [_P]),
    "lin_bind": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64]),
    "lin_set_stream": (C.c_int, [_P, _P]),
    "lin_set_hyper": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "lin_set_step": (C.c_int, [_P, C.c_int64]),
    "lin_set_dataset": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64]),
    "lin_train_step": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.c_int]),
    "lin_forward_backward": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.c_int, _P, C.c_double]),
    "lin_apply_gradients": (C.c_int, [_P, _P]),
    "lin_eval_step": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.c_int]),
    "lin_score": (C.c_int, [_P, _P, C.c_int, _P]),
    "lin_loss_slots": (C.c_int, [_P]),
    "lin_read_losses": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "lin_sync": (C.c_int, [_P]),
}

_lib = None


def load():
    """Load the shared library (once) and declare every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so, same soname as /opt/rocm's).  It must
    # be the one already loaded when libcae_hip.so resolves libamdhip64.so.7, otherwise the process
    # holds two runtimes and torch's streams / device pointers mean nothing to ours.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise CaeError(f"{LIB_PATH} is missing: build it with `python -m cae_tools_amd.build` "
                       "(hipcc --offload-arch=gfx950). cae_tools_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch, surfaced loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc is not None and rc < 0:
        msg = load().cae_last_error()
        raise CaeError(f"libcae_hip error {rc}: {msg.decode() if msg else '?'}")
    return rc
