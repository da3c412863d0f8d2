"""VaeEngine — Python owner of one libcae_hip 'var' engine (include/cae_vae.h) and of its device memory (torch tensors as
containers, as in engine.py / unet_engine.py)."""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from ._lib import CaeError, TensorInfoC, check
from .engine import _spec_layers, _to_c

TRAIN, TEST = 0, 1


class VaeEngine:

    def __init__(self, spec, fc_size, latent_size, max_batch, device=None):
        if not torch.cuda.is_available():
            raise CaeError("cae_tools_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = _lib.load()
        (enc, dec) = _spec_layers(spec)
        self.fc_size, self.latent_size, self.max_batch = int(fc_size), int(latent_size), int(max_batch)
        handle = C.c_void_p()
        check(self.lib.vae_engine_create(_to_c(enc), len(enc), _to_c(dec), len(dec), self.fc_size, self.latent_size,
                                         self.max_batch, C.byref(handle)))
        self.handle = handle
        self.n_param = int(self.lib.vae_param_count(handle))
        self.n_buffer = int(self.lib.vae_buffer_count(handle))
        self.workspace_bytes = int(self.lib.vae_workspace_bytes(handle))
        self.tensors = OrderedDict()
        info = TensorInfoC()
        for i in range(self.lib.vae_tensor_count(handle)):
            check(self.lib.vae_tensor_info(handle, i, C.byref(info)))
            self.tensors[info.name.decode()] = (int(info.arena), int(info.offset), int(info.numel),
                                                tuple(int(info.shape[d]) for d in range(info.ndim)))
        self.in_shape = tuple(enc[0]["input_dimensions"])
        self.out_shape = tuple(dec[-1]["output_dimensions"])
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream()
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.n_param, **f32)
        self.exp_avg = torch.zeros(self.n_param, **f32)
        self.exp_avg_sq = torch.zeros(self.n_param, **f32)
        self.buffers = torch.zeros(max(self.n_buffer, 4), **f32)
        self.workspace = torch.zeros(self.workspace_bytes + 256, dtype=torch.uint8, device=self.device)
        ws_ptr = (self.workspace.data_ptr() + 255) // 256 * 256
        check(self.lib.vae_bind(handle, self.params.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                self.buffers.data_ptr(), ws_ptr, self.workspace_bytes))
        check(self.lib.vae_set_stream(handle, self.stream.cuda_stream))
        torch.cuda.synchronize(self.device)
        self.num_batches_tracked = 0
        self.steps = 0
        self._keep = {}
        self.loss_slots = int(self.lib.vae_loss_slots(handle))

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.vae_engine_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def view(self, name):
        (arena, off, numel, shape) = self.tensors[name]
        return (self.params if arena == 0 else self.buffers)[off:off + numel].view(shape)

    def load_state(self, enc_state, dec_state):
        self.sync()
        nbt = None
        for prefix, sd in (("enc/", enc_state), ("dec/", dec_state)):
            for k, v in sd.items():
                if k.endswith("num_batches_tracked"):
                    nbt = int(np.asarray(v)) if nbt is None else nbt
                    continue
                name = prefix + k
                if name not in self.tensors:
                    raise CaeError(f"unexpected tensor '{k}' for this model geometry")
                t = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v).to(torch.float32)
                dst = self.view(name)
                if tuple(t.shape) != tuple(dst.shape):
                    raise CaeError(f"shape mismatch for '{k}': {tuple(t.shape)} vs {tuple(dst.shape)}")
                dst.copy_(t.to(self.device))
        missing = [n for n in self.tensors if (n[4:] not in (enc_state if n.startswith("enc/") else dec_state))]
        if missing:
            raise CaeError(f"state dict is missing {missing[:3]}...")
        if nbt is not None:
            self.num_batches_tracked = nbt
        torch.cuda.synchronize(self.device)

    def export_state(self):
        self.sync()
        enc, dec = OrderedDict(), OrderedDict()
        for name in self.tensors:
            side = enc if name.startswith("enc/") else dec
            side[name[4:]] = self.view(name).detach().cpu().clone()
            if name.endswith(".running_var"):
                side[name[4:-len("running_var")] + "num_batches_tracked"] = torch.tensor(self.num_batches_tracked, dtype=torch.int64)
        return enc, dec

    def reset_optimizer(self):
        self.sync()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.set_step(0)
        torch.cuda.synchronize(self.device)

    def set_hyper(self, lr=1e-3, weight_decay=1e-5, lambda_mse=1.0, lambda_kl=1.0, lambda_ssim=1.0, seed=0, betas=(0.9, 0.999),
                  eps=1e-8):
        check(self.lib.vae_set_hyper(self.handle, float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                     float(lambda_mse), float(lambda_kl), float(lambda_ssim), int(seed) & 0xFFFFFFFF))

    def set_kernel_mode(self, mode):
        """1 (default): row-streaming MS-SSIM kernels; 0: the LDS tile kernels (same results to fp32 rounding; A/B and tests)"""
        check(self.lib.vae_set_kernel_mode(self.handle, int(mode)))

    def set_step(self, step):
        self.steps = int(step)
        check(self.lib.vae_set_step(self.handle, self.steps))

    def set_dataset(self, which, x, t=None):
        def prep(a):
            return None if a is None else a.to(device=self.device, dtype=torch.float32).contiguous()
        (x, t) = (prep(x), prep(t))
        if tuple(x.shape[1:]) != self.in_shape or (t is not None and tuple(t.shape[1:]) != self.out_shape):
            raise CaeError(f"data set shapes do not match the model ({self.in_shape} -> {self.out_shape})")
        self._keep[which] = (x, t)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        check(self.lib.vae_set_dataset(self.handle, which, x.data_ptr(), None if t is None else t.data_ptr(), int(x.shape[0])))

    def upload_perm(self, perm):
        idx = torch.as_tensor(np.asarray(perm), dtype=torch.int32).to(self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        return idx

    def train_step(self, which, perm, start, batch, slot=0):
        check(self.lib.vae_train_step(self.handle, which, None if perm is None else perm.data_ptr(), int(start), int(batch), int(slot)))
        self.steps += 1
        self.num_batches_tracked += 1

    def forward_backward(self, which, perm, start, batch, slot=0, global_batch=None, out=None):
        grads = out if out is not None else torch.empty(self.n_param, dtype=torch.float32, device=self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        check(self.lib.vae_forward_backward(self.handle, which, None if perm is None else perm.data_ptr(), int(start), int(batch),
                                            int(slot), grads.data_ptr(),
                                            1.0 if global_batch is None else float(batch) / float(global_batch)))
        self.num_batches_tracked += 1
        if out is None:
            self.sync()
        return grads

    def apply_gradients(self, grads):
        """optimiser step from a flat fp32 gradient (the data-parallel half-step after the all-reduce)"""
        check(self.lib.vae_apply_gradients(self.handle, grads.data_ptr()))
        self.steps += 1

    def eval_step(self, which, perm, start, batch, slot=0):
        check(self.lib.vae_eval_step(self.handle, which, None if perm is None else perm.data_ptr(), int(start), int(batch), int(slot)))

    def run_batches(self, which, perm, n, batch_size, train):
        """[(mse, kl, 1 - ms_ssim, total)] per batch of one epoch"""
        out = []
        starts = list(range(0, n, batch_size))
        for lo in range(0, len(starts), self.loss_slots):
            chunk = starts[lo:lo + self.loss_slots]
            for (slot, start) in enumerate(chunk):
                (self.train_step if train else self.eval_step)(which, perm, start, min(batch_size, n - start), slot)
            out.extend(self.read_losses(0, len(chunk)))
        return out

    def read_losses(self, first, count):
        buf = (C.c_double * (4 * count))()
        check(self.lib.vae_read_losses(self.handle, int(first), int(count), buf))
        return [tuple(buf[4 * i + k] for k in range(4)) for i in range(count)]

    def score(self, x):
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.empty((x.shape[0],) + self.out_shape, dtype=torch.float32, device=self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        for lo in range(0, x.shape[0], self.max_batch):
            hi = min(x.shape[0], lo + self.max_batch)
            check(self.lib.vae_score(self.handle, x[lo:hi].data_ptr(), hi - lo, out[lo:hi].data_ptr()))
        self.sync()
        return out

    def sync(self):
        check(self.lib.vae_sync(self.handle))
