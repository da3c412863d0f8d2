"""LinearEngine — Python owner of one libcae_hip LinearModel engine (include/cae_linear.h)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import CaeError, check

TRAIN, TEST = 0, 1


class LinearEngine:

    def __init__(self, in_shape, out_shape, max_batch, device=None):
        if not torch.cuda.is_available():
            raise CaeError("cae_tools_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = _lib.load()
        self.in_shape, self.out_shape = tuple(int(v) for v in in_shape), tuple(int(v) for v in out_shape)
        self.nin, self.nout, self.max_batch = int(np.prod(self.in_shape)), int(np.prod(self.out_shape)), int(max_batch)
        handle = C.c_void_p()
        check(self.lib.lin_engine_create(self.nin, self.nout, self.max_batch, C.byref(handle)))
        self.handle = handle
        self.n_param = int(self.lib.lin_param_count(handle))
        self.workspace_bytes = int(self.lib.lin_workspace_bytes(handle))
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream()
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.n_param, **f32)
        self.exp_avg = torch.zeros(self.n_param, **f32)
        self.exp_avg_sq = torch.zeros(self.n_param, **f32)
        self.workspace = torch.zeros(self.workspace_bytes + 256, dtype=torch.uint8, device=self.device)
        ws_ptr = (self.workspace.data_ptr() + 255) // 256 * 256
        check(self.lib.lin_bind(handle, self.params.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), ws_ptr,
                                self.workspace_bytes))
        check(self.lib.lin_set_stream(handle, self.stream.cuda_stream))
        torch.cuda.synchronize(self.device)
        self.steps = 0
        self._keep = {}
        self.loss_slots = int(self.lib.lin_loss_slots(handle))

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.lin_engine_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state(self, state):
        w = torch.as_tensor(np.asarray(state["linear.1.weight"]) if not torch.is_tensor(state["linear.1.weight"]) else state["linear.1.weight"])
        b = torch.as_tensor(np.asarray(state["linear.1.bias"]) if not torch.is_tensor(state["linear.1.bias"]) else state["linear.1.bias"])
        if tuple(w.shape) != (self.nout, self.nin) or tuple(b.shape) != (self.nout,):
            raise CaeError(f"weights of shape {tuple(w.shape)} / {tuple(b.shape)} do not fit a {self.nin} -> {self.nout} model")
        self.sync()
        self.params[:self.nout * self.nin].copy_(w.reshape(-1).to(self.device, torch.float32))
        self.params[self.nout * self.nin:].copy_(b.to(self.device, torch.float32))
        torch.cuda.synchronize(self.device)

    def export_state(self):
        self.sync()
        from collections import OrderedDict
        return OrderedDict([("linear.1.weight", self.params[:self.nout * self.nin].view(self.nout, self.nin).cpu().clone()),
                            ("linear.1.bias", self.params[self.nout * self.nin:].cpu().clone())])

    def reset_optimizer(self):
        self.sync()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.steps = 0
        check(self.lib.lin_set_step(self.handle, 0))
        torch.cuda.synchronize(self.device)

    def set_hyper(self, lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999), eps=1e-8):
        check(self.lib.lin_set_hyper(self.handle, float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay)))

    def set_dataset(self, which, x, t=None):
        def prep(a):
            return None if a is None else a.to(device=self.device, dtype=torch.float32).contiguous()
        (x, t) = (prep(x), prep(t))
        if tuple(x.shape[1:]) != self.in_shape or (t is not None and tuple(t.shape[1:]) != self.out_shape):
            raise CaeError(f"data set shapes do not match the model ({self.in_shape} -> {self.out_shape})")
        self._keep[which] = (x, t)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        check(self.lib.lin_set_dataset(self.handle, which, x.data_ptr(), None if t is None else t.data_ptr(), int(x.shape[0])))

    def upload_perm(self, perm):
        idx = torch.as_tensor(np.asarray(perm), dtype=torch.int32).to(self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        return idx

    def train_step(self, which, perm, start, batch, slot=0):
        check(self.lib.lin_train_step(self.handle, which, None if perm is None else perm.data_ptr(), int(start), int(batch), int(slot)))
        self.steps += 1

    def forward_backward(self, which, perm, start, batch, slot=0, global_batch=None, out=None):
        grads = out if out is not None else torch.empty(self.n_param, dtype=torch.float32, device=self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        check(self.lib.lin_forward_backward(self.handle, which, None if perm is None else perm.data_ptr(), int(start), int(batch),
                                            int(slot), grads.data_ptr(),
                                            1.0 if global_batch is None else float(batch) / float(global_batch)))
        if out is None:
            self.sync()
        return grads

    def apply_gradients(self, grads):
        """optimiser step from a flat fp32 gradient (the data-parallel half-step after the all-reduce)"""
        check(self.lib.lin_apply_gradients(self.handle, grads.data_ptr()))
        self.steps += 1

    def eval_step(self, which, perm, start, batch, slot=0):
        check(self.lib.lin_eval_step(self.handle, which, None if perm is None else perm.data_ptr(), int(start), int(batch), int(slot)))

    def run_batches(self, which, perm, n, batch_size, train):
        out = []
        starts = list(range(0, n, batch_size))
        for lo in range(0, len(starts), self.loss_slots):
            chunk = starts[lo:lo + self.loss_slots]
            for (slot, start) in enumerate(chunk):
                (self.train_step if train else self.eval_step)(which, perm, start, min(batch_size, n - start), slot)
            out.extend(self.read_losses(0, len(chunk)))
        return out

    def read_losses(self, first, count):
        buf = (C.c_double * count)()
        check(self.lib.lin_read_losses(self.handle, int(first), int(count), buf))
        return [float(v) for v in buf]

    def score(self, x):
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.empty((x.shape[0],) + self.out_shape, dtype=torch.float32, device=self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        for lo in range(0, x.shape[0], self.max_batch):
            hi = min(x.shape[0], lo + self.max_batch)
            check(self.lib.lin_score(self.handle, x[lo:hi].data_ptr(), hi - lo, out[lo:hi].data_ptr()))
        self.sync()
        return out

    def sync(self):
        check(self.lib.lin_sync(self.handle))
