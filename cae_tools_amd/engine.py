"""HipEngine — Python owner of one libcae_hip engine and of the device memory it works on.

torch is used here as a container only: flat CUDA tensors for the parameter / gradient / Adam /
running-stat arenas and the workspace, a side stream, and (in dp.py) torch.distributed.  Every
FLOP of the model runs in the HIP kernels behind include/cae_hip.h.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from ._lib import CaeError, LayerSpecC, TensorInfoC, check

TRAIN, TEST = 0, 1


def _spec_layers(spec):
    """accept a ModelSpec-like object or its JSON dict; return two lists of plain dicts"""
    if hasattr(spec, "save"):
        spec = spec.save()
    return spec["input_layers"], spec["output_layers"]


def _to_c(layers):
    arr = (LayerSpecC * len(layers))()
    for i, l in enumerate(layers):
        k = l["kernel_size"]
        (kh, kw) = (int(k[0]), int(k[1])) if isinstance(k, (list, tuple)) else (int(k), int(k))
        (ic, ih, iw) = l["input_dimensions"]
        (oc, oh, ow) = l["output_dimensions"]
        arr[i] = LayerSpecC(ic, ih, iw, oc, oh, ow, kh, kw, int(l["stride"]), int(l.get("output_padding", 0)))
    return arr


class EnginePlan:
    """Geometry-only view of an engine (no GPU needed): tensor table, arena and workspace sizes."""

    def __init__(self, spec, fc_size, latent_size, max_batch):
        self.lib = _lib.load()
        (enc, dec) = _spec_layers(spec)
        self.enc_layers, self.dec_layers = enc, dec
        self.fc_size, self.latent_size, self.max_batch = int(fc_size), int(latent_size), int(max_batch)
        handle = C.c_void_p()
        check(self.lib.cae_engine_create(_to_c(enc), len(enc), _to_c(dec), len(dec), self.fc_size,
                                         self.latent_size, self.max_batch, C.byref(handle)))
        self.handle = handle
        self.n_param = int(self.lib.cae_param_count(handle))
        self.n_buffer = int(self.lib.cae_buffer_count(handle))
        self.workspace_bytes = int(self.lib.cae_workspace_bytes(handle))
        self.tensors = OrderedDict()
        info = TensorInfoC()
        for i in range(self.lib.cae_tensor_count(handle)):
            check(self.lib.cae_tensor_info(handle, i, C.byref(info)))
            shape = tuple(int(info.shape[d]) for d in range(info.ndim))
            self.tensors[info.name.decode()] = (int(info.arena), int(info.offset), int(info.numel), shape)
        self.in_shape = tuple(enc[0]["input_dimensions"])
        self.out_shape = tuple(dec[-1]["output_dimensions"])

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.cae_engine_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bn_prefixes(self):
        """state_dict prefixes ("enc/encoder_cnn.1") of the BatchNorm layers, in order"""
        return [n[: -len(".running_mean")] for n in self.tensors if n.endswith(".running_mean")]


class HipEngine(EnginePlan):

    def __init__(self, spec, fc_size, latent_size, max_batch, device=None, graph=True, specialised=True):
        if not torch.cuda.is_available():
            raise CaeError("cae_tools_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
        super().__init__(spec, fc_size, latent_size, max_batch)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream()
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.n_param, **f32)
        self.grads = torch.zeros(self.n_param, **f32)
        self.exp_avg = torch.zeros(self.n_param, **f32)
        self.exp_avg_sq = torch.zeros(self.n_param, **f32)
        self.buffers = torch.zeros(max(self.n_buffer, 4), **f32)
        self.workspace = torch.zeros(self.workspace_bytes + 256, dtype=torch.uint8, device=self.device)
        ws_ptr = (self.workspace.data_ptr() + 255) // 256 * 256
        check(self.lib.cae_bind(self.handle, self.params.data_ptr(), self.grads.data_ptr(),
                                self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.buffers.data_ptr(),
                                ws_ptr, self.workspace_bytes))
        check(self.lib.cae_set_stream(self.handle, self.stream.cuda_stream))
        check(self.lib.cae_set_graph_mode(self.handle, 1 if graph else 0))
        check(self.lib.cae_set_kernel_mode(self.handle, int(specialised) if isinstance(specialised, int) and not isinstance(specialised, bool) else (1 if specialised else 0)))
        torch.cuda.synchronize(self.device)
        self.num_batches_tracked = 0
        self.adam_steps = 0
        self._datasets = {}
        self._keep = []
        self.loss_slots = int(self.lib.cae_loss_slots(self.handle))
        self._slot = 0
        self._cursor = None     # host shadow of the device cursor (batch_start, loss_slot); None = unknown

    # ---- parameters ------------------------------------------------------------------------
    def _arena(self, arena):
        return self.params if arena == 0 else self.buffers

    def view(self, name):
        """torch view (device) of a named tensor, e.g. 'dec/decoder_conv.0.weight'"""
        (arena, off, numel, shape) = self.tensors[name]
        return self._arena(arena)[off:off + numel].view(shape)

    def grad_view(self, name):
        (arena, off, numel, shape) = self.tensors[name]
        assert arena == 0
        return self.grads[off:off + numel].view(shape)

    def load_state(self, enc_state, dec_state):
        """copy reference-format state dicts (encoder.weights / decoder.weights) into the arenas"""
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
        """(encoder_state, decoder_state) as CPU tensors under the reference's state_dict keys and
        order, including num_batches_tracked"""
        self.sync()
        enc, dec = OrderedDict(), OrderedDict()
        for name in self.tensors:
            side = enc if name.startswith("enc/") else dec
            side[name[4:]] = self.view(name).detach().cpu().clone()
            if name.endswith(".running_var"):
                side[name[4:-len("running_var")] + "num_batches_tracked"] = torch.tensor(
                    self.num_batches_tracked, dtype=torch.int64)
        # state_dict order of nn.BatchNorm2d: weight, bias, running_mean, running_var, num_batches_tracked
        return self._ordered(enc), self._ordered(dec)

    @staticmethod
    def _ordered(sd):
        order = {"weight": 0, "bias": 1, "running_mean": 2, "running_var": 3, "num_batches_tracked": 4}
        groups = OrderedDict()
        for k in sd:
            groups.setdefault(k.rsplit(".", 1)[0], []).append(k)
        out = OrderedDict()
        for g, keys in groups.items():
            for k in sorted(keys, key=lambda s: order[s.rsplit(".", 1)[1]]):
                out[k] = sd[k]
        return out

    def reset_optimizer(self):
        """a fresh torch.optim.Adam (conv_ae_model.py:310 re-creates it on every train())"""
        self.sync()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.adam_steps = 0
        torch.cuda.synchronize(self.device)
        check(self.lib.cae_set_adam_step(self.handle, 0))

    def load_optimizer_state(self, moments, step):
        """moments: {tensor name: (exp_avg, exp_avg_sq)} for parameter tensors; step: completed Adam steps"""
        self.sync()
        for name, (m, v) in moments.items():
            (arena, off, numel, shape) = self.tensors[name]
            self.exp_avg[off:off + numel].copy_(torch.as_tensor(m, dtype=torch.float32).reshape(-1).to(self.device))
            self.exp_avg_sq[off:off + numel].copy_(torch.as_tensor(v, dtype=torch.float32).reshape(-1).to(self.device))
        torch.cuda.synchronize(self.device)
        self.adam_steps = int(step)
        check(self.lib.cae_set_adam_step(self.handle, int(step)))

    def set_hyper(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5):
        check(self.lib.cae_set_hyper(self.handle, float(lr), float(betas[0]), float(betas[1]), float(eps),
                                     float(weight_decay)))

    # ---- data ------------------------------------------------------------------------------
    def _same_device(self, a, what):
        """the kernels dereference plain pointers: a tensor on another GPU is a memory fault, not a slow path"""
        if a.device != self.device:
            raise CaeError(f"{what} lives on {a.device} but this engine runs on {self.device} "
                           "(select the rank's GPU before creating data sets: cae_tools_amd.dp.select_device)")

    def set_dataset(self, which, x, t=None):
        """x (N,C,H,W) / t (N,C,H,W): fp32 CUDA tensors, already normalised; kept alive here"""
        for name, a, shp in (("x", x, self.in_shape), ("t", t, self.out_shape)):
            if a is None:
                continue
            if a.dtype != torch.float32 or not a.is_cuda or not a.is_contiguous():
                raise CaeError(f"dataset {name} must be a contiguous fp32 CUDA tensor")
            self._same_device(a, f"dataset {name}")
            if tuple(a.shape[1:]) != tuple(shp):
                raise CaeError(f"dataset {name} has sample shape {tuple(a.shape[1:])}, model expects {tuple(shp)}")
        self.sync()
        torch.cuda.synchronize(self.device)
        self._datasets[which] = (x, t)
        check(self.lib.cae_set_dataset(self.handle, which, x.data_ptr(), t.data_ptr() if t is not None else None,
                                       int(x.shape[0])))

    def upload_perm(self, perm):
        """int32 device copy of a sample permutation / index list"""
        p = torch.as_tensor(np.asarray(perm, dtype=np.int32)).to(self.device)
        torch.cuda.synchronize(self.device)
        self._keep.append(p)
        if len(self._keep) > 64:
            self.sync()
            self._keep = self._keep[-8:]
        return p

    # ---- steps -----------------------------------------------------------------------------
    def _claim_slots(self, n):
        if n > self.loss_slots:
            raise CaeError(f"{n} steps between loss reads exceeds {self.loss_slots} slots")
        if self._slot + n > self.loss_slots:
            self._slot = 0
        first = self._slot
        self._slot += n
        return first

    def _set_cursor(self, start, slot):
        """point the device cursor at (start, slot) unless the previous step already left it there (every step
        advances it by its batch size and one slot): one tiny launch less per step for sequential passes"""
        if self._cursor != (int(start), int(slot)):
            check(self.lib.cae_set_cursor(self.handle, int(start), int(slot)))
        self._cursor = (int(start), int(slot))

    def _advance(self, samples, steps):
        if self._cursor is not None:
            self._cursor = (self._cursor[0] + int(samples), self._cursor[1] + int(steps))

    def _read_losses(self, first, n):
        out = (C.c_double * n)()
        check(self.lib.cae_read_losses(self.handle, first, n, out))
        return [float(v) for v in out]

    def run_batches(self, which, perm_dev, n, batch_size, train=True):
        """One pass over perm[0:n] in batches of batch_size (last one partial, drop_last=False as in
        conv_ae_model.py:291-292).  Returns the per-batch mean losses
        (__train_epoch :185-203 / __test_epoch :205-221)."""
        nb = (n + batch_size - 1) // batch_size
        first = self._claim_slots(nb)
        ptr = perm_dev.data_ptr() if perm_dev is not None else None
        self._set_cursor(0, first)
        self._enqueue(which, ptr, n, batch_size, train)
        self._advance(n, nb)
        if train:
            self.num_batches_tracked += nb
            self.adam_steps += nb
        return self._read_losses(first, nb)

    STEPS_PER_GRAPH = 64

    def _enqueue(self, which, ptr, n, batch_size, train):
        """one pass over n samples: the full batches in multi-step graphs, then the partial last batch"""
        many = self.lib.cae_train_steps if train else self.lib.cae_eval_steps
        one = self.lib.cae_train_step if train else self.lib.cae_eval_step
        full = n // batch_size
        # at most three graph shapes per (data set, batch size): K steps, the remainder of the full batches, the partial
        # batch - all met in the first epoch, so nothing is captured mid-run (capture_graphs() builds them ahead of it)
        while full >= self.STEPS_PER_GRAPH:
            check(many(self.handle, which, ptr, batch_size, self.STEPS_PER_GRAPH))
            full -= self.STEPS_PER_GRAPH
        if full > 1:
            check(many(self.handle, which, ptr, batch_size, full))
        elif full == 1:
            check(one(self.handle, which, ptr, batch_size))
        if n % batch_size:
            check(one(self.handle, which, ptr, n % batch_size))

    def capture_graphs(self):
        """Context manager: step calls made inside capture and cache their hipGraphs and launch nothing
        (cae_set_capture_only); the host-side counters are put back on exit.  Used to build every graph shape of an epoch
        loop before its first (timed) step."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            keep = (self.num_batches_tracked, self.adam_steps, self._slot, self._cursor)
            check(self.lib.cae_set_capture_only(self.handle, 1))
            try:
                yield self
            finally:
                check(self.lib.cae_set_capture_only(self.handle, 0))
                (self.num_batches_tracked, self.adam_steps, self._slot, _) = keep
                self._cursor = None     # cae_set_cursor is not a step function: the device cursor may have moved
        return scope()

    def trace_range(self, name):
        """Context manager: a named roctx range around the host calls made inside (rocprofv3 --marker-trace); a no-op without
        a roctx library.  ConvAEModel.train marks its passes with it."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            pushed = self.lib.cae_trace_range_push(name.encode())
            try:
                yield
            finally:
                if pushed:
                    self.lib.cae_trace_range_pop()
        return scope()

    def graph_count(self):
        return int(self.lib.cae_graph_count(self.handle))

    def train_step(self, which, perm_dev, start, size):
        """a single training step on perm[start:start+size]; returns its loss (blocking)"""
        first = self._claim_slots(1)
        self._set_cursor(start, first)
        check(self.lib.cae_train_step(self.handle, which, perm_dev.data_ptr() if perm_dev is not None else None,
                                      int(size)))
        self._advance(size, 1)
        self.num_batches_tracked += 1
        self.adam_steps += 1
        return self._read_losses(first, 1)[0]

    def enqueue_train_steps(self, which, perm_dev, n, batch_size, slot_first):
        """bench helper: enqueue one pass without reading anything back"""
        nb = (n + batch_size - 1) // batch_size
        ptr = perm_dev.data_ptr() if perm_dev is not None else None
        self._set_cursor(0, slot_first)
        self._enqueue(which, ptr, n, batch_size, True)
        self._advance(n, nb)
        self.num_batches_tracked += nb
        self.adam_steps += nb
        return nb

    def forward_backward(self, which, perm_dev, start, size, global_batch):
        """DP half-step: gradients of sum/global_count into self.grads (caller all-reduces)"""
        first = self._claim_slots(1)
        self._set_cursor(start, first)
        check(self.lib.cae_forward_backward(self.handle, which,
                                            perm_dev.data_ptr() if perm_dev is not None else None, int(size),
                                            int(global_batch)))
        self._advance(size, 1)
        self.num_batches_tracked += 1
        return first

    def forward_backward_sync(self, which, perm_dev, start, size, global_batch, world, allreduce):
        """forward_backward with BatchNorm over the global batch: `allreduce(t)` is called with a float64
        CUDA view of each BatchNorm sum table and must sum it over the ranks in place, enqueued on
        self.stream (it is called inside `with torch.cuda.stream(self.stream)`)."""
        first = self._claim_slots(1)
        self._set_cursor(start, first)
        self._advance(size, 1)
        base = (self.workspace.data_ptr() + 255) // 256 * 256
        pad = base - self.workspace.data_ptr()
        failure = []

        def _cb(user, table_ptr, count):
            try:
                off = pad + (table_ptr - base)
                view = self.workspace[off:off + 8 * count].view(torch.float64)
                with torch.cuda.stream(self.stream):
                    allreduce(view)
                return 0
            except Exception as ex:  # surfaces after the C call returns
                failure.append(ex)
                return 1

        cb = _lib.ALLREDUCE_FN(_cb)
        rc = self.lib.cae_forward_backward_sync(self.handle, which, perm_dev.data_ptr() if perm_dev is not None else None,
                                                int(size), int(global_batch), int(world), cb, None)
        if failure:
            raise failure[0]
        check(rc)
        self.num_batches_tracked += 1
        return first

    def adam_step(self):
        check(self.lib.cae_adam_step(self.handle))
        self.adam_steps += 1

    # ---- data parallelism inside the library (include/cae_hip.h, cae_dp_*) ------------------------
    dp_world = 0
    dp_rank = 0

    def dp_init(self, dist, group=None):
        """Join the library's RCCL communicator: rank 0 draws the 128-byte rendezvous id, torch.distributed (any
        backend) carries it to the other ranks, cae_dp_init is the collective.  Returns True when data-parallel steps
        replay from one hipGraph with the collectives captured inside."""
        (world, rank) = (dist.get_world_size(group), dist.get_rank(group))
        ident = C.create_string_buffer(128)
        if rank == 0:
            check(self.lib.cae_dp_unique_id(ident))
        box = [ident.raw]
        with torch.cuda.device(self.device):
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            check(self.lib.cae_dp_init(self.handle, world, rank, box[0]))
        (self.dp_world, self.dp_rank) = (world, rank)
        self._cursor = None
        return self.dp_graph_capture()

    def dp_graph_capture(self):
        flag = C.c_int(0)
        check(self.lib.cae_dp_info(self.handle, None, None, C.byref(flag)))
        return bool(flag.value)

    def dp_shutdown(self):
        check(self.lib.cae_dp_shutdown(self.handle))
        self.dp_world = 0

    def dp_broadcast(self, root=0, params=True, buffers=True, moments=True):
        """rank `root`'s weights / BatchNorm running statistics / Adam moments to every rank (blocking)"""
        what = (1 if params else 0) | (2 if buffers else 0) | (4 if moments else 0)
        check(self.lib.cae_dp_broadcast_state(self.handle, int(root), what))

    def dp_set_overlap(self, enabled):
        check(self.lib.cae_dp_set_overlap(self.handle, 1 if enabled else 0))
        self.dp_overlap = bool(enabled)

    dp_overlap = True

    def dp_calibrate(self, dist, which, perm_dev, start, batch, global_batch, sync_bn, group=None, steps=48, warm=8):
        """Choose between the two launch structures of a data-parallel step - first gradient bucket all-reduced on the
        second stream beside the tail of backward, or both buckets on the main stream after it - by timing `steps` real
        steps of each on the live communicator, then putting every piece of training state back.  The arithmetic is the
        same either way; a fork/join between two hardware queues costs ~20 us in a replayed graph on this stack, which the
        overlap only earns back when the first bucket's all-reduce is slower than that.  Collective: every rank calls it
        with the same arguments; the slowest rank's times decide, so all ranks agree.  Returns {mode: seconds per step}."""
        import time
        if sync_bn:
            return {}   # SyncBN keeps every collective on the main stream
        self.sync()
        keep = [t.clone() for t in (self.params, self.buffers, self.exp_avg, self.exp_avg_sq, self.grads)]
        counters = (self.num_batches_tracked, self.adam_steps, self._slot)
        slot = self._claim_slots(1)
        times = {}
        for mode in (True, False):
            self.dp_set_overlap(mode)
            for (n, timed) in ((warm, False), (steps, True)):
                self.sync()
                t0 = time.perf_counter()
                for _ in range(n):
                    self._set_cursor(start, slot)       # the same batch and loss slot every time
                    self.dp_train_steps(which, perm_dev, batch, global_batch, False, 1)
                self.sync()
                if timed:
                    times[mode] = (time.perf_counter() - t0) / n
        t = torch.tensor([times[True], times[False]], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        (with_overlap, without) = (float(t[0]), float(t[1]))
        self.dp_set_overlap(with_overlap < without)
        # restore: parameters, running statistics, Adam moments and step count, host counters; drain the loss slot
        self.sync()
        for (dst, src) in zip((self.params, self.buffers, self.exp_avg, self.exp_avg_sq, self.grads), keep):
            dst.copy_(src)
        torch.cuda.synchronize(self.device)
        (self.num_batches_tracked, self.adam_steps, self._slot) = counters
        check(self.lib.cae_set_adam_step(self.handle, int(self.adam_steps)))
        self._read_losses(slot, 1)
        self._cursor = None
        return {"overlap": with_overlap, "serial": without}

    def claim_slots(self, n):
        return self._claim_slots(n)

    def set_cursor(self, start, slot):
        self._set_cursor(start, slot)

    def dp_train_steps(self, which, perm_dev, batch, global_batch, sync_bn, nsteps=1):
        """nsteps data-parallel optimiser steps: step k takes this rank's `batch` rows starting at cursor + k*global_batch"""
        ptr = perm_dev.data_ptr() if perm_dev is not None else None
        check(self.lib.cae_dp_train_steps(self.handle, which, ptr, int(batch), int(global_batch), 1 if sync_bn else 0,
                                          int(nsteps)))
        self._advance(int(global_batch) * int(nsteps), nsteps)
        self.num_batches_tracked += nsteps
        self.adam_steps += nsteps

    def dp_eval_steps(self, which, perm_dev, batch, global_batch, nsteps=1):
        ptr = perm_dev.data_ptr() if perm_dev is not None else None
        check(self.lib.cae_dp_eval_steps(self.handle, which, ptr, int(batch), int(global_batch), int(nsteps)))
        self._advance(int(global_batch) * int(nsteps), nsteps)

    def dp_read_losses(self, first, n):
        out = (C.c_double * n)()
        check(self.lib.cae_dp_read_losses(self.handle, first, n, out))
        return [float(v) for v in out]

    def score(self, x):
        """eval-mode forward of an explicit batch (B,C,H,W) fp32 CUDA tensor -> (B,C,H,W)"""
        if x.dtype != torch.float32 or not x.is_cuda:
            raise CaeError("score() needs an fp32 CUDA tensor")
        self._same_device(x, "score() input")
        x = x.contiguous()
        out = torch.empty((x.shape[0],) + tuple(self.out_shape), dtype=torch.float32, device=self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        done = 0
        while done < x.shape[0]:
            n = min(self.max_batch, x.shape[0] - done)
            check(self.lib.cae_score(self.handle, x[done:done + n].data_ptr(), n, out[done:done + n].data_ptr()))
            done += n
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        x.record_stream(self.stream)
        out.record_stream(self.stream)
        return out

    def _module_forward(self, fn, x, row_shape, what):
        if x.dtype != torch.float32 or not x.is_cuda:
            raise CaeError(f"{what}() needs an fp32 CUDA tensor")
        self._same_device(x, f"{what}() input")
        x = x.contiguous()
        out = torch.empty((x.shape[0],) + tuple(row_shape), dtype=torch.float32, device=self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        done = 0
        while done < x.shape[0]:
            n = min(self.max_batch, x.shape[0] - done)
            check(fn(self.handle, x[done:done + n].data_ptr(), n, out[done:done + n].data_ptr()))
            done += n
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        x.record_stream(self.stream)
        out.record_stream(self.stream)
        return out

    def encode(self, x):
        """Encoder.forward (encoder.py:60-64) in eval mode: (B,C,h,w) fp32 CUDA tensor -> (B, latent)"""
        if tuple(x.shape[1:]) != tuple(self.in_shape):
            raise CaeError(f"encode() expects rows of shape {tuple(self.in_shape)}, got {tuple(x.shape[1:])}")
        return self._module_forward(self.lib.cae_encode, x, (self.latent_size,), "encode")

    def decode(self, z):
        """Decoder.forward (decoder.py:73-78) in eval mode: (B, latent) fp32 CUDA tensor -> (B,C,H,W), sigmoid applied"""
        if z.dim() != 2 or z.shape[1] != self.latent_size:
            raise CaeError(f"decode() expects (batch, {self.latent_size}), got {tuple(z.shape)}")
        return self._module_forward(self.lib.cae_decode, z, self.out_shape, "decode")

    def sync(self):
        check(self.lib.cae_sync(self.handle))

    # ---- measurement -----------------------------------------------------------------------
    def profile_begin(self):
        check(self.lib.cae_profile_begin(self.handle))

    def profile_end(self, capacity=65536):
        """[(name, layer, microseconds, algorithmic_bytes)] for every launch since profile_begin"""
        recs = (_lib.ProfileRecC * capacity)()
        n = check(self.lib.cae_profile_end(self.handle, recs, capacity))
        return [(recs[i].name.decode(), int(recs[i].layer), float(recs[i].micros), float(recs[i].bytes))
                for i in range(n)]

    # ---- test hook -------------------------------------------------------------------------
    def debug_read(self, what, index=0, count=None, dtype=np.float32):
        cap = int(count) if count is not None else (1 << 28)
        buf = np.empty(cap, dtype=dtype)
        n = self.lib.cae_debug_read(self.handle, what.encode(), int(index), buf.ctypes.data_as(C.c_void_p), cap)
        check(n)
        return buf[:n]


# ---- stateless loader kernels ------------------------------------------------------------------

def scan_f32(x):
    """(nan_count, nanmin, nanmax) of an fp32 CUDA tensor as python numbers (ds_dataset.py:43-58)"""
    lib = _lib.load()
    x = x.contiguous()
    out = (C.c_double * 3)()
    torch.cuda.synchronize(x.device)
    check(lib.cae_scan_f32(x.data_ptr(), x.numel(), None, out))
    return int(out[0]), float(out[1]), float(out[2])


def normalise_pack(src, dst, c_off, vmin, vmax, enable=True, dst_rows=None):
    """dst[row(i), c_off:c_off+Cv] = normalise(src[i]) on the device (ds_dataset.py:99-113,137-147).
    The range is formed in fp64 from python floats and rounded to fp32, as numpy does.
    dst_rows: int32 CUDA tensor, row(i) = dst_rows[i] (inverse_permutation() of a frozen sample order: the samples land in
    batch order, which is the DataLoader + collate stacking of conv_ae_model.py:291-292,315-325); None: row(i) = i."""
    lib = _lib.load()
    (n, c_src) = (int(src.shape[0]), int(src.shape[1]))
    hw = int(np.prod(src.shape[2:]))
    rng = float(vmax) - float(vmin)
    if dst_rows is not None:
        if dst_rows.dtype != torch.int32 or dst_rows.device != src.device or dst_rows.numel() != n or int(dst.shape[0]) != n:
            raise CaeError("normalise_pack: dst_rows must be an int32 tensor of one entry per sample on the data's device")
    with torch.cuda.device(src.device):
        check(lib.cae_normalise_pack_rows(src.data_ptr(), n, c_src, hw, dst.data_ptr(), int(dst.shape[1]), int(c_off),
                                          C.c_float(float(np.float32(vmin))), C.c_float(float(np.float32(rng))),
                                          1 if enable else 0, dst_rows.data_ptr() if dst_rows is not None else None,
                                          torch.cuda.current_stream(src.device).cuda_stream))


def inverse_permutation(order, device):
    """int32 CUDA tensor inv with inv[order[i]] = i (cae_invert_permutation): where each sample goes when the data set is
    laid out in the frozen order `order` (host sequence of sample indices, every index exactly once)."""
    lib = _lib.load()
    order = np.ascontiguousarray(np.asarray(order, dtype=np.int32))
    n = int(order.size)
    if n < 1 or not np.array_equal(np.sort(order), np.arange(n, dtype=np.int32)):
        raise CaeError("inverse_permutation: `order` must hold every sample index 0..n-1 exactly once")
    perm = torch.from_numpy(order).to(device)
    inv = torch.empty(n, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        check(lib.cae_invert_permutation(perm.data_ptr(), n, inv.data_ptr(), torch.cuda.current_stream(device).cuda_stream))
    torch.cuda.synchronize(device)
    return inv


def denormalise_f64(y, vmin, vmax):
    """float64 CUDA tensor min + y*(max-min) (ds_dataset.py:131-135 on base_model.py:123's fp64 array)"""
    lib = _lib.load()
    y = y.contiguous()
    out = torch.empty(y.shape, dtype=torch.float64, device=y.device)
    check(lib.cae_denormalise_f64(y.data_ptr(), y.numel(), float(vmin), float(vmax) - float(vmin), out.data_ptr(),
                                  torch.cuda.current_stream(y.device).cuda_stream))
    return out


def metric_sums(y, actual, mask, vmin, vmax):
    """cae_metric_sums: per-instance fp64 sums (n, 8) of the denormalised scores `vmin + y*(vmax-vmin)` against
    `actual` over the pixels where `mask` (same shape, or None = all) is non-zero.  CUDA fp32 tensors (n, ...)."""
    lib = _lib.load()
    y, actual = y.contiguous(), actual.contiguous()
    if y.shape != actual.shape:
        raise ValueError("The shapes of 'actual' and 'estimates' must match.")
    n = int(y.shape[0])
    elems = y.numel() // max(n, 1)
    if mask is not None:
        mask = mask.to(device=y.device, dtype=torch.float32).expand(y.shape).contiguous()
    out = torch.empty((n, 8), dtype=torch.float64, device=y.device)
    stream = torch.cuda.current_stream(y.device).cuda_stream
    for lo in range(0, n, 65535):
        hi = min(n, lo + 65535)
        check(lib.cae_metric_sums(y[lo:hi].data_ptr(), actual[lo:hi].data_ptr(),
                                  mask[lo:hi].data_ptr() if mask is not None else None, hi - lo, elems,
                                  float(vmin), float(vmax) - float(vmin), out[lo:hi].data_ptr(), stream))
    return out


_STAGE_BYTES = 64 << 20


def upload_f32(arr, device):
    """numpy (N,...) array -> fp32 CUDA tensor.  A C-contiguous big-endian float32 array (a NetCDF-3 slab, usually a
    view of the file mapping) is copied as raw bytes through a pinned staging buffer and byte-swapped on the GPU
    (cae_bswap32); anything else is converted by numpy first."""
    import numpy as np
    arr = np.asarray(arr)
    if not (arr.dtype == np.dtype(">f4") and arr.dtype.byteorder == ">" and arr.flags.c_contiguous and arr.size > 0):
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(device)
    lib = _lib.load()
    out = torch.empty(arr.shape, dtype=torch.float32, device=device)
    words = out.view(-1).view(torch.int32)
    src = arr.reshape(-1).view(np.uint8)
    nbytes = src.size
    stage = torch.empty(min(nbytes, _STAGE_BYTES), dtype=torch.uint8).pin_memory()
    stage_np = stage.numpy()
    dst_bytes = words.view(torch.uint8)
    for lo in range(0, nbytes, _STAGE_BYTES):
        hi = min(nbytes, lo + _STAGE_BYTES)
        stage_np[:hi - lo] = src[lo:hi]
        dst_bytes[lo:hi].copy_(stage[:hi - lo], non_blocking=False)
    check(lib.cae_bswap32(words.data_ptr(), words.numel(), torch.cuda.current_stream(device).cuda_stream))
    return out
