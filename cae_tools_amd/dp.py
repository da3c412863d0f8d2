"""Data-parallel training: one process per GPU, minibatch sharded over ranks, one all-reduce of the
flat fp32 gradient arena per step (RCCL over xGMI through torch.distributed's "nccl" backend).

The reference has no multi-device path; this is the build's addition.  Semantics:
the global batch of a step is the union of the ranks' local batches; the loss is the mean over the
global batch, so each rank back-propagates sum(local terms)/global_count and the SUM all-reduce
yields the global-mean gradient on every rank; Adam then runs identically everywhere (weights stay
bit-identical across ranks because they start equal and see the same reduced gradient).
BatchNorm uses per-rank batch statistics by default (torch DDP's behaviour without SyncBatchNorm);
sync_bn=True all-reduces the BatchNorm sum tables between launches so that N ranks x batch/N give
the single-device batch-N result.

`engine` is anything with: .grads (flat tensor), .forward_backward(which, perm, start, size,
global_batch) and .adam_step() — HipEngine in production; the gloo CPU tests drive the same class
with a CPU stand-in to check the sharding / reduction logic.
"""
import contextlib

import torch


class DataParallel:

    def __init__(self, engine, dist, group=None, sync_bn=False):
        self.engine = engine
        self.dist = dist
        self.group = group
        self.sync_bn = sync_bn   # BatchNorm over the global batch (parity with the single-device reference)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._sizes = None

    def _stream_ctx(self):
        stream = getattr(self.engine, "stream", None)
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def broadcast_parameters(self, src=0):
        """make every rank start from rank `src`'s weights / running stats / Adam state"""
        with self._stream_ctx():
            for name in ("params", "buffers", "exp_avg", "exp_avg_sq"):
                t = getattr(self.engine, name, None)
                if t is not None:
                    self.dist.broadcast(t, src=src, group=self.group)

    def global_batch(self, local_size, equal=True):
        if equal:
            return local_size * self.world
        t = torch.tensor([local_size], dtype=torch.int64, device=self.engine.grads.device)
        self.dist.all_reduce(t, group=self.group)
        return int(t.item())

    def train_step(self, which, perm, start, size, global_batch=None):
        """one optimiser step on this rank's shard perm[start:start+size]"""
        gb = global_batch if global_batch is not None else size * self.world
        if self.sync_bn:
            slot = self.engine.forward_backward_sync(
                which, perm, start, size, gb, self.world,
                lambda table: self.dist.all_reduce(table, op=self.dist.ReduceOp.SUM, group=self.group))
        else:
            slot = self.engine.forward_backward(which, perm, start, size, gb)
        with self._stream_ctx():
            self.dist.all_reduce(self.engine.grads, op=self.dist.ReduceOp.SUM, group=self.group)
        self.engine.adam_step()
        return slot


class GradientHalfSteps:
    """UnetEngine / VaeEngine / LinearEngine behind the three-member interface DataParallel drives: a persistent flat
    gradient buffer, forward_backward into it with the local/global weight applied in the kernel that narrows the fp64
    accumulator (grad_scale of *_forward_backward), and the optimiser half-step (*_apply_gradients) after the all-reduce.
    BatchNorm statistics stay per rank.  The UNET's masked MSE divides by the LOCAL mask count, so with unequal mask
    coverage across ranks the reduced gradient weights shards by sample count rather than by valid-pixel count."""

    def __init__(self, engine):
        self.engine = engine
        self.grads = torch.zeros(engine.n_param, dtype=torch.float32, device=engine.device)
        self.stream = engine.stream
        (self.params, self.exp_avg, self.exp_avg_sq) = (engine.params, engine.exp_avg, engine.exp_avg_sq)
        self.buffers = getattr(engine, "buffers", None)
        self._slot = 0

    def forward_backward(self, which, perm, start, size, global_batch):
        slot = self._slot
        self._slot = (slot + 1) % self.engine.loss_slots
        self.engine.forward_backward(which, perm, start, size, slot=slot, global_batch=global_batch, out=self.grads)
        return slot

    def adam_step(self):
        self.engine.apply_gradients(self.grads)


def shard_bounds(n, world, rank):
    """contiguous split of a global batch of n samples: rank r takes rows [lo, hi) — the union over
    ranks is the reference's single-device batch"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
