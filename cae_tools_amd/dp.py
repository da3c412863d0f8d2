"""Data-parallel training: one process per GPU, every frozen global batch sharded over the ranks.

The reference has no multi-device path (it picks ONE device: conv_ae_model.py:294-297, 312-313);
this is the build's addition.  Semantics: the global batch of a step is the union of the ranks' local batches (rank r
takes rows shard_bounds(n, world, r) of the frozen batch, so the union IS the reference's batch); the loss is the mean
over the global batch, so each rank back-propagates sum(local terms)/global_count and a SUM all-reduce yields the
global-mean gradient on every rank; Adam then runs identically everywhere (weights stay bit-identical across ranks
because they start equal and see the same reduced gradient).  BatchNorm: sync_bn=True all-reduces the sum tables so that
N ranks x batch/N reproduce the single-device batch-N step (the parity mode); sync_bn=False keeps
per-rank batch statistics (torch DDP's behaviour without SyncBatchNorm; the throughput mode).

Two engine protocols:

* native (HipEngine): the collectives live INSIDE libcae_hip (include/cae_hip.h, cae_dp_*): an RCCL communicator owned
  by the library, gradient buckets all-reduced on a second HIP stream while backward still runs, the whole step replayed
  from one hipGraph.  torch.distributed is only the rendezvous that carries the 128-byte RCCL id.  The engine exposes
  set_cursor / claim_slots / dp_train_steps / dp_eval_steps / dp_read_losses / dp_broadcast.
* half-steps (UnetEngine / VaeEngine / LinearEngine behind GradientHalfSteps, and the CPU stand-ins of the gloo
  tests): .grads, .forward_backward(which, perm, start, size, global_batch), .adam_step(); the all-reduce is
  torch.distributed's.
"""
import contextlib

import torch


def shard_bounds(n, world, rank):
    """contiguous split of a global batch of n samples: rank r takes rows [lo, hi) — the union over
    ranks is the reference's single-device batch"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DataParallel:

    def __init__(self, engine, dist, group=None, sync_bn=False, overlap="auto"):
        self.engine = engine
        self.dist = dist
        self.group = group
        self.sync_bn = sync_bn   # BatchNorm over the global batch (parity with the single-device reference)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.native = hasattr(engine, "dp_train_steps")
        if self.native and getattr(engine, "dp_world", 0) != self.world:
            engine.dp_init(dist, group)
        # native engines: first gradient bucket on the second stream (True), both on the main stream (False), or whichever
        # HipEngine.dp_calibrate measures as faster on this communicator, decided once before the first training step
        self.overlap = overlap
        self.calibration = None
        if self.native and overlap != "auto":
            engine.dp_set_overlap(bool(overlap))

    def _calibrate(self, which, perm, start, batch, global_batch):
        """dp_calibrate is COLLECTIVE (RCCL steps + an all-reduce of the timings), so whether it runs is decided from values
        every rank shares: a global batch smaller than the world leaves some rank's shard empty, and then no rank calibrates
        (the default structure stays) rather than some entering the collective and the others not."""
        if not (self.native and self.overlap == "auto" and self.calibration is None and not self.sync_bn):
            return
        if int(global_batch) < self.world:
            self.calibration = {}
            return
        self.calibration = self.engine.dp_calibrate(self.dist, which, perm, start, batch, global_batch, False,
                                                    group=self.group)

    def _stream_ctx(self):
        stream = getattr(self.engine, "stream", None)
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def broadcast_parameters(self, src=0):
        """make every rank start from rank `src`'s weights / running stats / Adam state"""
        if self.native:
            self.engine.dp_broadcast(src)
            return
        with self._stream_ctx():
            for name in ("params", "buffers", "exp_avg", "exp_avg_sq"):
                t = getattr(self.engine, name, None)
                if t is not None:
                    self.dist.broadcast(t, src=src, group=self.group)

    def broadcast_buffers(self, src=0):
        """BatchNorm running statistics of rank `src` to every rank: with per-rank statistics the ranks' running
        averages differ (DDP broadcasts rank 0's every forward); done before a pass that scores with them and before
        saving, so that every rank evaluates one model"""
        if self.native:
            self.engine.dp_broadcast(src, params=False, buffers=True, moments=False)
        else:
            t = getattr(self.engine, "buffers", None)
            if t is not None:
                with self._stream_ctx():
                    self.dist.broadcast(t, src=src, group=self.group)

    def global_batch(self, local_size, equal=True):
        if equal:
            return local_size * self.world
        device = getattr(self.engine, "device", None) or self.engine.grads.device
        t = torch.tensor([local_size], dtype=torch.int64, device=device)
        self.dist.all_reduce(t, group=self.group)
        return int(t.item())

    # ---- one step -----------------------------------------------------------------------------
    def train_step(self, which, perm, start, size, global_batch=None):
        """one optimiser step on this rank's shard perm[start:start+size]; returns its loss slot"""
        gb = global_batch if global_batch is not None else size * self.world
        eng = self.engine
        if self.native:
            self._calibrate(which, perm, start, size, gb)
            slot = eng.claim_slots(1)
            eng.set_cursor(start, slot)
            eng.dp_train_steps(which, perm, size, gb, self.sync_bn, 1)
            return slot
        if self.sync_bn:
            slot = eng.forward_backward_sync(
                which, perm, start, size, gb, self.world,
                lambda table: self.dist.all_reduce(table, op=self.dist.ReduceOp.SUM, group=self.group))
        else:
            slot = eng.forward_backward(which, perm, start, size, gb)
        with self._stream_ctx():
            self.dist.all_reduce(eng.grads, op=self.dist.ReduceOp.SUM, group=self.group)
        eng.adam_step()
        return slot

    # ---- one pass (the epoch loops of conv_ae_model.py:185-221 over sharded global batches) ------------
    def run_batches(self, which, perm, n, global_batch, train=True):
        """One pass over perm[0:n] in GLOBAL batches of `global_batch` (the last one partial, drop_last=False as in
        conv_ae_model.py:291-292); this rank runs rows shard_bounds(size, world, rank) of each.  Returns the per-batch
        mean losses over the global batches, identical on every rank.  Native engines only."""
        if not self.native:
            raise TypeError("run_batches drives the in-library data-parallel path (HipEngine)")
        eng = self.engine
        (full, rem) = divmod(int(n), int(global_batch))
        nb = full + (1 if rem else 0)
        first = eng.claim_slots(nb)

        def steps(batch, gb, k):
            if train:
                eng.dp_train_steps(which, perm, batch, gb, self.sync_bn, k)
            else:
                eng.dp_eval_steps(which, perm, batch, gb, k)

        if full:
            (lo, hi) = shard_bounds(global_batch, self.world, self.rank)
            if train:
                self._calibrate(which, perm, lo, hi - lo, global_batch)
            eng.set_cursor(lo, first)      # the device cursor then moves one global batch per step
            left = full
            per_graph = getattr(eng, "STEPS_PER_GRAPH", 64)
            while left >= per_graph:       # three graph shapes per batch size at most: K steps, the remainder, 1 step
                steps(hi - lo, global_batch, per_graph)
                left -= per_graph
            if left:
                steps(hi - lo, global_batch, left)
        if rem:
            (lo, hi) = shard_bounds(rem, self.world, self.rank)
            eng.set_cursor(full * global_batch + lo, first + full)
            steps(hi - lo, rem, 1)         # hi == lo (an empty shard) still takes part in the collectives
        return eng.dp_read_losses(first, nb)


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


# ---- process-group plumbing for the model classes and CLIs ------------------------------------------

def env_world():
    """(rank, local_rank, world) from torch.distributed.run's environment; (0, 0, 1) outside it"""
    import os
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def select_device():
    """Make LOCAL_RANK's GPU the current device of a rank of a torch.distributed.run launch.  Every entry point that may
    run in a rank (train / apply / evaluate, the CLIs) calls this BEFORE its first CUDA allocation: DSDataset and the
    engines allocate on torch.cuda.current_device(), which is 0 in every rank until somebody says otherwise, and an engine
    on GPU r must not be handed GPU-0 pointers.  No-op outside a launch and without a GPU.  Returns the local rank."""
    (rank, local_rank, world) = env_world()
    if world > 1 and torch.cuda.is_available() and torch.cuda.current_device() != local_rank:
        torch.cuda.set_device(local_rank)
    return local_rank


def ensure_process_group():
    """torch.distributed handle when this process is one rank of a torch.distributed.run launch (initialising the
    default group on first use: RCCL when a GPU is present), else None.  The GPU of a rank is LOCAL_RANK."""
    import os
    import torch.distributed as dist
    select_device()
    if dist.is_available() and dist.is_initialized():
        # CAE_FORCE_DP=1: take the data-parallel code path on a one-rank group too (rehearsal on a one-GPU box and in tests)
        return dist if dist.get_world_size() > 1 or os.environ.get("CAE_FORCE_DP") == "1" else None
    (rank, local_rank, world) = env_world()
    if world <= 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist
