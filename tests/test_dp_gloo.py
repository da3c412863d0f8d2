"""Data-parallel host logic on CPU: two processes over the gloo backend drive
cae_tools_amd.dp.DataParallel with a CPU stand-in engine (the oracle), and must land on exactly
the parameters a single process gets by summing the two shards' gradients itself.
The stand-in lives here, in the tests: the product's DataParallel only sees the three-member
engine interface (.grads, .forward_backward, .adam_step)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


class OracleEngine:
    """CPU engine with the HipEngine members DataParallel uses"""

    def __init__(self, case):
        from helpers import oracle_model
        self.m = oracle_model(case)
        self.plist = [v for v in list(self.m.enc.values()) + list(self.m.dec.values()) if v.requires_grad]
        self.grads = torch.zeros(sum(p.numel() for p in self.plist))
        self.x = torch.from_numpy(np.concatenate([case.x, case.x2]))
        self.t = torch.from_numpy(np.concatenate([case.t, case.t2]))

    def forward_backward(self, which, perm, start, size, global_batch):
        idx = perm[start:start + size]
        y = self.m.forward(self.x[idx], train=True)
        # mean over the GLOBAL batch: sum of local squared errors / global element count
        loss = ((y - self.t[idx]) ** 2).sum() / (global_batch * y[0].numel())
        self.m.optim.zero_grad()
        loss.backward()
        self.grads.copy_(torch.cat([p.grad.reshape(-1) for p in self.plist]))
        return 0

    def adam_step(self):
        off = 0
        for p in self.plist:
            p.grad = self.grads[off:off + p.numel()].view_as(p).clone()
            off += p.numel()
        self.m.optim.step()

    def flat_params(self):
        return torch.cat([p.detach().reshape(-1) for p in self.plist])


def _worker(rank, world, port, name, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    from helpers import GoldenCase
    from cae_tools_amd.dp import DataParallel, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = GoldenCase(name)
    eng = OracleEngine(case)
    dp = DataParallel(eng, dist)
    assert not dp.native
    n = eng.x.shape[0]
    perm = torch.arange(n)
    for step in range(2):
        (lo, hi) = shard_bounds(n, world, rank)
        dp.train_step(0, perm, lo, hi - lo, global_batch=n)
    torch.save(eng.flat_params(), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_two_ranks_match_manual_shard_sum(tmp_path):
    sys.path.insert(0, HERE)
    from helpers import GoldenCase
    from cae_tools_amd.dp import shard_bounds
    name = "handspec_b4"
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0, r1), "ranks diverged: the reduced gradient or Adam differed"

    # single-process emulation of the same semantics: per-shard forward (per-shard BatchNorm batch
    # statistics), gradients of sum/global_count summed over the shards, one Adam step
    torch.set_num_threads(1)
    case = GoldenCase(name)
    ref = OracleEngine(case)
    n = ref.x.shape[0]
    perm = torch.arange(n)
    for step in range(2):
        total = torch.zeros_like(ref.grads)
        state_before = {k: v.clone() for k, v in list(ref.m.enc.items()) + list(ref.m.dec.items()) if "running" in k}
        for r in range(world):
            (lo, hi) = shard_bounds(n, world, r)
            ref.forward_backward(0, perm, lo, hi - lo, n)
            total += ref.grads
        ref.grads.copy_(total)
        ref.adam_step()
    np.testing.assert_allclose(r0.numpy(), ref.flat_params().numpy(), rtol=1e-6, atol=1e-7)


class OracleNativeEngine(OracleEngine):
    """CPU stand-in with the members DataParallel uses on a NATIVE engine (HipEngine's cae_dp_* surface): a device-side
    cursor that moves one GLOBAL batch per step, loss slots, collectives inside the engine (here: gloo)."""
    STEPS_PER_GRAPH = 2      # so that run_batches takes both its multi-step and its single-step route

    def __init__(self, case, world):
        super().__init__(case)
        self.dp_world = world            # "already joined": DataParallel does not call dp_init
        self.cursor = None
        self.slots = torch.zeros(64, dtype=torch.float64)
        self._next = 0
        self.calls = []

    def claim_slots(self, n):
        first = self._next
        self._next += n
        return first

    def set_cursor(self, start, slot):
        self.cursor = [int(start), int(slot)]

    def dp_set_overlap(self, enabled):
        pass

    def _step(self, perm, batch, gb, train):
        (start, slot) = self.cursor
        if batch > 0:
            idx = perm[start:start + batch]
            y = self.m.forward(self.x[idx], train=train)
            loss = ((y - self.t[idx]) ** 2).sum() / (gb * y[0].numel())
            self.slots[slot] += float(loss.detach())
        if train:
            self.m.optim.zero_grad()
            if batch > 0:
                loss.backward()
                self.grads.copy_(torch.cat([p.grad.reshape(-1) for p in self.plist]))
            else:
                self.grads.zero_()
            dist.all_reduce(self.grads)
            self.adam_step()
        self.cursor = [start + gb, slot + 1]

    def dp_train_steps(self, which, perm, batch, gb, sync_bn, nsteps=1):
        assert not sync_bn
        self.calls.append(("train", batch, gb, nsteps))
        for _ in range(nsteps):
            self._step(perm, batch, gb, True)

    def dp_eval_steps(self, which, perm, batch, gb, nsteps=1):
        self.calls.append(("eval", batch, gb, nsteps))
        with torch.no_grad():
            for _ in range(nsteps):
                self._step(perm, batch, gb, False)

    def _running(self):
        return [v for k, v in list(self.m.enc.items()) + list(self.m.dec.items()) if "running" in k]

    def dp_broadcast(self, src=0, params=True, buffers=True, moments=True):
        assert buffers and not params and not moments
        for v in self._running():
            dist.broadcast(v, src=src)

    def dp_read_losses(self, first, n):
        part = self.slots[first:first + n].clone()
        dist.all_reduce(part)
        self.slots[first:first + n] = 0
        return [float(v) for v in part]


def _pass_worker(rank, world, port, name, n, gb, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    from helpers import GoldenCase
    from cae_tools_amd.dp import DataParallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = OracleNativeEngine(GoldenCase(name), world)
    dp = DataParallel(eng, dist, overlap=False)
    assert dp.native
    perm = torch.from_numpy(np.random.default_rng(2).permutation(eng.x.shape[0])[:n])
    train = dp.run_batches(0, perm, n, gb, train=True)
    dp.broadcast_buffers(0)          # as ConvAEModel.train does before a test pass: rank 0's running statistics everywhere
    test = dp.run_batches(0, perm, n, gb, train=False)
    torch.save({"params": eng.flat_params(), "train": train, "test": test, "calls": eng.calls},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,n,gb", [(2, 7, 2), (3, 7, 3)])
def test_passes_over_sharded_global_batches(tmp_path, world, n, gb):
    """DataParallel.run_batches on `world` gloo ranks: global batches of gb over n samples (7 = 3 x 2 + 1 and 2 x 3 + 1:
    the short last batch leaves every rank but the first with an EMPTY shard), the device cursor stepping one global batch
    per step - against one process that walks the same global batches shard by shard."""
    sys.path.insert(0, HERE)
    from helpers import GoldenCase
    from cae_tools_amd.dp import shard_bounds
    name = "handspec_b4"
    mp.spawn(_pass_worker, args=(world, _free_port(), name, n, gb, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    for o in outs[1:]:
        assert torch.equal(o["params"], outs[0]["params"]), "ranks diverged"
        assert o["train"] == outs[0]["train"] and o["test"] == outs[0]["test"]      # every rank reports the global means
    # the last (1-sample) batch: rank 0 runs it, the others take part with an empty shard
    assert [c for c in outs[0]["calls"] if c[0] == "train"][-1] == ("train", 1, 1, 1)
    assert all([c for c in o["calls"] if c[0] == "train"][-1] == ("train", 0, 1, 1) for o in outs[1:])

    torch.set_num_threads(1)
    ref = OracleEngine(GoldenCase(name))
    perm = torch.from_numpy(np.random.default_rng(2).permutation(ref.x.shape[0])[:n])
    (train, test) = ([], [])
    for start in range(0, n, gb):
        size = min(gb, n - start)
        total = torch.zeros_like(ref.grads)
        loss = 0.0
        for r in range(world):
            (lo, hi) = shard_bounds(size, world, r)
            if hi > lo:
                idx = perm[start + lo:start + hi]
                running = [v for k, v in list(ref.m.enc.items()) + list(ref.m.dec.items()) if "running" in k]
                keep = [v.clone() for v in running]
                y = ref.m.forward(ref.x[idx], train=True)
                part = ((y - ref.t[idx]) ** 2).sum() / (size * y[0].numel())
                ref.m.optim.zero_grad()
                part.backward()
                if r > 0:       # the model that is scored afterwards carries rank 0's running statistics
                    with torch.no_grad():
                        for (v, k) in zip(running, keep):
                            v.copy_(k)
                total += torch.cat([p.grad.reshape(-1) for p in ref.plist])
                loss += float(part.detach())
        ref.grads.copy_(total)
        ref.adam_step()
        train.append(loss)
    with torch.no_grad():
        for start in range(0, n, gb):
            size = min(gb, n - start)
            loss = 0.0
            for r in range(world):
                (lo, hi) = shard_bounds(size, world, r)
                if hi > lo:
                    idx = perm[start + lo:start + hi]
                    y = ref.m.forward(ref.x[idx], train=False)
                    loss += float(((y - ref.t[idx]) ** 2).sum() / (size * y[0].numel()))
            test.append(loss)
    np.testing.assert_allclose(outs[0]["train"], train, rtol=1e-6)
    np.testing.assert_allclose(outs[0]["test"], test, rtol=5e-5)      # one-sample shards: the fp32 sum order of the reduced gradient shows
    # conv biases in front of a BatchNorm have an exactly-zero gradient; the ~1e-9 noise torch computes for them depends
    # on the summation order, and Adam turns noise into lr-sized steps: a handful of entries may differ by O(lr) per step
    d = np.abs(outs[0]["params"].numpy() - ref.flat_params().numpy())
    steps = -(-n // gb)
    assert np.quantile(d, 0.99) <= 1e-6 and d.max() <= GoldenCase(name).meta["lr"] * steps, (np.quantile(d, 0.99), d.max())


def test_gradient_half_steps_adapter_contract():
    """GradientHalfSteps hands DataParallel a persistent gradient buffer, rotates loss slots and forwards the
    local/global weight (host logic only: the engine here records the calls)"""
    from cae_tools_amd.dp import GradientHalfSteps

    class Recorder:
        n_param = 5
        device = torch.device("cpu")
        stream = None
        loss_slots = 3
        params = exp_avg = exp_avg_sq = torch.zeros(5)

        def __init__(self):
            self.calls = []

        def forward_backward(self, which, perm, start, size, slot=0, global_batch=None, out=None):
            out.fill_(size / global_batch)
            self.calls.append(("fb", which, start, size, slot, global_batch, out.data_ptr()))

        def apply_gradients(self, g):
            self.calls.append(("apply", g.data_ptr(), float(g[0])))

    rec = Recorder()
    half = GradientHalfSteps(rec)
    assert half.buffers is None and half.grads.shape == (5,)
    slots = [half.forward_backward(0, None, 4 * i, 2, 8) for i in range(4)]
    assert slots == [0, 1, 2, 0]
    half.adam_step()
    assert all(c[-1] == half.grads.data_ptr() for c in rec.calls if c[0] == "fb")
    assert rec.calls[-1] == ("apply", half.grads.data_ptr(), 0.25)
