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
