"""The in-library data-parallel path (include/cae_hip.h cae_dp_*: RCCL communicator owned by libcae_hip, two gradient
buckets on a second stream, the step replayed from one hipGraph) on ONE GPU: a world-size-1 RCCL group drives exactly the
code an 8-rank run drives (BASELINE cfg4), so its results must equal the fused single-device step's.

* DataParallel(HipEngine) without / with SyncBN == cae_train_step, step by step (local BatchNorm: bit for bit - both paths
  narrow the same fp64 accumulators to fp32 and run the same Adam kernel; SyncBN runs the per-layer launches: 2e-6).
* run_batches (global batches, partial last one, 64-step graphs) == HipEngine.run_batches, train and eval passes.
* the loss / gradient scale of a shard is 1/global count (a shard of half a global batch gives half the gradient).
* an empty shard takes part in the step without launching the model.
* ConvAEModel.train under the process group (CAE_FORCE_DP=1) == the single-process train().
The 2-rank SyncBN arithmetic itself (two shards == one device) is tests/test_syncbn_gpu.py; the multi-rank host logic
(sharding, cursor, partial and empty shards, loss reduction) is tests/test_dp_gloo.py.
"""
import os
import socket

import numpy as np
import pytest
import torch

from helpers import GoldenCase

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dist1():
    import torch.distributed as dist
    if dist.is_initialized():
        yield dist
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    yield dist
    dist.destroy_process_group()


def _make(case, x, t, max_batch=None):
    from cae_tools_amd.engine import HipEngine
    e = HipEngine(case.spec, case.meta["fc"], case.meta["latent"], max_batch=max_batch or x.shape[0])
    e.load_state(case.group("init/enc/"), case.group("init/dec/"))
    e.set_hyper(lr=case.meta["lr"], weight_decay=case.meta["weight_decay"])
    e.set_dataset(0, x, t)
    e.set_dataset(1, x, t)
    return e


def _data(case):
    x = torch.from_numpy(np.concatenate([case.x, case.x2])).cuda()
    t = torch.from_numpy(np.concatenate([case.t, case.t2])).cuda()
    return x, t


def _state(e):
    e.sync()
    return (e.params.cpu().numpy().copy(), e.buffers.cpu().numpy().copy(), e.exp_avg.cpu().numpy().copy(),
            e.exp_avg_sq.cpu().numpy().copy())


def _same(u, v, what, lr_steps):
    """two runs of the SAME arithmetic (only the arrival order of fp64 atomics differs): equal up to last-bit effects"""
    d = np.abs(u.astype(np.float64) - v.astype(np.float64))
    if what == "params":
        assert d.max() <= 1e-3 * lr_steps, (what, d.max())
    else:
        assert d.max() <= 1e-5 * max(np.abs(u).max(), 1e-30), (what, d.max())


@pytest.mark.parametrize("name", ["cfg2_b4", "handspec_b4", "tidal_b3"])
@pytest.mark.parametrize("sync_bn,overlap", [(False, True), (False, False), (True, True)])
def test_one_rank_group_equals_the_fused_step(dist1, name, sync_bn, overlap):
    from cae_tools_amd.dp import DataParallel
    case = GoldenCase(name)
    (x, t) = _data(case)
    n = x.shape[0]
    (a, b) = (_make(case, x, t), _make(case, x, t))
    dp = DataParallel(b, dist1, sync_bn=sync_bn, overlap=overlap)   # True: first bucket on the second stream
    assert dp.native and b.dp_world == 1
    dp.broadcast_parameters(0)
    (la, lb) = ([], [])
    for step in range(4):
        la.append(a.train_step(0, None, 0, n))
        slot = dp.train_step(0, None, 0, n)
        lb.append(b.dp_read_losses(slot, 1)[0])
    # fp64 atomics land in a different order from run to run: ~1e-9 on a loss, a last-bit flip here and there in fp32
    np.testing.assert_allclose(lb, la, rtol=2e-6 if sync_bn else 1e-7, atol=0)
    for (u, v, what) in zip(_state(a), _state(b), ("params", "buffers", "exp_avg", "exp_avg_sq")):
        if sync_bn:
            # per-layer launches instead of the fused head/tail: other summation orders; Adam turns a gradient
            # difference of relative size d into a parameter difference of at most ~lr*d per step
            d = np.abs(u - v)
            if what == "params":      # the bounds of test_hip_parity's Adam checks, over the 4 steps
                lr = case.meta["lr"]
                assert d.max() <= 0.25 * lr * 4 and np.quantile(d, 0.999) <= 0.02 * lr * 4, (d.max(), np.quantile(d, 0.999))
            elif what == "buffers":
                assert d.max() <= 2e-5 * np.abs(u).max(), d.max()
            else:
                assert d.max() <= 1e-3 * max(np.abs(u).max(), 1e-30), (what, d.max())
        else:
            _same(u, v, what, case.meta["lr"] * 4)
    assert b.adam_steps == a.adam_steps == 4 and b.num_batches_tracked == 4


def test_calibration_leaves_the_training_state_alone(dist1):
    from cae_tools_amd.dp import DataParallel
    case = GoldenCase("cfg2_b4")
    (x, t) = _data(case)
    n = x.shape[0]
    b = _make(case, x, t)
    dp = DataParallel(b, dist1, sync_bn=False)
    dp.train_step(0, None, 0, n)                     # calibrates first (overlap="auto"), then takes ONE step
    assert dp.calibration and b.adam_steps == 1 and b.num_batches_tracked == 1
    a = _make(case, x, t)
    a.train_step(0, None, 0, n)
    for (u, v, what) in zip(_state(a), _state(b), ("params", "buffers", "exp_avg", "exp_avg_sq")):
        _same(u, v, what, case.meta["lr"])


def test_passes_over_global_batches_match_the_single_device_passes(dist1):
    from cae_tools_amd.dp import DataParallel
    case = GoldenCase("cfg1_b3")
    (x, t) = _data(case)
    (n, gb) = (101, 8)                                       # 12 full global batches + a partial one of 5
    reps = n // x.shape[0] + 1
    x = x.repeat(reps, 1, 1, 1)[:n].contiguous()
    t = t.repeat(reps, 1, 1, 1)[:n].contiguous()
    x += 0.05 * torch.rand(x.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    t = (t + 0.05 * torch.rand(t.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))).clamp(0, 1)
    (a, b) = (_make(case, x, t, max_batch=gb), _make(case, x, t, max_batch=gb))
    a.STEPS_PER_GRAPH = b.STEPS_PER_GRAPH = 5                # 12 = 2 five-step graphs + 2 single steps
    dp = DataParallel(b, dist1, sync_bn=False)               # overlap="auto": times both structures, restores the state
    perm_np = np.random.default_rng(5).permutation(n).astype(np.int32)
    (pa, pb) = (a.upload_perm(perm_np), b.upload_perm(perm_np))
    for epoch in range(2):
        la = a.run_batches(0, pa, n, gb, train=True)
        lb = dp.run_batches(0, pb, n, gb, train=True)
        assert len(lb) == 13
        np.testing.assert_allclose(lb, la, rtol=2e-5)
        ea = a.run_batches(1, pa, n, gb, train=False)
        eb = dp.run_batches(1, pb, n, gb, train=False)
        np.testing.assert_allclose(eb, ea, rtol=2e-5)
    assert set(dp.calibration) == {"overlap", "serial"} and all(0 < v < 0.1 for v in dp.calibration.values())
    for (u, v, what) in zip(_state(a), _state(b), ("params", "buffers", "exp_avg", "exp_avg_sq")):
        d = np.abs(u - v)
        if what == "params":       # the same run up to the arrival order of fp64 atomics (bounds as in test_hip_parity's Adam checks)
            lr26 = case.meta["lr"] * 26
            assert d.max() <= 0.25 * lr26 and np.quantile(d, 0.999) <= 0.02 * lr26, (d.max(), np.quantile(d, 0.999))
        elif what == "buffers":
            assert d.max() <= 1e-4 * np.abs(u).max(), d.max()


def test_a_shard_contributes_its_share_of_the_global_mean(dist1):
    """loss and gradients of rows [0, h) with global_batch = 2h are exactly half of those with global_batch = h
    (same BatchNorm batch): the SUM all-reduce over equal shards then yields the global-batch mean gradient"""
    case = GoldenCase("cfg2_b4")
    (x, t) = _data(case)
    h = x.shape[0] // 2
    (a, b) = (_make(case, x, t), _make(case, x, t))
    sa = a.forward_backward(0, None, 0, h, h)
    sb = b.forward_backward(0, None, 0, h, 2 * h)
    (la, lb) = (a._read_losses(sa, 1)[0], b._read_losses(sb, 1)[0])
    assert abs(lb - 0.5 * la) <= 1e-7 * abs(la)
    (ga, gb) = (a.grads.cpu().numpy().astype(np.float64), b.grads.cpu().numpy().astype(np.float64))
    assert np.abs(ga).max() > 0
    assert np.abs(gb - 0.5 * ga).max() <= 2e-6 * np.abs(ga).max()


def test_an_empty_shard_takes_part_in_the_step(dist1):
    from cae_tools_amd.dp import DataParallel
    case = GoldenCase("cfg2_b4")
    (x, t) = _data(case)
    for sync_bn in (False, True):
        b = _make(case, x, t)
        DataParallel(b, dist1, sync_bn=sync_bn)
        before = _state(b)
        slot = b.claim_slots(1)
        b.set_cursor(0, slot)
        b.dp_train_steps(0, None, 0, 4, sync_bn, 1)      # batch 0 of a global batch of 4
        after = _state(b)
        assert b.dp_read_losses(slot, 1) == [0.0]
        np.testing.assert_array_equal(after[1], before[1])                     # no BatchNorm update without samples
        # zero gradient: Adam sees only the L2 term wd * w
        assert np.isfinite(after[0]).all() and np.abs(after[0] - before[0]).max() <= 1.01 * case.meta["lr"]
        assert b.adam_steps == 1
        # and the engine carries on normally afterwards
        assert np.isfinite(b.dp_read_losses(DataParallel(b, dist1, sync_bn=sync_bn).train_step(0, None, 0, 4), 1)[0])


def test_model_train_under_a_process_group(dist1, tmp_path, monkeypatch, capsys):
    """ConvAEModel.train() inside a torch.distributed launch (here: one rank, CAE_FORCE_DP=1) takes the data-parallel
    path - sharded global batches, in-library collectives, rank 0 saves - and lands where the plain train() lands"""
    from cae_tools_amd.data.arrays import DataArray, Dataset
    from cae_tools_amd.models.conv_ae_model import ConvAEModel
    rng = np.random.default_rng(11)

    def ds(n):
        d = Dataset()
        lo = rng.random((n, 1, 16, 16)).astype(np.float32) + 280.0
        d["lowres"] = DataArray(lo, dims=("n", "chan", "y1", "x1"))
        d["hires"] = DataArray(np.kron(lo, np.ones((1, 1, 4, 4), dtype=np.float32)) + 0.1 * rng.random((n, 1, 64, 64)).astype(np.float32),
                               dims=("n", "chan", "y2", "x2"))
        return d

    (train, test) = (ds(23), ds(9))

    def run(force, folder):
        if force:
            monkeypatch.setenv("CAE_FORCE_DP", "1")
        else:
            monkeypatch.delenv("CAE_FORCE_DP", raising=False)
        torch.manual_seed(4)
        m = ConvAEModel(batch_size=5, nr_epochs=3, test_interval=1, fc_size=16, encoded_dim_size=4, lr=1e-3)
        m.sync_bn = False
        metrics = m.train(["lowres"], "hires", train, test, model_path=str(folder))
        return m, metrics

    (plain, mp) = run(False, tmp_path / "plain")
    (par, md) = run(True, tmp_path / "dp")
    assert par._engine.dp_world == 1 and plain._engine.dp_world == 0
    np.testing.assert_allclose(par.history["train_loss"], plain.history["train_loss"], rtol=1e-4)
    np.testing.assert_allclose(par.history["test_loss"], plain.history["test_loss"], rtol=1e-4)
    for side in ("encoder.weights", "decoder.weights"):
        (sa, sb) = (torch.load(tmp_path / "plain" / side, weights_only=True), torch.load(tmp_path / "dp" / side, weights_only=True))
        assert list(sa) == list(sb)
        for k in sa:
            if sa[k].dtype == torch.int64:
                assert torch.equal(sa[k], sb[k]), k
            else:
                assert float((sa[k] - sb[k]).abs().max()) <= 0.1 * 1e-3 * 15, k      # 15 steps at lr 1e-3
    assert abs(md["test"]["mse"] - mp["test"]["mse"]) <= 1e-3 * mp["test"]["mse"]
    # apply() shards the cases over the ranks and gathers: same predictions either way
    (s1, s2) = (ds(7), None)
    s2 = Dataset({"lowres": s1["lowres"]})
    monkeypatch.delenv("CAE_FORCE_DP", raising=False)
    plain.apply(s1, ["lowres"])
    monkeypatch.setenv("CAE_FORCE_DP", "1")
    par.apply(s2, ["lowres"])
    np.testing.assert_allclose(s2["model_output"].values, s1["model_output"].values, rtol=0, atol=2e-3)
    assert s2["model_output"].values.shape == (7, 1, 64, 64) and s2["model_output"].values.dtype == np.float64
