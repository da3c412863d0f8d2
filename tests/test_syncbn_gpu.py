"""SyncBN data parallelism on ONE GPU: two engines stand in for two ranks (each on half of the
batch, each driven by its own host thread); the all-reduce callback adds the two engines'
BatchNorm sum tables.  Their summed gradients, running statistics and losses must equal a single
engine's result on the whole batch - the parity statement for the DP configuration (BASELINE cfg4:
global batch sharded over ranks vs the single-device reference)."""
import threading

import numpy as np
import pytest
import torch

from helpers import GoldenCase

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cfg2_b4", "odd_k5_b5", "tidal_b3"])
def test_two_ranks_with_syncbn_equal_one_device(name):
    from cae_tools_amd.engine import HipEngine
    case = GoldenCase(name)
    x = torch.from_numpy(np.concatenate([case.x, case.x2])).cuda()
    t = torch.from_numpy(np.concatenate([case.t, case.t2])).cuda()
    n = x.shape[0]
    half = n // 2
    sizes = [half, n - half]

    def make():
        e = HipEngine(case.spec, case.meta["fc"], case.meta["latent"], max_batch=n)
        e.load_state(case.group("init/enc/"), case.group("init/dec/"))
        e.set_hyper(lr=case.meta["lr"], weight_decay=case.meta["weight_decay"])
        e.set_dataset(0, x, t)
        return e

    full = make()
    slot = full.forward_backward(0, None, 0, n, n)
    loss_full = full._read_losses(slot, 1)[0]
    full.sync()
    g_full = full.grads.cpu().numpy().astype(np.float64)

    ranks = [make(), make()]
    barrier = threading.Barrier(2)
    tables = [None, None]
    slots = [None, None]
    errors = []

    def allreduce_for(r):
        def fn(table):
            torch.cuda.synchronize()
            tables[r] = table
            barrier.wait()
            if r == 0:
                total = tables[0] + tables[1]
                tables[0].copy_(total)
                tables[1].copy_(total)
                torch.cuda.synchronize()
            barrier.wait()
        return fn

    def run(r):
        try:
            start = 0 if r == 0 else sizes[0]
            slots[r] = ranks[r].forward_backward_sync(0, None, start, sizes[r], n, 2, allreduce_for(r))
            ranks[r].sync()
        except Exception as ex:  # pragma: no cover
            errors.append(ex)
            barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t_.start() for t_ in th]
    [t_.join(timeout=120) for t_ in th]
    assert not errors, errors
    loss = sum(ranks[r]._read_losses(slots[r], 1)[0] for r in range(2))
    assert abs(loss - loss_full) <= 1e-6 * abs(loss_full)
    g_sum = sum(ranks[r].grads.cpu().numpy().astype(np.float64) for r in range(2))
    scale = np.abs(g_full).max()
    assert np.abs(g_sum - g_full).max() <= 2e-5 * scale, np.abs(g_sum - g_full).max() / scale
    # running statistics: identical on both ranks and equal to the single-device ones
    b_full = full.buffers.cpu().numpy()
    for r in range(2):
        np.testing.assert_allclose(ranks[r].buffers.cpu().numpy(), b_full, rtol=1e-6, atol=1e-7)
    # and the local-BatchNorm half-steps do NOT reproduce it (the reason SyncBN exists)
    loc = make()
    loc.forward_backward(0, None, 0, sizes[0], n)
    loc.sync()
    g0 = loc.grads.cpu().numpy().astype(np.float64)
    loc2 = make()
    loc2.forward_backward(0, None, sizes[0], sizes[1], n)
    loc2.sync()
    g_local = g0 + loc2.grads.cpu().numpy().astype(np.float64)
    assert np.abs(g_local - g_full).max() > 1e-3 * scale
