"""Data-parallel half-steps of the UNET / var / linear engines (*_forward_backward with grad_scale + *_apply_gradients)
driven through cae_tools_amd.dp.DataParallel over a world-size-1 RCCL group, against the engines' own fused train_step,
and — for the linear model, which has no batch statistics — two shards summed against the full-batch step."""
import os
import socket

import numpy as np
import pytest
import torch

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


def _params(eng):
    eng.sync()
    return eng.params.cpu().numpy().astype(np.float64)


def _twin_check(make, dist, steps, lr):
    from cae_tools_amd.dp import DataParallel, GradientHalfSteps
    (a, b) = (make(), make())
    dp = DataParallel(GradientHalfSteps(b), dist)
    dp.broadcast_parameters()
    n = a._keep[0][0].shape[0] if hasattr(a, "_keep") else a.n_samples
    for i in range(steps):
        a.train_step(0, None, 0, n, slot=i)
        slot = dp.train_step(0, None, 0, n)
        assert slot == i
    (la, lb) = (np.array(a.read_losses(0, steps)), np.array(b.read_losses(0, steps)))
    np.testing.assert_allclose(lb, la, rtol=1e-4, atol=1e-7)
    assert b.steps == a.steps == steps
    # the half-step narrows the fp64 gradient to fp32 before Adam; agreement to a small fraction of the lr-sized update
    assert np.abs(_params(a) - _params(b)).max() <= 0.05 * lr * steps


def test_unet_half_steps_match_the_fused_step(dist1):
    from test_unet_hip_parity import _engine
    from unet_helpers import UnetCase
    c = UnetCase("u_k4_b3")
    (x, t, m) = c.step_batch(0)

    def make():
        eng = _engine(c)
        eng.set_dataset(0, x, t, m)
        eng.n_samples = x.shape[0]
        return eng
    _twin_check(make, dist1, 3, c.meta["lr"])


def test_vae_half_steps_match_the_fused_step(dist1):
    from test_vae_hip_parity import _engine, _setup
    (fc, latent, B) = (12, 4, 3)
    (spec, enc, dec, x, t) = _setup((12, 12), (176, 176), fc, latent, B, seed=21)

    def make():
        eng = _engine(spec, enc, dec, fc, latent, B, lr=1e-3, seed=4)
        eng.set_dataset(0, x, t)
        eng.n_samples = B
        return eng
    _twin_check(make, dist1, 3, 1e-3)


def _linear(n, seed=3):
    from cae_tools_amd.linear_engine import LinearEngine
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((n, 1, 8, 8), generator=g)
    t = torch.rand((n, 1, 16, 16), generator=g)
    w = {"linear.1.weight": 0.05 * torch.randn((256, 64), generator=g), "linear.1.bias": 0.05 * torch.randn(256, generator=g)}

    def make():
        eng = LinearEngine((1, 8, 8), (1, 16, 16), max_batch=n, device="cuda:0")
        eng.load_state(w)
        eng.set_hyper(lr=1e-3, weight_decay=1e-5)
        eng.set_dataset(0, x, t)
        eng.n_samples = n
        return eng
    return make


def test_linear_half_steps_match_the_fused_step(dist1):
    _twin_check(_linear(6), dist1, 3, 1e-3)


def test_linear_two_shards_sum_to_the_full_batch_step():
    """what two ranks do, on one device: shard gradients weighted local/global, summed, one Adam step"""
    from cae_tools_amd.dp import shard_bounds
    n = 7
    make = _linear(n)
    (full, halves) = (make(), make())
    for i in range(2):
        full.train_step(0, None, 0, n, slot=i)
        total = torch.zeros(halves.n_param, device="cuda:0")
        for r in range(2):
            (lo, hi) = shard_bounds(n, 2, r)
            total += halves.forward_backward(0, None, lo, hi - lo, slot=4 + r, global_batch=n)
        halves.apply_gradients(total)
    assert np.abs(_params(full) - _params(halves)).max() <= 0.05 * 1e-3 * 2
    # per-shard MSEs weighted by shard size give the full-batch MSE of the same step
    ls = halves.read_losses(4, 2)
    (lo, hi) = shard_bounds(n, 2, 0)
    assert (ls[0] * (hi - lo) + ls[1] * (n - hi + lo)) / n == pytest.approx(full.read_losses(1, 1)[0], rel=1e-3)


def _grad_scale_check(eng, n, rows):
    """The gradient a rank hands to the all-reduce: forward_backward(shard, global_batch = n) must be exactly the shard's own
    gradient times shard / n (Adam's early steps are almost blind to a wrong scale, so the parameter comparisons above cannot
    see it).  Same shard, same weights, same dropout / reparameterisation step: only the scale differs."""
    step = getattr(eng, "steps", 0)
    g_own = eng.forward_backward(0, None, 0, rows, slot=10).clone()
    if hasattr(eng, "set_step"):
        eng.set_step(step)
    g_dp = eng.forward_backward(0, None, 0, rows, slot=11, global_batch=n).clone()
    want = g_own * (rows / n)
    scale = float(want.abs().max())
    assert scale > 0
    assert float((g_dp - want).abs().max()) <= 1e-5 * scale


def test_unet_shard_gradient_carries_local_over_global(dist1):
    from test_unet_hip_parity import _engine
    from unet_helpers import UnetCase
    c = UnetCase("u_k4_b3")
    (x, t, m) = c.step_batch(0)
    eng = _engine(c)
    eng.set_dataset(0, x, t, m)
    _grad_scale_check(eng, 7, x.shape[0])


def test_vae_shard_gradient_carries_local_over_global(dist1):
    from test_vae_hip_parity import _engine, _setup
    (fc, latent, B) = (12, 4, 3)
    (spec, enc, dec, x, t) = _setup((12, 12), (176, 176), fc, latent, B, seed=21)
    eng = _engine(spec, enc, dec, fc, latent, B, lr=1e-3, seed=4)
    eng.set_dataset(0, x, t)
    _grad_scale_check(eng, 8, B)


def test_linear_shard_gradients_sum_to_the_full_batch_gradient():
    """no batch statistics in this model: the two shards' scaled gradients add up to the full-batch gradient itself"""
    from cae_tools_amd.dp import shard_bounds
    n = 7
    eng = _linear(n)()
    full = eng.forward_backward(0, None, 0, n, slot=20).clone()
    total = torch.zeros_like(full)
    for r in range(2):
        (lo, hi) = shard_bounds(n, 2, r)
        total += eng.forward_backward(0, None, lo, hi - lo, slot=21 + r, global_batch=n)
    assert float((total - full).abs().max()) <= 1e-5 * float(full.abs().max())
    _grad_scale_check(eng, n, 3)
