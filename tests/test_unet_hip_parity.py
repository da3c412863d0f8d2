"""UNET HIP path (include/cae_unet.h) against the reference-generated vectors (tests/golden/unet_*.npz) and, for
train-mode dropout (whose masks are a hash both sides share), against the pinned CPU oracle."""
import numpy as np
import pytest
import torch

from unet_helpers import hip_relu_decisions, TRAIN_CASES, UNET_CASES, UnetCase, unet_oracle

pytestmark = pytest.mark.gpu


def _engine(case, max_batch=None, dropout=None, specialised=True, seed=0):
    from cae_tools_amd.unet_engine import UnetEngine
    m = case.meta
    eng = UnetEngine(m["spec"], m["fc"], m["latent"], max_batch or m["batch"], device="cuda:0", specialised=specialised)
    eng.load_state(case.state("init", "enc"), case.state("init", "dec"))
    eng.set_hyper(lr=m["lr"], weight_decay=m["weight_decay"], dropout_rate=m["dropout"] if dropout is None else dropout,
                  lambda_pearson=m["lambda_pearson"], seed=seed)
    return eng


def _grad_dict(eng, flat):
    flat = flat.cpu()
    return {n: flat[off:off + numel].view(shape) for n, (arena, off, numel, shape) in eng.tensors.items() if arena == 0}


def _feeds_batchnorm(key):
    """biases added right before a BatchNorm: their exact gradient is 0; both sides hold rounding noise (~1e-8)"""
    return key.endswith(".bias") and (("encoder_cnn." in key and int(key.split(".")[1]) % 4 == 0)
                                      or "encoder_lin.0." in key or "decoder_lin.0." in key)


def _close(got, want, rel, floor, msg=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    tol = rel * max(np.abs(want).max(), floor)
    err = np.abs(got - want).max()
    assert err <= tol, f"{msg}: max err {err:.3e} > {tol:.3e}"


@pytest.mark.parametrize("specialised", [True, False])
@pytest.mark.parametrize("name", UNET_CASES)
def test_eval_forward_and_losses(name, specialised):
    c = UnetCase(name)
    eng = _engine(c, specialised=specialised)
    y = eng.score(c.t("x0")).cpu().numpy()
    np.testing.assert_allclose(y, c.z["eval/y"], rtol=0, atol=5e-6)
    eng.set_dataset(0, c.t("x0"), c.t("t0"), None if c.meta["mask"] == "ones" else c.t("m0"))
    eng.eval_step(0, None, 0, c.z["x0"].shape[0], slot=3)
    (mse, pl) = eng.read_losses(3, 1)[0]
    np.testing.assert_allclose([mse, pl], c.z["eval/losses"], rtol=2e-5)


@pytest.mark.parametrize("specialised", [True, False])
@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_forward_backward(name, specialised):
    c = UnetCase(name)
    eng = _engine(c, specialised=specialised)
    (x, t, m) = c.step_batch(0)
    eng.set_dataset(0, x, t, None if c.meta["mask"] == "ones" else m)
    g = _grad_dict(eng, eng.forward_backward(0, None, 0, x.shape[0], slot=0))
    (mse, pl) = eng.read_losses(0, 1)[0]
    np.testing.assert_allclose([mse, pl], c.z["step_losses"][0], rtol=2e-5)
    for name_, gv in g.items():
        want = c.z["grad/" + name_]
        if _feeds_batchnorm(name_):
            assert np.abs(gv.numpy()).max() < 2e-5 and np.abs(want).max() < 2e-5, name_
            continue
        # fp32 gradients of a 10-layer net: agreement to ~1e-3 of the tensor's largest entry
        _close(gv.numpy(), want, 2e-3, 1e-6, name_)


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_adamw_steps(name):
    c = UnetCase(name)
    eng = _engine(c)
    nsteps = c.meta["nsteps"]
    losses = []
    for i in range(nsteps):
        (x, t, m) = c.step_batch(i)
        eng.set_dataset(0, x, t, None if c.meta["mask"] == "ones" else m)
        eng.train_step(0, None, 0, x.shape[0], slot=i)
        if i == 0:
            (enc, dec) = eng.export_state()
            for (pre, sd) in (("enc/", enc), ("dec/", dec)):
                for k, v in sd.items():
                    want = c.z["step1/" + pre + k]
                    if k.endswith("num_batches_tracked"):
                        assert int(v) == int(want)
                    elif _feeds_batchnorm(pre + k):
                        assert np.abs(v.numpy() - want).max() <= 2.1 * c.meta["lr"], k   # Adam turns noise into +-lr
                    else:
                        # the first AdamW step is lr * g / (|g| + eps) = lr * sign(g) wherever |g| >> eps = 1e-8: entries
                        # whose reference gradient is significant must agree closely, the rest may differ by up to 2*lr
                        diff = np.abs(v.numpy() - want)
                        assert diff.max() <= 2.1 * c.meta["lr"], k
                        if ("grad/" + pre + k) in c.z:
                            sig = np.abs(c.z["grad/" + pre + k]) > 1e-5
                            assert (diff[sig] <= 2e-5 + 1e-4 * np.abs(want).max()).all(), k
                        else:   # running statistics
                            assert diff.max() <= 1e-5 + 1e-4 * np.abs(want).max(), k
    got = eng.read_losses(0, nsteps)
    np.testing.assert_allclose(np.array(got), c.z["step_losses"], rtol=5e-3)


def test_train_dropout_matches_oracle_hash_masks():
    """dropout 0.25 in train mode: the engine and the oracle draw the same hash masks"""
    c = UnetCase("u_k4_b3")
    eng = _engine(c, dropout=0.25, seed=77)
    eng.set_step(5)
    o = unet_oracle(c, dropout_rate=0.25, seed=77)
    o.step_count = 5
    (x, t, m) = c.step_batch(0)
    eng.set_dataset(0, x, t, m)
    g = _grad_dict(eng, eng.forward_backward(0, None, 0, x.shape[0], slot=1))
    (mse, pl, _) = o.loss_and_grads(x, t, m)
    np.testing.assert_allclose(eng.read_losses(1, 1)[0], [mse, pl], rtol=2e-5)
    for k, want in o.grads().items():
        if _feeds_batchnorm(k):
            assert np.abs(g[k].numpy()).max() < 2e-5, k
            continue
        _close(g[k].numpy(), want.numpy(), 2e-3, 1e-6, k)


def test_permutation_and_partial_batches():
    """samples are gathered through the permutation; a smaller batch than max_batch reuses the workspace"""
    c = UnetCase("u_rect_b4")
    eng = _engine(c, max_batch=4)
    (x, t, m) = c.step_batch(0)
    eng.set_dataset(0, x, t, m)
    perm = eng.upload_perm([2, 0, 3, 1])
    eng.eval_step(0, perm, 1, 3, slot=0)       # samples 0, 3, 1
    o = unet_oracle(c)
    idx = [0, 3, 1]
    want = o.eval_losses(x[idx], t[idx], m[idx])
    np.testing.assert_allclose(eng.read_losses(0, 1)[0], want, rtol=2e-5)


def test_errors():
    from cae_tools_amd._lib import CaeError
    from cae_tools_amd.unet_engine import UnetEngine, UnetPlan
    c = UnetCase("u_k4_b3")
    spec = c.meta["spec"]
    with pytest.raises(CaeError, match="one layer per encoder layer"):
        UnetPlan({"input_layers": spec["input_layers"], "output_layers": spec["output_layers"][:2]}, 8, 4, 2)
    bad = {"input_layers": spec["input_layers"], "output_layers": [dict(l) for l in spec["output_layers"]]}
    bad["output_layers"][1]["input_dimensions"] = [16, 4, 4]
    with pytest.raises(CaeError, match="2 x out_channels"):
        UnetPlan(bad, 8, 4, 2)
    eng = _engine(c)
    with pytest.raises(CaeError, match="not set"):
        eng.eval_step(0, None, 0, 2)
    eng.set_dataset(0, c.t("x0"), c.t("t0"), c.t("m0"))
    with pytest.raises(CaeError, match="outside the data set"):
        eng.eval_step(0, None, 2, 3)
    with pytest.raises(CaeError, match="outside 1"):
        eng.eval_step(0, None, 0, 9)
    # a one-sample TRAINING batch fails in the reference (BatchNorm1d: "Expected more than 1 value per channel")
    with pytest.raises(ValueError, match="Expected more than 1 value per channel when training"):
        eng.train_step(0, None, 0, 1)
    eng.eval_step(0, None, 2, 1, slot=5)          # ... while a one-sample eval batch is fine
    o = unet_oracle(c)
    np.testing.assert_allclose(eng.read_losses(5, 1)[0], o.eval_losses(c.t("x0")[2:3], c.t("t0")[2:3], c.t("m0")[2:3]), rtol=2e-5)


def test_ragged_epoch_matches_oracle_batch_by_batch():
    """an epoch over 7 samples in batches of 3 (3 + 3 + 1 in eval mode; training drops nothing either: 3 + 3 + ... the
    reference's DataLoader has drop_last=False): per-batch loss pairs through run_batches"""
    c = UnetCase("u_rect_b4")
    (x, t, m) = (torch.cat([c.step_batch(i)[k] for i in range(3)]) for k in range(3))     # 4 + 3 + 4 = 11 samples
    (x, t, m) = (x[:7], t[:7], m[:7])
    eng = _engine(c, max_batch=3)
    eng.set_dataset(1, x, t, m)
    perm = eng.upload_perm([6, 2, 5, 0, 4, 1, 3])
    got = eng.run_batches(1, perm, 7, 3, train=False)
    o = unet_oracle(c)
    order = [6, 2, 5, 0, 4, 1, 3]
    want = [o.eval_losses(x[order[i:i + 3]], t[order[i:i + 3]], m[order[i:i + 3]]) for i in (0, 3, 6)]
    np.testing.assert_allclose(np.array(got), np.array(want), rtol=3e-5)


@pytest.mark.parametrize("size,chans,fc,latent,B", [(64, [32, 64, 96], 24, 6, 5), (128, [16, 32, 64, 72], 20, 5, 3)],
                         ids=["64px_32-64-96", "128px_16-32-64-72"])
def test_mfma_path_at_medium_size_against_oracle_and_generic_kernels(size, chans, fc, latent, B):
    """wide layers, odd batch, dropout on: the specialised kernels (image-end layers from an LDS patch with the weights in
    registers, wide layers from a patch with weight tiles, the im2col tile engine for what those do not take, the Linear
    kernels) against the CPU oracle and the generic kernels.  64 px, 32 / 64 / 96 channels: full tiles, a 64-row tile and a
    partly filled 128-row tile, maps 32 / 16 / 8 wide.  128 px, 16 / 32 / 64 / 72 channels: maps 64 ... 8 wide, a 16-channel
    image-end layer, channel counts the patch kernels take (32, 64) next to ones they leave to the tile engine (72)"""
    from cae_tools_amd.models.unet import Decoder, Encoder, unet_layer_spec
    from cae_tools_amd.unet_engine import UnetEngine
    from oracle import unet_oracle as uo
    spec = unet_layer_spec(3, 3, (size, size), chans)
    torch.manual_seed(123)
    enc = Encoder(spec.get_input_layers(), latent, fc)
    dec = Decoder(spec.get_output_layers(), latent, fc)
    g = torch.Generator().manual_seed(9)
    x = torch.rand((B, 3, size, size), generator=g)
    t = torch.rand((B, 3, size, size), generator=g)
    m = (torch.rand((B, 1, size, size), generator=g) < 0.85).float()
    to64 = lambda sd: {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    res = {}
    for specialised in (True, False):
        eng = UnetEngine(spec, fc, latent, B, device="cuda:0", specialised=specialised)
        eng.load_state(enc.state_dict(), dec.state_dict())
        eng.set_hyper(dropout_rate=0.1, seed=4)
        eng.set_step(2)
        eng.set_dataset(0, x, t, m)
        grads = _grad_dict(eng, eng.forward_backward(0, None, 0, B, slot=0))
        # the oracle follows this run's ReLU decisions where its own pre-activation is within 1e-5 of zero (one BatchNorm
        # output landing on the other side of zero switches that element's whole upstream gradient: a summation-order
        # effect worth percents of a layer's weight gradient, oracle/unet_oracle.py ReluAlign), and nowhere else
        decisions = hip_relu_decisions(eng, spec.save(), fc, latent, B)      # (before score() overwrites the activations)
        res[specialised] = (grads, eng.read_losses(0, 1)[0], eng.score(x).cpu().numpy())
        o = uo.UnetOracle(spec.save(), enc.state_dict(), dec.state_dict(), dropout_rate=0.1, seed=4)
        o.step_count = 2
        with uo.ReluAlign(decisions) as al32:
            (mse, pl, _) = o.loss_and_grads(x, t, m)
        want = o.grads()
        # the same in fp64: how far the fp32 oracle itself is from the exact gradient on this model (BatchNorm over FIVE
        # rows behind the first Linear layer makes the Linear section ill-conditioned: 2e-2 of a tensor's maximum is rounding)
        o64 = uo.UnetOracle(spec.save(), to64(enc.state_dict()), to64(dec.state_dict()), dropout_rate=0.1, seed=4)
        o64.step_count = 2
        with uo.ReluAlign(decisions) as al64:
            o64.loss_and_grads(x.double(), t.double(), m.double())
        want64 = o64.grads()
        assert sum(al64.followed.values()) <= 8, f"ReLU decisions followed: {al64.followed}"
        np.testing.assert_allclose(res[specialised][1], [mse, pl], rtol=3e-5)
        for k, w in want.items():
            if _feeds_batchnorm(k):
                continue
            # the generic kernels sum K in one fp32 chain per output (error ~ K * 6e-8 of the operands' scale), the MFMA
            # ones in chunks; 1.5e-2 of the tensor's largest gradient covers both on this random, unnormalised model -
            # or, where the fp32 oracle itself is further than that from the fp64 answer, 3x the oracle's own error
            (got, w64) = (res[specialised][0][k].numpy().astype(np.float64), want64[k].numpy())
            own = float(np.abs(w.numpy().astype(np.float64) - w64).max())
            rel = 1.5e-2
            tol = max(rel * max(float(np.abs(w64).max()), 1e-6), 3.0 * own)
            err = float(np.abs(got - w64).max())
            assert err <= tol, f"{k} (specialised={specialised}): |hip - fp64| {err:.3e} > {tol:.3e} (the fp32 oracle's own {own:.3e})"
    # (o: the last trip's fp32 oracle - like each engine it has seen one training forward, so its running statistics moved once)
    np.testing.assert_allclose(res[True][2], res[False][2], rtol=0, atol=2e-5)
    np.testing.assert_allclose(res[True][2], o.eval_forward(x).numpy(), rtol=0, atol=2e-5)


def test_benchmark_geometry_full_size_against_oracle():
    """BASELINE cfg3 layers (3x256x256, channels 32/64/128/256, fc128/latent32) at batch 5: every MFMA tile shape, the
    split-K paths, the thin-layer kernels and the 65536-wide Linear layers at their real sizes against the CPU oracle
    (not batch 2: BatchNorm1d over two samples maps every feature to +-1 and amplifies fp32 rounding without bound)"""
    from cae_tools_amd.models.unet import Decoder, Encoder, unet_layer_spec
    from cae_tools_amd.unet_engine import UnetEngine
    from oracle import unet_oracle as uo
    torch.set_num_threads(8)
    spec = unet_layer_spec(3, 3, (256, 256), [32, 64, 128, 256])
    (fc, latent, B) = (128, 32, 5)
    torch.manual_seed(11)
    enc = Encoder(spec.get_input_layers(), latent, fc)
    dec = Decoder(spec.get_output_layers(), latent, fc)
    g = torch.Generator().manual_seed(12)
    x = torch.rand((B, 3, 256, 256), generator=g)
    t = torch.rand((B, 3, 256, 256), generator=g)
    m = (torch.rand((B, 1, 256, 256), generator=g) < 0.9).float()
    o = uo.UnetOracle(spec.save(), enc.state_dict(), dec.state_dict(), dropout_rate=0.1, seed=21)
    (mse, pl, _) = o.loss_and_grads(x, t, m)
    want = o.grads()
    eng = UnetEngine(spec, fc, latent, B, device="cuda:0")
    eng.load_state(enc.state_dict(), dec.state_dict())
    eng.set_hyper(dropout_rate=0.1, seed=21)
    eng.set_dataset(0, x, t, m)
    got = _grad_dict(eng, eng.forward_backward(0, None, 0, B, slot=0))
    np.testing.assert_allclose(eng.read_losses(0, 1)[0], [mse, pl], rtol=3e-5)
    # an untrained 10-layer net at this size is chaotic in fp32: a ReLU or a max-pool argmax that flips under a 1-ulp
    # difference (atomic summation order changes from run to run) moves individual gradient entries by several per cent.
    # So: the whole tensor in the L2 sense, and a loose bound per entry.
    for k, w in want.items():
        if _feeds_batchnorm(k):
            continue
        (gv, wv) = (got[k].numpy().astype(np.float64), w.numpy().astype(np.float64))
        assert np.linalg.norm(gv - wv) <= 2e-2 * max(np.linalg.norm(wv), 1e-9), k
        _close(gv, wv, 0.15, 1e-6, k)
    np.testing.assert_allclose(eng.score(x).cpu().numpy(), o.eval_forward(x).numpy(), rtol=0, atol=3e-5)


def test_benchmark_geometry_at_the_stated_batch():
    """BASELINE cfg3 at its STATED batch, 32 (the test above runs the same layers at batch 5): grid sizes, split-K choices and
    the pressure on the BatchNorm-sum shards change with the batch.  Train-mode loss parts and a sample of gradients against
    the CPU oracle at batch 32; eval rows equal to the same rows scored at batch 5 (eval mode is per sample); every gradient
    finite; one AdamW step on the batch lowers its loss."""
    from cae_tools_amd.models.unet import Decoder, Encoder, unet_layer_spec
    from cae_tools_amd.unet_engine import UnetEngine
    from oracle import unet_oracle as uo
    torch.set_num_threads(min(16, torch.get_num_threads() if torch.get_num_threads() > 8 else 8))
    spec = unet_layer_spec(3, 3, (256, 256), [32, 64, 128, 256])
    (fc, latent, B) = (128, 32, 32)
    torch.manual_seed(11)
    enc = Encoder(spec.get_input_layers(), latent, fc)
    dec = Decoder(spec.get_output_layers(), latent, fc)
    g = torch.Generator().manual_seed(14)
    x = torch.rand((B, 3, 256, 256), generator=g)
    t = torch.rand((B, 3, 256, 256), generator=g)
    m = (torch.rand((B, 1, 256, 256), generator=g) < 0.9).float()
    eng = UnetEngine(spec, fc, latent, B, device="cuda:0")
    eng.load_state(enc.state_dict(), dec.state_dict())
    eng.set_hyper(dropout_rate=0.0, seed=21, lr=1e-4)
    eng.set_dataset(0, x, t, m)
    # eval: rows of the batch-32 call == the same rows scored five at a time
    y32 = eng.score(x).cpu().numpy()
    small = UnetEngine(spec, fc, latent, 5, device="cuda:0")
    small.load_state(enc.state_dict(), dec.state_dict())
    y5 = small.score(x[:5]).cpu().numpy()
    assert np.abs(y32[:5] - y5).max() <= 1e-6
    assert np.isfinite(y32).all()
    # train-mode step at batch 32 against the oracle (dropout off: the reference's arithmetic exactly)
    o = uo.UnetOracle(spec.save(), enc.state_dict(), dec.state_dict(), dropout_rate=0.0, seed=21)
    (mse, pl, _) = o.loss_and_grads(x, t, m)
    want = o.grads()
    got = _grad_dict(eng, eng.forward_backward(0, None, 0, B, slot=0))
    first = eng.read_losses(0, 1)[0]
    np.testing.assert_allclose(first, [mse, pl], rtol=3e-5)
    for k, w in want.items():
        gv = got[k].numpy().astype(np.float64)
        assert np.isfinite(gv).all(), k
        if _feeds_batchnorm(k):
            continue
        wv = w.numpy().astype(np.float64)
        assert np.linalg.norm(gv - wv) <= 2e-2 * max(np.linalg.norm(wv), 1e-9), k
    # one small optimiser step on this batch (first-order regime), then the same batch again: the loss went down
    eng.train_step(0, None, 0, B, slot=1)
    eng.forward_backward(0, None, 0, B, slot=2)
    after = eng.read_losses(2, 1)[0]
    assert after[0] + after[1] < first[0] + first[1]


def test_mfma_path_non_square_odd_channel_counts():
    """96x160 maps, channels 16 / 40 / 72 (every tile shape partly filled), 2 -> 5 channels, batch 3, per-channel mask"""
    from cae_tools_amd.models.unet import Decoder, Encoder, unet_layer_spec
    from cae_tools_amd.unet_engine import UnetEngine
    from oracle import unet_oracle as uo
    spec = unet_layer_spec(2, 5, (96, 160), [16, 40, 72])
    (fc, latent, B) = (20, 7, 3)
    torch.manual_seed(31)
    enc = Encoder(spec.get_input_layers(), latent, fc)
    dec = Decoder(spec.get_output_layers(), latent, fc)
    g = torch.Generator().manual_seed(32)
    x = torch.rand((B, 2, 96, 160), generator=g)
    t = torch.rand((B, 5, 96, 160), generator=g)
    m = (torch.rand((B, 5, 96, 160), generator=g) < 0.8).float()
    o = uo.UnetOracle(spec.save(), enc.state_dict(), dec.state_dict(), dropout_rate=0.0)
    (mse, pl, _) = o.loss_and_grads(x, t, m)
    eng = UnetEngine(spec, fc, latent, B, device="cuda:0")
    eng.load_state(enc.state_dict(), dec.state_dict())
    eng.set_hyper(dropout_rate=0.0)
    eng.set_dataset(0, x, t, m)
    got = _grad_dict(eng, eng.forward_backward(0, None, 0, B, slot=0))
    np.testing.assert_allclose(eng.read_losses(0, 1)[0], [mse, pl], rtol=3e-5)
    for k, w in o.grads().items():
        if _feeds_batchnorm(k):
            continue
        (gv, wv) = (got[k].numpy().astype(np.float64), w.numpy().astype(np.float64))
        # (a transposed-conv bias in front of a nearly constant attention gate has a gradient of ~1e-9: absolute floor)
        assert np.linalg.norm(gv - wv) <= 2e-2 * np.linalg.norm(wv) + 1e-7 * np.sqrt(wv.size), k
    np.testing.assert_allclose(eng.score(x).cpu().numpy(), o.eval_forward(x).numpy(), rtol=0, atol=3e-5)
