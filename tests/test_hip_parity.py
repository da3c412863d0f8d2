"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference-generated
golden vectors.  Needs a GPU.

Tolerances (fp32 path, stated per SURVEY.md §7):
  forward sigmoid outputs            1e-5 absolute (values in [0,1])
  loss                               1e-6 relative
  one-step gradients                 distance to the fp64 answer <= 3x the reference's own fp32
                                     distance to it on the same tensor + 1e-5 of the tensor's max
                                     (small-batch BatchNorm makes some cases ill-conditioned: the
                                     reference itself is off by 3e-3 relative on cfg1_b3)
  Adam steps                         see test_adam_step_by_step / test_adam_steps_free_running; conv
                                     biases that feed a BatchNorm only |delta| <= 2.5*lr*steps
                                     (their gradient is rounding noise in the reference - DESIGN.md)
"""
import numpy as np
import pytest
import torch

from helpers import (MODEL_CASES, GoldenCase, assert_close_as_reference, bn_bias_keys, oracle_model,
                     projections, subsample)

pytestmark = pytest.mark.gpu


def _engine(case, graph=True, max_batch=None, specialised=True):
    from cae_tools_amd.engine import HipEngine
    eng = HipEngine(case.spec, case.meta["fc"], case.meta["latent"],
                    max_batch=max_batch or max(8, case.meta["batch"]), graph=graph, specialised=specialised)
    eng.load_state(case.group("init/enc/"), case.group("init/dec/"))
    eng.set_hyper(lr=case.meta["lr"], weight_decay=case.meta["weight_decay"])
    return eng


def _oracle(case):
    from oracle import cae_oracle as orc
    return orc.OracleModel(case.spec, case.group("init/enc/"), case.group("init/dec/"),
                           lr=case.meta["lr"], weight_decay=case.meta["weight_decay"])


def _dataset(eng, case):
    x = torch.from_numpy(np.concatenate([case.x, case.x2])).cuda()
    t = torch.from_numpy(np.concatenate([case.t, case.t2])).cuda()
    eng.set_dataset(0, x, t)
    return x, t


@pytest.mark.parametrize("specialised", [False, True])
@pytest.mark.parametrize("name", MODEL_CASES)
def test_eval_forward(name, specialised):
    case = GoldenCase(name)
    eng = _engine(case, specialised=specialised)
    y = eng.score(torch.from_numpy(case.x).cuda()).cpu().numpy()
    ref = _oracle(case).eval_forward(torch.from_numpy(case.x)).numpy()
    assert np.abs(y - ref).max() <= 1e-5
    np.testing.assert_allclose(subsample(y), case["eval0/y_sub"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(projections(y, 77), case["eval0/y_proj"], rtol=0, atol=2e-3)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_module_level_forward(name):
    """Encoder.forward(x) -> z and Decoder.forward(z) -> y on their own (reference encoder.py:60-64, decoder.py:73-78), eval
    mode, through cae_encode / cae_decode: against the oracle's two module functions, the reference's stored latent, and the
    fused scoring path."""
    from oracle import cae_oracle as orc
    from cae_tools_amd.models.encoder import Encoder
    from cae_tools_amd.models.decoder import Decoder
    case = GoldenCase(name)
    eng = _engine(case)
    x = torch.from_numpy(case.x).cuda()
    z = eng.encode(x)
    om = _oracle(case)
    with torch.no_grad():
        z_ref = orc.encoder_forward(om.spec, om.enc, torch.from_numpy(case.x), train=False)
        y_ref = orc.decoder_forward(om.spec, om.dec, z_ref, train=False)
    tol_z = 2e-5 * max(1.0, float(z_ref.abs().max()))
    assert tuple(z.shape) == tuple(z_ref.shape)
    assert np.abs(z.cpu().numpy() - z_ref.numpy()).max() <= tol_z
    np.testing.assert_allclose(z.cpu().numpy(), case["eval0/latent"], rtol=0, atol=tol_z)
    # the decoder alone, from the ORACLE's latent (so that an encoder error cannot hide in it), and composed
    y = eng.decode(z_ref.cuda())
    assert np.abs(y.cpu().numpy() - y_ref.numpy()).max() <= 1e-5
    y2 = eng.decode(z).cpu().numpy()
    assert np.abs(y2 - eng.score(x).cpu().numpy()).max() <= 1e-6
    np.testing.assert_allclose(subsample(y2), case["eval0/y_sub"], rtol=0, atol=1e-5)
    # the module classes bind to the same entry points once attached; unattached they refuse (no CPU path)
    layers = case.spec
    from cae_tools_amd.models.model_sizer import ModelSpec
    ms = ModelSpec()
    ms.load(layers)
    enc = Encoder(ms.get_input_layers(), encoded_space_dim=case.meta["latent"], fc_size=case.meta["fc"])
    dec = Decoder(ms.get_output_layers(), encoded_space_dim=case.meta["latent"], fc_size=case.meta["fc"])
    with pytest.raises(RuntimeError):
        enc(x)
    with pytest.raises(RuntimeError):
        dec(z)
    enc.attach(eng)
    dec.attach(eng)
    assert torch.equal(enc(x), z)
    assert np.abs(dec(enc.forward(x)).cpu().numpy() - y2).max() == 0.0
    # partial batches larger than the engine's max_batch are walked in chunks
    xx = torch.cat([x, x, x])[: eng.max_batch + 1]
    assert torch.equal(eng.encode(xx)[: x.shape[0]], z)


@pytest.mark.parametrize("graph,specialised", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("name", MODEL_CASES)
def test_train_forward_backward(name, graph, specialised):
    case = GoldenCase(name)
    eng = _engine(case, graph=graph, specialised=specialised)
    _dataset(eng, case)
    b = case.meta["batch"]
    slot = eng.forward_backward(0, None, 0, b, b)
    loss = eng._read_losses(slot, 1)[0]
    assert abs(loss - float(case["train0/loss"])) <= 1e-6 * abs(float(case["train0/loss"])) + 1e-9

    # raw conv outputs of every layer against the oracle's trace
    orc = _oracle(case)
    trace = {}
    orc.loss_and_grads(torch.from_numpy(case.x), torch.from_numpy(case.t), trace=trace)
    n_enc = len(case.spec["input_layers"])
    n_dec = len(case.spec["output_layers"])
    for i in range(n_enc + n_dec - 1):
        ref = trace[f"enc_conv{i}" if i < n_enc else f"dec_conv{i - n_enc}"].numpy()
        got = eng.debug_read("act", i, ref.size).reshape(ref.shape)
        tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= tol, f"layer {i}"
    got = eng.debug_read("latent", 0, trace["latent"].numel()).reshape(trace["latent"].shape)
    assert np.abs(got - trace["latent"].numpy()).max() <= 2e-5 * max(1.0, float(trace["latent"].abs().max()))

    eng.sync()
    noisy = bn_bias_keys(case.spec)
    m64 = oracle_model(case, "float64")
    m64.loss_and_grads(torch.from_numpy(case.x).double(), torch.from_numpy(case.t).double())
    exact = m64.grads()
    for k, ref in case.group("train0/grad/").items():
        g = eng.grad_view(k).cpu().numpy()
        if k in noisy:
            # exactly zero in exact arithmetic; the HIP path computes it as zero
            assert np.abs(g).max() <= 1e-6 + 1e-4 * float(np.abs(ref).max())
            continue
        assert_close_as_reference(g, ref, exact[k].numpy(), k)
    # running statistics after one train-mode forward
    for k, ref in case.group("train0/buf/").items():
        if k.endswith("num_batches_tracked"):
            assert eng.num_batches_tracked == int(ref)
            continue
        np.testing.assert_allclose(eng.view(k).cpu().numpy(), ref, rtol=1e-5, atol=1e-6, err_msg=k)


def _oracle_moments(orc):
    out = {}
    for side, group in (("enc/", orc.enc), ("dec/", orc.dec)):
        for k, p in group.items():
            st = orc.optim.state.get(p) if torch.is_tensor(p) and p.requires_grad else None
            if st:
                out[side + k] = (st["exp_avg"], st["exp_avg_sq"])
    return out


@pytest.mark.parametrize("name", MODEL_CASES)
def test_adam_step_by_step(name):
    """Every optimiser step checked on its own: before step s the engine is given the oracle's
    weights, running statistics and Adam moments, both take the step, and the UPDATE is compared.
    (Free-running fp32 trajectories separate chaotically - one ReLU flipping under a 1-ulp weight
    difference changes gradients by 1e-3 - so the free-running check below is loose.)
    Tolerance: 99.9 % of the elements of every tensor move within 2 % of lr of the oracle's move,
    all of them within 25 % of lr: Adam's m/(sqrt(v)+eps) turns a 1e-6 gradient error into up to
    that on the few elements whose gradient is itself ~1e-6 of the tensor's largest."""
    case = GoldenCase(name)
    lr = case.meta["lr"]
    eng = _engine(case)
    _dataset(eng, case)
    b, b2 = case.meta["batch"], case.x2.shape[0]
    orc = _oracle(case)
    batches = [(torch.from_numpy(case.x), torch.from_numpy(case.t)), (torch.from_numpy(case.x2), torch.from_numpy(case.t2))]
    noisy = bn_bias_keys(case.spec)
    for s in range(case.meta["nsteps"]):
        before = orc.state()
        enc = {k[4:]: v for k, v in before.items() if k.startswith("enc/")}
        dec = {k[4:]: v for k, v in before.items() if k.startswith("dec/")}
        eng.load_state(enc, dec)
        eng.load_optimizer_state(_oracle_moments(orc), s)
        loss_ref = orc.train_step(*batches[s % 2])
        loss = eng.train_step(0, None, 0 if s % 2 == 0 else b, b if s % 2 == 0 else b2)
        assert abs(loss - loss_ref) <= 2e-6 * abs(loss_ref) + 1e-9
        after = orc.state()
        (e2, d2) = eng.export_state()
        for side, sd in (("enc/", e2), ("dec/", d2)):
            for k, v in sd.items():
                key = side + k
                if k.endswith("num_batches_tracked"):
                    continue
                ref = after[key].numpy()
                if "running_" in k:
                    np.testing.assert_allclose(v.numpy(), ref, rtol=2e-5, atol=1e-6, err_msg=key)
                    continue
                if key in noisy:
                    assert np.abs(v.numpy() - ref).max() <= 2.2 * lr, key
                    continue
                d = np.abs((v.numpy() - before[key].numpy()) - (ref - before[key].numpy())).reshape(-1)
                assert d.max() <= 0.25 * lr, f"step {s} {key}: {d.max():.3e}"
                if d.size >= 2000:
                    assert np.quantile(d, 0.999) <= 0.02 * lr, f"step {s} {key}: q99.9 {np.quantile(d, 0.999):.3e}"
                else:
                    assert np.quantile(d, 0.9) <= 0.02 * lr, f"step {s} {key}: q90 {np.quantile(d, 0.9):.3e}"


@pytest.mark.parametrize("name", MODEL_CASES)
def test_adam_step_no_further_from_fp64_than_the_reference(name):
    """The step-by-step check above bounds the update in units of lr; this one says what those units hide.  Before every step
    the engine, the fp32 oracle (the reference's arithmetic) and an fp64 oracle start from the same weights, running statistics
    and Adam moments; all three take the step.  Per tensor, the engine's update may be no further (max norm) from the fp64
    update than 3x the fp32 reference's own update is, plus 1e-3 of lr: where Adam's m / (sqrt(v) + eps) amplifies a gradient
    rounding error, it amplifies the reference's just as much."""
    from oracle import cae_oracle as orc_mod
    case = GoldenCase(name)
    lr = case.meta["lr"]
    eng = _engine(case)
    _dataset(eng, case)
    b, b2 = case.meta["batch"], case.x2.shape[0]
    orc = _oracle(case)
    batches = [(torch.from_numpy(case.x), torch.from_numpy(case.t)), (torch.from_numpy(case.x2), torch.from_numpy(case.t2))]
    noisy = bn_bias_keys(case.spec)
    worst = 0.0
    for s in range(case.meta["nsteps"]):
        before = orc.state()
        moments = _oracle_moments(orc)
        enc = {k[4:]: v for k, v in before.items() if k.startswith("enc/")}
        dec = {k[4:]: v for k, v in before.items() if k.startswith("dec/")}
        eng.load_state(enc, dec)
        eng.load_optimizer_state(moments, s)
        # the same state in fp64
        to64 = lambda sd: {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        o64 = orc_mod.OracleModel(case.spec, to64(enc), to64(dec), lr=lr, weight_decay=case.meta["weight_decay"])
        for side, group in (("enc/", o64.enc), ("dec/", o64.dec)):
            for k, p64 in group.items():
                if side + k in moments:
                    (m, v) = moments[side + k]
                    o64.optim.state[p64] = {"step": torch.tensor(float(s)), "exp_avg": m.double().clone(), "exp_avg_sq": v.double().clone()}
        (xb, tb) = batches[s % 2]
        orc.train_step(xb, tb)
        o64.train_step(xb.double(), tb.double())
        eng.train_step(0, None, 0 if s % 2 == 0 else b, b if s % 2 == 0 else b2)
        (after32, after64) = (orc.state(), o64.state())
        (e2, d2) = eng.export_state()
        for side, sd in (("enc/", e2), ("dec/", d2)):
            for k, v in sd.items():
                key = side + k
                if k.endswith("num_batches_tracked") or "running_" in k or key in noisy:
                    continue
                b0 = before[key].numpy().astype(np.float64)
                d64 = after64[key].numpy() - b0
                d32 = after32[key].numpy().astype(np.float64) - b0
                dh = v.numpy().astype(np.float64) - b0
                (err_ref, err_hip) = (float(np.abs(d32 - d64).max()), float(np.abs(dh - d64).max()))
                worst = max(worst, err_hip / (3.0 * err_ref + 1e-3 * lr))
                assert err_hip <= 3.0 * err_ref + 1e-3 * lr, \
                    f"step {s} {key}: |hip - fp64| = {err_hip:.3e}, the reference's own {err_ref:.3e} (lr {lr:g})"
    print(f"{name}: worst ratio to the bound {worst:.2f}")


@pytest.mark.parametrize("name", MODEL_CASES)
def test_adam_steps_free_running(name):
    """4 steps from the initial state without re-synchronisation, against the reference's stored
    trajectory (golden steps/*): losses tight, 99 % of every parameter tensor within a tenth of
    Adam's displacement bound (lr per step) and all within it, eval-mode output 2e-3."""
    case = GoldenCase(name)
    eng = _engine(case)
    _dataset(eng, case)
    b, b2 = case.meta["batch"], case.x2.shape[0]
    losses = []
    for s in range(case.meta["nsteps"]):
        losses.append(eng.train_step(0, None, 0 if s % 2 == 0 else b, b if s % 2 == 0 else b2))
    np.testing.assert_allclose(losses, case["steps/loss"], rtol=2e-5, atol=1e-7)
    (enc, dec) = eng.export_state()
    noisy = bn_bias_keys(case.spec)
    bound = case.meta["lr"] * case.meta["nsteps"]
    for side, sd in (("enc/", enc), ("dec/", dec)):
        for k, v in sd.items():
            ref = case["steps/" + side + k]
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(ref)
            elif "running_mean" in k:
                # carries the reference's noise-driven bias walk (bias is part of the batch mean)
                np.testing.assert_allclose(v.numpy(), ref, rtol=1e-3, atol=0.5 * bound, err_msg=side + k)
            elif "running_var" in k:
                np.testing.assert_allclose(v.numpy(), ref, rtol=2e-3, atol=1e-5, err_msg=side + k)
            elif side + k in noisy:
                assert np.abs(v.numpy() - ref).max() <= 2.5 * bound, k
            else:
                d = np.abs(v.numpy() - ref).reshape(-1)
                assert d.max() <= bound and np.quantile(d, 0.99) <= 0.1 * bound, side + k
    y = eng.score(torch.from_numpy(case.x).cuda()).cpu().numpy()
    np.testing.assert_allclose(subsample(y), case["steps/eval_y_sub"], rtol=0, atol=2e-3)


def test_epoch_with_permutation_and_partial_batch():
    """run_batches: gathered samples through a permutation, last batch partial (drop_last=False)"""
    case = GoldenCase("cfg1_b3")
    eng = _engine(case)
    x, t = _dataset(eng, case)
    n = x.shape[0]  # 5 samples
    perm = np.array([3, 0, 4, 1, 2], dtype=np.int32)
    pd = eng.upload_perm(perm)
    got = eng.run_batches(0, pd, n, 2, train=True)
    orc = _oracle(case)
    ref = []
    xc, tc = x.cpu(), t.cpu()
    for s in range(0, n, 2):
        idx = perm[s:s + 2]
        ref.append(orc.train_step(xc[idx], tc[idx]))
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-7)
    # eval pass (test epoch) with the updated weights / running stats
    got_eval = eng.run_batches(0, pd, n, 2, train=False)
    ref_eval = [orc.eval_loss(xc[perm[s:s + 2]], tc[perm[s:s + 2]]) for s in range(0, n, 2)]
    np.testing.assert_allclose(got_eval, ref_eval, rtol=1e-4, atol=1e-7)


def test_loader_kernels_bit_exact():
    import json, os
    from helpers import GOLDEN
    from cae_tools_amd.engine import scan_f32, normalise_pack, denormalise_f64
    npz = np.load(os.path.join(GOLDEN, "ds_dataset.npz"), allow_pickle=False)
    meta = json.load(open(os.path.join(GOLDEN, "ds_dataset.json")))
    (mins, maxs, omin, omax) = meta["normalisation_parameters"]
    names = meta["input_names"]
    n = npz["lowres"].shape[0]
    ctot = sum(npz[k].shape[1] for k in names)
    dst = torch.zeros((n, ctot) + npz["lowres"].shape[2:], dtype=torch.float32, device="cuda")
    off = 0
    for k in names:
        src = torch.from_numpy(npz[k]).cuda()
        (nans, lo, hi) = scan_f32(src)
        assert (nans, lo, hi) == (0, mins[k], maxs[k])
        normalise_pack(src, dst, off, lo, hi)
        off += src.shape[1]
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dst.cpu().numpy(), npz["norm_in"])
    hires = torch.from_numpy(npz["hires"]).cuda()
    out = torch.zeros_like(hires)
    normalise_pack(hires, out, 0, omin, omax)
    np.testing.assert_array_equal(out.cpu().numpy(), npz["norm_out"])
    raw = torch.zeros_like(dst)
    off = 0
    for k in names:
        src = torch.from_numpy(npz[k]).cuda()
        normalise_pack(src, raw, off, 0.0, 0.0, enable=False)
        off += src.shape[1]
    np.testing.assert_array_equal(raw.cpu().numpy()[2], npz["raw_in2"])
    # NaN counting
    bad = npz["lowres"].copy(); bad[0, 0, 0, 0] = np.nan; bad[3, 0, 1, 1] = np.nan
    (nans, lo, hi) = scan_f32(torch.from_numpy(bad).cuda())
    assert nans == 2 and lo == float(np.nanmin(bad)) and hi == float(np.nanmax(bad))
    # denormalise: float32 scores -> float64, bit exact with numpy's fp64 expression
    y32 = npz["denorm_in"].astype(np.float32)
    got = denormalise_f64(torch.from_numpy(y32).cuda(), omin, omax).cpu().numpy()
    np.testing.assert_array_equal(got, omin + (y32.astype(np.float64) * (omax - omin)))


def test_batch_stacking_in_the_normalise_pass_bit_exact():
    """DataLoader(shuffle=True) + default collate of the reference (conv_ae_model.py:291-292, 315-325) = the normalisation
    kernel writing sample i to row inverse[i] of the frozen order (cae_normalise_pack_rows + cae_invert_permutation): the
    rows of DSDataset.device_batches(order) are, bit for bit, the reference DSDataset's normalised samples in that order
    (tests/golden/ds_dataset.npz), for a multi-variable input (channel offsets) and the single-variable target."""
    import json, os
    from helpers import GOLDEN
    from cae_tools_amd.engine import normalise_pack, inverse_permutation, CaeError
    from cae_tools_amd.models.ds_dataset import DSDataset
    from cae_tools_amd.data.arrays import DataArray, Dataset
    npz = np.load(os.path.join(GOLDEN, "ds_dataset.npz"), allow_pickle=False)
    meta = json.load(open(os.path.join(GOLDEN, "ds_dataset.json")))
    names = meta["input_names"]
    n = npz["lowres"].shape[0]
    order = np.random.default_rng(4).permutation(n)
    inv = inverse_permutation(order, torch.device("cuda", torch.cuda.current_device()))
    assert inv.dtype == torch.int32 and np.array_equal(inv.cpu().numpy()[order], np.arange(n))
    ds = Dataset()
    for k in names:
        ds[k] = DataArray(npz[k], dims=("n", "c_" + k, "y", "x"))
    ds["hires"] = DataArray(npz["hires"], dims=("n", "c", "Y", "X"))
    d = DSDataset(ds, names, "hires")
    assert d.get_normalisation_parameters() == meta["normalisation_parameters"]
    (x, t) = d.device_batches(order)
    np.testing.assert_array_equal(x.cpu().numpy(), npz["norm_in"][order])
    np.testing.assert_array_equal(t.cpu().numpy(), npz["norm_out"][order])
    # the un-permuted arrays are unchanged by it, and the raw (un-normalised) target is the uploaded variable itself
    np.testing.assert_array_equal(d.device_inputs().cpu().numpy(), npz["norm_in"])
    d.set_normalise_output(False)
    np.testing.assert_array_equal(d.device_outputs().cpu().numpy(), npz["hires"])
    # a table that is not a permutation is refused on the host
    with pytest.raises(CaeError):
        inverse_permutation(np.array([0, 0, 1]), torch.device("cuda"))
    with pytest.raises(CaeError):
        normalise_pack(torch.zeros((3, 1, 2, 2), device="cuda"), torch.zeros((3, 1, 2, 2), device="cuda"), 0, 0.0, 1.0,
                       dst_rows=torch.zeros(2, dtype=torch.int32, device="cuda"))


def test_multi_step_graph_equals_single_steps():
    """cae_train_steps / cae_eval_steps (64 steps per captured graph) walk the cursor exactly like 64
    single-step launches: same per-batch losses, same weights afterwards."""
    from cae_tools_amd.engine import HipEngine
    case = GoldenCase("handspec_b4")
    rng = np.random.default_rng(5)
    n, bs = 2 * 66 + 1, 2                     # 66 full batches (one 64-step graph + 2 singles) + a partial one
    (ic, ih, iw) = case.spec["input_layers"][0]["input_dimensions"]
    (oc, oh, ow) = case.spec["output_layers"][-1]["output_dimensions"]
    x = torch.from_numpy(rng.random((n, ic, ih, iw), dtype=np.float32)).cuda()
    t = torch.from_numpy(rng.random((n, oc, oh, ow), dtype=np.float32)).cuda()
    perm = rng.permutation(n)
    results = []
    for steps_per_graph in (64, 10 ** 9):
        eng = _engine(case, max_batch=bs)
        eng.STEPS_PER_GRAPH = steps_per_graph
        eng.set_dataset(0, x, t)
        pd = eng.upload_perm(perm)
        tr = eng.run_batches(0, pd, n, bs, train=True)
        ev = eng.run_batches(0, pd, n, bs, train=False)
        eng.sync()
        results.append((np.array(tr), np.array(ev), eng.params.cpu().numpy().copy(), eng.buffers.cpu().numpy().copy()))
        assert len(tr) == 67 and eng.num_batches_tracked == 67
    (a, b) = results
    np.testing.assert_allclose(a[0], b[0], rtol=1e-6)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-6)
    # 67 Adam steps apart the two runs differ only by the arrival order of the fp64 atomics
    np.testing.assert_allclose(a[2], b[2], rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(a[3], b[3], rtol=1e-5, atol=5e-6)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_fused_optimiser_launch_sees_the_same_gradients(name):
    """The training step's last launch (k_adam) also computes the first encoder layer's weight gradient and that layer's
    BatchNorm parameter gradients (kernels_generic.h AdamConv0); forward_backward() keeps them as separate launches.  After
    ONE step from zero moments exp_avg = (1 - beta1) (g + wd w): every parameter's g recovered from the fused step must be
    the gradient the separate launches produce.  (This is the test that catches a workgroup of the fused launch reading a
    parameter another workgroup of the same launch has already updated: 1e-3 relative on gamma-dependent entries.)"""
    case = GoldenCase(name)
    n = case.x.shape[0] + case.x2.shape[0]
    a = _engine(case, max_batch=max(8, n))
    _dataset(a, case)
    a.forward_backward(0, None, 0, n, n)
    a.sync()
    g_sep = a.grads.cpu().numpy().astype(np.float64)
    w0 = a.params.cpu().numpy().astype(np.float64)
    b = _engine(case, max_batch=max(8, n))
    _dataset(b, case)
    b.train_step(0, None, 0, n)
    b.sync()
    g_fused = b.exp_avg.cpu().numpy().astype(np.float64) / 0.1 - case.meta["weight_decay"] * w0
    # exp_avg is fp32: its rounding (6e-8 relative to g + wd w) is what the comparison can resolve
    scale = np.abs(g_sep) + case.meta["weight_decay"] * np.abs(w0)
    assert float(np.max(np.abs(g_fused - g_sep) - 1e-6 * scale)) <= 1e-7 * float(np.abs(g_sep).max())
