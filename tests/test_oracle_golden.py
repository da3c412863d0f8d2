"""Pin oracle/cae_oracle.py to the vectors produced by the reference's own modules
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, MODEL_CASES, GoldenCase, projections, subsample
from oracle import cae_oracle as orc

torch.set_num_threads(1)


def _model(case, prefix="init/"):
    return orc.OracleModel(case.spec, case.group(prefix + "enc/"), case.group(prefix + "dec/"),
                           lr=case.meta["lr"], weight_decay=case.meta["weight_decay"])


@pytest.mark.parametrize("name", MODEL_CASES)
def test_eval_forward(name):
    case = GoldenCase(name)
    m = _model(case)
    trace = {}
    with torch.no_grad():
        y = m.forward(torch.from_numpy(case.x), train=False, trace=trace)
    np.testing.assert_allclose(trace["latent"].numpy(), case["eval0/latent"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(subsample(y.numpy()), case["eval0/y_sub"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(projections(y.numpy(), 77), case["eval0/y_proj"], rtol=0, atol=1e-4)
    assert abs(m.eval_loss(torch.from_numpy(case.x), torch.from_numpy(case.t)) - float(case["eval0/loss"])) < 1e-7


@pytest.mark.parametrize("name", MODEL_CASES)
def test_train_forward_backward(name):
    case = GoldenCase(name)
    m = _model(case)
    loss, y = m.loss_and_grads(torch.from_numpy(case.x), torch.from_numpy(case.t))
    assert abs(loss - float(case["train0/loss"])) < 1e-7
    np.testing.assert_allclose(subsample(y.numpy()), case["train0/y_sub"], rtol=0, atol=1e-6)
    if "train0/y_full" in case.keys():
        np.testing.assert_allclose(y.numpy(), case["train0/y_full"], rtol=0, atol=1e-6)
    for k, g in m.grads().items():
        ref = case["train0/grad/" + k]
        scale = max(1e-30, float(np.abs(ref).max()))
        # same ATen kernels, one thread: expect (near) bit equality; allow 1e-5 of the tensor's max
        assert float(np.abs(g.numpy() - ref).max()) <= 1e-5 * scale + 1e-9, k
    st = m.state()
    for k in case.keys():
        if k.startswith("train0/buf/"):
            np.testing.assert_allclose(st[k[len("train0/buf/"):]].numpy(), case[k], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_adam_steps(name):
    case = GoldenCase(name)
    m = _model(case)
    batches = [(torch.from_numpy(case.x), torch.from_numpy(case.t)),
               (torch.from_numpy(case.x2), torch.from_numpy(case.t2))]
    losses = [m.train_step(*batches[s % 2]) for s in range(case.meta["nsteps"])]
    np.testing.assert_allclose(losses, case["steps/loss"], rtol=1e-5, atol=1e-7)
    st = m.state()
    from helpers import bn_bias_keys
    noisy = bn_bias_keys(case.spec)
    for k, v in st.items():
        ref = case["steps/" + k]
        if k in noisy:
            assert float(np.abs(v.numpy() - ref).max()) <= 2.5 * case.meta["lr"] * case.meta["nsteps"], k
        else:
            np.testing.assert_allclose(v.numpy(), ref, rtol=2e-4, atol=2e-6, err_msg=k)
    y = m.eval_forward(batches[0][0])
    np.testing.assert_allclose(subsample(y.numpy()), case["steps/eval_y_sub"], rtol=0, atol=2e-5)
    assert abs(m.eval_loss(*batches[0]) - float(case["steps/eval_loss"])) < 1e-5


def test_loader_arithmetic():
    npz = np.load(os.path.join(GOLDEN, "ds_dataset.npz"), allow_pickle=False)
    with open(os.path.join(GOLDEN, "ds_dataset.json")) as f:
        meta = json.load(f)
    (mins, maxs, omin, omax) = meta["normalisation_parameters"]
    names = meta["input_names"]
    for n in names:
        (nans, lo, hi) = orc.scan_variable(npz[n])
        assert nans == 0 and lo == mins[n] and hi == maxs[n]
    (nans, lo, hi) = orc.scan_variable(npz["hires"])
    assert (lo, hi) == (omin, omax)
    packed = orc.pack_inputs([npz[n] for n in names], [mins[n] for n in names], [maxs[n] for n in names])
    assert packed.dtype == np.float32
    np.testing.assert_array_equal(packed, npz["norm_in"])            # bit exact
    np.testing.assert_array_equal(orc.normalise_variable(npz["hires"], omin, omax), npz["norm_out"])
    np.testing.assert_array_equal(orc.denormalise_output(npz["denorm_in"], omin, omax), npz["denorm_out"])
    raw = orc.pack_inputs([npz[n] for n in names], None or [0] * 3, [0] * 3, normalise=False)
    np.testing.assert_array_equal(raw[2], npz["raw_in2"])
    assert np.all(npz["mask"] == 1.0) and npz["mask"].shape == npz["norm_in"].shape


def test_relu_decision_hooks_of_the_oracle():
    """oracle/cae_oracle.py _relu: relu_inputs() leaves no trace on the model; an all-zero relu_fix changes nothing; a +1 / -1
    entry changes the DERIVATIVE at that position only (the value of the forward pass, hence the loss, stays)."""
    import torch
    from helpers import oracle_model
    case = GoldenCase("cfg2_b4")
    (x, t) = (torch.from_numpy(case.x), torch.from_numpy(case.t))
    o = oracle_model(case)
    before = {k: v.clone() for k, v in o.state().items()}
    z = o.relu_inputs(x)
    assert set(z) == {"enc_conv0", "enc_conv1", "enc_fc0", "dec_fc0"} | {f"dec_conv{i}" for i in range(len(case.spec["output_layers"]) - 1)}
    for k, v in o.state().items():
        assert torch.equal(v, before[k]), k
    (loss0, _) = oracle_model(case).loss_and_grads(x, t)
    a = oracle_model(case)
    (loss1, _) = a.loss_and_grads(x, t, relu_fix={k: torch.zeros_like(v) for k, v in z.items()})
    ref = oracle_model(case)
    ref.loss_and_grads(x, t)
    assert loss1 == loss0 and all(torch.equal(g, ref.grads()[k]) for k, g in a.grads().items())
    # block one passing position and pass one blocked position of the first decoder layer
    zz = z["dec_conv0"]
    d = torch.zeros_like(zz)
    pos = tuple(int(i) for i in (zz > 0.1).nonzero()[0])
    neg = tuple(int(i) for i in (zz < -0.1).nonzero()[0])
    d[pos] = -1.0
    d[neg] = 1.0
    b = oracle_model(case)
    (loss2, _) = b.loss_and_grads(x, t, relu_fix={"dec_conv0": d})
    assert loss2 == loss0
    changed = [k for k, g in b.grads().items() if not torch.equal(g, ref.grads()[k])]
    assert "dec/decoder_conv.0.weight" in changed and "enc/encoder_cnn.0.weight" in changed
    assert "dec/decoder_conv.3.weight" not in changed      # downstream of the edited ReLU: untouched
