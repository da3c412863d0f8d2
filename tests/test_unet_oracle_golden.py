"""The UNET oracle (oracle/unet_oracle.py, a restatement from the text of unet.py) against vectors produced by the
reference's own Encoder / Decoder / ChannelAttention / loss definitions (tests/golden/make_golden_unet.py)."""
import numpy as np
import pytest
import torch

from unet_helpers import TRAIN_CASES, UNET_CASES, UnetCase, unet_oracle

torch.set_num_threads(1)


@pytest.mark.parametrize("name", UNET_CASES)
def test_eval_forward_and_losses(name):
    c = UnetCase(name)
    o = unet_oracle(c)
    y = o.eval_forward(c.t("x0"))
    np.testing.assert_allclose(y.numpy(), c.z["eval/y"], rtol=0, atol=2e-7)
    (mse, pl) = o.eval_losses(c.t("x0"), c.t("t0"), c.t("m0"))
    np.testing.assert_allclose([mse, pl], c.z["eval/losses"], rtol=1e-6)
    from oracle import unet_oracle as uo
    np.testing.assert_allclose(uo.pearson_corr(y, c.t("t0"), c.t("m0")).numpy(), c.z["eval/pearson"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_training_steps(name):
    c = UnetCase(name)
    o = unet_oracle(c)
    for i in range(c.meta["nsteps"]):
        (x, t, m) = c.step_batch(i)
        if i == 0:
            (mse, pl, y) = o.loss_and_grads(x, t, m)
            np.testing.assert_allclose(y.numpy(), c.z["train/y"], rtol=0, atol=2e-7)
            for k, g in o.grads().items():
                want = c.z["grad/" + k]
                np.testing.assert_allclose(g.numpy(), want, rtol=1e-4, atol=1e-6 * max(1e-3, np.abs(want).max()), err_msg=k)
            o.optim.step()
            o.step_count += 1
            for k, v in o.state().items():
                np.testing.assert_allclose(v.numpy(), c.z["step1/" + k], rtol=1e-5, atol=1e-6, err_msg=k)
        else:
            (mse, pl) = o.train_step(x, t, m)
        np.testing.assert_allclose([mse, pl], c.z["step_losses"][i], rtol=2e-5)
    for k, v in o.state().items():
        want = c.z["steps/" + k]
        # 3 AdamW steps of lr 1e-3: a sign flip of a ~0 gradient moves a weight by up to 2*lr per step
        np.testing.assert_allclose(v.numpy(), want, rtol=1e-4, atol=1e-5, err_msg=k)


def test_dropout_hash_properties():
    from oracle import unet_oracle as uo
    keep = uo.dropout_keep(seed=3, step=7, site=uo.SITE_DEC_CONV + 1, shape=(4, 8, 32, 32), p=0.1)
    assert keep.dtype == bool and abs(keep.mean() - 0.9) < 0.01
    again = uo.dropout_keep(seed=3, step=7, site=uo.SITE_DEC_CONV + 1, shape=(4, 8, 32, 32), p=0.1)
    assert (keep == again).all()
    other = uo.dropout_keep(seed=3, step=8, site=uo.SITE_DEC_CONV + 1, shape=(4, 8, 32, 32), p=0.1)
    assert (keep != other).mean() > 0.1
    assert uo.dropout_keep(1, 1, 1, (1000,), 0.0).all()
    # known answers of the hash (pin for the device implementation)
    assert [int(v) for v in uo._pcg(np.array([0, 1, 2, 0xFFFFFFFF], dtype=np.uint32))] == \
        [129708002, 2831084092, 2055130248, 3861530882]
    assert int(uo.dropout_key(3, 7, 201)) == 3624308049
    assert uo.dropout_keep(3, 7, 201, (2, 3, 5), 0.5).astype(int).reshape(-1).tolist() == \
        [1, 1, 0, 1, 1, 1, 0, 0, 1, 0, 1, 0, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0, 1]


def test_relu_align_follows_foreign_decisions_only_near_zero():
    """oracle.unet_oracle.ReluAlign (what lets a parity test take the ReLU decisions of the implementation under test): with the
    oracle's own decisions nothing changes; a foreign decision is followed where the pre-activation is within tol of zero and
    ignored anywhere else"""
    from oracle import unet_oracle as uo
    x = torch.tensor([[-2.0, -3e-6, 4e-6, 1.5], [0.3, -0.2, 2e-7, -1e-7]], requires_grad=True)
    own = (x > 0).numpy()
    with uo.ReluAlign({"site": own}) as al:
        y = uo._relu(x, "site")
    assert torch.equal(y, torch.relu(x)) and al.followed == {"site": 0}
    foreign = own.copy()
    foreign[0, 1] = True       # -3e-6: inside tol, followed (the value passes, its gradient too)
    foreign[0, 2] = False      # +4e-6: inside tol, followed (blocked)
    foreign[0, 0] = True       # -2.0: far from zero, ignored
    foreign[1, 0] = False      # +0.3: far from zero, ignored
    with uo.ReluAlign({"site": foreign}, tol=1e-5) as al:
        y = uo._relu(x, "site")
        other = uo._relu(x, "another site")          # no decisions for this name: plain ReLU
    y.sum().backward()
    assert al.followed == {"site": 2}
    assert torch.equal(other, torch.relu(x))
    np.testing.assert_array_equal(y.detach().numpy(), np.array([[0.0, -3e-6, 0.0, 1.5], [0.3, 0.0, 2e-7, 0.0]], dtype=np.float32))
    np.testing.assert_array_equal(x.grad.numpy(), np.array([[0, 1, 0, 1], [1, 0, 1, 0]], dtype=np.float32))
    assert uo._relu_hook is None
