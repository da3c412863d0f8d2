"""The launch sequence bench.py and ConvAEModel.train() actually time - cae_train_step(s): graph-replayed, k_head_fwd leaving
the batch copy and the pre-update gamma behind, the first encoder layer's weight gradient inside k_adam (AdamConv0), the 16->8
layer's backward on k_ct_bwd_band - held to the CPU oracle AT THE BENCHMARK SIZE (BASELINE cfg2: 16x16 -> 256x256, fc128 /
latent32, batch 64, and the reference's ragged last batch of 36).  forward_backward() (tests/test_full_size_gpu.py) keeps the
first encoder layer's weight gradient as its own launch, so it cannot see a fused-launch bug; these tests can:

* one cae_train_step from zero moments: exp_avg = (1 - beta1)(g + wd w), so every parameter's gradient is recovered from the
  fused step and compared with the oracle's by the fp64-anchored criterion (no further from the fp64 answer than 3x the fp32
  reference itself is), the oracles taking the HIP step's ReLU decisions at the handful of positions whose input is within
  rounding of zero (helpers.relu_fix_for: at 2e6 BatchNorm outputs per layer two correct fp32 implementations disagree on a
  mask bit about once per step, and one bit moves every upstream gradient by ~1e-3);
* four graph-replayed steps (the state re-synchronised to the oracle's before each, as test_adam_step_no_further_from_fp64...):
  losses 2e-5 relative, every parameter tensor's update no further from the fp64 oracle's than 3x the fp32 reference's own;
* four FREE-RUNNING graph-replayed steps: the loss trajectory against the oracle's;
* the same step through DataParallel on a one-rank RCCL group at 64 rows per rank (the 8-GPU configuration's per-rank work).
"""
import os
import socket

import numpy as np
import pytest
import torch

from helpers import assert_close_as_reference, bn_bias_keys, hip_relu_decisions, relu_fix_for

pytestmark = pytest.mark.gpu

LR, WD = 1e-3, 1e-5


def _model(seed):
    from cae_tools_amd.models.model_sizer import create_model_spec
    from cae_tools_amd.models.encoder import Encoder
    from cae_tools_amd.models.decoder import Decoder
    spec = create_model_spec(input_size=(16, 16), input_channels=1, output_size=(256, 256), output_channels=1)
    torch.manual_seed(seed)
    enc = Encoder(spec.get_input_layers(), encoded_space_dim=32, fc_size=128)
    dec = Decoder(spec.get_output_layers(), encoded_space_dim=32, fc_size=128)
    return spec, enc.state_dict(), dec.state_dict()


def _data(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((n, 1, 16, 16), generator=g), torch.rand((n, 1, 256, 256), generator=g)


def _engine(spec, enc, dec, x, t, specialised=True):
    from cae_tools_amd.engine import HipEngine
    eng = HipEngine(spec, 128, 32, max_batch=64, graph=True, specialised=specialised)
    eng.load_state(enc, dec)
    eng.set_hyper(lr=LR, weight_decay=WD)
    eng.set_dataset(0, x.cuda(), t.cuda())
    return eng


def _oracles(spec, enc, dec):
    from oracle import cae_oracle as orc
    to64 = lambda sd: {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    o32 = orc.OracleModel(spec.save(), enc, dec, lr=LR, weight_decay=WD)
    o64 = orc.OracleModel(spec.save(), to64(enc), to64(dec), lr=LR, weight_decay=WD)
    return o32, o64


@pytest.mark.parametrize("batch", [64, 36])
@pytest.mark.parametrize("specialised", [1, 5], ids=["lds-forward", "gather-forward"])
def test_fused_graph_step_gradients_against_the_oracle(batch, specialised):
    """g recovered from exp_avg after ONE graph-replayed cae_train_step against the oracle's gradients, fp64-anchored.
    specialised=5: the channel-rich decoder layers' forward on the gather kernel k_ig_fwd_s2 (what layers with Cin % 4 != 0
    always run) - the configuration that sat 6.8e-4 from the oracle until its BatchNorm sums were folded in fp64."""
    torch.set_num_threads(8)
    spec, enc, dec = _model(3)
    x, t = _data(batch, 4)
    eng = _engine(spec, enc, dec, x, t, specialised=specialised)
    w0 = eng.params.cpu().numpy().astype(np.float64)
    loss = eng.train_step(0, None, 0, batch)
    eng.sync()
    g_fused = eng.exp_avg.cpu().numpy().astype(np.float64) / 0.1 - WD * w0
    decisions = hip_relu_decisions(eng, batch)
    o32, o64 = _oracles(spec, enc, dec)
    (fix32, _) = relu_fix_for(o32, x, decisions, "fp32 oracle")
    (fix64, flips) = relu_fix_for(o64, x.double(), decisions, "fp64 oracle")
    print(f"B={batch}: {flips} ReLU decisions taken from the HIP step")
    loss32, _ = o32.loss_and_grads(x, t, relu_fix=fix32)
    loss64, _ = o64.loss_and_grads(x.double(), t.double(), relu_fix=fix64)
    assert abs(loss - loss64) <= 3.0 * abs(loss32 - loss64) + 2e-6 * abs(loss64)
    (g32, g64) = (o32.grads(), o64.grads())
    noisy = bn_bias_keys(spec.save())
    for k in g32:
        if k in noisy:
            continue
        (arena, off, numel, shape) = eng.tensors[k]
        # exp_avg is fp32: recovering g from it costs 6e-8 of |g + wd w|, well under the criterion's relative floor
        assert_close_as_reference(g_fused[off:off + numel].reshape(shape), g32[k].numpy(), g64[k].numpy(), f"B={batch} {k}")
    # and this step did run the launches this test is about
    eng.profile_begin()
    eng.train_step(0, None, 0, batch)
    labels = [name for (name, layer, us, nbytes) in eng.profile_end() if name != "event_pair"]
    assert "enc_conv_wgrad" not in labels, labels          # the first encoder layer's weight gradient lives in k_adam
    assert labels[0] == "head_fwd" and labels[-1] == "adam", labels
    assert ("ig_convt_fwd" in labels) == (specialised == 5) and ("ct_convt_fwd" in labels) == (specialised == 1), labels


@pytest.mark.parametrize("batch", [64, 36])
def test_graph_replayed_steps_no_further_from_fp64_than_the_reference(batch):
    """Four graph-replayed training steps at the benchmark size, each from the oracle's state: the loss within 2e-5 and every
    parameter tensor's update no further from an fp64 oracle's update than 3x the fp32 reference's own is (+ 1e-3 lr), both
    oracles stepping with the HIP step's ReLU decisions where their own input is within rounding of zero."""
    from oracle import cae_oracle as orc_mod
    torch.set_num_threads(8)
    spec, enc, dec = _model(5)
    x, t = _data(2 * batch if batch == 64 else 64 + batch, 6)
    eng = _engine(spec, enc, dec, x, t)
    o32, _ = _oracles(spec, enc, dec)
    noisy = bn_bias_keys(spec.save())
    starts = [(0, batch), (x.shape[0] - batch, batch)]     # two different batches, alternating
    (worst, total_flips) = (0.0, 0)
    for s in range(4):
        before = o32.state()
        (e0, d0) = ({k[4:]: v for k, v in before.items() if k.startswith("enc/")},
                    {k[4:]: v for k, v in before.items() if k.startswith("dec/")})
        moments = {}
        for side, group in (("enc/", o32.enc), ("dec/", o32.dec)):
            for k, p in group.items():
                st = o32.optim.state.get(p)
                if st:
                    moments[side + k] = (st["exp_avg"].clone(), st["exp_avg_sq"].clone())
        eng.load_state(e0, d0)
        eng.load_optimizer_state(moments, s)
        to64 = lambda sd: {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        o64 = orc_mod.OracleModel(spec.save(), to64(e0), to64(d0), lr=LR, weight_decay=WD)
        for side, group in (("enc/", o64.enc), ("dec/", o64.dec)):
            for k, p64 in group.items():
                if side + k in moments:
                    (m, v) = moments[side + k]
                    o64.optim.state[p64] = {"step": torch.tensor(float(s)), "exp_avg": m.double().clone(),
                                            "exp_avg_sq": v.double().clone()}
        (lo, n) = starts[s % 2]
        (xb, tb) = (x[lo:lo + n], t[lo:lo + n])
        loss = eng.train_step(0, None, lo, n)
        decisions = hip_relu_decisions(eng, n)
        (fix32, _) = relu_fix_for(o32, xb, decisions, f"step {s} fp32 oracle")
        (fix64, flips) = relu_fix_for(o64, xb.double(), decisions, f"step {s} fp64 oracle")
        total_flips += flips
        loss32 = o32.train_step(xb, tb, relu_fix=fix32)
        o64.train_step(xb.double(), tb.double(), relu_fix=fix64)
        assert abs(loss - loss32) <= 2e-5 * abs(loss32), (s, loss, loss32)
        (after32, after64) = (o32.state(), o64.state())
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
                worst = max(worst, err_hip / (3.0 * err_ref + 1e-3 * LR))
                assert err_hip <= 3.0 * err_ref + 1e-3 * LR, \
                    f"step {s} {key}: |hip - fp64| = {err_hip:.3e}, the reference's own {err_ref:.3e}"
    print(f"B={batch}: worst ratio to the bound {worst:.2f}; {total_flips} ReLU decisions taken from the HIP steps")


def test_free_running_graph_steps_follow_the_oracle_trajectory():
    """Six free-running steps over three batches of 64 replayed from the multi-step graph (cae_train_steps): the loss of
    every step against OracleModel.train_step on the same rows."""
    torch.set_num_threads(8)
    spec, enc, dec = _model(7)
    x, t = _data(192, 8)
    eng = _engine(spec, enc, dec, x, t)
    o32, _ = _oracles(spec, enc, dec)
    got = eng.run_batches(0, None, 192, 64, train=True) + eng.run_batches(0, None, 192, 64, train=True)
    ref = [o32.train_step(x[s:s + 64], t[s:s + 64]) for s in (0, 64, 128, 0, 64, 128)]
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=0)
    assert eng.graph_count() == 1, eng.graph_count()       # one 3-step graph served both passes


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


@pytest.mark.parametrize("overlap", [True, False], ids=["overlap", "serial"])
def test_data_parallel_graph_at_64_rows_per_rank(dist1, overlap):
    """BASELINE cfg4's per-rank work (64 rows per rank) through the in-library data-parallel step on a one-rank RCCL group,
    graph-captured: the all-reduced fp32 gradient arena against the oracle's gradients (fp64-anchored) and the weights after
    the step against the fused single-device step's."""
    from cae_tools_amd.dp import DataParallel
    torch.set_num_threads(8)
    spec, enc, dec = _model(9)
    x, t = _data(64, 10)
    (a, b) = (_engine(spec, enc, dec, x, t), _engine(spec, enc, dec, x, t))
    dp = DataParallel(b, dist1, sync_bn=False, overlap=overlap)
    dp.broadcast_parameters(0)
    assert dp.native and b.dp_world == 1 and b.dp_graph_capture()
    la = a.train_step(0, None, 0, 64)
    slot = dp.train_step(0, None, 0, 64)
    lb = b.dp_read_losses(slot, 1)[0]
    b.sync()
    assert abs(la - lb) <= 1e-7 * abs(la)
    decisions = hip_relu_decisions(b, 64)
    o32, o64 = _oracles(spec, enc, dec)
    (fix32, _) = relu_fix_for(o32, x, decisions, "fp32 oracle")
    (fix64, _) = relu_fix_for(o64, x.double(), decisions, "fp64 oracle")
    o32.loss_and_grads(x, t, relu_fix=fix32)
    o64.loss_and_grads(x.double(), t.double(), relu_fix=fix64)
    (g32, g64) = (o32.grads(), o64.grads())
    noisy = bn_bias_keys(spec.save())
    for k in g32:
        if k not in noisy:
            assert_close_as_reference(b.grad_view(k).cpu().numpy(), g32[k].numpy(), g64[k].numpy(), f"dp {k}")
    d = np.abs(a.params.cpu().numpy().astype(np.float64) - b.params.cpu().numpy().astype(np.float64))
    assert d.max() <= 1e-3 * LR, d.max()      # the same arithmetic: only the arrival order of fp64 atomics differs


def test_graph_cache_keeps_syncbn_and_local_bn_steps_apart(dist1):
    """A SyncBN step and a per-rank-BatchNorm step of the same sizes are different launch sequences (table all-reduces,
    bn_batch, the 1/world scale): each gets its own cached hipGraph, and replaying one never serves the other."""
    from cae_tools_amd.dp import DataParallel
    from helpers import GoldenCase
    case = GoldenCase("cfg2_b4")
    from cae_tools_amd.engine import HipEngine
    x = torch.from_numpy(case.x).cuda()
    t = torch.from_numpy(case.t).cuda()
    e = HipEngine(case.spec, case.meta["fc"], case.meta["latent"], max_batch=x.shape[0])
    e.load_state(case.group("init/enc/"), case.group("init/dec/"))
    e.set_hyper(lr=case.meta["lr"], weight_decay=case.meta["weight_decay"])
    e.set_dataset(0, x, t)
    dp = DataParallel(e, dist1, sync_bn=False, overlap=False)
    n = x.shape[0]
    before = e.graph_count()
    e.set_cursor(0, e.claim_slots(1)); e.dp_train_steps(0, None, n, n, False, 1)
    one = e.graph_count()
    e.set_cursor(0, e.claim_slots(1)); e.dp_train_steps(0, None, n, n, True, 1)
    two = e.graph_count()
    e.set_cursor(0, e.claim_slots(1)); e.dp_train_steps(0, None, n, n, False, 1)
    e.sync()
    assert one == before + 1 and two == one + 1 and e.graph_count() == two, (before, one, two, e.graph_count())
    # and the labels of the two sequences differ: SyncBN runs the per-layer launches (no fused head / tail)
    seqs = []
    for sync in (False, True):
        e.profile_begin()
        e.set_cursor(0, e.claim_slots(1)); e.dp_train_steps(0, None, n, n, sync, 1)
        seqs.append([name for (name, layer, us, nbytes) in e.profile_end() if name != "event_pair"])
    assert seqs[0] != seqs[1] and "head_fwd" in seqs[0] and "head_fwd" not in seqs[1], seqs
