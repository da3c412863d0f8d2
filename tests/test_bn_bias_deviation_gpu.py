"""The one documented place where the HIP path's saved weights differ from the reference's by design (DESIGN.md §2,
"Bias before BatchNorm"): a conv bias that feeds a BatchNorm has an exactly-zero gradient; the reference evaluates it as a sum
of fp32 terms (~1e-9 of rounding noise) and Adam turns that noise into +-lr steps of arbitrary sign, so those seven biases
random-walk (and running_mean follows them); the HIP path gets 0 and leaves them to weight decay.

This test bounds what that does to what a user sees, over 400 optimiser steps (40 epochs of BASELINE cfg1: gen.py circle data,
100 cases, batch 10, fc16 / latent4):
  A  the oracle (the reference's arithmetic),
  B  the oracle with exactly that deviation applied (those gradients zeroed before Adam) - the deviation in isolation,
  C  the HIP path,
  A' the oracle again from weights that differ from A's in one last bit: the rounding-level ("chaos") baseline.
fp32 training is chaotic: A, A', B and C are four roundings of the same run.  The assertion is that B and C are
no further from A - in held-out eval outputs and loss - than 3x what A' is."""
import numpy as np
import pytest
import torch

from helpers import bn_bias_keys

pytestmark = pytest.mark.gpu

STEPS, BATCH, LR, WD = 400, 10, 1e-3, 1e-5


def _data():
    from cae_tools_amd.data import datagen
    from oracle import cae_oracle as orc
    out = []
    for (seed, n) in ((1234, 100), (4321, 40)):
        ds = datagen.generate("circle", n, seed=seed)
        if not out:
            (_, imin, imax) = orc.scan_variable(ds["lowres"].values)
            (_, omin, omax) = orc.scan_variable(ds["hires"].values)
        out.append((torch.from_numpy(orc.pack_inputs([ds["lowres"].values], [imin], [imax])),
                    torch.from_numpy(orc.normalise_variable(ds["hires"].values, omin, omax))))
    return out


def test_zero_gradient_bn_biases_do_not_change_what_the_model_predicts():
    from cae_tools_amd.engine import HipEngine
    from cae_tools_amd.models.decoder import Decoder
    from cae_tools_amd.models.encoder import Encoder
    from cae_tools_amd.models.model_sizer import create_model_spec
    from oracle import cae_oracle as orc
    torch.set_num_threads(8)
    ((xtr, ttr), (xte, tte)) = _data()
    spec = create_model_spec(input_size=(16, 16), input_channels=1, output_size=(256, 256), output_channels=1)
    torch.manual_seed(0)
    enc = Encoder(spec.get_input_layers(), encoded_space_dim=4, fc_size=16)
    dec = Decoder(spec.get_output_layers(), encoded_space_dim=4, fc_size=16)
    order = np.concatenate([np.random.default_rng(e).permutation(100) for e in range(STEPS * BATCH // 100)]).astype(np.int32)
    noisy = bn_bias_keys(spec.save())

    def run_oracle(zero_noisy, nudge=False):
        es = enc.state_dict()
        if nudge:   # the chaos baseline: the reference's own arithmetic from weights that differ in the last bit
            es = {k: (v * (1.0 + 2.0 ** -23) if k == "encoder_lin.0.weight" else v) for k, v in es.items()}
        o = orc.OracleModel(spec.save(), es, dec.state_dict(), lr=LR, weight_decay=WD)
        for s in range(STEPS):
            idx = torch.from_numpy(order[s * BATCH:(s + 1) * BATCH].astype(np.int64))
            o.loss_and_grads(xtr[idx], ttr[idx])
            if zero_noisy:
                for (side, group) in (("enc/", o.enc), ("dec/", o.dec)):
                    for k, v in group.items():
                        if side + k in noisy:
                            v.grad.zero_()
            o.optim.step()
        return o

    a = run_oracle(False)
    a2 = run_oracle(False, nudge=True)
    b = run_oracle(True)
    eng = HipEngine(spec, 16, 4, max_batch=BATCH, device="cuda:0")
    eng.load_state(enc.state_dict(), dec.state_dict())
    eng.set_hyper(lr=LR, weight_decay=WD)
    eng.set_dataset(0, xtr.cuda(), ttr.cuda())
    perm = eng.upload_perm(order)
    for s in range(STEPS):
        eng.train_step(0, perm, s * BATCH, BATCH)
    (ya, yb, ya2) = (a.eval_forward(xte).numpy(), b.eval_forward(xte).numpy(), a2.eval_forward(xte).numpy())
    yc = eng.score(xte.cuda()).cpu().numpy()
    t = tte.numpy()
    (la, lb, lc, la2) = (float(np.mean((y - t) ** 2)) for y in (ya, yb, yc, ya2))
    print(f"chaos baseline (reference vs reference from 1-ulp-different weights): eval MSE {la2:.6f}, max |dy| {np.abs(ya - ya2).max():.2e}, "
          f"mean |dy| {np.abs(ya - ya2).mean():.2e}; mean |A-B| {np.abs(ya - yb).mean():.2e}")
    (sa, sc) = (a.state(), dict())
    (e_sd, d_sd) = eng.export_state()
    for (pre, sd) in (("enc/", e_sd), ("dec/", d_sd)):
        for k, v in sd.items():
            sc[pre + k] = v
    init = {("enc/" + k): v for k, v in enc.state_dict().items()}
    init.update({("dec/" + k): v for k, v in dec.state_dict().items()})
    walk_a = max(float((sa[k] - init[k]).abs().max()) for k in noisy)
    walk_c = max(float((sc[k].float() - init[k]).abs().max()) for k in noisy)
    d_ab = float(np.abs(ya - yb).max())
    d_bc = float(np.abs(yb - yc).max())
    d_ac = float(np.abs(ya - yc).max())
    print(f"eval MSE on 40 held-out cases after {STEPS} steps: reference {la:.6f}  reference+deviation {lb:.6f}  HIP {lc:.6f}")
    print(f"max |eval output difference|: A-B {d_ab:.2e}  B-C {d_bc:.2e}  A-C {d_ac:.2e};  mean |A-C| {np.abs(ya - yc).mean():.2e}")
    print(f"largest move of a BatchNorm-fed bias: reference {walk_a:.2e} (random walk, <= lr * steps = {LR * STEPS:.2e}), HIP {walk_c:.2e}")
    # Both sets of biases first move towards zero at ~lr per step (Adam normalises the weight-decay gradient wd * w, which is
    # all the HIP path sees and which dominates the reference's 1e-9 of noise until |w| ~ 1e-4); then the reference's random-walk
    # and the HIP path's stay.  Neither can move further than lr per step.
    assert walk_a <= 1.05 * LR * STEPS and walk_c <= 1.05 * LR * STEPS
    apart = max(float((sa[k] - sc[k].float()).abs().max()) for k in noisy)
    print(f"largest distance between a reference and a HIP BatchNorm-fed bias after {STEPS} steps: {apart:.2e}")
    assert apart <= LR * STEPS
    # What the user sees.  fp32 training is chaotic: the reference started from weights that differ in ONE last bit (A') ends
    # 9 % away in held-out loss and 1.5e-2 away in mean output (measured, MI355X box, 400 steps).  The deviation - alone (B) or
    # inside the HIP path (C) - must not move the model further than that rounding-level baseline does (measured: 0.6-0.75x).
    base_dy = float(np.abs(ya - ya2).mean())
    base_dl = abs(la2 - la)
    for (name, y, l) in (("reference + deviation", yb, lb), ("HIP path", yc, lc)):
        dy = float(np.abs(ya - y).mean())
        assert dy <= 3.0 * base_dy + 2e-3, f"{name}: mean |eval output - reference| = {dy:.2e}, rounding-level baseline {base_dy:.2e}"
        assert abs(l - la) <= 3.0 * base_dl + 0.05 * la, f"{name}: held-out MSE {l:.6f} vs reference {la:.6f} (baseline {la2:.6f})"
    # and all of them learnt: the held-out loss fell by more than half from the untrained model's
    l0 = float(np.mean((orc.OracleModel(spec.save(), enc.state_dict(), dec.state_dict()).eval_forward(xte).numpy() - t) ** 2))
    assert max(la, lb, lc) < 0.5 * l0
