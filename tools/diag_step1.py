#!/usr/bin/env python3
"""Second graph-replayed training step at the benchmark size from the oracle's state: per tensor, the gradient the engine's
Adam consumed (recovered from exp_avg: g = (m1 - 0.9 m0) / 0.1 - wd w) and the parameter update, against the fp32 and fp64
oracles.  Diagnostic for tests/test_timed_path_gpu.py.   python tools/diag_step1.py [tensor-name-substring]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_timed_path_gpu as T
    from oracle import cae_oracle as orc_mod
    focus = sys.argv[1] if len(sys.argv) > 1 else "decoder_conv.9.weight"
    torch.set_num_threads(int(os.environ.get("DIAG_THREADS", "8")))
    batch = 64
    spec, enc, dec = T._model(5)
    x, t = T._data(2 * batch, 6)
    eng = T._engine(spec, enc, dec, x, t)
    o32, _ = T._oracles(spec, enc, dec)
    starts = [(0, batch), (x.shape[0] - batch, batch)]
    for s in range(2):
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
        o64 = orc_mod.OracleModel(spec.save(), to64(e0), to64(d0), lr=T.LR, weight_decay=T.WD)
        for side, group in (("enc/", o64.enc), ("dec/", o64.dec)):
            for k, p64 in group.items():
                if side + k in moments:
                    (m, v) = moments[side + k]
                    o64.optim.state[p64] = {"step": torch.tensor(float(s)), "exp_avg": m.double().clone(), "exp_avg_sq": v.double().clone()}
        (lo, n) = starts[s % 2]
        (xb, tb) = (x[lo:lo + n], t[lo:lo + n])
        m0_all = eng.exp_avg.cpu().numpy().astype(np.float64)
        o32.train_step(xb, tb)
        tr64 = {}
        o64.loss_and_grads(xb.double(), tb.double(), trace=tr64)
        o64.optim.step()
        for l in range(5):
            y = tr64[f"dec_conv{l}"]
            (mu, sd) = (y.mean((0, 2, 3)).numpy(), y.std((0, 2, 3)).numpy())
            print(f"  step {s} dec conv {l}: |mean|/std per channel max {np.abs(mu / sd).max():.2f}; std min {sd.min():.3e} max {sd.max():.3e}; "
                  f"last channels mean {mu[-2:]} std {sd[-2:]}")
        (g32, g64) = (o32.grads(), o64.grads())
        eng.train_step(0, None, lo, n)
        eng.sync()
        m1_all = eng.exp_avg.cpu().numpy().astype(np.float64)
        v1_all = eng.exp_avg_sq.cpu().numpy().astype(np.float64)
        p1_all = eng.params.cpu().numpy().astype(np.float64)
        (after32, after64) = (o32.state(), o64.state())
        print(f"step {s}:")
        for key in g32:
            (arena, off, numel, shape) = eng.tensors[key]
            w0 = before[key].numpy().astype(np.float64).reshape(-1)
            gh = (m1_all[off:off + numel] - 0.9 * m0_all[off:off + numel]) / 0.1 - T.WD * w0
            (a32, a64) = (g32[key].numpy().astype(np.float64).reshape(-1), g64[key].numpy().reshape(-1))
            sc = np.abs(a64).max()
            dh = p1_all[off:off + numel] - w0
            d32 = after32[key].numpy().astype(np.float64).reshape(-1) - w0
            d64 = after64[key].numpy().reshape(-1) - w0
            line = (f"  {key:32s} g: hip {np.abs(gh - a64).max() / sc:.1e} o32 {np.abs(a32 - a64).max() / sc:.1e}   "
                    f"update/lr: hip {np.abs(dh - d64).max() / T.LR:.2e} o32 {np.abs(d32 - d64).max() / T.LR:.2e}")
            print(line)
            if focus in key:
                err = (gh - a64).reshape(shape)
                np.set_printoptions(linewidth=220, precision=2)
                print("    g error / max|g| per element, hip:")
                print((err / sc).reshape(shape[0], -1))
                print("    the fp32 oracle's:")
                print(((a32 - a64) / sc).reshape(shape[0], -1))
                i = int(np.argmax(np.abs(dh - d64)))
                mo = after32 and o32.optim.state[(o32.enc if key.startswith('enc/') else o32.dec)[key[4:]]]
                print(f"    worst element {i}: w0 {w0[i]:.9e} g64 {a64[i]:.9e} g32 {a32[i]:.9e} ghip {gh[i]:.9e}")
                print(f"      m0 {m0_all[off + i]:.9e} m1 hip {m1_all[off + i]:.9e} o32 {float(mo['exp_avg'].reshape(-1)[i]):.9e}")
                print(f"      v1 hip {v1_all[off + i]:.9e} o32 {float(mo['exp_avg_sq'].reshape(-1)[i]):.9e}")
                print(f"      update hip {dh[i]:.9e} o32 {d32[i]:.9e} o64 {d64[i]:.9e}")


if __name__ == "__main__":
    main()
