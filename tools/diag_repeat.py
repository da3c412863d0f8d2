#!/usr/bin/env python3
"""Run the SAME graph-replayed training step R times from the same state and compare the gradients Adam consumed (recovered
from exp_avg) across the repetitions, per tensor: correct code differs by fp64-atomic arrival order only (<= 1 ulp of fp32
here and there); anything larger is a race.   python tools/diag_repeat.py [reps] [batch]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_timed_path_gpu as T
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    spec, enc, dec = T._model(5)
    x, t = T._data(2 * batch, 6)
    eng = T._engine(spec, enc, dec, x, t)
    eng.train_step(0, None, 0, batch)       # a first step so that the moments are not zero (the update then depends on g)
    eng.sync()
    keep = [a.clone() for a in (eng.params, eng.buffers, eng.exp_avg, eng.exp_avg_sq)]
    runs = []
    for r in range(reps):
        eng.sync()
        for (dst, src) in zip((eng.params, eng.buffers, eng.exp_avg, eng.exp_avg_sq), keep):
            dst.copy_(src)
        torch.cuda.synchronize()
        eng.lib.cae_set_adam_step(eng.handle, 1)
        eng.train_step(0, None, batch, batch)
        eng.sync()
        runs.append((eng.exp_avg.cpu().numpy().astype(np.float64), eng.params.cpu().numpy().astype(np.float64)))
    m0 = keep[2].cpu().numpy().astype(np.float64)
    g = [(m - 0.9 * m0) / 0.1 for (m, p) in runs]
    bad = 0
    for name, (arena, off, numel, shape) in eng.tensors.items():
        if arena != 0:
            continue
        ref = g[0][off:off + numel]
        sc = np.abs(ref).max() + 1e-30
        devs = [np.abs(gr[off:off + numel] - ref).max() / sc for gr in g[1:]]
        pd = [np.abs(runs[i][1][off:off + numel] - runs[0][1][off:off + numel]).max() / T.LR for i in range(1, reps)]
        flag = "  <-- " if max(devs) > 3e-6 or max(pd) > 1e-4 else ""
        bad += bool(flag)
        print(f"{name:34s} max dev of g over {reps - 1} repeats / max|g|: {max(devs):.2e} (median {np.median(devs):.1e})   "
              f"update dev / lr {max(pd):.2e}{flag}")
    print("tensors flagged:", bad)


if __name__ == "__main__":
    main()
