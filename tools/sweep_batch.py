"""Throughput and per-kernel time of the training step vs batch size (not the graded bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cae_tools_amd.engine import HipEngine
from bench import build_model, FC, LATENT


def synthetic(n, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand((n, 1, 16, 16), generator=g, dtype=torch.float32)
    t = torch.rand((n, 1, 256, 256), generator=g, dtype=torch.float32)
    return x.to(device), t.to(device)


spec, enc, dec = build_model(0)
dev = torch.device("cuda", 0)
N = 2048
x, t = synthetic(N, dev, 1)
table_at = int(os.environ.get("TABLE_AT", "0"))
for B in (int(b) for b in (sys.argv[1:] or ["64", "256"])):
    eng = HipEngine(spec, FC, LATENT, max_batch=B, device=dev)
    eng.load_state(enc.state_dict(), dec.state_dict()); eng.set_hyper(); eng.set_dataset(0, x, t)
    perm = None      # batches are contiguous rows (what ConvAEModel.train and bench.py do)
    n = (N // B) * B
    for _ in range(3): eng.enqueue_train_steps(0, perm, n, B, 0)
    eng.sync(); t0 = time.perf_counter()
    reps = 10
    for _ in range(reps): eng.enqueue_train_steps(0, perm, n, B, 0)
    eng.sync(); dt = time.perf_counter() - t0
    print(f"B={B:4d}: {dt / (reps * n / B) * 1e6:8.1f} us/step  {reps * n / dt:10.0f} img/s", flush=True)
    if B == table_at:
        eng.profile_begin(); eng.enqueue_train_steps(0, perm, n, B, 0); recs = eng.profile_end()
        agg = {}
        for (name, layer, us, nb) in recs:
            a = agg.setdefault((name, layer), [0.0, 0, nb]); a[0] += us; a[1] += 1
        for (k, v) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
            print(f"   {k[0]:26s} L{k[1]} {v[0]/v[1]:8.1f} us {v[2]/1e6:8.1f} MB {v[2]/(v[0]/v[1])/1e3:7.0f} GB/s")
    del eng
