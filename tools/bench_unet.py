#!/usr/bin/env python3
"""UNET training throughput on one MI355X (BASELINE.json configs[2], "cfg3": 256x256 -> 256x256, 3 channels,
'unet' with skip connections, batch 32).  Not the round's headline bench (bench.py is); prints one JSON line.

    python tools/bench_unet.py [--steps 20] [--warmup 3] [--batch 32] [--channels 32,64,128,256] [--generic] [--cpu]

Layer definitions are hand-written (k4 s2 p1: the auto-sizer cannot produce a skip-compatible spec);
synthetic data, torch.manual_seed(0) default initialisation, dropout 0.1, AdamW.  FLOPs are the algorithmic
conv / conv-transpose / linear MACs x 2 x 3 (forward, input gradient, weight gradient); MFMA fp32 peak 157.3 TF.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cae_tools_amd.models.unet import Decoder, Encoder, unet_layer_spec  # noqa: E402


def conv_macs(spec_json, fc, latent):
    macs = 0
    for l in spec_json["input_layers"]:
        (ci, _, _), (co, oh, ow) = l["input_dimensions"], l["output_dimensions"]
        macs += co * oh * ow * ci * l["kernel_size"] ** 2
    for l in spec_json["output_layers"]:
        (ci, ih, iw), (co, _, _) = l["input_dimensions"], l["output_dimensions"]
        macs += ci * ih * iw * co * l["kernel_size"] ** 2
    (c, h, w) = spec_json["input_layers"][-1]["output_dimensions"]
    (c2, h2, w2) = spec_json["output_layers"][0]["input_dimensions"]
    macs += c * h * w * fc + fc * latent + latent * fc + fc * c2 * h2 * w2
    return macs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--channels", default="32,64,128,256")
    ap.add_argument("--generic", action="store_true", help="shape-generic kernels instead of the MFMA ones")
    ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle (one step)")
    args = ap.parse_args()
    chans = [int(c) for c in args.channels.split(",")]
    spec = unet_layer_spec(3, 3, (args.size, args.size), chans)
    (fc, latent, B) = (128, 32, args.batch)
    torch.manual_seed(0)
    enc = Encoder(spec.get_input_layers(), latent, fc)
    dec = Decoder(spec.get_output_layers(), latent, fc)
    from cae_tools_amd.unet_engine import UnetEngine
    eng = UnetEngine(spec, fc, latent, B, device="cuda:0", specialised=not args.generic)
    eng.load_state(enc.state_dict(), dec.state_dict())
    eng.set_hyper(lr=1e-3, weight_decay=1e-5, dropout_rate=0.1, lambda_pearson=1.0, seed=1)
    n = 2 * B
    g = torch.Generator().manual_seed(1)
    x = torch.rand((n, 3, args.size, args.size), generator=g)
    t = torch.rand((n, 3, args.size, args.size), generator=g)
    eng.set_dataset(0, x.cuda(), t.cuda(), None)
    perm = eng.upload_perm(np.random.default_rng(0).permutation(n))

    def run(k):
        for s in range(k):
            eng.train_step(0, perm, (s % 2) * B, B, slot=s % 64)
    run(args.warmup)
    eng.sync()
    t0 = time.perf_counter()
    run(args.steps)
    eng.sync()
    dt = (time.perf_counter() - t0) / args.steps
    macs = conv_macs(spec.save(), fc, latent)
    flops = 2 * 3 * macs * B
    out = {"metric": "unet train images/sec (3x256x256 -> 3x256x256, batch 32)", "value": B / dt, "unit": "images/s",
           "ms_per_step": dt * 1e3, "n_gpus": 1, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"cfg3: UNET {args.size}x{args.size} 3->3 ch, channels {chans}, k4 s2 p1, fc128/latent32, "
                                  f"batch {B}, dropout 0.1, masked MSE + Pearson, AdamW", "kernels": "generic" if args.generic else "mfma"},
           "algorithmic_gflop_per_step": flops / 1e9,
           "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                        "frac": flops / dt / 1e12 / 157.3, "traffic": None},
           "losses_last": eng.read_losses((args.steps - 1) % 64, 1)[0]}
    if args.cpu:
        from oracle import unet_oracle as uo
        torch.set_num_threads(16)
        o = uo.UnetOracle(spec.save(), enc.state_dict(), dec.state_dict(), dropout_rate=0.0)
        m = torch.ones((B, 3, args.size, args.size))
        o.train_step(x[:B], t[:B], m)
        c0 = time.perf_counter()
        o.train_step(x[:B], t[:B], m)
        cdt = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": B / cdt, "unit": "images/s", "cores": 16, "kind": "port",
                               "sample": "1 training step at batch %d after 1 warm-up, dropout 0, torch CPU" % B}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
