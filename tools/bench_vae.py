#!/usr/bin/env python3
"""'var' (VAE + MS-SSIM) training throughput on one MI355X (BASELINE.json configs[4], "cfg5": 64x64 -> 512x512, 1 channel,
batch 16).  Not the headline bench; prints one JSON line.   python tools/bench_vae.py [--steps 20] [--warmup 3] [--cpu]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    from cae_tools_amd.models.decoder import Decoder
    from cae_tools_amd.models.model_sizer import create_model_spec
    from cae_tools_amd.models.var_ae_model import VarEncoder
    from cae_tools_amd.vae_engine import VaeEngine
    spec = create_model_spec(input_size=(64, 64), input_channels=1, output_size=(512, 512), output_channels=1)
    (fc, latent, B) = (128, 32, args.batch)
    torch.manual_seed(0)
    enc = VarEncoder(spec.get_input_layers(), latent, fc)
    dec = Decoder(spec.get_output_layers(), latent, fc)
    eng = VaeEngine(spec, fc, latent, B, device="cuda:0")
    eng.load_state(enc.state_dict(), dec.state_dict())
    eng.set_hyper(seed=1)
    n = 2 * B
    g = torch.Generator().manual_seed(1)
    x = torch.rand((n, 1, 64, 64), generator=g)
    t = torch.rand((n, 1, 512, 512), generator=g)
    eng.set_dataset(0, x.cuda(), t.cuda())
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
    out = {"metric": "var (VAE + MS-SSIM) train images/sec (64x64 -> 512x512, batch 16)", "value": B / dt, "unit": "images/s",
           "ms_per_step": dt * 1e3, "n_gpus": 1, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "cfg5: VarAEModel 64x64->512x512 1-ch, fc128/latent32, batch %d, MSE + KL + MS-SSIM, Adam" % B,
                      "params": sum(t_[2] for t_ in eng.tensors.values() if t_[0] == 0)},
           "losses_last": eng.read_losses((args.steps - 1) % 64, 1)[0]}
    if args.cpu:
        from oracle import vae_oracle as vo
        torch.set_num_threads(16)
        o = vo.VaeOracle(spec.save(), enc.state_dict(), dec.state_dict())
        o.train_step(x[:B], t[:B])
        c0 = time.perf_counter()
        o.train_step(x[:B], t[:B])
        cdt = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": B / cdt, "unit": "images/s", "cores": 16, "kind": "port",
                               "sample": "1 training step at batch %d after 1 warm-up, torch CPU (own definition)" % B}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
