#!/usr/bin/env python3
"""End-to-end agreement beyond the step-level parity tests: train BASELINE cfg1 (the reference's own CPU-runnable case:
gen.py 'circle' data, 100 train / 100 test cases, 16x16 -> 256x256, batch 10, CLI hyper-parameters fc16 / latent4) for E
epochs with the HIP path and with the CPU oracle from the same seed, and compare the loss curves.  fp32 trajectories are
chaotic, so the curves agree statistically, not bitwise.   python tools/convergence_check.py [--epochs 40]"""
import argparse
import io
import json
import os
import sys
import time
from contextlib import redirect_stdout

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from cae_tools_amd.data import datagen
    from cae_tools_amd.models.conv_ae_model import ConvAEModel
    from cae_tools_amd.models.decoder import Decoder
    from cae_tools_amd.models.encoder import Encoder
    from cae_tools_amd.models.model_sizer import create_model_spec
    from oracle import cae_oracle as orc
    train = datagen.generate("circle", 100, seed=1234)
    test = datagen.generate("circle", 100, seed=4321)
    kw = dict(batch_size=10, nr_epochs=args.epochs, test_interval=1, fc_size=16, encoded_dim_size=4)
    torch.manual_seed(args.seed)
    mt = ConvAEModel(**kw)
    t0 = time.time()
    with redirect_stdout(io.StringIO()):
        mt.train(["lowres"], "hires", train, test)
    t_gpu = time.time() - t0
    # the oracle, driven as conv_ae_model.py drives the reference modules
    (_, imin, imax) = orc.scan_variable(train["lowres"].values)
    (_, omin, omax) = orc.scan_variable(train["hires"].values)
    (xtr, ttr) = (torch.from_numpy(orc.pack_inputs([train["lowres"].values], [imin], [imax])),
                  torch.from_numpy(orc.normalise_variable(train["hires"].values, omin, omax)))
    (xte, tte) = (torch.from_numpy(orc.pack_inputs([test["lowres"].values], [imin], [imax])),
                  torch.from_numpy(orc.normalise_variable(test["hires"].values, omin, omax)))
    spec = create_model_spec(input_size=(16, 16), input_channels=1, output_size=(256, 256), output_channels=1)
    torch.manual_seed(args.seed)
    enc = Encoder(spec.get_input_layers(), encoded_space_dim=4, fc_size=16)
    dec = Decoder(spec.get_output_layers(), encoded_space_dim=4, fc_size=16)
    trb = [b for b in torch.utils.data.DataLoader(torch.arange(100), batch_size=10, shuffle=True)]
    teb = [b for b in torch.utils.data.DataLoader(torch.arange(100), batch_size=10, shuffle=True)]
    o = orc.OracleModel(spec.save(), enc.state_dict(), dec.state_dict(), lr=1e-3, weight_decay=1e-5)
    torch.set_num_threads(16)
    (tr, te) = ([], [])
    t0 = time.time()
    for _ in range(args.epochs):
        tr.append(float(np.mean([o.train_step(xtr[i], ttr[i]) for i in trb])))
        te.append(float(np.mean([o.eval_loss(xte[i], tte[i]) for i in teb])))
    t_cpu = time.time() - t0
    (gtr, gte) = (np.array(mt.history["train_loss"]), np.array(mt.history["test_loss"]))
    (tr, te) = (np.array(tr), np.array(te))
    out = {"epochs": args.epochs, "hip_train_loss": [gtr[0], gtr[len(gtr) // 2], gtr[-1]], "cpu_train_loss": [tr[0], tr[len(tr) // 2], tr[-1]],
           "hip_test_loss_last": gte[-1], "cpu_test_loss_last": te[-1],
           "max_rel_diff_train_curve": float(np.max(np.abs(gtr - tr) / tr)), "max_rel_diff_test_curve": float(np.max(np.abs(gte - te) / te)),
           "first_epoch_rel_diff": float(abs(gtr[0] - tr[0]) / tr[0]), "hip_seconds_incl_setup": t_gpu, "cpu_oracle_seconds": t_cpu}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
