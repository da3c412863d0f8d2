"""shared helpers of the UNET tests: golden-case loading (tests/golden/unet_*.npz|json, written by
tests/golden/make_golden_unet.py from the reference's own class bodies)"""
import glob
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
UNET_CASES = sorted(os.path.basename(p)[5:-5] for p in glob.glob(os.path.join(GOLDEN, "unet_*.json")))
TRAIN_CASES = [c for c in UNET_CASES if "eval" not in c]


class UnetCase:

    def __init__(self, name):
        with open(os.path.join(GOLDEN, f"unet_{name}.json")) as f:
            self.meta = json.load(f)
        self.z = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
        self.name = name

    def state(self, prefix, which):
        keys = self.meta["enc_keys"] if which == "enc" else self.meta["dec_keys"]
        return {k: torch.from_numpy(self.z[f"{prefix}/{which}/{k}"].copy()) for k in keys}

    def t(self, key):
        return torch.from_numpy(self.z[key].copy())

    def step_batch(self, i):
        return self.t(f"step{i}/x"), self.t(f"step{i}/t"), self.t(f"step{i}/m")


def unet_oracle(case, prefix="init", **kw):
    from oracle import unet_oracle as uo
    m = case.meta
    args = dict(lr=m["lr"], weight_decay=m["weight_decay"], dropout_rate=m["dropout"], lambda_pearson=m["lambda_pearson"])
    args.update(kw)
    return uo.UnetOracle(m["spec"], case.state(prefix, "enc"), case.state(prefix, "dec"), **args)


def hip_relu_decisions(eng, spec_json, fc, latent, B):
    """(output > 0) at every ReLU site of the HIP engine's last training forward, for oracle.unet_oracle.ReluAlign: the
    encoder skips (ReLU outputs), the four Linear activations and the decoder inputs (after their dropout: a False there may
    also mean 'dropped', which ReluAlign is indifferent to)"""
    dec = {}
    for i, l in enumerate(spec_json["input_layers"]):
        (c, h, w) = l["output_dimensions"]
        dec[f"enc{i}"] = eng.debug_read(f"enc_s{i}", B * c * h * w) > 0
    (c2, h2, w2) = spec_json["output_layers"][0]["input_dimensions"]
    for k, (name, n) in enumerate((("efc0", fc), ("efc1", latent), ("dfc0", fc), ("dfc1", c2 * h2 * w2))):
        dec[name] = eng.debug_read(f"fc_a{k}", B * n) > 0
    for j, l in enumerate(spec_json["output_layers"][:-1]):
        (c, h, w) = l["output_dimensions"]
        dec[f"dec{j}"] = eng.debug_read(f"dec_din{j + 1}", B * 2 * c * h * w) > 0
    return dec
