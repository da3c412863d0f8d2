#!/usr/bin/env python3
"""Golden vectors for the LinearModel path from the reference's own Linear module (src/cae_tools/models/linear.py, which
needs torch alone and imports here; linear_model.py does not - it pulls in torchvision / xarray through base_model - so its
training step, linear_model.py:146-153 with MSELoss :241 and Adam(lr, weight_decay) :247, is driven by the loop below).
Runs ONLY in the build container.      python tests/golden/make_golden_linear.py"""
import json
import os
import sys

import numpy as np
import torch

REF_SRC = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
if not os.path.isdir(REF_SRC):
    sys.exit("reference not mounted: this script only runs in the build container")
sys.path.insert(0, REF_SRC)
from cae_tools.models.linear import Linear  # noqa: E402

torch.set_num_threads(1)
CASES = {"lin_8_32_b5": dict(in_shape=(1, 8, 8), out_shape=(1, 32, 32), batch=5, seed=41),
         "lin_2ch_b3": dict(in_shape=(2, 6, 5), out_shape=(3, 9, 7), batch=3, seed=42)}
LR, WD, NSTEPS = 1e-3, 1e-5, 3

for name, cfg in CASES.items():
    torch.manual_seed(cfg["seed"])
    mod = Linear(cfg["in_shape"], cfg["out_shape"])
    out = {"init/" + k: v.detach().numpy().copy() for k, v in mod.state_dict().items()}
    rng = np.random.default_rng(cfg["seed"])
    optim = torch.optim.Adam([{"params": mod.parameters()}], lr=LR, weight_decay=WD)
    loss_fn = torch.nn.MSELoss()
    losses = []
    for i in range(NSTEPS):
        b = cfg["batch"] if i % 2 == 0 else cfg["batch"] - 1
        x = torch.from_numpy(rng.random((b,) + cfg["in_shape"], dtype=np.float32))
        t = torch.from_numpy(rng.random((b,) + cfg["out_shape"], dtype=np.float32))
        out[f"step{i}/x"], out[f"step{i}/t"] = x.numpy(), t.numpy()
        mod.train()
        y = mod(x)
        loss = loss_fn(y, t)
        optim.zero_grad()
        loss.backward()
        if i == 0:
            out["fwd/y"] = y.detach().numpy().copy()
            for k, v in mod.named_parameters():
                out["grad/" + k] = v.grad.detach().numpy().copy()
        optim.step()
        losses.append(float(loss))
    for k, v in mod.state_dict().items():
        out["steps/" + k] = v.detach().numpy().copy()
    out["losses"] = np.array(losses)
    np.savez_compressed(os.path.join(HERE, f"linear_{name}.npz"), **out)
    with open(os.path.join(HERE, f"linear_{name}.json"), "w") as f:
        json.dump({"name": name, "in_shape": cfg["in_shape"], "out_shape": cfg["out_shape"], "seed": cfg["seed"], "lr": LR,
                   "weight_decay": WD, "nsteps": NSTEPS, "keys": list(mod.state_dict().keys())}, f, indent=1)
    print(name, losses)
