#!/usr/bin/env python3
"""Generate the golden vectors that pin oracle/ and the HIP path to the reference.

Runs ONLY in the build container: it imports the reference's importable arithmetic
modules (model_sizer / encoder / decoder / ds_dataset) from /root/reference/src,
drives them with a short loop equivalent to the reference step
(conv_ae_model.py:189-197: forward -> MSELoss -> zero_grad -> backward -> Adam.step)
and stores inputs + expected outputs as data.  No reference source is stored.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz|*.json

The reference has no golden vectors or numeric assertions of its own,
so these files are the parity pin.  Everything that needs a random number takes it
from torch.manual_seed(<seed in the case table>) or numpy default_rng(<seed>).
"""
import gzip
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

from cae_tools.models.model_sizer import create_model_spec, ModelSpec  # noqa: E402
from cae_tools.models.encoder import Encoder  # noqa: E402
from cae_tools.models.decoder import Decoder  # noqa: E402
from cae_tools.models.ds_dataset import DSDataset  # noqa: E402

torch.set_num_threads(1)  # one thread: summation order of the stored vectors is fixed

# ----------------------------------------------------------------------------------
# shared helpers (the tests re-implement these three small functions identically)
# ----------------------------------------------------------------------------------

def synth_input(rng, b, c, h, w):
    """normalised-looking low-res input in [0,1)"""
    return rng.random((b, c, h, w), dtype=np.float32)


def synth_target_u8(rng, b, c, h, w):
    """smooth target quantised to k/256 so it stores as uint8 and is exact in fp32"""
    yy, xx = np.meshgrid(np.linspace(-3, 3, h), np.linspace(-2, 2, w), indexing="ij")
    out = np.zeros((b, c, h, w), dtype=np.uint8)
    for i in range(b):
        for j in range(c):
            mu = 0.6 + 1.2 * rng.random()
            amp = 0.3 + 0.6 * rng.random()
            base = 0.1 * rng.random()
            d = np.sqrt(yy * yy + xx * xx)
            g = base + amp * np.exp(-((d - mu) ** 2) / (2 * 0.2 ** 2))
            out[i, j] = np.clip(np.floor(g * 256), 0, 255).astype(np.uint8)
    return out


def projections(arr, seed, nproj=8):
    """fp64 dot products of the flattened array with fixed +-1 vectors"""
    flat = np.asarray(arr, dtype=np.float64).reshape(-1)
    rng = np.random.default_rng(seed)
    out = np.zeros(nproj, dtype=np.float64)
    for k in range(nproj):
        signs = rng.integers(0, 2, size=flat.size, dtype=np.int8).astype(np.float64) * 2.0 - 1.0
        out[k] = float(np.dot(flat, signs))
    return out


def subsample(arr, step=5):
    return np.ascontiguousarray(arr[..., ::step, ::step])


# ----------------------------------------------------------------------------------
# model cases
# ----------------------------------------------------------------------------------

CASES = {
    # BASELINE cfg1 geometry and CLI hyper-parameters (fc16 / latent4), plumbing batch
    "cfg1_b3": dict(in_size=(16, 16), in_ch=1, out_size=(256, 256), out_ch=1, fc=16, latent=4,
                    batch=3, seed=11, full_output=True),
    # BASELINE cfg2 geometry and API defaults (fc128 / latent32)
    "cfg2_b4": dict(in_size=(16, 16), in_ch=1, out_size=(256, 256), out_ch=1, fc=128, latent=32,
                    batch=4, seed=12),
    # test_specs.py "circle2": non-square 24x20 -> 280x256
    "circle2_b2": dict(in_size=(24, 20), in_ch=1, out_size=(280, 256), out_ch=1, fc=32, latent=8,
                       batch=2, seed=13),
    # test_specs.py "tidal_circle1": two input variables, 6x6 -> 256x256, its hyper-parameters
    "tidal_b3": dict(in_size=(6, 6), in_ch=2, out_size=(256, 256), out_ch=1, fc=32, latent=8,
                     batch=3, seed=14),
    # odd sizes, 2 output channels, kernel 5: decoder kernels grow per dimension
    "odd_k5_b5": dict(in_size=(21, 18), in_ch=2, out_size=(77, 61), out_ch=2, fc=24, latent=6,
                      batch=5, seed=15, kernel=5, stride=2, full_output=True),
    # stride 3
    "s3_b4": dict(in_size=(20, 20), in_ch=1, out_size=(64, 64), out_ch=1, fc=20, latent=5,
                  batch=4, seed=16, kernel=3, stride=3, full_output=True),
    # explicit layer counts (conv_input_layer_count / conv_output_layer_count)
    "counts_b3": dict(in_size=(32, 32), in_ch=1, out_size=(64, 64), out_ch=1, fc=40, latent=10,
                      batch=3, seed=17, in_layers=1, out_layers=2, full_output=True),
}

# a hand-written layer definition file (--layer-definitions-path) with output_padding
HANDSPEC = {
    "input_layers": [
        {"is_input": True, "kernel_size": 3, "stride": 2, "output_padding": 0,
         "input_dimensions": [1, 12, 12], "output_dimensions": [3, 5, 5]},
    ],
    "output_layers": [
        {"is_input": False, "kernel_size": 3, "stride": 2, "output_padding": 1,
         "input_dimensions": [6, 5, 5], "output_dimensions": [3, 12, 12]},
        {"is_input": False, "kernel_size": [4, 3], "stride": 2, "output_padding": 0,
         "input_dimensions": [3, 12, 12], "output_dimensions": [1, 26, 25]},
    ],
}

LR = 1e-3
WD = 1e-5
NSTEPS = 4


def state_to_np(prefix, module, out):
    for k, v in module.state_dict().items():
        out[f"{prefix}{k}"] = v.detach().cpu().numpy().copy()


def run_case(name, cfg, spec=None):
    torch.manual_seed(cfg["seed"])
    if spec is None:
        spec = create_model_spec(input_size=cfg["in_size"], input_channels=cfg["in_ch"],
                                 output_size=cfg["out_size"], output_channels=cfg["out_ch"],
                                 kernel_size=cfg.get("kernel", 3), stride=cfg.get("stride", 2),
                                 input_layer_count=cfg.get("in_layers"),
                                 output_layer_count=cfg.get("out_layers"))
    enc = Encoder(spec.get_input_layers(), encoded_space_dim=cfg["latent"], fc_size=cfg["fc"])
    dec = Decoder(spec.get_output_layers(), encoded_space_dim=cfg["latent"], fc_size=cfg["fc"])

    out = {}
    meta = {"name": name, "seed": cfg["seed"], "fc": cfg["fc"], "latent": cfg["latent"],
            "batch": cfg["batch"], "lr": LR, "weight_decay": WD, "nsteps": NSTEPS,
            "spec": spec.save()}
    state_to_np("init/enc/", enc, out)
    state_to_np("init/dec/", dec, out)

    rng = np.random.default_rng(1000 + cfg["seed"])
    b = cfg["batch"]
    (ic, ih, iw) = spec.get_input_layers()[0].get_input_dimensions()
    (oc, oh, ow) = spec.get_output_layers()[-1].get_output_dimensions()
    x_np = synth_input(rng, b, ic, ih, iw)
    t_u8 = synth_target_u8(rng, b, oc, oh, ow)
    # a second, smaller batch: the partial last batch of conv_ae_model.py:291 (drop_last=False)
    b2 = max(1, b - 1)
    x2_np = synth_input(rng, b2, ic, ih, iw)
    t2_u8 = synth_target_u8(rng, b2, oc, oh, ow)
    out["x"] = x_np
    out["t_u8"] = t_u8
    out["x2"] = x2_np
    out["t2_u8"] = t2_u8
    x = torch.from_numpy(x_np)
    t = torch.from_numpy(t_u8.astype(np.float32) / 256.0)
    x2 = torch.from_numpy(x2_np)
    t2 = torch.from_numpy(t2_u8.astype(np.float32) / 256.0)

    loss_fn = torch.nn.MSELoss()

    # --- eval-mode forward with the initial running statistics (score path, :223-239)
    enc.eval(); dec.eval()
    with torch.no_grad():
        z = enc(x)
        y_eval = dec(z)
    out["eval0/latent"] = z.numpy().copy()
    out["eval0/y_sub"] = subsample(y_eval.numpy())
    out["eval0/y_proj"] = projections(y_eval.numpy(), 77)
    out["eval0/loss"] = np.float64(loss_fn(y_eval, t).item())

    # --- one train-mode forward/backward (conv_ae_model.py:189-196)
    enc.train(); dec.train()
    z = enc(x)
    y = dec(z)
    loss = loss_fn(y, t)
    for p in list(enc.parameters()) + list(dec.parameters()):
        p.grad = None
    loss.backward()
    out["train0/latent"] = z.detach().numpy().copy()
    out["train0/y_sub"] = subsample(y.detach().numpy())
    out["train0/y_proj"] = projections(y.detach().numpy(), 78)
    if cfg.get("full_output"):
        out["train0/y_full"] = y.detach().numpy().copy()
    out["train0/loss"] = np.float64(loss.item())
    for k, p in enc.named_parameters():
        out[f"train0/grad/enc/{k}"] = p.grad.numpy().copy()
    for k, p in dec.named_parameters():
        out[f"train0/grad/dec/{k}"] = p.grad.numpy().copy()
    # running statistics after exactly one train-mode forward
    for k, v in enc.state_dict().items():
        if "running" in k or "num_batches" in k:
            out[f"train0/buf/enc/{k}"] = v.numpy().copy()
    for k, v in dec.state_dict().items():
        if "running" in k or "num_batches" in k:
            out[f"train0/buf/dec/{k}"] = v.numpy().copy()

    # --- NSTEPS optimiser steps from the INITIAL state (rebuild: same seed => same init)
    torch.manual_seed(cfg["seed"])
    enc = Encoder(spec.get_input_layers(), encoded_space_dim=cfg["latent"], fc_size=cfg["fc"])
    dec = Decoder(spec.get_output_layers(), encoded_space_dim=cfg["latent"], fc_size=cfg["fc"])
    optim = torch.optim.Adam([{"params": enc.parameters()}, {"params": dec.parameters()}],
                             lr=LR, weight_decay=WD)
    enc.train(); dec.train()
    losses = []
    batches = [(x, t), (x2, t2)]
    for step in range(NSTEPS):
        (xb, tb) = batches[step % 2]
        yb = dec(enc(xb))
        loss = loss_fn(yb, tb)
        optim.zero_grad()
        loss.backward()
        optim.step()
        losses.append(loss.item())
    out["steps/loss"] = np.array(losses, dtype=np.float64)
    state_to_np("steps/enc/", enc, out)
    state_to_np("steps/dec/", dec, out)
    # eval forward + test loss with the trained weights and updated running stats (:205-221)
    enc.eval(); dec.eval()
    with torch.no_grad():
        y_eval = dec(enc(x))
    out["steps/eval_y_sub"] = subsample(y_eval.numpy())
    out["steps/eval_y_proj"] = projections(y_eval.numpy(), 79)
    out["steps/eval_loss"] = np.float64(loss_fn(y_eval, t).item())

    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    with open(os.path.join(HERE, f"{name}.json"), "w") as f:
        json.dump(meta, f, indent=1)
    nparam = sum(p.numel() for p in enc.parameters()) + sum(p.numel() for p in dec.parameters())
    print(f"{name}: {nparam} params, losses {losses}")


# ----------------------------------------------------------------------------------
# model_sizer sweep (integer logic, model_sizer.py:112-162)
# ----------------------------------------------------------------------------------

def sizer_sweep():
    rows = []
    geoms = [((16, 16), (256, 256)), ((6, 6), (256, 256)), ((24, 20), (280, 256)), ((7, 7), (28, 28)),
             ((64, 64), (512, 512)), ((21, 18), (77, 61)), ((20, 20), (64, 64)), ((32, 32), (64, 64)),
             ((256, 256), (256, 256)), ((5, 9), (33, 47)), ((100, 60), (300, 200)), ((3, 3), (8, 8)),
             ((12, 12), (100, 100)), ((16, 16), (255, 255)), ((16, 16), (257, 250))]
    for (isz, osz) in geoms:
        for (ic, oc) in [(1, 1), (2, 1), (3, 3)]:
            for (k, s) in [(3, 2), (5, 2), (3, 3), (4, 2), (2, 2), (3, 1)]:
                for (ilc, olc) in [(None, None), (1, 2), (2, None), (None, 3)]:
                    if s == 1 and (ilc is None or olc is None):
                        continue  # stride 1 without explicit counts makes very deep specs; skip
                    try:
                        spec = create_model_spec(input_size=isz, input_channels=ic, output_size=osz,
                                                 output_channels=oc, stride=s, kernel_size=k,
                                                 input_layer_count=ilc, output_layer_count=olc)
                        result = spec.save()
                    except Exception as ex:  # record that the reference raises
                        result = {"raises": type(ex).__name__}
                    rows.append({"args": {"input_size": list(isz), "input_channels": ic,
                                          "output_size": list(osz), "output_channels": oc, "stride": s,
                                          "kernel_size": k, "input_layer_count": ilc,
                                          "output_layer_count": olc},
                                 "spec": result})
    # LayerSpec/ModelSpec repr text is part of summary.txt (conv_ae_model.py:362-380)
    spec = create_model_spec(input_size=(24, 20), input_channels=1, output_size=(280, 256), output_channels=1)
    reprs = {"circle2_repr": repr(spec)}
    hs = ModelSpec()
    hs.load(HANDSPEC)
    reprs["handspec_repr"] = repr(hs)
    reprs["handspec_roundtrip"] = hs.save()
    with gzip.open(os.path.join(HERE, "model_sizer.json.gz"), "wt") as f:
        json.dump({"rows": rows, "reprs": reprs}, f)
    print(f"model_sizer: {len(rows)} rows")


# ----------------------------------------------------------------------------------
# DSDataset (ds_dataset.py:22-159) on a duck-typed stub dataset
# ----------------------------------------------------------------------------------

class StubVar:
    """the attributes DSDataset touches on an xarray.DataArray: shape/values/data/size/[]"""

    def __init__(self, arr):
        self._a = arr

    @property
    def shape(self):
        return self._a.shape

    @property
    def values(self):
        return self._a

    @property
    def data(self):
        return self._a

    @property
    def size(self):
        return self._a.size

    def __getitem__(self, key):
        return StubVar(self._a[key])


def dataset_case():
    rng = np.random.default_rng(4242)
    n = 7
    lowres = (288 + 10 * rng.random((n, 1, 6, 5))).astype(np.float32)
    tide = np.broadcast_to(rng.random((n, 1, 1, 1)).astype(np.float32) * 2 - 1, (n, 1, 6, 5)).copy()
    const = np.full((n, 2, 6, 5), 3.25, dtype=np.float32)  # range 0 -> normalises to 0.0 (:102-103)
    hires = (288 + 10 * rng.random((n, 1, 24, 20))).astype(np.float32)
    ds = {"lowres": StubVar(lowres), "tide": StubVar(tide), "const": StubVar(const), "hires": StubVar(hires)}
    names = ["lowres", "tide", "const"]
    d = DSDataset(ds, names, "hires", normalise_in=True, normalise_out=True)
    out = {"lowres": lowres, "tide": tide, "const": const, "hires": hires}
    params = d.get_normalisation_parameters()
    ins, outs, masks, labels = [], [], [], []
    for i in range(n):
        (a, b, m, lab) = d[i]
        ins.append(a); outs.append(b); masks.append(m); labels.append(lab)
    out["norm_in"] = np.stack(ins)
    out["norm_out"] = np.stack(outs)
    out["mask"] = np.stack(masks)
    y = rng.random((3, 1, 24, 20))  # float64, like score_arr in base_model.py:123
    out["denorm_in"] = y
    out["denorm_out"] = d.denormalise_output(y)
    # normalisation disabled
    d2 = DSDataset(ds, names, "hires", normalise_in=False, normalise_out=False)
    (a, b, m, lab) = d2[2]
    out["raw_in2"] = a
    out["raw_out2"] = b
    np.savez_compressed(os.path.join(HERE, "ds_dataset.npz"), **out)
    with open(os.path.join(HERE, "ds_dataset.json"), "w") as f:
        json.dump({"input_names": names, "output_name": "hires", "normalisation_parameters": params,
                   "labels": labels, "input_shape": list(d.get_input_shape()),
                   "output_shape": list(d.get_output_shape()), "input_spec": d.get_input_spec(),
                   "output_spec": d.get_output_spec()}, f, indent=1)
    # NaN behaviour (:43-46, :56-58)
    bad = hires.copy(); bad[1, 0, 2, 3] = np.nan
    try:
        DSDataset({"lowres": StubVar(lowres), "hires": StubVar(bad)}, ["lowres"], "hires")
        msg = None
    except ValueError as ex:
        msg = str(ex)
    bad_in = lowres.copy(); bad_in[0, 0, 0, 0] = np.nan; bad_in[3, 0, 1, 1] = np.nan
    try:
        DSDataset({"lowres": StubVar(bad_in), "hires": StubVar(hires)}, ["lowres"], "hires")
        msg_in = None
    except ValueError as ex:
        msg_in = str(ex)
    with open(os.path.join(HERE, "ds_dataset_errors.json"), "w") as f:
        json.dump({"nan_output_message": msg, "nan_input_message": msg_in}, f, indent=1)
    print("ds_dataset: params", params)


def main():
    for name, cfg in CASES.items():
        run_case(name, cfg)
    hs = ModelSpec()
    hs.load(HANDSPEC)
    run_case("handspec_b4", dict(fc=12, latent=5, batch=4, seed=18, full_output=True), spec=hs)
    sizer_sweep()
    dataset_case()


if __name__ == "__main__":
    main()
