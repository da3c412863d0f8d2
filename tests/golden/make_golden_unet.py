#!/usr/bin/env python3
"""Golden vectors for the UNET path from the reference's own class bodies.

Runs ONLY in the build container.  src/cae_tools/models/unet.py cannot be imported as a module here: its
first lines import torchvision and xarray, which the image does not have and which are not stubbed.  The
arithmetic of the path, however, lives in definitions that need torch alone:
    class ChannelAttention, class Encoder, class Decoder                      (unet.py:23-39,73-163)
    UNET.masked_mse_loss, UNET.pearson_corr_torch                             (unet.py:635-678)
This script parses the file, compiles exactly those five definitions from its syntax tree (read at run time
from /root/reference, never stored) in a namespace holding torch / nn / F, and drives them with a loop
equivalent to the reference's training step (unet.py:307-325: zero_grad, forward, masked MSE +
lambda_pearson * (1 - mean Pearson), backward, AdamW.step; AdamW(lr, weight_decay) :457; the cosine schedule
has eta_min == lr, i.e. a constant rate :459).  Nothing else of the file runs (no VGG weights, no transforms).
Stored: layer specs, seeds, inputs, masks, initial state, eval/train outputs, losses, gradients, parameters
and BatchNorm buffers after the AdamW steps.  Train-mode cases use dropout_rate = 0 (the reference's dropout
masks come from torch's global generator and cannot be matched by any other implementation); the dropout > 0
case stores EVAL-mode results only.

    python tests/golden/make_golden_unet.py
"""
import ast
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

REF_SRC = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
if not os.path.isdir(REF_SRC):
    sys.exit("reference not mounted: this script only runs in the build container")
sys.path.insert(0, REF_SRC)
from cae_tools.models.model_sizer import ModelSpec  # noqa: E402

torch.set_num_threads(1)


def load_reference_definitions():
    path = os.path.join(REF_SRC, "cae_tools", "models", "unet.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    wanted_classes = {"ChannelAttention", "Encoder", "Decoder"}
    wanted_methods = {"masked_mse_loss", "pearson_corr_torch"}
    body = []
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in wanted_classes:
            body.append(node)
        if isinstance(node, ast.ClassDef) and node.name == "UNET":
            body.extend(n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in wanted_methods)
    ns = {"torch": torch, "nn": nn, "F": F}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


REF = load_reference_definitions()


def unet_spec(in_ch, out_ch, size, channels, kernel=4, stride=2, pad=1):
    """hand-written layer definitions (cli/train_cae.py:143-147): `output_padding` is what the UNET modules pass
    as `padding` (unet.py:82,140); decoder inputs after the first carry the skip concat (2 x previous output)"""
    (h, w) = size
    enc, dims = [], [(in_ch, h, w)]
    for c in channels:
        (_, ph, pw) = dims[-1]
        dims.append((c, (ph + 2 * pad - kernel) // stride + 1, (pw + 2 * pad - kernel) // stride + 1))
        enc.append({"is_input": True, "kernel_size": kernel, "stride": stride, "output_padding": pad,
                    "input_dimensions": list(dims[-2]), "output_dimensions": list(dims[-1])})
    dec = []
    n = len(channels)
    for j in range(n):
        src = dims[n - j]
        dst = dims[n - j - 1]
        cin = src[0] if j == 0 else 2 * src[0]
        cout = dst[0] if j < n - 1 else out_ch
        oh = (src[1] - 1) * stride - 2 * pad + kernel
        ow = (src[2] - 1) * stride - 2 * pad + kernel
        assert (oh, ow) == (dst[1], dst[2]), "decoder does not land on the skip's size"
        dec.append({"is_input": False, "kernel_size": kernel, "stride": stride, "output_padding": pad,
                    "input_dimensions": [cin, src[1], src[2]], "output_dimensions": [cout, oh, ow]})
    return {"input_layers": enc, "output_layers": dec}


CASES = {
    # the benchmark layer pattern (k4 s2 p1) in miniature, 3 -> 3 channels, mask with one channel
    "u_k4_b3": dict(spec=unet_spec(3, 3, (16, 16), [8, 16, 16]), fc=12, latent=5, batch=3, seed=31, mask="b1hw"),
    # non-square, 2 -> 1 channels, per-channel mask, channel counts that are not multiples of 8 / 16
    "u_rect_b4": dict(spec=unet_spec(2, 1, (24, 16), [8, 24]), fc=10, latent=4, batch=4, seed=32, mask="bchw"),
    # kernel 3 stride 1 pad 1 (size-preserving) layers: the general-geometry path
    "u_k3s1_b2": dict(spec=unet_spec(1, 2, (10, 12), [8, 8], kernel=3, stride=1, pad=1), fc=8, latent=3, batch=2,
                      seed=33, mask="ones"),
    # one encoder / one decoder layer: no skip connection, no attention
    "u_single_b3": dict(spec=unet_spec(2, 2, (8, 8), [8]), fc=6, latent=3, batch=3, seed=34, mask="b1hw"),
    # dropout 0.1 model: eval-mode only
    "u_drop_eval_b3": dict(spec=unet_spec(3, 3, (16, 16), [8, 16]), fc=12, latent=5, batch=3, seed=35, mask="b1hw",
                           dropout=0.1, eval_only=True),
}
LR, WD, LAMBDA_P, NSTEPS = 1e-3, 1e-5, 1.0, 3


def make_batch(rng, spec, b, mask_kind):
    (ic, ih, iw) = spec["input_layers"][0]["input_dimensions"]
    (oc, oh, ow) = spec["output_layers"][-1]["output_dimensions"]
    x = rng.random((b, ic, ih, iw), dtype=np.float32)
    yy, xx = np.meshgrid(np.linspace(-1, 1, oh), np.linspace(-1, 1, ow), indexing="ij")
    t = np.zeros((b, oc, oh, ow), dtype=np.float32)
    for i in range(b):
        for c in range(oc):
            t[i, c] = 0.5 + 0.4 * np.sin(3 * yy * rng.random() + 2 * xx * rng.random() + rng.random())
    t = (t + 0.05 * rng.standard_normal(t.shape)).clip(0, 1).astype(np.float32)
    if mask_kind == "ones":
        m = np.ones((b, oc, oh, ow), dtype=np.float32)
    elif mask_kind == "b1hw":
        m = (rng.random((b, 1, oh, ow)) < 0.8).astype(np.float32)
    else:
        m = (rng.random((b, oc, oh, ow)) < 0.8).astype(np.float32)
    return x, t, m


def state_to_np(prefix, module, out):
    for k, v in module.state_dict().items():
        out[prefix + k] = v.detach().cpu().numpy().copy()


def run_case(name, cfg):
    spec_json = cfg["spec"]
    spec = ModelSpec()
    spec.load(spec_json)
    p = cfg.get("dropout", 0.0)
    torch.manual_seed(cfg["seed"])
    enc = REF["Encoder"](spec.get_input_layers(), encoded_space_dim=cfg["latent"], fc_size=cfg["fc"], dropout_rate=p)
    dec = REF["Decoder"](spec.get_output_layers(), encoded_space_dim=cfg["latent"], fc_size=cfg["fc"], dropout_rate=p)
    out = {}
    state_to_np("init/enc/", enc, out)
    state_to_np("init/dec/", dec, out)
    rng = np.random.default_rng(cfg["seed"])
    batches = [make_batch(rng, spec_json, cfg["batch"] if i % 2 == 0 else max(cfg["batch"] - 1, 2), cfg["mask"])
               for i in range(NSTEPS)]
    (x0, t0, m0) = (torch.from_numpy(a) for a in batches[0])
    out["x0"], out["t0"], out["m0"] = batches[0]

    def forward(x):
        (z, skip) = enc(x)
        return dec(z, skip)

    def losses(y, t, m):
        mse = REF["masked_mse_loss"](None, y, t, m)
        pl = 1 - torch.mean(REF["pearson_corr_torch"](None, y, t, m))
        return mse, pl

    enc.eval(), dec.eval()
    with torch.no_grad():
        y = forward(x0)
        (mse, pl) = losses(y, t0, m0)
    out["eval/y"] = y.numpy().copy()
    out["eval/losses"] = np.array([float(mse), float(pl)])
    with torch.no_grad():
        out["eval/pearson"] = REF["pearson_corr_torch"](None, y, t0, m0).numpy().copy()
    meta = {"name": name, "spec": spec_json, "fc": cfg["fc"], "latent": cfg["latent"], "batch": cfg["batch"],
            "seed": cfg["seed"], "dropout": p, "lr": LR, "weight_decay": WD, "lambda_pearson": LAMBDA_P,
            "nsteps": 0 if cfg.get("eval_only") else NSTEPS, "mask": cfg["mask"],
            "enc_keys": list(enc.state_dict().keys()), "dec_keys": list(dec.state_dict().keys())}
    if not cfg.get("eval_only"):
        params = list(enc.parameters()) + list(dec.parameters())
        optim = torch.optim.AdamW(params, lr=LR, weight_decay=WD)
        step_losses = []
        for (i, (x, t, m)) in enumerate(batches):
            (x, t, m) = (torch.from_numpy(a) for a in (x, t, m))
            out[f"step{i}/x"], out[f"step{i}/t"], out[f"step{i}/m"] = batches[i]
            enc.train(), dec.train()
            optim.zero_grad()
            y = forward(x)
            (mse, pl) = losses(y, t, m)
            (mse + LAMBDA_P * pl).backward()
            if i == 0:
                out["train/y"] = y.detach().numpy().copy()
                for (pre, mod) in (("enc/", enc), ("dec/", dec)):
                    for k, v in mod.named_parameters():
                        out["grad/" + pre + k] = v.grad.detach().numpy().copy()
            optim.step()
            step_losses.append([float(mse), float(pl)])
            if i == 0:
                state_to_np("step1/enc/", enc, out)
                state_to_np("step1/dec/", dec, out)
        state_to_np("steps/enc/", enc, out)
        state_to_np("steps/dec/", dec, out)
        out["step_losses"] = np.array(step_losses)
        print(f"{name}: losses {step_losses}")
    nparam = sum(p_.numel() for p_ in enc.parameters()) + sum(p_.numel() for p_ in dec.parameters())
    meta["params"] = nparam
    np.savez_compressed(os.path.join(HERE, f"unet_{name}.npz"), **out)
    with open(os.path.join(HERE, f"unet_{name}.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(f"{name}: {nparam} params, eval losses {out['eval/losses']}")


if __name__ == "__main__":
    for (name, cfg) in CASES.items():
        run_case(name, cfg)
