"""Shared helpers for the parity tests: golden-fixture access and the three small
functions tests/golden/make_golden.py used to build inputs and digests."""
import gzip
import json
import os
from collections import OrderedDict

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

MODEL_CASES = ["cfg1_b3", "cfg2_b4", "circle2_b2", "tidal_b3", "odd_k5_b5", "s3_b4", "counts_b3",
               "handspec_b4"]


def projections(arr, seed, nproj=8):
    flat = np.asarray(arr, dtype=np.float64).reshape(-1)
    rng = np.random.default_rng(seed)
    out = np.zeros(nproj, dtype=np.float64)
    for k in range(nproj):
        signs = rng.integers(0, 2, size=flat.size, dtype=np.int8).astype(np.float64) * 2.0 - 1.0
        out[k] = float(np.dot(flat, signs))
    return out


def subsample(arr, step=5):
    return np.ascontiguousarray(np.asarray(arr)[..., ::step, ::step])


class GoldenCase:
    def __init__(self, name):
        self.name = name
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            self.meta = json.load(f)
        self.npz = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.spec = self.meta["spec"]

    def __getitem__(self, key):
        return self.npz[key]

    def keys(self):
        return self.npz.files

    def group(self, prefix):
        """OrderedDict of arrays under `prefix`, keys stripped of it, in file order"""
        out = OrderedDict()
        for k in self.npz.files:
            if k.startswith(prefix):
                out[k[len(prefix):]] = self.npz[k]
        return out

    @property
    def x(self):
        return self.npz["x"]

    @property
    def t(self):
        return self.npz["t_u8"].astype(np.float32) / 256.0

    @property
    def x2(self):
        return self.npz["x2"]

    @property
    def t2(self):
        return self.npz["t2_u8"].astype(np.float32) / 256.0


def load_sizer_sweep():
    with gzip.open(os.path.join(GOLDEN, "model_sizer.json.gz"), "rt") as f:
        return json.load(f)


def bn_bias_keys(spec):
    """state_dict keys of conv biases that feed a BatchNorm: their gradient is exactly zero in
    exact arithmetic (BN removes any per-channel constant), so the reference's value is pure
    rounding noise and Adam turns it into +-lr steps of arbitrary sign.  Parity for these keys
    is checked on magnitude only (see DESIGN.md 'Bias before BatchNorm')."""
    keys = set()
    for i in range(len(spec["input_layers"])):
        keys.add(f"enc/encoder_cnn.{3 * i}.bias")
    n = len(spec["output_layers"])
    for i in range(n - 1):
        keys.add(f"dec/decoder_conv.{3 * i}.bias")
    return keys


def oracle_model(case, dtype="float32"):
    """OracleModel on the case's initial state; dtype float64 gives the 'exact' answer used to
    express tolerances as a multiple of the reference's own fp32 error"""
    import torch
    from oracle import cae_oracle as orc
    def conv(group):
        out = OrderedDict()
        for k, v in group.items():
            t = torch.as_tensor(np.array(v))
            out[k] = t.double() if (dtype == "float64" and t.is_floating_point()) else t
        return out
    return orc.OracleModel(case.spec, conv(case.group("init/enc/")), conv(case.group("init/dec/")),
                           lr=case.meta["lr"], weight_decay=case.meta["weight_decay"])


def assert_close_as_reference(got, ref32, exact64, name, factor=3.0, floor_rel=1e-5, floor_abs=1e-9):
    """|got - exact| <= factor * |reference_fp32 - exact| (max norm) + floor_rel * max|exact| + floor_abs:
    the HIP result may be no further from the fp64 answer than `factor` times the reference's
    own fp32 rounding error on the same tensor."""
    got = np.asarray(got, dtype=np.float64)
    ref32 = np.asarray(ref32, dtype=np.float64)
    exact64 = np.asarray(exact64, dtype=np.float64)
    scale = float(np.abs(exact64).max())
    err_ref = float(np.abs(ref32 - exact64).max())
    err_got = float(np.abs(got - exact64).max())
    bound = factor * err_ref + floor_rel * scale + floor_abs
    assert err_got <= bound, f"{name}: |hip-exact|={err_got:.3e} > {bound:.3e} (reference's own error {err_ref:.3e}, scale {scale:.3e})"


def hip_relu_decisions(eng, batch):
    """{oracle ReLU name: bool array} - which ReLUs of the engine's LAST train-mode step passed their input.  BatchNorm'd conv
    layers: the masked gradient the backward kernels leave in the workspace is zero exactly where the mask is (test hook
    cae_debug_read "grad"); Linear layers: the stored post-ReLU activation is positive exactly where it is not."""
    import torch
    n_enc = len(eng.enc_layers)
    out = {}
    layers = [(f"enc_conv{i}", i, l) for i, l in enumerate(eng.enc_layers)]
    layers += [(f"dec_conv{i}", n_enc + i, l) for i, l in enumerate(eng.dec_layers[:-1])]
    for (name, idx, l) in layers:
        (c, h, w) = l["output_dimensions"]
        out[name] = torch.from_numpy(eng.debug_read("grad", idx, count=batch * c * h * w).reshape(batch, c, h, w) != 0)
    for (name, idx) in (("enc_fc0", 0), ("dec_fc0", 2)):
        out[name] = torch.from_numpy(eng.debug_read("fc", idx, count=batch * eng.fc_size).reshape(batch, eng.fc_size) > 0)
    return out


def relu_fix_for(oracle, x, decisions, what, z_tol=2e-5, max_flips=16):
    """relu_fix argument (oracle/cae_oracle.py _relu) that gives `oracle` the ReLU decisions `decisions` (hip_relu_decisions)
    on input batch x - after checking that every position where the oracle decides otherwise is one it cannot decide: its own
    pre-activation there is rounding-sized (<= z_tol of the layer's scale), and there are only a handful of them.
    Why: at the benchmark batch a layer has up to 2e6 BatchNorm outputs and fp32 noise of ~1e-7, so about once per step one
    of them lies within the noise of zero and two correct fp32 implementations disagree on that mask bit; because a weight
    gradient is a sum of ~N terms that cancel to ~sqrt(N) of one term, ONE flipped bit moves every upstream gradient by
    1e-4..1e-3 of its maximum (measured: tools/diag_dp64.py - a single mismatch at the 4->2 layer, |z64| = 7e-8, put the HIP
    path 1e-3 from an fp64 oracle that the fp32 oracle, which happened not to flip there, followed to 1e-6).  With the
    decisions aligned what is left is rounding, and the strict fp64-anchored bound applies again.  Returns (fix, flips)."""
    import torch
    z = oracle.relu_inputs(x)
    (fix, flips) = ({}, 0)
    for name, passed in decisions.items():
        zk = z[name]
        d = passed.to(zk.dtype) - (zk > 0).to(zk.dtype)
        differs = d != 0
        n = int(differs.sum())
        if n:
            worst = float(zk[differs].abs().max())
            scale = max(1.0, float(zk.abs().max()) / 8.0)
            assert worst <= z_tol * scale, f"{what}: ReLU decision differs at {name} where the oracle's input is {worst:.3e} - not a rounding-sized input"
            fix[name] = d
            flips += n
    assert flips <= max_flips, f"{what}: {flips} ReLU decisions differ from the oracle's"
    return fix, flips
