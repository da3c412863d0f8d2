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


def assert_close_up_to_relu_flips(got, ref32, exact64, name, factor=3.0, floor_rel=1e-5, flip_rel=1e-3, q=0.9, unit=None):
    """assert_close_as_reference for LARGE batches, where single ReLU flips are part of fp32 arithmetic: with ~1e6 BatchNorm
    outputs per layer and a relative rounding noise of 1e-7, some output lies within the noise of zero in about one layer
    per step, and two correct fp32 implementations then disagree on that one mask bit.  The flipped position changes the
    BatchNorm-backward sums of its channel, i.e. shifts that channel's whole gradient slice by ~1e-5 of the tensor's
    maximum, and adds its own product to a few taps (measured on the benchmark geometry at batch 64: one flip at the 8->4
    layer's output moved a quarter of that layer's weight gradient by 8e-6 of its maximum and two taps by 1e-4, while the
    fp32 oracle, which did not flip there, sat at 7e-7 - tools/diag_step1.py).  The reference's own error cannot predict
    the other implementation's flips, so the bound has two parts:
      * the q-quantile of |got - exact| over the tensor's elements <= factor * the same quantile of the reference's own error
        + floor_rel * max|exact|   (what a systematic error - a wrong constant, a stale operand - cannot hide from);
      * the maximum <= factor * the reference's maximum + flip_rel * max|exact|   (room for isolated flips).
    `unit`: express both floors in this absolute unit instead of max|exact| (parameter updates: the learning rate)."""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    ref32 = np.asarray(ref32, dtype=np.float64).reshape(-1)
    exact64 = np.asarray(exact64, dtype=np.float64).reshape(-1)
    scale = float(unit) if unit is not None else float(np.abs(exact64).max())
    (e_got, e_ref) = (np.abs(got - exact64), np.abs(ref32 - exact64))
    (qg, qr) = (float(np.quantile(e_got, q)), float(np.quantile(e_ref, q)))
    # (a tensor of a few elements has no quantile apart from its maximum: BatchNorm vectors of 2..32 channels)
    assert e_got.size < 32 or qg <= factor * qr + floor_rel * scale + 1e-12, \
        f"{name}: {int(q * 100)}th percentile of |hip-exact| = {qg:.3e} > {factor} x {qr:.3e} (the reference's) + {floor_rel * scale:.3e}"
    (mg, mr) = (float(e_got.max()), float(e_ref.max()))
    assert mg <= factor * mr + flip_rel * scale + 1e-12, \
        f"{name}: max |hip-exact| = {mg:.3e} > {factor} x {mr:.3e} (the reference's) + {flip_rel * scale:.3e}"
    return mg / (factor * mr + flip_rel * scale + 1e-12)
