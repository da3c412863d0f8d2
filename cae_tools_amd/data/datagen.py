"""Seeded synthetic data in the shape of the reference's test generator.

Same patterns, value ranges, variable names and dimension names as test/datagen/gen.py:24-103
and test/test_specs.py (ring on a tilted grid, block-mean coarsening of one high-resolution
field into the low- and high-resolution variables), but with explicit seeds (the reference
draws from the unseeded `random` module) and NetCDF-3 output.  Used by bench.py, the tests and
`python -m cae_tools_amd.data.datagen <folder>`.
"""
import math
import os
import random

import numpy as np

from .arrays import DataArray, Dataset

SPECS = {
    "circle": dict(input_size=(16, 16), output_size=(256, 256), inputs=["lowres"], output="hires", pattern="circle"),
    "tidal_circle1": dict(input_size=(6, 6), output_size=(256, 256), inputs=["lowres", "tide_3d"], output="hires",
                          pattern="tidal_circle"),
    "curve1": dict(input_size=(16, 16), output_size=(256, 256), inputs=["lowres"], output="hires", pattern="curve"),
    "circle2": dict(input_size=(24, 20), output_size=(280, 256), inputs=["lowres"], output="hires", pattern="circle"),
}


def _lcm(a, b):
    return a * b // math.gcd(a, b)


def _block_mean(arr, fy, fx):
    (h, w) = arr.shape
    return arr.reshape(h // fy, fy, w // fx, fx).mean(axis=(1, 3))


def _pattern(kind, height, width, rng, mu=1.0):
    """(field, aux) for one sample; aux = tidal height or None"""
    from scipy import ndimage
    if kind == "circle":
        (yy, xx) = np.meshgrid(np.linspace(-2, 2, width), np.linspace(-3, 3, height))
        d = np.sqrt(yy * yy + xx * xx)
        g = np.exp(-((d - mu) ** 2 / (2.0 * 0.2 ** 2)))
        return ndimage.rotate(g, 15)[0:height, 0:width], None
    if kind == "tidal_circle":
        tide = math.sin(rng.random() * 2 * math.pi)
        (yy, xx) = np.meshgrid(np.linspace(-8, 8, width), np.linspace(-10, 10, height))
        d = np.sqrt(yy * yy + xx * xx)
        sigma = 0.2 + 0.1 * tide
        g = np.exp(-((d - mu) ** 2 / (2.0 * sigma ** 2)))
        return ndimage.rotate(g, 15)[0:height, 0:width], tide
    if kind == "curve":
        (yy, xx) = np.meshgrid(np.linspace(0, 100, width), np.linspace(0, 100, height))
        return np.sqrt((yy - 50) ** 2 + (xx - 50) ** 2) / math.sqrt(50 ** 2 + 50 ** 2), None
    raise ValueError(kind)


def generate(spec_name, n, seed):
    """Dataset with variables spec['output'] (n,1,H,W), spec['inputs'][0] (n,1,h,w) (+ aux inputs)"""
    spec = SPECS[spec_name]
    (ih, iw), (oh, ow) = spec["input_size"], spec["output_size"]
    (sh, sw) = (_lcm(oh, ih), _lcm(ow, iw))
    rng = random.Random(seed)
    lo = np.zeros((n, 1, ih, iw), dtype=np.float32)
    hi = np.zeros((n, 1, oh, ow), dtype=np.float32)
    tides = np.zeros((n,), dtype=np.float32)
    cache = None
    for i in range(n):
        if spec["pattern"] == "tidal_circle" or cache is None:
            (field, aux) = _pattern(spec["pattern"], sh, sw, rng)
            cache = field
        else:
            (field, aux) = (cache, None)   # the circle / curve field does not depend on the sample
        if aux is not None:
            tides[i] = aux
        arr = 288 + 5 * rng.random() + field * rng.random() * 5
        lo[i, 0] = _block_mean(arr, sh // ih, sw // iw)
        hi[i, 0] = _block_mean(arr, sh // oh, sw // ow)
    ds = Dataset()
    ds[spec["output"]] = DataArray(hi, dims=("n", "chan", "y2", "x2"))
    ds[spec["inputs"][0]] = DataArray(lo, dims=("n", "chan", "y1", "x1"))
    if spec["pattern"] == "tidal_circle":
        ds["tide_1d"] = DataArray(tides, dims=("n",), attrs={"type": "auxilary-predictor", "min-value": -1.0,
                                                              "max-value": 1.0})
        ds[spec["inputs"][1]] = DataArray(np.broadcast_to(tides.reshape(n, 1, 1, 1), (n, 1, ih, iw)).copy(),
                                          dims=("n", "chan", "y1", "x1"))
    return ds


def write_all(root, n=100, train_seed=1234, test_seed=4321):
    for name, spec in SPECS.items():
        (ih, iw), (oh, ow) = spec["input_size"], spec["output_size"]
        folder = os.path.join(root, name, f"{ih}x{iw}_{oh}x{ow}")
        for fname, seed in (("train.nc", train_seed), ("test.nc", test_seed)):
            path = os.path.join(folder, fname)
            if not os.path.exists(path):
                generate(name, n, seed).to_netcdf(path)
                print("Written", path)


if __name__ == "__main__":
    import sys
    write_all(sys.argv[1] if len(sys.argv) > 1 else "test_data")
