"""ModelMetric parity: tests/golden/model_metric.npz holds seeded truth / normalised
scores / masks and the metrics the reference's ModelMetric (model_metric.py:25-71) returned for them."""
import os

import numpy as np
import pytest

from cae_tools_amd.models.model_metric import ModelMetric, metrics_from_sums

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_metric.npz"))
KEYS = ("mse", "rmse", "mae", "mean_pearson_correlation")


def _expected(tag):
    return {k: float(G[f"{tag}/{k}"]) for k in KEYS}


def _scores():
    return float(G["vmin"]) + (G["y"].astype(np.float64) * (float(G["vmax"]) - float(G["vmin"])))


@pytest.mark.parametrize("tag", ["masked", "all"])
def test_host_model_metric_matches_reference(tag):
    mask = G["mask"] if tag == "masked" else np.ones_like(G["mask"])
    mm = ModelMetric()
    for i in range(G["truth"].shape[0]):
        mm.accumulate(G["truth"][i], _scores()[i], mask[i])
    got = mm.get_metrics()
    for k in KEYS:
        assert got[k] == pytest.approx(_expected(tag)[k], rel=1e-13, abs=0)


def test_host_model_metric_errors():
    mm = ModelMetric()
    with pytest.raises(ValueError, match="No data accumulated"):
        mm.get_metrics()
    with pytest.raises(ValueError, match="must match"):
        mm.accumulate(np.zeros((2, 2)), np.zeros((2, 3)), np.ones((2, 2)))


@pytest.mark.parametrize("tag", ["masked", "all"])
def test_sums_to_metrics_matches_reference(tag):
    """the host half of the device path, fed with sums formed in numpy exactly as k_metric_sums defines them"""
    mask = G["mask"] if tag == "masked" else np.ones_like(G["mask"])
    vmin = float(G["vmin"])
    rows = []
    for i in range(G["truth"].shape[0]):
        keep = mask[i].reshape(-1) != 0
        a = G["truth"][i].reshape(-1)[keep].astype(np.float64)
        e = _scores()[i].reshape(-1)[keep]
        (a_, e_, d) = (a - vmin, e - vmin, a - e)
        rows.append([keep.sum(), a_.sum(), e_.sum(), (a_ * a_).sum(), (e_ * e_).sum(), (a_ * e_).sum(),
                     np.abs(d).sum(), (d * d).sum()])
    got = metrics_from_sums(np.array(rows))
    for k in KEYS:
        assert got[k] == pytest.approx(_expected(tag)[k], rel=1e-11, abs=0)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["masked", "all"])
def test_device_model_metric_matches_reference(tag):
    import torch
    from cae_tools_amd.models.model_metric import DeviceModelMetric
    dev = torch.device("cuda:0")
    mask = torch.from_numpy(G["mask"]).to(dev) if tag == "masked" else None
    mm = DeviceModelMetric()
    (truth, y) = (torch.from_numpy(G["truth"]).to(dev), torch.from_numpy(G["y"]).to(dev))
    # two accumulate calls (as evaluate would do per data set chunk)
    mm.accumulate(truth[:4], y[:4], None if mask is None else mask[:4], float(G["vmin"]), float(G["vmax"]))
    mm.accumulate(truth[4:], y[4:], None if mask is None else mask[4:], float(G["vmin"]), float(G["vmax"]))
    got = mm.get_metrics()
    # fp64 sums over <= 480 pixels per case: only the summation order differs from numpy's
    for k in KEYS:
        assert got[k] == pytest.approx(_expected(tag)[k], rel=1e-11, abs=0)


@pytest.mark.gpu
def test_device_metric_large_instances_and_counts():
    """instances larger than one block sweep (several chunks + grid-stride), count column exact"""
    import torch
    from cae_tools_amd import engine as eng
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    (n, c, h, w) = (3, 2, 300, 301)
    y = torch.rand((n, c, h, w), generator=g)
    a = 10.0 + 5.0 * torch.rand((n, c, h, w), generator=g)
    m = (torch.rand((n, c, h, w), generator=g) < 0.5).float()
    s = eng.metric_sums(y.to(dev), a.to(dev), m.to(dev), 10.0, 15.0).cpu().numpy()
    e = 10.0 + y.double().numpy() * 5.0
    ad = a.double().numpy()
    keep = m.numpy() != 0
    for i in range(n):
        k = keep[i]
        assert s[i, 0] == k.sum()
        np.testing.assert_allclose(s[i, 1], (ad[i][k] - 10.0).sum(), rtol=1e-12)
        np.testing.assert_allclose(s[i, 5], ((ad[i][k] - 10.0) * (e[i][k] - 10.0)).sum(), rtol=1e-12)
        np.testing.assert_allclose(s[i, 6], np.abs(ad[i][k] - e[i][k]).sum(), rtol=1e-12)
        np.testing.assert_allclose(s[i, 7], ((ad[i][k] - e[i][k]) ** 2).sum(), rtol=1e-12)
