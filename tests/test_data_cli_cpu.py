"""CPU-side host logic: named-array containers, NetCDF-3 round trip, seeded data generator,
CLI flag surface (against the reference's flag table), ModelMetric, DP shard bounds."""
import os

import numpy as np
import pytest

from cae_tools_amd.data import datagen
from cae_tools_amd.data.arrays import DataArray, Dataset, open_dataset, open_mfdataset


def test_netcdf3_roundtrip_and_concat(tmp_path):
    ds = datagen.generate("circle2", 4, seed=3)
    assert ds["lowres"].shape == (4, 1, 24, 20) and ds["hires"].shape == (4, 1, 280, 256)
    assert ds["hires"].dims == ("n", "chan", "y2", "x2") and ds["lowres"].dims == ("n", "chan", "y1", "x1")
    p1, p2 = str(tmp_path / "a.nc"), str(tmp_path / "b.nc")
    ds.to_netcdf(p1)
    back = open_dataset(p1)
    for k in ds:
        assert back[k].dims == ds[k].dims and back[k].dtype == np.float32
        np.testing.assert_array_equal(back[k].values, ds[k].values)
    # multi-file nested concat along an existing case dimension
    box = Dataset()
    box["v"] = DataArray(np.arange(24, dtype=np.float32).reshape(2, 1, 3, 4), dims=("box", "channel", "y", "x"))
    box["s"] = DataArray(np.array([1.5, 2.5], dtype=np.float32), dims=("box",))
    box.to_netcdf(p1)
    box.to_netcdf(p2)
    cat = open_mfdataset([p1, p2], concat_dim="box", combine="nested")
    assert cat["v"].shape == (4, 1, 3, 4) and cat["s"].shape == (4,) and cat.dims["y"] == 3
    assert cat["box"].shape == (4,)


def test_datagen_is_seeded_and_in_range():
    a = datagen.generate("circle", 6, seed=11)
    b = datagen.generate("circle", 6, seed=11)
    c = datagen.generate("circle", 6, seed=12)
    np.testing.assert_array_equal(a["hires"].values, b["hires"].values)
    assert not np.array_equal(a["hires"].values, c["hires"].values)
    # gen.py: 288 + 5*u1 + pattern*u2*5 with pattern in [0,1]  ->  [288, 298]
    assert 288.0 <= float(a["hires"].values.min()) and float(a["hires"].values.max()) <= 298.0
    # the low-res variable is the block mean of the same field: means agree
    np.testing.assert_allclose(a["lowres"].values.mean(axis=(1, 2, 3)), a["hires"].values.mean(axis=(1, 2, 3)), rtol=1e-5)
    t = datagen.generate("tidal_circle1", 3, seed=1)
    assert t["tide_3d"].shape == (3, 1, 6, 6) and np.all(np.abs(t["tide_1d"].values) <= 1.0)
    np.testing.assert_array_equal(t["tide_3d"].values[:, 0, 0, 0], t["tide_1d"].values)


# flag -> default, transcribed from the reference's argparse table (cli/train_cae.py:19-53)
REFERENCE_TRAIN_FLAGS = {
    "--train-inputs": None, "--test-inputs": None, "--model-folder": None, "--continue-training": False,
    "--input-variables": None, "--output-variable": None, "--nr-epochs": 500, "--latent-size": 4, "--fc-size": 16,
    "--batch-size": 10, "--learning-rate": 0.001, "--lr-step-size": 500, "--lr-gamma": 0.5, "--lambda-mse": 1,
    "--lambda-kl": 1, "--lambda-l1": 0.001, "--lambda-pearson": 1, "--lambda-ssim": 1, "--lambda-additional": 1,
    "--weight-decay": 1e-5, "--dropout-rate": 1e-1, "--additional-loss": None, "--scheduler-type": None,
    "--method": "var", "--layer-definitions-path": None, "--stride": 2, "--kernel-size": 3,
    "--input-layer-count": None, "--output-layer-count": None, "--model-id": None, "--database-path": None,
    "--chunk-size": 1000, "--include-coasts": False, "--mask-variable": None,
}


def test_train_cli_flag_surface():
    from cae_tools_amd.cli.train_cae import build_parser
    p = build_parser()
    got = {}
    for a in p._actions:
        for o in a.option_strings:
            if o.startswith("--") and o != "--help":
                got[o] = a.default
    build_only = {"--gpus": 1, "--local-bn": False}     # data-parallel flags the reference does not have
    assert {k: v for k, v in got.items() if k in build_only} == build_only
    assert {k: v for k, v in got.items() if k not in build_only} == REFERENCE_TRAIN_FLAGS
    required = {o for a in p._actions if a.required for o in a.option_strings}
    assert required == {"--train-inputs", "--test-inputs", "--model-folder", "--input-variables", "--output-variable"}
    args = p.parse_args(["--train-inputs", "a", "b", "--test-inputs", "c", "--model-folder", "m", "--input-variables",
                         "x", "y", "--output-variable", "z", "--method", "conv"])
    assert args.train_inputs == ["a", "b"] and args.input_variables == ["x", "y"] and args.method == "conv"


def test_apply_cli_flag_surface():
    from cae_tools_amd.cli.apply_cae import build_parser
    p = build_parser()
    a = p.parse_args(["in1.nc", "in2.nc", "out.nc", "--model-folder", "m"])
    assert a.data_paths == ["in1.nc", "in2.nc"] and a.output_path == "out.nc"
    assert a.prediction_variable == "model_output" and a.input_variables is None and a.mask_variable is None
    assert a.gpus == 1


def test_gpus_flag_spawns_one_rank_per_gpu(monkeypatch):
    """--gpus N re-runs the command under torch.distributed.run as a CHILD process (this process never touches the GPU),
    with the rendezvous on 127.0.0.1; inside a rank it is a no-op, and only rank 0 keeps its stdout"""
    import sys
    from cae_tools_amd.cli import _launch
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"] = cmd
        seen["env"] = env
        return 7

    monkeypatch.setattr(_launch.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert _launch.maybe_spawn_ranks("cae_tools_amd.cli.train_cae", 1, ["--x"]) is None and not seen
    assert _launch.maybe_spawn_ranks("cae_tools_amd.cli.train_cae", 4, ["--method", "conv", "--gpus", "4"]) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["-m", "cae_tools_amd.cli.train_cae", "--method", "conv", "--gpus", "4"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    seen.clear()
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    assert _launch.maybe_spawn_ranks("cae_tools_amd.cli.train_cae", 4, []) is None and not seen
    out = sys.stdout
    monkeypatch.setenv("RANK", "2")
    try:
        assert _launch.maybe_spawn_ranks("cae_tools_amd.cli.train_cae", 4, []) is None
        assert sys.stdout is not out
    finally:
        sys.stdout = out


def test_model_metric_matches_direct_formulas():
    from scipy.stats import pearsonr
    from cae_tools_amd.models.model_metric import ModelMetric
    rng = np.random.default_rng(0)
    mm = ModelMetric()
    acts, ests, rs = [], [], []
    for i in range(4):
        a = rng.random((1, 6, 5)) * 10 + 288
        e = a + rng.normal(0, 0.3, a.shape)
        m = np.ones_like(a)
        m[0, 0, :i] = 0
        mm.accumulate(a, e, m)
        keep = m.reshape(-1).astype(bool)
        acts.append(a.reshape(-1)[keep]); ests.append(e.reshape(-1)[keep])
        rs.append(pearsonr(acts[-1], ests[-1])[0])
    got = mm.get_metrics()
    A, E = np.concatenate(acts), np.concatenate(ests)
    assert got["mse"] == pytest.approx(np.mean((A - E) ** 2)) and got["rmse"] == pytest.approx(np.sqrt(got["mse"]))
    assert got["mae"] == pytest.approx(np.mean(np.abs(A - E))) and got["mean_pearson_correlation"] == pytest.approx(np.mean(rs))
    with pytest.raises(ValueError):
        mm.accumulate(np.zeros((2, 2)), np.zeros((3, 2)), np.ones((2, 2)))
    with pytest.raises(ValueError):
        ModelMetric().get_metrics()


def test_shard_bounds_cover_the_batch():
    from cae_tools_amd.dp import shard_bounds
    for n in (1, 7, 36, 64, 100, 512):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
