"""LinearModel path on the GPU (include/cae_linear.h) against vectors from the reference's Linear module and, for the
model / CLI surface, against the oracle driven like linear_model.py drives its module."""
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from test_linear_cpu import CASES, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CASES)
def test_engine_matches_reference_vectors(name):
    from cae_tools_amd.linear_engine import LinearEngine
    (meta, z) = load(name)
    eng = LinearEngine(meta["in_shape"], meta["out_shape"], max_batch=8, device="cuda:0")
    eng.load_state({k: z["init/" + k] for k in meta["keys"]})
    eng.set_hyper(lr=meta["lr"], weight_decay=meta["weight_decay"])
    (x0, t0) = (torch.from_numpy(z["step0/x"]), torch.from_numpy(z["step0/t"]))
    np.testing.assert_allclose(eng.score(x0).cpu().numpy(), z["fwd/y"], rtol=0, atol=2e-6)
    eng.set_dataset(0, x0, t0)
    g = eng.forward_backward(0, None, 0, x0.shape[0], slot=1).cpu().numpy()
    nw = eng.nout * eng.nin
    np.testing.assert_allclose(g[:nw].reshape(eng.nout, eng.nin), z["grad/linear.1.weight"], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(g[nw:], z["grad/linear.1.bias"], rtol=2e-5, atol=1e-9)
    assert eng.read_losses(1, 1)[0] == pytest.approx(float(z["losses"][0]), rel=2e-6)
    for i in range(meta["nsteps"]):
        (x, t) = (torch.from_numpy(z[f"step{i}/x"]), torch.from_numpy(z[f"step{i}/t"]))
        eng.set_dataset(0, x, t)
        eng.train_step(0, None, 0, x.shape[0], slot=10 + i)
    np.testing.assert_allclose(eng.read_losses(10, meta["nsteps"]), z["losses"], rtol=1e-5)
    sd = eng.export_state()
    for k in meta["keys"]:
        # Adam's first steps are +-lr wherever |g| >> eps; agreement to a fraction of lr
        assert np.abs(sd[k].numpy() - z["steps/" + k]).max() <= 2e-5, k


def _data(n, seed):
    from cae_tools_amd.data.arrays import DataArray, Dataset
    rng = np.random.default_rng(seed)
    lo = (280 + 10 * rng.random((n, 1, 8, 8))).astype(np.float32)
    hi = np.repeat(np.repeat(lo, 3, axis=2), 3, axis=3) + rng.standard_normal((n, 1, 24, 24)).astype(np.float32) * 0.1
    ds = Dataset()
    ds["lowres"] = DataArray(lo, dims=("n", "chan", "y", "x"))
    ds["hires"] = DataArray(hi.astype(np.float32), dims=("n", "chan", "y2", "x2"))
    return ds


def test_model_train_save_load_apply_and_cli(tmp_path):
    from cae_tools_amd.cli import apply_cae, train_cae
    from cae_tools_amd.data.arrays import open_dataset
    from cae_tools_amd.models.linear import Linear
    from cae_tools_amd.models.linear_model import LinearModel
    from oracle import cae_oracle as orc
    from oracle.linear_oracle import LinearOracle
    (train, test) = (_data(11, 1), _data(5, 2))
    (_, imin, imax) = orc.scan_variable(train["lowres"].values)
    (_, omin, omax) = orc.scan_variable(train["hires"].values)
    xtr = torch.from_numpy(orc.pack_inputs([train["lowres"].values], [imin], [imax]))
    ttr = torch.from_numpy(orc.normalise_variable(train["hires"].values, omin, omax))
    xte = torch.from_numpy(orc.pack_inputs([test["lowres"].values], [imin], [imax]))
    tte = torch.from_numpy(orc.normalise_variable(test["hires"].values, omin, omax))
    torch.manual_seed(4)
    mod = Linear((1, 8, 8), (1, 24, 24))
    trb = [b for b in torch.utils.data.DataLoader(torch.arange(11), batch_size=4, shuffle=True)]
    teb = [b for b in torch.utils.data.DataLoader(torch.arange(5), batch_size=4, shuffle=True)]
    o = LinearOracle((1, 8, 8), (1, 24, 24), mod.state_dict(), lr=1e-3, weight_decay=1e-5)
    hist = {"train_loss": [], "test_loss": []}
    for epoch in range(3):
        hist["train_loss"].append(float(np.mean([o.train_step(xtr[i], ttr[i]) for i in trb])))
        hist["test_loss"].append(float(np.mean([o.eval_loss(xte[i], tte[i]) for i in teb])))

    torch.manual_seed(4)
    mt = LinearModel(batch_size=4, nr_epochs=3, test_interval=1)
    folder = str(tmp_path / "m")
    with redirect_stdout(io.StringIO()) as out:
        mt.train(["lowres"], "hires", train, test, model_path=folder)
    assert "Running on device: cuda" in out.getvalue() and "Train Metrics" in out.getvalue()
    np.testing.assert_allclose(mt.history["train_loss"], hist["train_loss"], rtol=1e-4)
    np.testing.assert_allclose(mt.history["test_loss"], hist["test_loss"], rtol=1e-4)
    assert sorted(os.listdir(folder)) == sorted(["weights", "normalisation.weights", "parameters.json", "history.json",
                                                 "summary.txt", "input_spec.json", "output_spec.json"])
    sd = torch.load(os.path.join(folder, "weights"), weights_only=True)
    assert list(sd) == ["linear.1.weight", "linear.1.bias"]
    assert np.abs(sd["linear.1.weight"].numpy() - o.state()["linear.1.weight"].numpy()).max() < 5e-5
    with open(os.path.join(folder, "summary.txt")) as f:
        assert f.read() == "Model Summary:\n\tInput shape:\n\t\tsize=(1, 8, 8)\n\tOutput shape:\n\t\tsize=(1, 24, 24)\n"
    m2 = LinearModel()
    m2.load(folder)
    (a, b) = (_data(5, 2), _data(5, 2))
    mt.apply(a, ["lowres"])
    m2.apply(b, ["lowres"])
    np.testing.assert_allclose(a["model_output"].values, b["model_output"].values, rtol=0, atol=1e-9)
    want = omin + o.eval_forward(xte).double().numpy() * (omax - omin)
    np.testing.assert_allclose(a["model_output"].values, want, rtol=0, atol=2e-3 * (omax - omin))
    # CLI: --method linear, then apply_cae picks the model type from parameters.json
    (ptr, pte, f2, out_nc) = (str(tmp_path / n) for n in ("tr.nc", "te.nc", "m2", "scored.nc"))
    train.to_netcdf(ptr)
    test.to_netcdf(pte)
    with redirect_stdout(io.StringIO()):
        train_cae.main(["--train-inputs", ptr, "--test-inputs", pte, "--model-folder", f2, "--input-variables", "lowres",
                        "--output-variable", "hires", "--method", "linear", "--nr-epochs", "2", "--batch-size", "4"])
        apply_cae.main([pte, out_nc, "--model-folder", f2])
    with open(os.path.join(f2, "parameters.json")) as f:
        assert json.load(f)["type"] == "LinearModel"
    assert open_dataset(out_nc)["model_output"].shape == (5, 1, 24, 24)
