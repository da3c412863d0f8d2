"""UNET drop-in surface on the GPU (unet.py:200-633): train() / save() / load() / apply() / CLI against the CPU oracle
driven the way the reference drives its modules (same seed, same shuffles, dropout 0 so that no mask generator matters)."""
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _data(n, seed, cin=2, cout=1, size=16):
    from cae_tools_amd.data.arrays import DataArray, Dataset
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, size), np.linspace(-1, 1, size), indexing="ij")
    x = np.zeros((n, cin, size, size), dtype=np.float32)
    y = np.zeros((n, cout, size, size), dtype=np.float32)
    for i in range(n):
        (a, b, c) = rng.random(3)
        base = 280 + 10 * np.sin(3 * a * yy + 2 * b * xx + 6 * c)
        for k in range(cin):
            x[i, k] = base + rng.standard_normal((size, size)) * (0.5 + k)
        for k in range(cout):
            y[i, k] = base * (1 + 0.01 * k)
    ds = Dataset()
    ds["lo"] = DataArray(x, dims=("n", "chan_in", "y", "x"))
    ds["hi"] = DataArray(y, dims=("n", "chan_out", "y", "x"))
    ds["valid"] = DataArray((rng.random((n, 1, size, size)) < 0.85).astype(np.float32), dims=("n", "one", "y", "x"))
    return ds


def _reference_flow(train, test, spec, seed, batch_size, nr_epochs, test_interval, fc, latent, lr, wd, lam):
    """unet.py:388-509 restated with the oracle: DSDataset scan + normalise, modules from the seed, frozen shuffles of
    both loaders, AdamW epochs, eval-mode test epochs"""
    from oracle import cae_oracle as orc
    from oracle import unet_oracle as uo
    from cae_tools_amd.models.unet import Decoder, Encoder
    (_, imin, imax) = orc.scan_variable(train["lo"].values)
    (_, omin, omax) = orc.scan_variable(train["hi"].values)

    def pack(ds):
        return (torch.from_numpy(orc.pack_inputs([ds["lo"].values], [imin], [imax])),
                torch.from_numpy(orc.normalise_variable(ds["hi"].values, omin, omax)), torch.from_numpy(ds["valid"].values))
    (xtr, ttr, mtr) = pack(train)
    (xte, tte, mte) = pack(test)
    torch.manual_seed(seed)
    enc = Encoder(spec.get_input_layers(), encoded_space_dim=latent, fc_size=fc)
    dec = Decoder(spec.get_output_layers(), encoded_space_dim=latent, fc_size=fc)
    tr_batches = [b for b in torch.utils.data.DataLoader(torch.arange(len(xtr)), batch_size=batch_size, shuffle=True)]
    te_batches = [b for b in torch.utils.data.DataLoader(torch.arange(len(xte)), batch_size=batch_size, shuffle=True)]
    m = uo.UnetOracle(spec.save(), enc.state_dict(), dec.state_dict(), lr=lr, weight_decay=wd, dropout_rate=0.0,
                      lambda_pearson=lam)
    hist = {"train_loss": [], "test_loss": []}
    for epoch in range(nr_epochs):
        tl = float(np.mean([m.train_step(xtr[i], ttr[i], mtr[i])[0] for i in tr_batches]))
        if epoch % test_interval == 0:
            hist["train_loss"].append(tl)
            hist["test_loss"].append(float(np.mean([m.eval_losses(xte[i], tte[i], mte[i])[0] for i in te_batches])))
    return m, hist, (imin, imax, omin, omax), (xte, tte, mte)


def test_train_save_load_apply(tmp_path):
    from cae_tools_amd.models.unet import UNET, unet_layer_spec
    (train, test) = (_data(14, 1), _data(6, 2))
    spec = unet_layer_spec(2, 1, (16, 16), [8, 16])
    kw = dict(batch_size=4, nr_epochs=3, test_interval=1, fc_size=10, encoded_dim_size=4, lr=1e-3, weight_decay=1e-5,
              dropout_rate=0.0, lambda_pearson=0.5)
    (ref, hist, norm, (xte, tte, mte)) = _reference_flow(train, test, spec, 7, 4, 3, 1, 10, 4, 1e-3, 1e-5, 0.5)

    torch.manual_seed(7)
    mt = UNET(**kw)
    mt.spec = spec                                  # what --layer-definitions-path does (cli/train_cae.py:143-147)
    folder = str(tmp_path / "model")
    buf = io.StringIO()
    with redirect_stdout(buf):
        metrics = mt.train(["lo"], "hi", train, test, model_path=folder, mask_variable_name="valid")
    out = buf.getvalue()
    for line in ("initiating train method", "finished train_loarder and test_loader", "Running on device: cuda",
                 "time used for training one epoch:", "learn rate: 0.001000", "elapsed:", "Test Metrics", "Train Metrics"):
        assert line in out, line
    rows = [l for l in out.splitlines() if l.startswith("epoch: ")]
    assert len(rows) == 3 and rows[0].startswith("epoch: 0, train_mse: ") and "test_pearson_loss: " in rows[0]
    np.testing.assert_allclose(mt.history["train_loss"], hist["train_loss"], rtol=2e-3)
    np.testing.assert_allclose(mt.history["test_loss"], hist["test_loss"], rtol=5e-3)
    assert mt.history["nr_epochs"] == 3
    assert mt.normalisation_parameters == [{"lo": norm[0]}, {"lo": norm[1]}, norm[2], norm[3]]
    assert set(metrics) == {"test", "train"} and set(metrics["test"]) == {"mse", "rmse", "mae", "mean_pearson_correlation"}

    # model folder (unet.py:551-583)
    assert sorted(os.listdir(folder)) == sorted(["encoder.weights", "decoder.weights", "normalisation.weights", "parameters.json",
                                                 "spec.json", "history.json", "summary.txt", "input_spec.json",
                                                 "output_spec.json"])
    enc_sd = torch.load(os.path.join(folder, "encoder.weights"), weights_only=True)
    dec_sd = torch.load(os.path.join(folder, "decoder.weights"), weights_only=True)
    assert list(enc_sd) == list(ref.enc) and list(dec_sd) == list(ref.dec)
    assert int(enc_sd["encoder_cnn.1.num_batches_tracked"]) == 3 * 4 and enc_sd["encoder_cnn.1.num_batches_tracked"].dtype == torch.int64
    for (sd, want) in ((enc_sd, ref.enc), (dec_sd, ref.dec)):
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                continue
            # 12 AdamW steps of lr 1e-3 on both sides: weights with ~0 gradient random-walk by up to lr per step
            assert np.abs(v.numpy() - want[k].detach().numpy()).max() <= 12 * 2.1e-3, k
    with open(os.path.join(folder, "parameters.json")) as f:
        p = json.load(f)
    assert p["type"] == "UNET" and p["lambda_pearson"] == 0.5 and p["dropout_rate"] == 0.0 and p["input_shape"] == [2, 16, 16]

    # load + apply (base_model.py:102-152): float64 variable, (case, channel, y, x) dims, denormalised
    m2 = UNET()
    m2.load(folder)
    assert m2.get_model_id() == mt.get_model_id() and m2.get_input_variable_names() == ["lo"]
    score_ds = _data(6, 2)
    m2.apply(score_ds, ["lo"], prediction_variable="pred")
    pred = score_ds["pred"]
    assert pred.dims == ("n", "model_output_channel", "model_output_y", "model_output_x") and pred.values.dtype == np.float64
    again = _data(6, 2)
    mt.apply(again, ["lo"], prediction_variable="pred")
    np.testing.assert_allclose(pred.values, again["pred"].values, rtol=0, atol=1e-9)
    want = norm[2] + ref.eval_forward(xte).double().numpy() * (norm[3] - norm[2])
    np.testing.assert_allclose(pred.values, want, rtol=0, atol=0.05 * (norm[3] - norm[2]))   # 12 steps of fp32 training apart


def test_cli_train_then_apply_unet(tmp_path):
    from cae_tools_amd.cli import apply_cae, train_cae
    from cae_tools_amd.data.arrays import open_dataset
    from cae_tools_amd.models.unet import unet_layer_spec
    (train, test) = (_data(10, 3), _data(4, 4))
    (ptr, pte, spec_path, folder, out_nc) = (str(tmp_path / n) for n in ("train.nc", "test.nc", "layers.json", "m", "scored.nc"))
    train.to_netcdf(ptr)
    test.to_netcdf(pte)
    with open(spec_path, "w") as f:
        json.dump(unet_layer_spec(2, 1, (16, 16), [8, 16]).save(), f)
    buf = io.StringIO()
    with redirect_stdout(buf):
        train_cae.main(["--train-inputs", ptr, "--test-inputs", pte, "--model-folder", folder, "--input-variables", "lo",
                        "--output-variable", "hi", "--method", "unet", "--nr-epochs", "2", "--batch-size", "4",
                        "--fc-size", "8", "--latent-size", "3", "--layer-definitions-path", spec_path,
                        "--mask-variable", "valid", "--dropout-rate", "0.1", "--database-path", str(tmp_path / "db.sqlite")])
        apply_cae.main([pte, out_nc, "--model-folder", folder, "--input-variables", "lo"])
    assert "Training cases: 10, Test cases: 4" in buf.getvalue()
    scored = open_dataset(out_nc)
    assert scored["model_output"].shape == (4, 1, 16, 16) and np.isfinite(scored["model_output"].values).all()
    import sqlite3
    conn = sqlite3.connect(str(tmp_path / "db.sqlite"))
    assert conn.execute("SELECT model_type FROM MODEL_TRAINING").fetchone()[0] == "UNET"
    assert conn.execute("SELECT COUNT(*) FROM MODEL_EVALUATIONS").fetchone()[0] == 1
