#!/usr/bin/env python3
"""Golden vectors for the rows after the training path: ModelMetric and ModelDatabase.

Runs ONLY in the build container (imports the reference's model_metric / model_database modules from
/root/reference/src); stores inputs + expected outputs as data, no reference source.

    python tests/golden/make_golden_aux.py
"""
import json
import os
import sqlite3
import sys
import tempfile

import numpy as np

REF_SRC = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
if not os.path.isdir(REF_SRC):
    sys.exit("reference not mounted: this script only runs in the build container")
sys.path.insert(0, REF_SRC)

from cae_tools.models.model_metric import ModelMetric  # noqa: E402
from cae_tools.utils.model_database import ModelDatabase  # noqa: E402


def metric_case():
    """base_model.py:69-100 in miniature: fp32 truth, fp32 normalised scores denormalised in fp64, binary masks"""
    rng = np.random.default_rng(2024)
    (n, c, h, w) = (7, 1, 24, 20)
    (vmin, vmax) = (287.90234375, 298.11328125)
    truth = (vmin + (vmax - vmin) * rng.random((n, c, h, w))).astype(np.float32)
    y = np.clip((truth - vmin) / (vmax - vmin) + 0.05 * rng.standard_normal((n, c, h, w)), 0, 1).astype(np.float32)
    mask = (rng.random((n, c, h, w)) < 0.7).astype(np.float32)
    mask[3] = 0.0          # an instance with no valid pixel is skipped by the Pearson average (:60-61)
    mask[5] = 1.0
    out = {"truth": truth, "y": y, "mask": mask, "vmin": vmin, "vmax": vmax}
    for (tag, mk) in (("masked", mask), ("all", np.ones_like(mask))):
        mm = ModelMetric()
        scores = vmin + (y.astype(np.float64) * (vmax - vmin))   # ds_dataset.py:131-135 on a float64 array
        for i in range(n):
            mm.accumulate(truth[i], scores[i], mk[i])
        m = mm.get_metrics()
        for k, v in m.items():
            out[f"{tag}/{k}"] = np.float64(v)
        print(tag, m)
    np.savez_compressed(os.path.join(HERE, "model_metric.npz"), **out)


def database_case():
    """the rows and schema the reference's ModelDatabase writes (utils/model_database.py:11-39)"""
    calls = {
        "training": [
            dict(model_id="11111111-2222-3333-4444-555555555555", model_type="ConvAE", target_variable="hires",
                 input_variables=["lowres"], description="Model Summary:\n\t...", model_path="/tmp/m1",
                 train_path="train.nc", train_loss=0.0123, test_path="test.nc", test_loss=0.0234,
                 hyperparameters={"type": "ConvAEModel", "batch_size": 10, "lr": 0.001}, spec={"input_layers": []}),
            dict(model_id="aaaaaaaa-bbbb-cccc-dddd-eeeeeeeeeeee", model_type="ConvAE", target_variable="hires",
                 input_variables=["u", "v"], description="second", model_path="", train_path="a.nc,b.nc",
                 train_loss=0.5, test_path="c.nc", test_loss=0.004, hyperparameters={"fc_size": 128},
                 spec={"output_layers": []}),
        ],
        "evaluation": [
            dict(model_id="11111111-2222-3333-4444-555555555555", train_path="train.nc", test_path="test.nc",
                 metrics={"test": {"mse": 1.5, "rmse": 1.2247, "mae": 1.0, "mean_pearson_correlation": 0.9},
                          "train": {"mse": 1.25, "rmse": 1.118, "mae": 0.9, "mean_pearson_correlation": 0.95}}),
        ],
    }
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "models.db")
        db = ModelDatabase(path)
        for c in calls["training"]:
            db.add_training_result(c["model_id"], c["model_type"], c["target_variable"], c["input_variables"],
                                   c["description"], c["model_path"], c["train_path"], c["train_loss"], c["test_path"],
                                   c["test_loss"], c["hyperparameters"], c["spec"])
        for c in calls["evaluation"]:
            db.add_evaluation_result(c["model_id"], c["train_path"], c["test_path"], c["metrics"])
        import contextlib
        import io
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            db.dump()
            db.dump_model("aaaaaaaa-bbbb-cccc-dddd-eeeeeeeeeeee")
            db.dump_model("no-such-model")
        db.conn.close()
        conn = sqlite3.connect(path)
        schema = [r[0] for r in conn.execute("SELECT sql FROM sqlite_master WHERE type='table' ORDER BY name")]
        version = conn.execute("SELECT * FROM MODEL_SCHEMA").fetchall()
        cols = "model_id, model_type, target_variable, input_variables, model_description, model_path, train_path, " \
               "train_loss, test_path, test_loss, hyperparameters, spec"
        training = conn.execute(f"SELECT {cols} FROM MODEL_TRAINING ORDER BY rowid").fetchall()
        evaluation = conn.execute("SELECT model_id, train_path, test_path, metrics FROM MODEL_EVALUATIONS "
                                  "ORDER BY rowid").fetchall()
        conn.close()
    dump = buf.getvalue().split("\n")
    # the timestamp column of dump_model is wall-clock: keep the line, blank the value
    dump = [("  timestamp: <now>" if ln.strip().startswith("timestamp:") else ln) for ln in dump]
    with open(os.path.join(HERE, "model_database.json"), "w") as f:
        json.dump({"calls": calls, "schema": schema, "version": version, "training_rows": training,
                   "evaluation_rows": evaluation, "dump": dump}, f, indent=1)
    print("model_database:", len(training), "training rows,", len(evaluation), "evaluation rows,", len(dump), "dump lines")


if __name__ == "__main__":
    metric_case()
    database_case()
