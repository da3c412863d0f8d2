"""ModelDatabase parity: tests/golden/model_database.json holds the calls made to the
reference's ModelDatabase, the sqlite schema / rows it wrote and the reports it printed."""
import contextlib
import io
import json
import os
import sqlite3

from cae_tools_amd.utils.model_database import ModelDatabase
from cae_tools_amd.cli import query_database

with open(os.path.join(os.path.dirname(__file__), "golden", "model_database.json")) as f:
    G = json.load(f)


def _fill(path):
    db = ModelDatabase(path)
    for c in G["calls"]["training"]:
        db.add_training_result(c["model_id"], c["model_type"], c["target_variable"], c["input_variables"],
                               c["description"], c["model_path"], c["train_path"], c["train_loss"], c["test_path"],
                               c["test_loss"], c["hyperparameters"], c["spec"])
    with contextlib.redirect_stdout(io.StringIO()) as echoed:
        for c in G["calls"]["evaluation"]:
            db.add_evaluation_result(c["model_id"], c["train_path"], c["test_path"], c["metrics"])
    return db, echoed.getvalue()


def test_schema_and_rows_match_reference(tmp_path):
    path = str(tmp_path / "models.db")
    (db, echoed) = _fill(path)
    assert echoed.startswith(G["calls"]["evaluation"][0]["model_id"] + " train.nc test.nc {")   # :36 prints the row
    db.conn.close()
    conn = sqlite3.connect(path)
    schema = [r[0] for r in conn.execute("SELECT sql FROM sqlite_master WHERE type='table' ORDER BY name")]
    assert schema == G["schema"]
    assert [list(r) for r in conn.execute("SELECT * FROM MODEL_SCHEMA")] == G["version"]
    cols = "model_id, model_type, target_variable, input_variables, model_description, model_path, train_path, " \
           "train_loss, test_path, test_loss, hyperparameters, spec"
    assert [list(r) for r in conn.execute(f"SELECT {cols} FROM MODEL_TRAINING ORDER BY rowid")] == G["training_rows"]
    assert [list(r) for r in conn.execute("SELECT model_id, train_path, test_path, metrics FROM MODEL_EVALUATIONS "
                                          "ORDER BY rowid")] == G["evaluation_rows"]
    stamp = conn.execute("SELECT timestamp FROM MODEL_TRAINING").fetchone()[0]
    assert len(stamp) >= 19 and stamp[4] == "-" and stamp[10] == " "      # 'YYYY-MM-DD HH:MM:SS[.ffffff]'
    conn.close()


def test_reports_match_reference(tmp_path, capsys):
    path = str(tmp_path / "models.db")
    (db, _) = _fill(path)
    db.conn.close()
    capsys.readouterr()
    query_database.main([path])
    query_database.main([path, "--model-id", "aaaaaaaa-bbbb-cccc-dddd-eeeeeeeeeeee"])
    query_database.main([path, "--model-id", "no-such-model"])
    lines = capsys.readouterr().out.split("\n")
    lines = [("  timestamp: <now>" if ln.strip().startswith("timestamp:") else ln) for ln in lines]
    assert lines == G["dump"]


def test_reopen_appends(tmp_path):
    path = str(tmp_path / "models.db")
    (db, _) = _fill(path)
    db.conn.close()
    (db, _) = _fill(path)       # existing file: no CREATE TABLE, rows appended (:23-24)
    assert db.conn.execute("SELECT COUNT(*) FROM MODEL_TRAINING").fetchone()[0] == 2 * len(G["training_rows"])
    assert db.conn.execute("SELECT COUNT(*) FROM MODEL_SCHEMA").fetchone()[0] == 1
