"""ModelDatabase — the sqlite experiment log of the reference (src/cae_tools/utils/model_database.py):
same file format (tables MODEL_SCHEMA / MODEL_TRAINING / MODEL_EVALUATIONS with the reference's column
declarations, :16-21), same call surface (add_training_result :26, add_evaluation_result :33, dump :120,
dump_model :145, dump_schema :86) and the same printed reports, so a database written by either side is
readable by the other (tests/test_model_database.py compares rows and reports with tests/golden/)."""
import datetime
import json
import os
import sqlite3

SCHEMA_VERSION = "0.1"

_TABLES = {
    "MODEL_SCHEMA": "version STRING",
    "MODEL_TRAINING": "timestamp DATE, model_id STRING, model_type STRING, target_variable STRING, "
                      "input_variables STRING, model_description STRING, model_path STRING, train_path STRING, "
                      "train_loss FLOAT, test_path STRING, test_loss FLOAT, hyperparameters STRING, spec STRING",
    "MODEL_EVALUATIONS": "timestamp DATE, model_id STRING, train_path STRING, test_path STRING, metrics STRING",
}
_REPORT_COLUMNS = (("model_id", 36), ("model_type", 9), ("test_loss", 10), ("train_loss", 10), ("test_mse", 10),
                   ("train_mse", 10), ("test_mae", 10), ("train_mae", 10))
_REPORT_TITLES = {"model_id": "ModelID", "model_type": "ModelType", "test_loss": "Test Loss", "train_loss": "Train Loss",
                  "input_variables": "Inputs", "test_mse": "Test MSE", "train_mse": "Train MSE", "test_mae": "Test MAE",
                  "train_mae": "Train MAE"}


def _insert(conn, table, values):
    conn.execute(f"INSERT INTO {table} VALUES({','.join('?' * len(values))})", values)
    conn.commit()


def _select(conn, sql, args=()):
    cur = conn.execute(sql, args)
    names = [d[0] for d in cur.description]
    return [dict(zip(names, row)) for row in cur.fetchall()]


class ModelDatabase:

    def __init__(self, database_path):
        fresh = not os.path.exists(database_path)
        self.conn = sqlite3.connect(database_path)
        if fresh:
            for (name, columns) in _TABLES.items():
                self.conn.execute(f"CREATE TABLE {name}({columns})")
            _insert(self.conn, "MODEL_SCHEMA", (SCHEMA_VERSION,))

    # ---- writers ---------------------------------------------------------------------------------
    def add_training_result(self, model_id, model_type, target_variable, input_variables, description, model_path,
                            train_path, train_loss, test_path, test_loss, hyperparameters, spec):
        _insert(self.conn, "MODEL_TRAINING",
                (str(datetime.datetime.now()), model_id, model_type, target_variable, json.dumps(input_variables),
                 description, model_path, train_path, train_loss, test_path, test_loss, json.dumps(hyperparameters),
                 json.dumps(spec)))

    def add_evaluation_result(self, model_id, train_path, test_path, metrics):
        print(model_id, train_path, test_path, metrics)
        _insert(self.conn, "MODEL_EVALUATIONS",
                (str(datetime.datetime.now()), model_id, train_path, test_path, json.dumps(metrics)))

    # ---- reports ---------------------------------------------------------------------------------
    @staticmethod
    def _report_line(fields):
        cells = " | ".join(f"{fields.get(key, ''):{width}s}" for (key, width) in _REPORT_COLUMNS)
        print(f"| {cells} | {fields.get('input_variables', '')}")

    def dump_schema(self):
        print("MODEL_SCHEMA")
        for row in self.conn.execute("SELECT * FROM MODEL_SCHEMA").fetchall():
            print(json.dumps(row))
        print()

    def dump(self):
        self._report_line(_REPORT_TITLES)
        for tr in _select(self.conn, "SELECT * FROM MODEL_TRAINING ORDER BY test_loss ASC"):
            line = {"model_id": tr["model_id"], "model_type": tr["model_type"],
                    "input_variables": ", ".join(json.loads(tr["input_variables"])),
                    "test_loss": "%0.2f" % tr["test_loss"], "train_loss": "%0.2f" % tr["train_loss"]}
            evaluations = _select(self.conn, "SELECT * FROM MODEL_EVALUATIONS WHERE model_id=?", [tr["model_id"]])
            if not evaluations:
                self._report_line(line)
            for (idx, ev) in enumerate(evaluations):
                m = json.loads(ev["metrics"])
                scores = {f"{part}_{q}": "%0.2f" % m[part][q] for part in ("test", "train") for q in ("mse", "mae")}
                self._report_line({**(line if idx == 0 else {}), **scores})   # continuation rows carry scores only
        print()

    @staticmethod
    def _dump_record(record, titles):
        width = max(len(titles.get(key, key)) for key in record)
        for (key, value) in record.items():
            if isinstance(value, str) and value.startswith("{"):
                text = json.dumps(json.loads(value), indent=4)
            else:
                text = str(value)
            (head, *rest) = text.split("\n")
            print(titles.get(key, key).rjust(width) + ": " + head)
            for line in rest:
                print(" " * (width + 2) + line)

    def dump_model(self, model_id):
        print("\n\nModel:")
        trained = _select(self.conn, "SELECT * FROM MODEL_TRAINING WHERE model_id=?", [model_id])
        if not trained:
            print("Model not found")
            return
        for record in trained:
            self._dump_record(record, {"model_id": "Model ID"})
        print("\n\nModel Evaluations:")
        evaluations = _select(self.conn, "SELECT * FROM MODEL_EVALUATIONS WHERE model_id=?", [model_id])
        for record in evaluations:
            self._dump_record(record, {"model_id": "Model ID"})
        if not evaluations:
            print("No evaluations found")
