"""LinearModel — drop-in for the reference's `--method linear` model (src/cae_tools/models/linear_model.py) on libcae_hip.

Same constructor keywords (:32-34), train / apply / score / save / load / summary / get_parameters, the same model folder
(`weights` = torch-saved state_dict with linear.1.weight / linear.1.bias, normalisation.weights, parameters.json,
history.json, summary.txt, input_spec.json, output_spec.json) and printed lines.  Where the reference's HEAD is inconsistent
(its epoch loops unpack three values from DSDataset's four-tuple, :146,166; `test_paths` is undefined at :281) the evident
intent is implemented.  The step (forward, MSELoss, backward, Adam(lr, weight_decay): :146-153,241,247) runs in the HIP
kernels behind include/cae_linear.h."""
import json
import os
import time

import numpy as np
import torch

from .. import linear_engine as _le
from ..utils.model_database import ModelDatabase
from .base_model import BaseModel
from .conv_ae_model import _index_batches
from .ds_dataset import DSDataset
from .linear import Linear


class LinearModel(BaseModel):

    def __init__(self, normalise_input=True, normalise_output=True, batch_size=10, nr_epochs=500, test_interval=10, lr=0.001,
                 weight_decay=1e-5, use_gpu=True, database_path=None):
        super().__init__()
        self.normalise_input, self.normalise_output = normalise_input, normalise_output
        self.normalisation_parameters = None
        self.input_shape = self.output_shape = None
        self.weights = None
        (self.batch_size, self.nr_epochs, self.test_interval) = (batch_size, nr_epochs, test_interval)
        (self.lr, self.weight_decay, self.use_gpu) = (lr, weight_decay, use_gpu)
        self.history = {"train_loss": [], "test_loss": [], "nr_epochs": 0}
        self.optim = None
        self.db = ModelDatabase(database_path) if database_path else None
        self._engine = None

    def get_parameters(self):
        return {"model_id": self.get_model_id(), "type": "LinearModel", "input_shape": list(self.input_shape),
                "output_shape": list(self.output_shape), "batch_size": self.batch_size, "test_interval": self.test_interval,
                "lr": self.lr, "weight_decay": self.weight_decay, "normalise_input": self.normalise_input,
                "normalise_output": self.normalise_output}

    def summary(self):
        if not self.input_shape:
            return "Model has not been trained"
        return (f"Model Summary:\n\tInput shape:\n\t\tsize={tuple(self.input_shape)}\n\tOutput shape:\n"
                f"\t\tsize={tuple(self.output_shape)}\n")

    def _pull_weights(self):
        if self._engine is not None:
            self.weights.load_state_dict(self._engine.export_state())

    def save(self, to_folder):
        os.makedirs(to_folder, exist_ok=True)
        self._pull_weights()
        torch.save(self.weights.state_dict(), os.path.join(to_folder, "weights"))
        for fname, text in {"normalisation.weights": json.dumps(self.normalisation_parameters),
                            "parameters.json": json.dumps(self.get_parameters()), "history.json": json.dumps(self.history),
                            "summary.txt": self.summary()}.items():
            with open(os.path.join(to_folder, fname), "w") as f:
                f.write(text)
        super().save(to_folder)

    def load(self, from_folder):
        with open(os.path.join(from_folder, "normalisation.weights")) as f:
            self.normalisation_parameters = json.loads(f.read())
        with open(os.path.join(from_folder, "parameters.json")) as f:
            p = json.loads(f.read())
        if "model_id" in p:
            self.set_model_id(p["model_id"])
        self.input_shape, self.output_shape = tuple(p["input_shape"]), tuple(p["output_shape"])
        for key in ("batch_size", "test_interval", "lr", "weight_decay", "normalise_input", "normalise_output"):
            setattr(self, key, p[key])
        with open(os.path.join(from_folder, "history.json")) as f:
            self.history = json.loads(f.read())
        self.weights = Linear(self.input_shape, self.output_shape)
        self.weights.load_state_dict(self.torch_load(os.path.join(from_folder, "weights")))
        self._engine = None
        super().load(from_folder)

    def _get_engine(self, max_batch):
        if self._engine is None or self._engine.max_batch < max_batch:
            if self._engine is not None:
                self._pull_weights()
            eng = _le.LinearEngine(self.input_shape, self.output_shape, max_batch=max_batch)
            eng.load_state(self.weights.state_dict())
            self._engine = eng
        return self._engine

    def _score_device(self, x):
        return self._get_engine(max(1, min(int(self.batch_size), int(x.shape[0])))).score(x)

    def score(self, batches, save_arr):
        ctr = 0
        for batch in batches:
            x = torch.as_tensor(batch, dtype=torch.float32)
            y = self._score_device(x.cuda() if not x.is_cuda else x).cpu().numpy()
            save_arr[ctr:ctr + y.shape[0], :, :, :] = y
            ctr += self.batch_size

    def train(self, input_variables, output_variable, training_ds, testing_ds, model_path="", training_paths="",
              testing_paths="", mask_variable_name=None):
        train_ds = DSDataset(training_ds, input_variables, output_variable, normalise_in=self.normalise_input,
                             normalise_out=self.normalise_output)
        self.set_input_spec(train_ds.get_input_spec())
        self.set_output_spec(train_ds.get_output_spec())
        self.normalisation_parameters = train_ds.get_normalisation_parameters()
        test_ds = DSDataset(testing_ds, input_variables, output_variable, normalise_in=self.normalise_input,
                            normalise_out=self.normalise_output)
        test_ds.set_normalisation_parameters(self.normalisation_parameters)
        self.input_shape, self.output_shape = tuple(train_ds.get_input_shape()), tuple(train_ds.get_output_shape())
        if not self.weights:
            self.weights = Linear(self.input_shape, self.output_shape)
        train_perm = _index_batches(len(train_ds), self.batch_size)      # both loaders shuffle (:228-229), frozen once
        test_perm = _index_batches(len(test_ds), self.batch_size)
        print(f"Running on device: {torch.device('cuda')}")
        start = time.time()
        eng = self._get_engine(int(self.batch_size))
        eng.set_hyper(lr=self.lr, weight_decay=self.weight_decay)
        eng.reset_optimizer()
        eng.set_dataset(_le.TRAIN, train_ds.device_inputs(), train_ds.device_outputs())
        eng.set_dataset(_le.TEST, test_ds.device_inputs(), test_ds.device_outputs())
        (train_idx, test_idx) = (eng.upload_perm(train_perm), eng.upload_perm(test_perm))
        train_loss = test_loss = 0.0
        for epoch in range(self.nr_epochs):
            train_loss = float(np.mean(eng.run_batches(_le.TRAIN, train_idx, len(train_ds), self.batch_size, True)))
            if epoch % self.test_interval == 0:
                test_loss = float(np.mean(eng.run_batches(_le.TEST, test_idx, len(test_ds), self.batch_size, False)))
                self.history["train_loss"].append(train_loss)
                self.history["test_loss"].append(test_loss)
                print("%5d %.6f %.6f" % (epoch, train_loss, test_loss))
        self.history["nr_epochs"] = self.history["nr_epochs"] + self.nr_epochs
        print("elapsed:" + str(time.time() - start))
        if self.db:
            self.db.add_training_result(self.get_model_id(), "Linear", output_variable, input_variables, self.summary(), model_path,
                                        training_paths, train_loss, testing_paths, test_loss, self.get_parameters(), {})
        if model_path:
            self.save(model_path)
        else:
            self._pull_weights()
        metrics = {"test": self.evaluate(test_ds), "train": self.evaluate(train_ds)}
        self.dump_metrics("Test Metrics", metrics["test"])
        self.dump_metrics("Train Metrics", metrics["train"])
        if self.db:
            self.db.add_evaluation_result(self.get_model_id(), training_paths, testing_paths, metrics)
        return metrics
