"""BaseModel — shared model plumbing with the reference's public surface
(src/cae_tools/models/base_model.py): model id (:33,56-61), io-spec accessors (:35-54),
evaluate (:69-100), apply (:102-152), dump_metrics (:154-157), save/load of input_spec.json /
output_spec.json (:162-180).  Scoring, denormalisation and the metric reductions run on the GPU
through libcae_hip; only per-case sums and the final fp64 predictions cross PCIe."""
import json
import os
import uuid

import numpy as np
import torch

from ..data.arrays import DataArray
from .ds_dataset import DSDataset
from .model_metric import DeviceModelMetric, ModelMetric  # noqa: F401


def _make_data_array(like_ds, data, dims):
    """DataArray of the same family as the dataset it is assigned into"""
    if type(like_ds).__module__.startswith("xarray"):
        import xarray as xr
        return xr.DataArray(data, dims=dims)
    return DataArray(data, dims=dims)


class BaseModel:

    def __init__(self):
        self.input_spec = None
        self.output_spec = None
        self.model_id = str(uuid.uuid4())

    def set_input_spec(self, input_spec):
        self.input_spec = input_spec

    def get_input_spec(self):
        return self.input_spec

    def set_output_spec(self, output_spec):
        self.output_spec = output_spec

    def get_output_spec(self):
        return self.output_spec

    def get_input_variable_names(self):
        return None if self.input_spec is None else [item["name"] for item in self.input_spec]

    def get_output_variable_name(self):
        return None if self.output_spec is None else self.output_spec["name"]

    def set_model_id(self, model_id):
        self.model_id = model_id

    def get_model_id(self):
        return self.model_id

    def torch_load(self, from_path):
        return torch.load(from_path, map_location=torch.device("cpu"), weights_only=True)

    # ---- scoring helpers -------------------------------------------------------------------
    def _score_device(self, x):
        """eval-mode forward of an (N,C,H,W) fp32 CUDA tensor; implemented by the sub-class"""
        raise NotImplementedError

    def _score_all(self, x):
        """_score_device over the whole array; under a torch.distributed.run launch the cases are sharded over the
        ranks (no exchange while scoring: SURVEY.md §8e) and the scores all-gathered, so every rank returns all of
        them.  Collective: every rank must call it with the same number of cases."""
        from .. import dp as _dp
        dist = _dp.ensure_process_group()
        n = int(x.shape[0])
        if dist is None or n < dist.get_world_size():
            return self._score_device(x)
        (world, rank) = (dist.get_world_size(), dist.get_rank())
        (lo, hi) = _dp.shard_bounds(n, world, rank)
        mine = self._score_device(x[lo:hi])
        per = -(-n // world)
        pad = torch.zeros((per,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        pad[:hi - lo] = mine
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        sizes = [b - a for (a, b) in (_dp.shard_bounds(n, world, r) for r in range(world))]
        return torch.cat([p[:k] for p, k in zip(parts, sizes)])

    def evaluate(self, dataset, device=None):
        """score every case, denormalise, and pool the reference's metrics (:69-100).  Scores, truth and mask
        stay on the GPU; cae_metric_sums reduces each case to eight fp64 sums.  The mask is all ones unless the
        dataset carries a mask variable shaped like the output (the reference builds the default mask with the
        INPUT's shape, which cannot index the output; the intended all-pixels mask is used - SURVEY.md headline 3)."""
        from .. import dp as _dp
        _dp.select_device()
        dataset.set_normalise_output(False)
        truth = dataset.device_outputs()
        scores = self._score_all(dataset.device_inputs())
        mm = DeviceModelMetric()
        mm.accumulate(truth, scores, dataset.device_mask(), dataset.min_output, dataset.max_output)
        return mm.get_metrics()

    def apply(self, score_ds, input_variables, prediction_variable="model_output",
              channel_dimension="model_output_channel", y_dimension="model_output_y",
              x_dimension="model_output_x", mask_variable_name=None):
        """Add `prediction_variable` (float64, denormalised, dims (case, channel, y, x)) to score_ds
        in place (:102-152)."""
        from .. import dp as _dp
        _dp.ensure_process_group()      # a rank's GPU is selected before the data set is uploaded
        first = score_ds[input_variables[0]]
        n_dimension = first.dims[0]
        ds = DSDataset(score_ds, input_variables, input_variables[0], normalise_in=self.normalise_input,
                       mask_variable_name=mask_variable_name)
        ds.set_normalisation_parameters(self.normalisation_parameters)
        y = self._score_all(ds.device_inputs())     # cases sharded over the GPUs of a torch.distributed.run launch
        out = ds.denormalise_device(y)   # fp64 on the device: min + y*(max-min), then one D2H copy
        score_ds[prediction_variable] = _make_data_array(score_ds, out.cpu().numpy(),
                                                         (n_dimension, channel_dimension, y_dimension, x_dimension))

    def dump_metrics(self, title, metrics):
        print("\n" + title)
        for key in metrics:
            print(f"\t{key:30s}:{metrics[key]}")

    def score(self, batches, save_arr):
        pass  # implement in sub-class

    def save(self, to_folder):
        for (spec, fname) in ((self.input_spec, "input_spec.json"), (self.output_spec, "output_spec.json")):
            if spec is not None:
                with open(os.path.join(to_folder, fname), "w") as f:
                    f.write(json.dumps(spec))

    def load(self, from_folder):
        for attr, fname in (("input_spec", "input_spec.json"), ("output_spec", "output_spec.json")):
            path = os.path.join(from_folder, fname)
            if os.path.exists(path):
                with open(path) as f:
                    setattr(self, attr, json.loads(f.read()))

    def train(self, input_variables, output_variable, training_ds, testing_ds, model_path="", training_paths="",
              testing_paths=""):
        pass  # implement in sub-class

    def summary(self):
        pass  # implement in sub-class

    def get_parameters(self):
        pass  # implement in sub-class
