"""VarAEModel — the 'var' method of the train_cae CLI (its default, cli/train_cae.py:42) on libcae_hip.

The reference ships NO source for this model (cae_tools.models.var_ae_model is imported by model_evaluator.py:35 but the
file is absent); what survives is the CLI surface: --lambda-mse / --lambda-kl / --lambda-ssim (cli/train_cae.py:32-36)
and the pytorch_msssim requirement (README.md:29).  This class is therefore the build's own definition, shaped like
ConvAEModel (same constructor keywords plus the three lambdas, same train / apply / score / save / load surface and model
folder), with the arithmetic published in oracle/vae_oracle.py and computed by the HIP kernels behind include/cae_vae.h:
ConvAE encoder stack -> Linear -> ReLU -> (mu, logvar) heads -> z = mu + eps*exp(logvar/2) -> ConvAE decoder;
loss = lambda_mse*MSE + lambda_kl*KL + lambda_ssim*(1 - MS-SSIM); Adam with L2 weight decay."""
import json
import os
import time

import numpy as np
import torch

from .. import vae_engine as _ve
from ..utils.model_database import ModelDatabase
from ._params import ParamBag, add_batchnorm, add_conv
from .base_model import BaseModel
from .conv_ae_model import _index_batches
from .decoder import Decoder
from .ds_dataset import DSDataset
from .model_sizer import ModelSpec, create_model_spec


class VarEncoder(ParamBag):
    """Conv2d -> BatchNorm2d -> ReLU per layer, Linear(F, fc), heads encoder_mu / encoder_logvar; PyTorch default init"""

    def __init__(self, layers, encoded_space_dim, fc_size):
        super().__init__()
        self.layers = list(layers)
        for i, layer in enumerate(self.layers):
            (cin, _, _) = layer.get_input_dimensions()
            (cout, _, _) = layer.get_output_dimensions()
            (kh, kw) = layer.kernel_hw()
            add_conv(self, f"encoder_cnn.{3 * i}", (cout, cin, kh, kw), cout)
            add_batchnorm(self, f"encoder_cnn.{3 * i + 1}", cout)
        (chan, y, x) = self.layers[-1].get_output_dimensions()
        add_conv(self, "encoder_lin.0", (fc_size, chan * y * x), fc_size)
        add_conv(self, "encoder_mu", (encoded_space_dim, fc_size), encoded_space_dim)
        add_conv(self, "encoder_logvar", (encoded_space_dim, fc_size), encoded_space_dim)

    def forward(self, x):
        raise RuntimeError("VarEncoder.forward on its own is not a product path: use VarAEModel.score / apply / train")


class VarAEModel(BaseModel):

    def __init__(self, normalise_input=True, normalise_output=True, batch_size=10, nr_epochs=500, test_interval=10,
                 encoded_dim_size=32, fc_size=128, lr=0.001, weight_decay=1e-5, use_gpu=True, conv_kernel_size=3, conv_stride=2,
                 conv_input_layer_count=None, conv_output_layer_count=None, database_path=None, lambda_mse=1, lambda_kl=1,
                 lambda_ssim=1, noise_seed=0):
        super().__init__()
        self.normalise_input, self.normalise_output = normalise_input, normalise_output
        self.normalisation_parameters = None
        self.input_shape = self.output_shape = None
        self.encoder = self.decoder = None
        (self.batch_size, self.nr_epochs, self.test_interval) = (batch_size, nr_epochs, test_interval)
        (self.encoded_dim_size, self.fc_size, self.lr, self.weight_decay, self.use_gpu) = (encoded_dim_size, fc_size, lr,
                                                                                          weight_decay, use_gpu)
        (self.conv_kernel_size, self.conv_stride) = (conv_kernel_size, conv_stride)
        (self.conv_input_layer_count, self.conv_output_layer_count) = (conv_input_layer_count, conv_output_layer_count)
        (self.lambda_mse, self.lambda_kl, self.lambda_ssim, self.noise_seed) = (lambda_mse, lambda_kl, lambda_ssim, noise_seed)
        self.spec = None
        self.history = {"train_loss": [], "test_loss": [], "nr_epochs": 0}
        self.db = ModelDatabase(database_path) if database_path else None
        self._engine = None

    def get_parameters(self):
        return {"type": "VarAEModel", "input_shape": list(self.input_shape), "output_shape": list(self.output_shape),
                "batch_size": self.batch_size, "test_interval": self.test_interval, "encoded_dim_size": self.encoded_dim_size,
                "fc_size": self.fc_size, "lr": self.lr, "weight_decay": self.weight_decay, "lambda_mse": self.lambda_mse,
                "lambda_kl": self.lambda_kl, "lambda_ssim": self.lambda_ssim, "normalise_input": self.normalise_input,
                "normalise_output": self.normalise_output, "conv_kernel_size": self.conv_kernel_size,
                "conv_stride": self.conv_stride, "conv_input_layer_count": self.conv_input_layer_count,
                "conv_output_layer_count": self.conv_output_layer_count, "model_id": self.get_model_id()}

    def summary(self):
        if not self.spec:
            return "Model has not been trained - no layers assigned yet"
        fc = f"\tFully Connected Layer:\n\t\tsize={self.fc_size}\n"
        return ("Model Summary:\n" + "".join(str(l) for l in self.spec.input_layers) + fc
                + f"\tLatent Vector (mu, logvar):\n\t\tsize={self.encoded_dim_size}\n" + fc
                + "".join(str(l) for l in self.spec.output_layers))

    def _modules(self):
        self.encoder = VarEncoder(self.spec.get_input_layers(), encoded_space_dim=self.encoded_dim_size, fc_size=self.fc_size)
        self.decoder = Decoder(self.spec.get_output_layers(), encoded_space_dim=self.encoded_dim_size, fc_size=self.fc_size)

    def _pull_weights(self):
        if self._engine is not None:
            (enc, dec) = self._engine.export_state()
            self.encoder.load_state_dict(enc)
            self.decoder.load_state_dict(dec)

    def save(self, to_folder):
        os.makedirs(to_folder, exist_ok=True)
        self._pull_weights()
        torch.save(self.encoder.state_dict(), os.path.join(to_folder, "encoder.weights"))
        torch.save(self.decoder.state_dict(), os.path.join(to_folder, "decoder.weights"))
        for fname, text in {"normalisation.weights": json.dumps(self.normalisation_parameters),
                            "parameters.json": json.dumps(self.get_parameters()), "spec.json": json.dumps(self.spec.save()),
                            "history.json": json.dumps(self.history), "summary.txt": self.summary()}.items():
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
        for key in ("batch_size", "test_interval", "encoded_dim_size", "fc_size", "lr", "weight_decay", "normalise_input",
                    "normalise_output", "lambda_mse", "lambda_kl", "lambda_ssim"):
            setattr(self, key, p[key])
        for key in ("conv_kernel_size", "conv_stride", "conv_input_layer_count", "conv_output_layer_count"):
            setattr(self, key, p.get(key, None))
        with open(os.path.join(from_folder, "history.json")) as f:
            self.history = json.loads(f.read())
        with open(os.path.join(from_folder, "spec.json")) as f:
            self.spec = ModelSpec()
            self.spec.load(json.loads(f.read()))
        self._modules()
        self.encoder.load_state_dict(self.torch_load(os.path.join(from_folder, "encoder.weights")))
        self.decoder.load_state_dict(self.torch_load(os.path.join(from_folder, "decoder.weights")))
        self._engine = None
        super().load(from_folder)

    def _get_engine(self, max_batch):
        if self._engine is None or self._engine.max_batch < max_batch:
            if self._engine is not None:
                self._pull_weights()
            eng = _ve.VaeEngine(self.spec, self.fc_size, self.encoded_dim_size, max_batch=max_batch)
            eng.load_state(self.encoder.state_dict(), self.decoder.state_dict())
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
        self.normalisation_parameters = train_ds.get_normalisation_parameters()
        self.set_input_spec(train_ds.get_input_spec())
        self.set_output_spec(train_ds.get_output_spec())
        test_ds = DSDataset(testing_ds, input_variables, output_variable, normalise_in=self.normalise_input,
                            normalise_out=self.normalise_output)
        test_ds.set_normalisation_parameters(self.normalisation_parameters)
        self.input_shape, self.output_shape = tuple(train_ds.get_input_shape()), tuple(train_ds.get_output_shape())
        if not self.spec:
            self.spec = create_model_spec(input_size=self.input_shape[1:], input_channels=self.input_shape[0],
                                          output_size=self.output_shape[1:], output_channels=self.output_shape[0],
                                          kernel_size=self.conv_kernel_size, stride=self.conv_stride,
                                          input_layer_count=self.conv_input_layer_count,
                                          output_layer_count=self.conv_output_layer_count)
        if not self.encoder or not self.decoder:
            self._modules()
        train_perm = _index_batches(len(train_ds), self.batch_size)
        test_perm = _index_batches(len(test_ds), self.batch_size)
        print(f"Running on device: {torch.device('cuda')}")
        start = time.time()
        eng = self._get_engine(int(self.batch_size))
        eng.set_hyper(lr=self.lr, weight_decay=self.weight_decay, lambda_mse=self.lambda_mse, lambda_kl=self.lambda_kl,
                      lambda_ssim=self.lambda_ssim, seed=self.noise_seed)
        eng.reset_optimizer()
        eng.set_dataset(_ve.TRAIN, train_ds.device_inputs(), train_ds.device_outputs())
        eng.set_dataset(_ve.TEST, test_ds.device_inputs(), test_ds.device_outputs())
        (train_idx, test_idx) = (eng.upload_perm(train_perm), eng.upload_perm(test_perm))
        train_loss = test_loss = 0.0
        for epoch in range(self.nr_epochs):
            train_loss = float(np.mean([l[3] for l in eng.run_batches(_ve.TRAIN, train_idx, len(train_ds), self.batch_size, True)]))
            if epoch % self.test_interval == 0:
                test_loss = float(np.mean([l[3] for l in eng.run_batches(_ve.TEST, test_idx, len(test_ds), self.batch_size, False)]))
                self.history["train_loss"].append(train_loss)
                self.history["test_loss"].append(test_loss)
                print("%5d %.6f %.6f" % (epoch, train_loss, test_loss))
        self.history["nr_epochs"] += self.nr_epochs
        print("elapsed:" + str(time.time() - start))
        if self.db:
            self.db.add_training_result(self.get_model_id(), "VarAE", output_variable, input_variables, self.summary(), model_path,
                                        training_paths, train_loss, testing_paths, test_loss, self.get_parameters(), self.spec.save())
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
