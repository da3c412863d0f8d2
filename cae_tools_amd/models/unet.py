"""UNET — drop-in for the reference's `--method unet` model (src/cae_tools/models/unet.py) on libcae_hip.

Same constructor keywords (:201-204), train / score / save / load / summary / get_parameters surface, the same
model folder (encoder.weights / decoder.weights are torch-saved state_dicts with the reference's keys:
encoder_cnn.{4i} Conv2d, {4i+1} BatchNorm2d, encoder_lin.{0,1,4}; decoder_lin.{0,1,4},
attention_layers.{j}.fc{1,2}, decoder_conv.{4j} ConvTranspose2d, {4j+1} BatchNorm2d(2C)) and the same printed
progress lines.  Encoder / Decoder below are parameter containers that replay the reference modules' default
initialisation draw for draw (same seed => bit-identical initial weights; tests/test_unet_cpu.py); every FLOP of
training and scoring runs in the HIP kernels behind include/cae_unet.h.

What differs from the reference, on purpose:
  * the VGG perceptual loss is not built: the reference constructs it (:253, downloading VGG19 weights) but never adds
    it to the loss (:316-322);
  * dropout masks come from the engine's counter-based hash, not from torch's generator (dropout_seed keyword);
  * with mask_variable_name=None and a one-channel input the reference's all-ones mask has the input's shape
    (ds_dataset.py:152-156); it is materialised here so that torch.sum(mask) counts what the reference counts.
"""
import json
import os
import time

import numpy as np
import torch

from .. import unet_engine as _ue
from ..utils.model_database import ModelDatabase
from ._params import ParamBag, add_batchnorm, default_layer_init
from .base_model import BaseModel
from .conv_ae_model import _index_batches
from .ds_dataset import DSDataset
from .model_sizer import ModelSpec, create_model_spec


def _drawn(shape, n_bias):
    """a layer's default initialisation: weight, then bias (nn.Conv2d / ConvTranspose2d / Linear reset_parameters)"""
    w = torch.empty(shape)
    b = torch.empty(n_bias) if n_bias else None
    if b is None:
        import math
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    else:
        default_layer_init(w, b)
    return w, b


class Encoder(ParamBag):
    """unet.py:73-112"""

    def __init__(self, layers, encoded_space_dim, fc_size, dropout_rate=0.1):
        super().__init__()
        self.layers = list(layers)
        self.dropout_rate = dropout_rate
        for i, layer in enumerate(self.layers):
            (cin, _, _) = layer.get_input_dimensions()
            (cout, _, _) = layer.get_output_dimensions()
            (kh, kw) = layer.kernel_hw()
            (w, b) = _drawn((cout, cin, kh, kw), cout)
            self.add_param(f"encoder_cnn.{4 * i}.weight", w)
            self.add_param(f"encoder_cnn.{4 * i}.bias", b)
            add_batchnorm(self, f"encoder_cnn.{4 * i + 1}", cout)
        (chan, y, x) = self.layers[-1].get_output_dimensions()
        (w, b) = _drawn((fc_size, chan * y * x), fc_size)
        self.add_param("encoder_lin.0.weight", w)
        self.add_param("encoder_lin.0.bias", b)
        add_batchnorm(self, "encoder_lin.1", fc_size)
        (w, b) = _drawn((encoded_space_dim, fc_size), encoded_space_dim)
        self.add_param("encoder_lin.4.weight", w)
        self.add_param("encoder_lin.4.bias", b)

    def forward(self, x):
        raise RuntimeError("unet Encoder.forward on its own is not a product path: use UNET.score / apply / train")


class Decoder(ParamBag):
    """unet.py:114-163.  The reference constructs decoder_lin, then per layer ConvTranspose2d and (all but the last)
    ChannelAttention, and registers attention_layers before decoder_conv: draws follow construction order, the
    state_dict follows registration order."""

    def __init__(self, layers, encoded_space_dim, fc_size, dropout_rate=0.1):
        super().__init__()
        self.layers = list(layers)
        self.dropout_rate = dropout_rate
        (self.chan, self.y, self.x) = self.layers[0].get_input_dimensions()
        flat = self.chan * self.y * self.x
        lin0 = _drawn((fc_size, encoded_space_dim), fc_size)
        lin4 = _drawn((flat, fc_size), flat)
        convs, atts = [], []
        last = len(self.layers) - 1
        for j, layer in enumerate(self.layers):
            (cin, _, _) = layer.get_input_dimensions()
            (cout, _, _) = layer.get_output_dimensions()
            (kh, kw) = layer.kernel_hw()
            convs.append(_drawn((cin, cout, kh, kw), cout))
            if j != last:
                fc1 = _drawn((cout // 8, cout, 1, 1), 0)[0]
                fc2 = _drawn((cout, cout // 8, 1, 1), 0)[0]
                atts.append((fc1, fc2))
        self.add_param("decoder_lin.0.weight", lin0[0])
        self.add_param("decoder_lin.0.bias", lin0[1])
        add_batchnorm(self, "decoder_lin.1", fc_size)
        self.add_param("decoder_lin.4.weight", lin4[0])
        self.add_param("decoder_lin.4.bias", lin4[1])
        for j, (fc1, fc2) in enumerate(atts):
            self.add_param(f"attention_layers.{j}.fc1.weight", fc1)
            self.add_param(f"attention_layers.{j}.fc2.weight", fc2)
        for j, (w, b) in enumerate(convs):
            self.add_param(f"decoder_conv.{4 * j}.weight", w)
            self.add_param(f"decoder_conv.{4 * j}.bias", b)
            if j != last:
                add_batchnorm(self, f"decoder_conv.{4 * j + 1}", 2 * w.shape[1])

    def forward(self, x, x_skip):
        raise RuntimeError("unet Decoder.forward on its own is not a product path: use UNET.score / apply / train")


def unet_layer_spec(input_channels, output_channels, size, channels, kernel_size=4, stride=2, padding=1):
    """Hand-written layer definitions for a UNET (what --layer-definitions-path carries, cli/train_cae.py:143-147):
    `output_padding` is the PADDING the UNET modules pass to Conv2d / ConvTranspose2d (unet.py:82,140); decoder layer
    j > 0 takes 2 x channels (skip concat, :160); create_model_spec cannot produce such a spec."""
    from .model_sizer import LayerSpec
    (h, w) = size
    dims = [(input_channels, h, w)]
    enc = []
    for c in channels:
        (_, ph, pw) = dims[-1]
        dims.append((c, (ph + 2 * padding - kernel_size) // stride + 1, (pw + 2 * padding - kernel_size) // stride + 1))
        enc.append(LayerSpec(True, kernel_size, stride, dims[-2], dims[-1], padding))
    dec = []
    n = len(channels)
    for j in range(n):
        (src, dst) = (dims[n - j], dims[n - j - 1])
        cin = src[0] if j == 0 else 2 * src[0]
        cout = dst[0] if j < n - 1 else output_channels
        oh = (src[1] - 1) * stride - 2 * padding + kernel_size
        ow = (src[2] - 1) * stride - 2 * padding + kernel_size
        if (oh, ow) != (dst[1], dst[2]):
            raise ValueError(f"decoder layer {j} lands on {(oh, ow)}, the skip connection has {(dst[1], dst[2])}")
        dec.append(LayerSpec(False, kernel_size, stride, (cin, src[1], src[2]), (cout, oh, ow), padding))
    return ModelSpec(enc, dec)


class UNET(BaseModel):

    def __init__(self, normalise_input=True, normalise_output=True, batch_size=10, nr_epochs=500, test_interval=10,
                 encoded_dim_size=32, fc_size=128, lr=0.001, weight_decay=1e-5, dropout_rate=0.1, use_gpu=True,
                 conv_kernel_size=3, conv_stride=2, conv_input_layer_count=None, conv_output_layer_count=None,
                 database_path=None, lambda_l1=0.001, lambda_pearson=1, dropout_seed=0):
        super().__init__()
        self.normalise_input = normalise_input
        self.normalise_output = normalise_output
        self.normalisation_parameters = None
        self.input_shape = self.output_shape = None
        self.encoder = self.decoder = None
        self.batch_size = batch_size
        self.nr_epochs = nr_epochs
        self.test_interval = test_interval
        self.encoded_dim_size = encoded_dim_size
        self.fc_size = fc_size
        self.lr = lr
        self.weight_decay = weight_decay
        self.dropout_rate = dropout_rate
        self.use_gpu = use_gpu
        self.conv_kernel_size = conv_kernel_size
        self.conv_stride = conv_stride
        self.conv_input_layer_count = conv_input_layer_count
        self.conv_output_layer_count = conv_output_layer_count
        self.spec = None
        self.history = {"train_loss": [], "test_loss": [], "nr_epochs": 0}
        self.db = ModelDatabase(database_path) if database_path else None
        self.lambda_l1 = lambda_l1            # accepted and stored like the reference; unused there too
        self.lambda_pearson = lambda_pearson
        self.dropout_seed = dropout_seed
        self._engine = None
        self._steps_done = 0

    # ---- persistence ---------------------------------------------------------------------------------
    def get_parameters(self):
        return {
            "type": "UNET",
            "input_shape": list(self.input_shape),
            "output_shape": list(self.output_shape),
            "batch_size": self.batch_size,
            "test_interval": self.test_interval,
            "encoded_dim_size": self.encoded_dim_size,
            "fc_size": self.fc_size,
            "lr": self.lr,
            "lambda_pearson": self.lambda_pearson,
            "weight_decay": self.weight_decay,
            "dropout_rate": self.dropout_rate,
            "normalise_input": self.normalise_input,
            "normalise_output": self.normalise_output,
            "conv_kernel_size": self.conv_kernel_size,
            "conv_stride": self.conv_stride,
            "conv_input_layer_count": self.conv_input_layer_count,
            "conv_output_layer_count": self.conv_output_layer_count,
            "model_id": self.get_model_id(),
        }

    def summary(self):
        if not self.spec:
            return "Model has not been trained - no layers assigned yet"
        fc = f"\tFully Connected Layer:\n\t\tsize={self.fc_size}\n"
        return ("Model Summary:\n" + "".join(str(l) for l in self.spec.input_layers) + fc
                + f"\tLatent Vector:\n\t\tsize={self.encoded_dim_size}\n" + fc
                + "".join(str(l) for l in self.spec.output_layers))

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
                            "parameters.json": json.dumps(self.get_parameters()),
                            "spec.json": json.dumps(self.spec.save()),
                            "history.json": json.dumps(self.history),
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
        self.input_shape = tuple(p["input_shape"])
        self.output_shape = tuple(p["output_shape"])
        for key in ("batch_size", "test_interval", "encoded_dim_size", "fc_size", "lr", "weight_decay", "normalise_input",
                    "normalise_output"):
            setattr(self, key, p[key])
        for key in ("conv_kernel_size", "conv_stride", "conv_input_layer_count", "conv_output_layer_count"):
            setattr(self, key, p.get(key, None))
        with open(os.path.join(from_folder, "history.json")) as f:
            self.history = json.loads(f.read())
        with open(os.path.join(from_folder, "spec.json")) as f:
            self.spec = ModelSpec()
            self.spec.load(json.loads(f.read()))
        self.encoder = Encoder(self.spec.get_input_layers(), encoded_space_dim=self.encoded_dim_size, fc_size=self.fc_size,
                               dropout_rate=self.dropout_rate)
        self.decoder = Decoder(self.spec.get_output_layers(), encoded_space_dim=self.encoded_dim_size, fc_size=self.fc_size,
                               dropout_rate=self.dropout_rate)
        self.encoder.load_state_dict(self.torch_load(os.path.join(from_folder, "encoder.weights")))
        self.decoder.load_state_dict(self.torch_load(os.path.join(from_folder, "decoder.weights")))
        self._engine = None
        super().load(from_folder)

    # ---- engine ----------------------------------------------------------------------------------------
    def _get_engine(self, max_batch):
        if self._engine is None or self._engine.max_batch < max_batch:
            if self._engine is not None:
                self._pull_weights()
            eng = _ue.UnetEngine(self.spec, self.fc_size, self.encoded_dim_size, max_batch=max_batch)
            eng.load_state(self.encoder.state_dict(), self.decoder.state_dict())
            self._engine = eng
        return self._engine

    def _score_device(self, x):
        eng = self._get_engine(max(1, min(int(self.batch_size), int(x.shape[0]))))
        return eng.score(x)

    def score(self, batches, save_arr):
        """eval-mode forward of a list of (B,C,H,W) batches into save_arr (:373-382)"""
        ctr = 0
        for batch in batches:
            x = torch.as_tensor(batch, dtype=torch.float32)
            y = self._score_device(x.cuda() if not x.is_cuda else x).cpu().numpy()
            save_arr[ctr:ctr + y.shape[0], :, :, :] = y
            ctr += self.batch_size

    # ---- training --------------------------------------------------------------------------------------
    def train(self, input_variables, output_variable, training_ds, testing_ds, model_path="", training_paths="",
              testing_paths="", mask_variable_name=None):
        print("initiating train method")
        train_ds = DSDataset(training_ds, input_variables, output_variable, normalise_in=self.normalise_input,
                             normalise_out=self.normalise_output, mask_variable_name=mask_variable_name)
        print("loaded train_ds to train method")
        self.set_input_spec(train_ds.get_input_spec())
        self.set_output_spec(train_ds.get_output_spec())
        self.normalisation_parameters = train_ds.get_normalisation_parameters()
        test_ds = DSDataset(testing_ds, input_variables, output_variable, normalise_in=self.normalise_input,
                            normalise_out=self.normalise_output, mask_variable_name=mask_variable_name)
        test_ds.set_normalisation_parameters(self.normalisation_parameters)
        self.input_shape = tuple(train_ds.get_input_shape())
        self.output_shape = tuple(train_ds.get_output_shape())
        (input_chan, input_y, input_x) = self.input_shape
        (output_chan, output_y, output_x) = self.output_shape
        print("finished loading train_ds and test_ds from DSDataset")
        if not self.spec:
            self.spec = create_model_spec(input_size=(input_y, input_x), input_channels=input_chan,
                                          output_size=(output_y, output_x), output_channels=output_chan,
                                          kernel_size=self.conv_kernel_size, stride=self.conv_stride,
                                          input_layer_count=self.conv_input_layer_count,
                                          output_layer_count=self.conv_output_layer_count)
        if not self.encoder:
            self.encoder = Encoder(self.spec.get_input_layers(), encoded_space_dim=self.encoded_dim_size,
                                   fc_size=self.fc_size, dropout_rate=self.dropout_rate)
        if not self.decoder:
            self.decoder = Decoder(self.spec.get_output_layers(), encoded_space_dim=self.encoded_dim_size,
                                   fc_size=self.fc_size, dropout_rate=self.dropout_rate)
        # frozen shuffles in the reference's order (:440-441, both loaders shuffle)
        train_perm = _index_batches(len(train_ds), self.batch_size)
        test_perm = _index_batches(len(test_ds), self.batch_size)
        print("finished train_loarder and test_loader")
        print(f"Running on device: {torch.device('cuda')}")
        start = time.time()

        eng = self._get_engine(int(self.batch_size))
        eng.set_hyper(lr=self.lr, weight_decay=self.weight_decay, dropout_rate=self.dropout_rate,
                      lambda_pearson=self.lambda_pearson, seed=self.dropout_seed)
        eng.reset_optimizer()       # AdamW is re-created on every train() (:457)
        eng.set_step(0)
        t0 = time.time()
        for (which, ds) in ((_ue.TRAIN, train_ds), (_ue.TEST, test_ds)):
            eng.set_dataset(which, ds.device_inputs(), ds.device_outputs(), self._loss_mask(ds))
        train_idx = eng.upload_perm(train_perm)
        test_idx = eng.upload_perm(test_perm)
        print(f"finished batching in {time.time() - t0:.2f} seconds")

        train_loss = test_loss = 0.0
        try:
            for epoch in range(self.nr_epochs):
                e0 = time.time()
                losses = eng.run_batches(_ue.TRAIN, train_idx, len(train_ds), self.batch_size, train=True)
                print(f"time used for training one epoch: {time.time() - e0:.2f}")
                train_loss = float(np.mean([l[0] for l in losses]))
                train_pearson_loss = float(np.mean([l[1] for l in losses]))
                if epoch % self.test_interval == 0:
                    tl = eng.run_batches(_ue.TEST, test_idx, len(test_ds), self.batch_size, train=False)
                    test_loss = float(np.mean([l[0] for l in tl]))
                    test_pearson_loss = float(np.mean([l[1] for l in tl]))
                    self.history["train_loss"].append(train_loss)
                    self.history["test_loss"].append(test_loss)
                    print(f"epoch: {epoch}, train_mse: {train_loss:.6f}, train_pearson_loss: {train_pearson_loss:.4f}, "
                          f"test_mse: {test_loss:.6f}, test_pearson_loss: {test_pearson_loss:.4f}")
                    print(f"learn rate: {self.lr:.6f}")
        except KeyboardInterrupt:
            print("Training interrupted. Performing cleanup...")
        elapsed = time.time() - start
        self.history["nr_epochs"] += self.nr_epochs
        print("elapsed:" + str(elapsed))

        if self.db:
            self.db.add_training_result(self.get_model_id(), "UNET", output_variable, input_variables, self.summary(),
                                        model_path, training_paths, train_loss, testing_paths, test_loss,
                                        self.get_parameters(), self.spec.save())
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

    def _loss_mask(self, ds):
        """(N, 1|C, H, W) fp32 mask for the loss, or None for all ones counted over the output's channels"""
        m = ds.device_mask()
        if m is not None:
            return m
        (cin, _, _) = self.input_shape
        (cout, oy, ox) = self.output_shape
        if cin == cout:
            return None
        if cin == 1:
            return torch.ones((len(ds), 1, oy, ox), dtype=torch.float32, device="cuda")
        raise ValueError(f"without a mask variable the reference's mask has the input's shape ({cin} channels), which "
                         f"does not broadcast against {cout} output channels")
