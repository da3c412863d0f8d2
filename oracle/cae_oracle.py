"""CPU oracle for the ConvAE hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (cae_tools_amd) never does and fails loudly without its HIP
library.

What it is: a functional (torch.nn.functional, fp32, CPU) restatement of the arithmetic
the reference runs for one training / scoring step, written from the reference's
behaviour, parameterised by a spec dict (the JSON form of model_sizer.ModelSpec.save(),
reference model_sizer.py:85-89) and a flat dict of named tensors that uses the reference's
state_dict keys.

Parity pin: tests/test_oracle_golden.py checks every function here against the vectors in
tests/golden/*.npz, which tests/golden/make_golden.py produced by importing and running the
reference's own Encoder / Decoder / create_model_spec / DSDataset.  The
third-party arithmetic underneath both is PyTorch (version unpinned by the reference;
2.10.0 in this image).

Reference sites restated:
  Encoder.forward            src/cae_tools/models/encoder.py:40-64
  Decoder.forward            src/cae_tools/models/decoder.py:31-50,73-78
  train step                 src/cae_tools/models/conv_ae_model.py:189-200 (MSELoss :303, Adam :310)
  test / score step          src/cae_tools/models/conv_ae_model.py:205-239
  DSDataset normalisation    src/cae_tools/models/ds_dataset.py:49-67,99-113,131-135
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # torch.nn.BatchNorm2d default, encoder.py:45 / decoder.py:47
BN_MOMENTUM = 0.1   # torch.nn.BatchNorm2d default


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (list, tuple)) else (int(v), int(v))


def layer_table(spec):
    """[(kind, key_prefix, layer_dict, has_bn)] in execution order, with state_dict prefixes.

    nn.Sequential indices: encoder_cnn has (conv, bn, relu) triples -> conv at 3i, bn at 3i+1
    (encoder.py:40-46); decoder_conv the same except the last layer has no bn/relu
    (decoder.py:40-48)."""
    rows = []
    for i, l in enumerate(spec["input_layers"]):
        rows.append(("conv", f"encoder_cnn.{3 * i}", f"encoder_cnn.{3 * i + 1}", l, True))
    n_out = len(spec["output_layers"])
    for i, l in enumerate(spec["output_layers"]):
        rows.append(("convt", f"decoder_conv.{3 * i}", f"decoder_conv.{3 * i + 1}", l, i != n_out - 1))
    return rows


def _relu(z, name, trace, relu_fix):
    """F.relu, with two test hooks: the pre-activation is recorded as trace["z:" + name], and relu_fix[name] (a tensor d of
    z's shape with entries -1 / 0 / +1) overrides the DERIVATIVE at single positions - +1: pass the gradient although z <= 0,
    -1: block it although z > 0 - without changing the value.  At a large batch some BatchNorm output lies within fp32
    rounding of zero every few steps, and two correct fp32 implementations then disagree on that one mask bit; a single
    flipped bit moves every upstream gradient by ~1/sqrt(N) of itself (the sums cancel that far), which says nothing about
    either implementation.  The parity tests therefore give the oracle the HIP path's decisions at exactly those positions
    (after checking that its own |z| there is rounding-sized) and compare what is left: tests/helpers.py."""
    if trace is not None:
        trace["z:" + name] = z.detach()
    h = F.relu(z)
    if relu_fix is not None and name in relu_fix:
        h = h + relu_fix[name].to(z.dtype) * (z - z.detach())
    return h


def encoder_forward(spec, enc, x, train, trace=None, relu_fix=None):
    """encoder.py:60-64.  `enc` maps reference state_dict keys to tensors; running stats are
    updated in place when train=True (F.batch_norm semantics)."""
    h = x
    for i, l in enumerate(spec["input_layers"]):
        c, b = f"encoder_cnn.{3 * i}", f"encoder_cnn.{3 * i + 1}"
        h = F.conv2d(h, enc[c + ".weight"], enc[c + ".bias"], stride=int(l["stride"]))
        if trace is not None:
            trace[f"enc_conv{i}"] = h.detach()
        h = F.batch_norm(h, enc[b + ".running_mean"], enc[b + ".running_var"], enc[b + ".weight"],
                         enc[b + ".bias"], training=train, momentum=BN_MOMENTUM, eps=BN_EPS)
        if train:
            enc[b + ".num_batches_tracked"] += 1
        h = _relu(h, f"enc_conv{i}", trace, relu_fix)
    h = h.flatten(1)
    h = _relu(F.linear(h, enc["encoder_lin.0.weight"], enc["encoder_lin.0.bias"]), "enc_fc0", trace, relu_fix)
    if trace is not None:
        trace["enc_fc0"] = h.detach()
    z = F.linear(h, enc["encoder_lin.2.weight"], enc["encoder_lin.2.bias"])
    return z


def decoder_forward(spec, dec, z, train, trace=None, relu_fix=None):
    """decoder.py:73-78."""
    layers = spec["output_layers"]
    (c0, y0, x0) = layers[0]["input_dimensions"]
    h = _relu(F.linear(z, dec["decoder_lin.0.weight"], dec["decoder_lin.0.bias"]), "dec_fc0", trace, relu_fix)
    if trace is not None:
        trace["dec_fc0"] = h.detach()
    h = F.linear(h, dec["decoder_lin.2.weight"], dec["decoder_lin.2.bias"])
    if trace is not None:
        trace["dec_fc1"] = h.detach()
    h = h.view(h.shape[0], c0, y0, x0)
    for i, l in enumerate(layers):
        c, b = f"decoder_conv.{3 * i}", f"decoder_conv.{3 * i + 1}"
        h = F.conv_transpose2d(h, dec[c + ".weight"], dec[c + ".bias"], stride=int(l["stride"]),
                               padding=0, output_padding=int(l["output_padding"]))
        if trace is not None:
            trace[f"dec_conv{i}"] = h.detach()
        if i != len(layers) - 1:
            h = F.batch_norm(h, dec[b + ".running_mean"], dec[b + ".running_var"], dec[b + ".weight"],
                             dec[b + ".bias"], training=train, momentum=BN_MOMENTUM, eps=BN_EPS)
            if train:
                dec[b + ".num_batches_tracked"] += 1
            h = _relu(h, f"dec_conv{i}", trace, relu_fix)
    return torch.sigmoid(h)


def is_param(key):
    return not ("running_" in key or "num_batches_tracked" in key)


class OracleModel:
    """Holds encoder/decoder tensors under the reference's state_dict keys and steps them the
    way conv_ae_model.py does.  Parameters are autograd leaves; Adam is torch.optim.Adam over
    the two parameter groups in the reference's order (conv_ae_model.py:305-310)."""

    def __init__(self, spec, enc_state, dec_state, lr=1e-3, weight_decay=1e-5):
        self.spec = spec
        self.enc = OrderedDict()
        self.dec = OrderedDict()
        for k, v in enc_state.items():
            t = torch.as_tensor(np.array(v)) if not torch.is_tensor(v) else v.detach().clone()
            self.enc[k] = t.requires_grad_(True) if is_param(k) else t
        for k, v in dec_state.items():
            t = torch.as_tensor(np.array(v)) if not torch.is_tensor(v) else v.detach().clone()
            self.dec[k] = t.requires_grad_(True) if is_param(k) else t
        self.optim = torch.optim.Adam(
            [{"params": [v for k, v in self.enc.items() if is_param(k)]},
             {"params": [v for k, v in self.dec.items() if is_param(k)]}],
            lr=lr, weight_decay=weight_decay)

    # -- forward variants ---------------------------------------------------------------
    def forward(self, x, train, trace=None, relu_fix=None):
        z = encoder_forward(self.spec, self.enc, x, train, trace, relu_fix)
        if trace is not None:
            trace["latent"] = z.detach()
        return decoder_forward(self.spec, self.dec, z, train, trace, relu_fix)

    def relu_inputs(self, x):
        """{name: pre-activation of every ReLU} of a train-mode forward that leaves no trace on the model (running statistics
        and batch counters put back): what tests/helpers.py compares the HIP path's mask decisions with"""
        keep = {k: v.clone() for side in (self.enc, self.dec) for k, v in side.items() if not is_param(k)}
        trace = {}
        with torch.no_grad():
            self.forward(x, train=True, trace=trace)
        for side in (self.enc, self.dec):
            for k in side:
                if not is_param(k):
                    side[k].copy_(keep[k])
        return {k[2:]: v for k, v in trace.items() if k.startswith("z:")}

    def eval_forward(self, x):
        """score(): conv_ae_model.py:223-239 (eval mode, no grad)."""
        with torch.no_grad():
            return self.forward(x, train=False)

    def eval_loss(self, x, t):
        """one batch of __test_epoch: conv_ae_model.py:205-221."""
        with torch.no_grad():
            return float(F.mse_loss(self.forward(x, train=False), t))

    # -- training -----------------------------------------------------------------------
    def loss_and_grads(self, x, t, trace=None, relu_fix=None):
        """forward(train) + MSELoss + backward: conv_ae_model.py:191-196.  Returns the loss and
        leaves .grad on every parameter."""
        y = self.forward(x, train=True, trace=trace, relu_fix=relu_fix)
        loss = F.mse_loss(y, t)
        self.optim.zero_grad()
        loss.backward()
        return float(loss.detach()), y.detach()

    def train_step(self, x, t, relu_fix=None):
        """one iteration of __train_epoch: conv_ae_model.py:189-200."""
        loss, _ = self.loss_and_grads(x, t, relu_fix=relu_fix)
        self.optim.step()
        return loss

    def grads(self):
        out = OrderedDict()
        for k, v in self.enc.items():
            if is_param(k):
                out["enc/" + k] = v.grad.detach().clone()
        for k, v in self.dec.items():
            if is_param(k):
                out["dec/" + k] = v.grad.detach().clone()
        return out

    def state(self):
        out = OrderedDict()
        for k, v in self.enc.items():
            out["enc/" + k] = v.detach().clone()
        for k, v in self.dec.items():
            out["dec/" + k] = v.detach().clone()
        return out


# ---------------------------------------------------------------------------------------
# loader arithmetic (ds_dataset.py)
# ---------------------------------------------------------------------------------------

def scan_variable(values):
    """(nan_count, min, max) as the reference computes them (ds_dataset.py:43-46,53-58):
    python floats of np.nanmin / np.nanmax over the whole variable."""
    values = np.asarray(values)
    return (int(np.sum(np.where(np.isnan(values), 1, 0))), float(np.nanmin(values)),
            float(np.nanmax(values)))


def normalise_variable(arr, vmin, vmax):
    """ds_dataset.py:99-113 for a float32 array and python-float min/max: the subtraction and
    the division happen in float32 with the fp64 range rounded to float32; range 0 -> 0.0."""
    arr = np.asarray(arr, dtype=np.float32)
    rng = vmax - vmin
    if rng == 0:
        return np.zeros_like(arr)
    return ((arr - np.float32(vmin)) / np.float32(rng)).astype(np.float32)


def pack_inputs(variables, mins, maxs, normalise=True):
    """ds_dataset.py:137-147: channel-concatenate the (normalised) input variables into one
    float32 (N, C, H, W) array.  `variables` is a list of (N, Cv, H, W) arrays."""
    parts = []
    for v, lo, hi in zip(variables, mins, maxs):
        parts.append(normalise_variable(v, lo, hi) if normalise else np.asarray(v, dtype=np.float32))
    return np.concatenate(parts, axis=1)


def denormalise_output(arr, vmin, vmax):
    """ds_dataset.py:131-135 on the float64 score array of base_model.py:123,151."""
    return vmin + (np.asarray(arr, dtype=np.float64) * (vmax - vmin))
