"""CPU oracle for the UNET path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/ (and bench legs that time a CPU baseline) may import this module.

Parity pin: the reference's unet module cannot be imported in the build image (its first lines import
torchvision and xarray; both are absent and are not stubbed), and the reference holds no fixtures for it.
tests/golden/make_golden_unet.py therefore compiles, from the file's syntax tree, only the definitions that
need torch alone — class ChannelAttention :23-39, class Encoder :73-112, class Decoder :114-163,
UNET.masked_mse_loss :635-639, UNET.pearson_corr_torch :641-678 — and drives them with the reference's
step (:307-325; AdamW :457; constant learning rate: CosineAnnealingLR with eta_min == lr, :459).  The
vectors it stored (tests/golden/unet_*.npz) pin every function of this file (tests/test_unet_oracle_golden.py).
This file itself is a restatement from the text of unet.py, composed of the same torch.nn.functional
operations the reference's torch.nn modules run.  The VGG perceptual loss is constructed by the reference
but never enters the loss (:316-322) and is omitted.

Dropout: the reference draws masks from torch's global generator, which no other implementation can
reproduce.  Here (and in the HIP path) a mask is a pure function of (seed, step, site, element index):
`dropout_keep` below, a PCG-style integer hash.  With dropout_rate = 0 the step is the reference's
arithmetic exactly; with dropout_rate > 0 it is the reference's arithmetic for THESE masks.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# dropout sites
SITE_ENC_CONV = 0      # + layer index
SITE_ENC_FC0 = 100
SITE_ENC_FC1 = 101
SITE_DEC_FC0 = 102
SITE_DEC_FC1 = 103
SITE_DEC_CONV = 200    # + layer index


def _pcg(v):
    """PCG-RXS-M-XS 32-bit output permutation used as a hash; numpy uint32 arithmetic wraps mod 2^32"""
    v = np.asarray(v, dtype=np.uint32)
    with np.errstate(over="ignore"):
        state = v * np.uint32(747796405) + np.uint32(2891336453)
        word = ((state >> ((state >> np.uint32(28)) + np.uint32(4))) ^ state) * np.uint32(277803737)
    return (word >> np.uint32(22)) ^ word


def dropout_key(seed, step, site):
    with np.errstate(over="ignore"):
        k = _pcg(np.uint32(seed & 0xFFFFFFFF) + np.uint32(0x9E3779B9) * np.uint32(site))
        return _pcg(k ^ np.uint32(step & 0xFFFFFFFF))


def dropout_keep(seed, step, site, shape, p):
    """boolean keep-mask for a tensor of `shape`, element i (row-major) kept iff hash >= floor(p * 2^32)"""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    key = dropout_key(seed, step, site)
    with np.errstate(over="ignore"):
        hi = _pcg(key + (idx >> np.uint64(32)).astype(np.uint32))
        r = _pcg(idx.astype(np.uint32) ^ hi)
    thr = np.uint32(min(int(p * 4294967296.0), 0xFFFFFFFF))
    return (r >= thr).reshape(shape)


class Dropper:
    """applies F.dropout's arithmetic (x * keep * 1/(1-p)) with hash masks; identity in eval mode or at p = 0"""

    def __init__(self, p, seed, step, train):
        self.p, self.seed, self.step, self.train = float(p), int(seed), int(step), bool(train)

    def __call__(self, x, site):
        if not self.train or self.p == 0.0:
            return x
        keep = torch.from_numpy(dropout_keep(self.seed, self.step, site, tuple(x.shape), self.p))
        return x * (keep.to(x.dtype) * (1.0 / (1.0 - self.p)))


# ReLU decisions.  An fp32 implementation that sums in another order can land a BatchNorm output on the other side of zero
# where the exact value is within rounding of it; the element's whole upstream gradient then switches on or off, and one
# such element moves a layer's weight gradient by percents.  A parity test can hand the oracle the decisions the
# implementation under test took (`ReluAlign`): they are followed only where the oracle's own pre-activation is within
# `tol` of zero, so a wrong decision anywhere else still shows.
_relu_hook = None


class ReluAlign:
    """`decisions[name]`: boolean array (output > 0) per ReLU site — 'enc{i}', 'efc0', 'efc1', 'dfc0', 'dfc1', 'dec{j}'.
    Taken from outputs AFTER a dropout, a False may also mean 'dropped': harmless, the element is zero either way."""

    def __init__(self, decisions, tol=1e-5):
        self.decisions, self.tol, self.followed = decisions, float(tol), {}
        self.disagree = {}      # per site: (elements decided differently, the largest |pre-activation| among them)

    def __call__(self, name, x):
        own = x > 0
        want = self.decisions.get(name)
        if want is None:
            return F.relu(x)
        want = torch.as_tensor(np.asarray(want)).reshape(x.shape)
        diff = own != want
        if name.startswith(("enc", "efc")) or not bool(diff.any()):      # (sites read before their dropout: exact)
            self.disagree[name] = (int(diff.sum()), float(x.detach().abs()[diff].max()) if bool(diff.any()) else 0.0)
        follow = diff & (x.detach().abs() < self.tol)
        self.followed[name] = self.followed.get(name, 0) + int(follow.sum())
        return x * torch.where(follow, want, own).to(x.dtype)

    def __enter__(self):
        global _relu_hook
        _relu_hook = self
        return self

    def __exit__(self, *exc):
        global _relu_hook
        _relu_hook = None
        return False


def _relu(x, name):
    return F.relu(x) if _relu_hook is None else _relu_hook(name, x)


def _bn(x, st, key, train):
    y = F.batch_norm(x, st[key + ".running_mean"], st[key + ".running_var"], st[key + ".weight"], st[key + ".bias"],
                     training=train, momentum=BN_MOMENTUM, eps=BN_EPS)
    if train:
        st[key + ".num_batches_tracked"] += 1
    return y


def encoder_forward(spec, enc, x, train, drop):
    """unet.py:102-112.  ModuleList indices: (conv, bn, relu, dropout) per layer -> conv 4i, bn 4i+1 (:77-85);
    encoder_lin: Linear 0, BatchNorm1d 1, ReLU 2, Dropout 3, Linear 4, ReLU 5, Dropout 6 (:92-100).
    The skip is the ReLU output (taken before the dropout, which is not in-place) (:106-107)."""
    skips = []
    h = x
    for i, l in enumerate(spec["input_layers"]):
        c, b = f"encoder_cnn.{4 * i}", f"encoder_cnn.{4 * i + 1}"
        h = F.conv2d(h, enc[c + ".weight"], enc[c + ".bias"], stride=int(l["stride"]), padding=int(l["output_padding"]))
        h = _relu(_bn(h, enc, b, train), f"enc{i}")
        skips.append(h)
        h = drop(h, SITE_ENC_CONV + i)
    h = h.flatten(1)
    h = F.linear(h, enc["encoder_lin.0.weight"], enc["encoder_lin.0.bias"])
    h = drop(_relu(_bn(h, enc, "encoder_lin.1", train), "efc0"), SITE_ENC_FC0)
    h = drop(_relu(F.linear(h, enc["encoder_lin.4.weight"], enc["encoder_lin.4.bias"]), "efc1"), SITE_ENC_FC1)
    skips.pop()
    return h, skips


def channel_attention(dec, j, x):
    """unet.py:35-39: sigmoid(fc2(relu(fc1(avgpool(x)))) + fc2(relu(fc1(maxpool(x)))))"""
    w1, w2 = dec[f"attention_layers.{j}.fc1.weight"], dec[f"attention_layers.{j}.fc2.weight"]
    avg = F.adaptive_avg_pool2d(x, 1)
    mx = F.adaptive_max_pool2d(x, 1)
    a = F.conv2d(F.relu(F.conv2d(avg, w1)), w2)
    m = F.conv2d(F.relu(F.conv2d(mx, w1)), w2)
    return torch.sigmoid(a + m)


def decoder_forward(spec, dec, z, skips, train, drop):
    """unet.py:149-163.  decoder_lin as encoder_lin; decoder_conv ModuleList: ConvTranspose2d 4j, BatchNorm2d(2C)
    4j+1, ReLU, Dropout for every layer but the last (:133-147)."""
    layers = spec["output_layers"]
    (c0, y0, x0) = layers[0]["input_dimensions"]
    h = F.linear(z, dec["decoder_lin.0.weight"], dec["decoder_lin.0.bias"])
    h = drop(_relu(_bn(h, dec, "decoder_lin.1", train), "dfc0"), SITE_DEC_FC0)
    h = drop(_relu(F.linear(h, dec["decoder_lin.4.weight"], dec["decoder_lin.4.bias"]), "dfc1"), SITE_DEC_FC1)
    h = h.view(h.shape[0], c0, y0, x0)
    rev = skips[::-1]
    for j, l in enumerate(layers):
        c, b = f"decoder_conv.{4 * j}", f"decoder_conv.{4 * j + 1}"
        h = F.conv_transpose2d(h, dec[c + ".weight"], dec[c + ".bias"], stride=int(l["stride"]),
                               padding=int(l["output_padding"]))
        if j < len(rev):
            h = h * channel_attention(dec, j, h)
            h = torch.cat((h, rev[j]), 1)
        if j != len(layers) - 1:
            h = drop(_relu(_bn(h, dec, b, train), f"dec{j}"), SITE_DEC_CONV + j)
    return torch.sigmoid(h)


def masked_mse_loss(pred, target, mask):
    """unet.py:635-639"""
    diff = (pred - target) * mask
    return torch.sum(diff ** 2) / torch.sum(mask)


def pearson_corr(decoded, high_res, mask):
    """unet.py:641-678: masked Pearson correlation per (batch, channel)"""
    d = decoded.reshape(decoded.size(0), decoded.size(1), -1)
    t = high_res.reshape(high_res.size(0), high_res.size(1), -1)
    m = mask.reshape(mask.size(0), mask.size(1), -1).float()
    n = torch.sum(m, dim=2, keepdim=True)
    mean_d = torch.sum(d * m, dim=2, keepdim=True) / (n + 1e-8)
    mean_t = torch.sum(t * m, dim=2, keepdim=True) / (n + 1e-8)
    dc = d - mean_d
    tc = t - mean_t
    std_d = torch.sqrt(torch.sum(m * (d - mean_d) ** 2, dim=2, keepdim=True) / (n + 1e-8) + 1e-8)
    std_t = torch.sqrt(torch.sum(m * (t - mean_t) ** 2, dim=2, keepdim=True) / (n + 1e-8) + 1e-8)
    num = torch.sum(m * (dc / std_d) * (tc / std_t), dim=2)
    return num / torch.sum(m, dim=2)


def is_param(key):
    return not ("running_" in key or "num_batches_tracked" in key)


class UnetOracle:
    """encoder / decoder tensors under the reference's state_dict keys, stepped as unet.py:307-325 does"""

    def __init__(self, spec, enc_state, dec_state, lr=1e-3, weight_decay=1e-5, dropout_rate=0.1, lambda_pearson=1.0,
                 seed=0):
        self.spec = spec
        self.dropout_rate, self.lambda_pearson, self.seed = float(dropout_rate), float(lambda_pearson), int(seed)
        self.step_count = 0
        self.enc, self.dec = OrderedDict(), OrderedDict()
        for (dst, src) in ((self.enc, enc_state), (self.dec, dec_state)):
            for k, v in src.items():
                t = torch.as_tensor(np.array(v)) if not torch.is_tensor(v) else v.detach().clone()
                dst[k] = t.requires_grad_(True) if is_param(k) else t
        self.optim = torch.optim.AdamW([v for k, v in self.enc.items() if is_param(k)]
                                       + [v for k, v in self.dec.items() if is_param(k)], lr=lr, weight_decay=weight_decay)

    def forward(self, x, train, step=None):
        drop = Dropper(self.dropout_rate, self.seed, self.step_count if step is None else step, train)
        (z, skips) = encoder_forward(self.spec, self.enc, x, train, drop)
        return decoder_forward(self.spec, self.dec, z, skips, train, drop)

    def eval_forward(self, x):
        with torch.no_grad():
            return self.forward(x, train=False)

    def losses(self, y, t, mask):
        mse = masked_mse_loss(y, t, mask)
        pl = 1 - torch.mean(pearson_corr(y, t, mask))
        return mse, pl

    def eval_losses(self, x, t, mask):
        with torch.no_grad():
            (mse, pl) = self.losses(self.forward(x, train=False), t, mask)
        return float(mse), float(pl)

    def loss_and_grads(self, x, t, mask):
        y = self.forward(x, train=True)
        (mse, pl) = self.losses(y, t, mask)
        self.optim.zero_grad()
        (mse + self.lambda_pearson * pl).backward()
        return float(mse.detach()), float(pl.detach()), y.detach()

    def train_step(self, x, t, mask):
        (mse, pl, _) = self.loss_and_grads(x, t, mask)
        self.optim.step()
        self.step_count += 1
        return mse, pl

    def grads(self):
        out = OrderedDict()
        for (pre, st) in (("enc/", self.enc), ("dec/", self.dec)):
            for k, v in st.items():
                if is_param(k):
                    out[pre + k] = v.grad.detach().clone()
        return out

    def state(self):
        out = OrderedDict()
        for (pre, st) in (("enc/", self.enc), ("dec/", self.dec)):
            for k, v in st.items():
                out[pre + k] = v.detach().clone()
        return out
