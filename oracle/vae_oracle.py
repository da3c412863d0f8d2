"""CPU definition of the 'var' (variational) autoencoder path and its MS-SSIM loss  --  TEST INFRASTRUCTURE.

PARITY UNPINNED: the reference has NO source for this path.  `--method var` is the default of its train_cae CLI
(cli/train_cae.py:42) and `cae_tools.models.var_ae_model` is imported by model_evaluator.py:35, but the file is missing
from the repository; only the CLI flags survive (--lambda-mse / --lambda-kl / --lambda-ssim, cli/train_cae.py:32-36) and
README.md:29 names the `pytorch_msssim` package (not installed here, no call site in the reference).  This module is
therefore the build's OWN published definition; the HIP path is tested against it and
nothing here can be checked against the reference.

Definition
  encoder   the ConvAE encoder stack (Conv2d k s -> BatchNorm2d -> ReLU per layer, encoder.py:40-46), flatten,
            Linear(F, fc) -> ReLU, then two heads Linear(fc, latent): mu and logvar
  sample    z = mu + eps * exp(0.5 * logvar) in training (eps ~ N(0,1) from the hash below), z = mu in eval / scoring
  decoder   the ConvAE decoder (decoder.py:31-50,73-78): Linear -> ReLU -> Linear, ConvTranspose2d (-> BN -> ReLU) stack, sigmoid
  loss      lambda_mse * mean((y - t)^2) + lambda_kl * (-0.5 * mean(1 + logvar - mu^2 - exp(logvar)))
            + lambda_ssim * (1 - MS-SSIM(y, t))
  MS-SSIM   Wang et al. 2003 as implemented by pytorch_msssim: 11-tap gaussian window (sigma 1.5), valid convolution,
            data_range 1, K = (0.01, 0.03), 5 scales with weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333), 2x2 average
            pooling (padding = size mod 2) between scales, ReLU on the per-scale terms, mean over (batch, channel)
  optimiser Adam with L2 weight decay (as the ConvAE path, conv_ae_model.py:310)
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .cae_oracle import BN_EPS, BN_MOMENTUM, decoder_forward, is_param
from .unet_oracle import _pcg

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)
WIN_SIZE, WIN_SIGMA, K1, K2 = 11, 1.5, 0.01, 0.03


def gaussian_window():
    c = torch.arange(WIN_SIZE, dtype=torch.float32) - WIN_SIZE // 2
    g = torch.exp(-(c ** 2) / (2 * WIN_SIGMA ** 2))
    return g / g.sum()


def _filter(x, g):
    """separable valid gaussian filtering of (B,C,H,W)"""
    c = x.shape[1]
    x = F.conv2d(x, g.view(1, 1, -1, 1).repeat(c, 1, 1, 1), groups=c)
    return F.conv2d(x, g.view(1, 1, 1, -1).repeat(c, 1, 1, 1), groups=c)


def _ssim_cs(x, y, g):
    (c1, c2) = (K1 ** 2, K2 ** 2)
    (mu1, mu2) = (_filter(x, g), _filter(y, g))
    (s11, s22, s12) = (_filter(x * x, g) - mu1 * mu1, _filter(y * y, g) - mu2 * mu2, _filter(x * y, g) - mu1 * mu2)
    cs_map = (2 * s12 + c2) / (s11 + s22 + c2)
    ssim_map = ((2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ms_ssim(x, y):
    """mean over (batch, channel) of prod_s cs_s^w_s * ssim_last^w_last; x, y (B,C,H,W) in [0,1]"""
    g = gaussian_window()
    terms = []
    for s in range(len(MS_WEIGHTS)):
        (ssim_c, cs) = _ssim_cs(x, y, g)
        if s < len(MS_WEIGHTS) - 1:
            terms.append(torch.relu(cs))
            pad = [d % 2 for d in x.shape[2:]]
            x = F.avg_pool2d(x, kernel_size=2, padding=pad)
            y = F.avg_pool2d(y, kernel_size=2, padding=pad)
    terms.append(torch.relu(ssim_c))
    stack = torch.stack(terms, dim=0)
    w = torch.tensor(MS_WEIGHTS, dtype=stack.dtype).view(-1, 1, 1)
    return torch.prod(stack ** w, dim=0).mean()


def normal_noise(seed, step, shape):
    """eps ~ N(0,1): Box-Muller on two hashes of (seed, step, element index); numpy float32, same bits as the HIP kernel"""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint32)
    with np.errstate(over="ignore"):
        key = _pcg(_pcg(np.uint32(seed & 0xFFFFFFFF) + np.uint32(0x9E3779B9) * np.uint32(977)) ^ np.uint32(step & 0xFFFFFFFF))
        h1 = _pcg(idx * np.uint32(2) ^ key)
        h2 = _pcg((idx * np.uint32(2) + np.uint32(1)) ^ key)
    u1 = (h1.astype(np.float64) + 1.0) / 4294967296.0         # (0, 1]
    u2 = h2.astype(np.float64) / 4294967296.0
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)).astype(np.float32).reshape(shape)


def encoder_forward(spec, enc, x, train):
    h = x
    for i, l in enumerate(spec["input_layers"]):
        c, b = f"encoder_cnn.{3 * i}", f"encoder_cnn.{3 * i + 1}"
        h = F.conv2d(h, enc[c + ".weight"], enc[c + ".bias"], stride=int(l["stride"]))
        h = F.batch_norm(h, enc[b + ".running_mean"], enc[b + ".running_var"], enc[b + ".weight"], enc[b + ".bias"],
                         training=train, momentum=BN_MOMENTUM, eps=BN_EPS)
        if train:
            enc[b + ".num_batches_tracked"] += 1
        h = F.relu(h)
    h = F.relu(F.linear(h.flatten(1), enc["encoder_lin.0.weight"], enc["encoder_lin.0.bias"]))
    return (F.linear(h, enc["encoder_mu.weight"], enc["encoder_mu.bias"]),
            F.linear(h, enc["encoder_logvar.weight"], enc["encoder_logvar.bias"]))


class VaeOracle:

    def __init__(self, spec, enc_state, dec_state, lr=1e-3, weight_decay=1e-5, lambda_mse=1.0, lambda_kl=1.0, lambda_ssim=1.0,
                 seed=0):
        self.spec = spec
        (self.lambda_mse, self.lambda_kl, self.lambda_ssim, self.seed) = (lambda_mse, lambda_kl, lambda_ssim, seed)
        self.step_count = 0
        self.enc, self.dec = OrderedDict(), OrderedDict()
        for (dst, src) in ((self.enc, enc_state), (self.dec, dec_state)):
            for k, v in src.items():
                t = torch.as_tensor(np.array(v)) if not torch.is_tensor(v) else v.detach().clone()
                dst[k] = t.requires_grad_(True) if is_param(k) else t
        self.optim = torch.optim.Adam([{"params": [v for k, v in self.enc.items() if is_param(k)]},
                                       {"params": [v for k, v in self.dec.items() if is_param(k)]}], lr=lr,
                                      weight_decay=weight_decay)

    def forward(self, x, train):
        (mu, logvar) = encoder_forward(self.spec, self.enc, x, train)
        z = mu
        if train:
            z = mu + torch.from_numpy(normal_noise(self.seed, self.step_count, tuple(mu.shape))) * torch.exp(0.5 * logvar)
        return decoder_forward(self.spec, self.dec, z, train), mu, logvar

    def losses(self, y, t, mu, logvar):
        mse = F.mse_loss(y, t)
        kl = -0.5 * torch.mean(1 + logvar - mu ** 2 - torch.exp(logvar))
        ssim_loss = 1 - ms_ssim(y, t)
        return mse, kl, ssim_loss

    def total(self, parts):
        return self.lambda_mse * parts[0] + self.lambda_kl * parts[1] + self.lambda_ssim * parts[2]

    def eval_forward(self, x):
        with torch.no_grad():
            return self.forward(x, train=False)[0]

    def eval_losses(self, x, t):
        with torch.no_grad():
            (y, mu, logvar) = self.forward(x, train=False)
            return [float(v) for v in self.losses(y, t, mu, logvar)]

    def loss_and_grads(self, x, t):
        (y, mu, logvar) = self.forward(x, train=True)
        parts = self.losses(y, t, mu, logvar)
        self.optim.zero_grad()
        self.total(parts).backward()
        return [float(v.detach()) for v in parts], y.detach()

    def train_step(self, x, t):
        (parts, _) = self.loss_and_grads(x, t)
        self.optim.step()
        self.step_count += 1
        return parts

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
