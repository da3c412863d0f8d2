"""Which ReLU decisions of the HIP UNET step at the medium test geometry differ from the fp64 oracle (and how far from zero the
oracle pre-activation is there), and the gradients with the oracle following them: tests/test_unet_hip_parity.py, ReluAlign.

    python tools/diag_unet_relu_flips.py        (on the GPU box)
"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from cae_tools_amd.models.unet import Decoder, Encoder, unet_layer_spec
from cae_tools_amd.unet_engine import UnetEngine
from oracle import unet_oracle as uo
from test_unet_hip_parity import _grad_dict, _feeds_batchnorm
from unet_helpers import hip_relu_decisions
spec = unet_layer_spec(3, 3, (64, 64), [32, 64, 96])
(fc, latent, B) = (24, 6, 5)
torch.manual_seed(123)
enc = Encoder(spec.get_input_layers(), latent, fc)
dec = Decoder(spec.get_output_layers(), latent, fc)
g = torch.Generator().manual_seed(9)
x = torch.rand((B, 3, 64, 64), generator=g); t = torch.rand((B, 3, 64, 64), generator=g)
m = (torch.rand((B, 1, 64, 64), generator=g) < 0.85).float()
to64 = lambda sd: {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
eng = UnetEngine(spec, fc, latent, B, device="cuda:0", specialised=True)
eng.load_state(enc.state_dict(), dec.state_dict())
eng.set_hyper(dropout_rate=0.1, seed=4); eng.set_step(2); eng.set_dataset(0, x, t, m)
gd = _grad_dict(eng, eng.forward_backward(0, None, 0, B, slot=0))
decisions = hip_relu_decisions(eng, spec.save(), fc, latent, B)
o64 = uo.UnetOracle(spec.save(), to64(enc.state_dict()), to64(dec.state_dict()), dropout_rate=0.1, seed=4)
o64.step_count = 2
with uo.ReluAlign(decisions) as al:
    o64.loss_and_grads(x.double(), t.double(), m.double())
print("followed", al.followed)
print("disagree", al.disagree)
want64 = o64.grads()
worst = []
for k, w64 in want64.items():
    if _feeds_batchnorm(k): continue
    err = float(np.abs(gd[k].numpy().astype(np.float64) - w64.numpy()).max()); mx = float(np.abs(w64.numpy()).max())
    worst.append((err / max(mx, 1e-6), k))
worst.sort(reverse=True)
print([(f"{r:.2e}", k) for r, k in worst[:5]])
