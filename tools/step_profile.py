"""Per-launch durations of one eager training step at the benchmark geometry (HIP-event brackets around every launch:
tens of percent high per launch, good for comparing kernels with each other).

    python tools/step_profile.py [batch]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cae_tools_amd.engine import HipEngine                     # noqa: E402
from cae_tools_amd.models.model_sizer import create_model_spec  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    spec = create_model_spec(input_size=(16, 16), input_channels=1, output_size=(256, 256), output_channels=1)
    eng = HipEngine(spec, 128, 32, B, device="cuda:0", graph=False)
    torch.manual_seed(0)
    eng.params.normal_(0, 0.05)
    eng.set_dataset(0, torch.rand((B, 1, 16, 16), device="cuda:0"), torch.rand((B, 1, 256, 256), device="cuda:0"))
    for it in range(10):
        eng.forward_backward(0, None, 0, B, B)
    eng.sync()
    acc = {}
    reps = 20
    for it in range(reps):
        eng.profile_begin()
        eng.forward_backward(0, None, 0, B, B)
        for (name, layer, us, nbytes) in eng.profile_end():
            acc.setdefault((name, layer), []).append(us)
    tot = 0.0
    for (name, layer), v in acc.items():
        m = float(np.median(v))
        tot += m
        print(f"{name:28s} layer {layer}: {m:7.2f} us")
    print(f"sum {tot:.1f} us")


if __name__ == "__main__":
    main()
