"""Where k_s2_last_fused (kernels_last.h) or k_s2_bwd_rows (kernels_rows.h) spends its time: wall-clock stamps (100 MHz) of
thread 0 of the first 384 workgroups at the phase boundaries (CAE_HEAD_DBG=4 / 5), read back from the loader's scratch buffer.

    python tools/last_phases.py [batch]                 # the fused last layer
    python tools/last_phases.py [batch] rows [layer]    # the row-streaming backward of decoder layer `layer` (default 4)
    python tools/last_phases.py [batch] ctb [layer]     # the LDS-staged backward (kernels_ctbwd.h) of decoder layer 0..2 (CAE_HEAD_DBG=6)
"""
import os
import sys

import numpy as np
import torch

ROWS = len(sys.argv) > 2 and sys.argv[2] == "rows"
CTB = len(sys.argv) > 2 and sys.argv[2] == "ctb"
os.environ["CAE_HEAD_DBG"] = "6" if CTB else "5" if ROWS else "4"
if (ROWS or CTB) and len(sys.argv) > 3:
    os.environ["CAE_DBG_LAYER"] = sys.argv[3]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cae_tools_amd.engine import HipEngine                     # noqa: E402
from cae_tools_amd.models.model_sizer import create_model_spec  # noqa: E402

PHASES = (["issue loads", "constants + small staging", "gradient maps to LDS", "MFMA tasks", "producer sums"] if CTB else
          ["consts + issue loads", "barrier (weights in LDS)", "first row", "remaining rows", "reductions + atomics"] if ROWS else
          ["issue loads", "BatchNorm consts + weights", "first row", "remaining rows", "reductions + atomics"])


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    spec = create_model_spec(input_size=(16, 16), input_channels=1, output_size=(256, 256), output_channels=1)
    eng = HipEngine(spec, 128, 32, B, device="cuda:0", graph=False)
    torch.manual_seed(0)
    eng.params.normal_(0, 0.05)
    eng.set_dataset(0, torch.rand((B, 1, 16, 16), device="cuda:0"), torch.rand((B, 1, 256, 256), device="cuda:0"))
    for it in range(12):
        eng.forward_backward(0, None, 0, B, B)
        eng.sync()
    raw = eng.debug_read("scan", 0, count=3072, dtype=np.float64).view(np.int64).astype(np.float64)
    st = raw[:384 * 8].reshape(384, 8)[:, :6]
    ok = (st[:, 0] > 0) & (np.diff(st, axis=1) >= 0).all(axis=1) & (st[:, 5] - st[:, 0] < 1e5)
    st = st[ok]
    d = np.diff(st, axis=1) / 100.0
    print(f"{len(st)} workgroups with a complete stamp set")
    for i, name in enumerate(PHASES):
        print(f"{name:28s} median {np.median(d[:, i]):6.2f} us   max {d[:, i].max():6.2f}")
    print(f"workgroup lifetime: median {np.median(st[:, 5] - st[:, 0]) / 100:.2f} us; first start .. last end "
          f"{(st[:, 5].max() - st[:, 0].min()) / 100:.2f} us; start spread {(st[:, 0].max() - st[:, 0].min()) / 100:.2f} us")


if __name__ == "__main__":
    main()
