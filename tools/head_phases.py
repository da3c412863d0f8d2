"""Where the fused encoder+Linear forward kernel (kernels_head.h k_head_fwd) spends its time: wall-clock stamps (100 MHz)
taken by thread 0 of every workgroup at the phase boundaries, read back from the spare buffer they are written to when
CAE_HEAD_DBG=1.  Eval-mode forward of one batch, repeated; prints the median per-phase durations in microseconds."""
import os
import sys

import numpy as np
import torch

os.environ["CAE_HEAD_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cae_tools_amd.engine import HipEngine                     # noqa: E402
from cae_tools_amd.models.model_sizer import create_model_spec  # noqa: E402

PHASES = ["prefetch", "bias+zero", "stage inputs", "encoder", "fc0 prologue", "fc0", "fc1", "fc2", "fc3"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    spec = create_model_spec(input_size=(16, 16), input_channels=1, output_size=(256, 256), output_channels=1)
    eng = HipEngine(spec, 128, 32, B, device="cuda:0")
    torch.manual_seed(0)
    eng.params.normal_(0, 0.05)
    x = torch.rand((B, 1, 16, 16), device="cuda:0")
    rows = []
    inner = []
    for it in range(20):
        eng.score(x)
        eng.sync()
        raw = eng.debug_read("fcgrad", 3, count=B * 576)
        full = raw.view(np.int64)[: 36 * 16].reshape(36, 16).astype(np.float64)
        inner.append(full[:, [2, 9, 10, 12, 13, 3]])
        st = full[:, :9]
        rows.append(st)
    st = np.median(np.stack(rows[5:]), axis=0)
    d = np.diff(st, axis=1) / 100.0   # 100 MHz -> us
    print("phase            median over workgroups   workgroup 0")
    for i, name in enumerate(PHASES[1:]):
        print(f"{name:16s} {np.median(d[:, i]):8.2f} us            {d[0, i]:8.2f} us")
    di = np.diff(np.median(np.stack(inner[5:]), axis=0), axis=1) / 100.0
    for i, name in enumerate(["  conv 0", "  stats 0 + publish", "  in-place act + conv 1", "  stats 1 + publish", "  tail"]):
        print(f"{name:24s} {np.median(di[:, i]):8.2f} us")
    print(f"{'first..last':16s} {np.median(st[:, -1] - st[:, 0]) / 100.0:8.2f} us")
    print(f"launch spread (first stamp, max-min over workgroups): {(st[:, 0].max() - st[:, 0].min()) / 100.0:.2f} us")
    print(f"whole grid (max last - min first): {(st[:, -1].max() - st[:, 0].min()) / 100.0:.2f} us")


def tail():
    """the same for k_tail_bwd: one training step, stamps of row group 0's three task workgroups"""
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    spec = create_model_spec(input_size=(16, 16), input_channels=1, output_size=(256, 256), output_channels=1)
    eng = HipEngine(spec, 128, 32, B, device="cuda:0", graph=False)
    torch.manual_seed(0)
    eng.params.normal_(0, 0.05)
    eng.set_dataset(0, torch.rand((B, 1, 16, 16), device="cuda:0"), torch.rand((B, 1, 256, 256), device="cuda:0"))
    groups = (B + 15) // 16
    rows = []
    for it in range(20):
        eng.train_step(0, None, 0, B)
        eng.sync()
        raw = eng.debug_read("fcgrad", 3, count=B * 576)
        rows.append(raw.view(np.int64)[: 4 * groups * 16].reshape(4, groups, 16).astype(np.float64))
    st = np.median(np.stack(rows[5:]), axis=0)
    names = ["zero + consts", "panels to LDS", "g1", "g0", "gx", "BatchNorm sums"]
    print("k_tail_bwd, us per phase; columns = tasks 0 (gx + sums), 1 (dW1), 2 (dW0), 3 (dW2) of row group 0")
    for i, name in enumerate(names):
        vals = []
        for task in range(4):
            (t0, t1) = (st[task, 0, i], st[task, 0, i + 1])
            vals.append("    -   " if (t1 <= t0 or t0 == 0) else f"{(t1 - t0) / 100.0:8.2f}")
        print(f"{name:18s}" + " ".join(vals))
    last = [6, 3, 4, 2]
    print(f"{'weight gradient':18s}" + " ".join(f"{(st[t, 0, 7] - st[t, 0, last[t]]) / 100.0:8.2f}" for t in range(4)))
    print(f"{'workgroup total':18s}" + " ".join(f"{(st[t, 0, 7] - st[t, 0, 0]) / 100.0:8.2f}" for t in range(4)))
    print(f"whole grid: {(st[:, :, 7].max() - st[:, :, 0].min()) / 100.0:.2f} us")


def ig():
    """k_ig_fwd_s2 of the last gather-MFMA decoder layer (the stamps of later launches overwrite earlier ones): parity-0
    workgroups' stamps"""
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    spec = create_model_spec(input_size=(16, 16), input_channels=1, output_size=(256, 256), output_channels=1)
    eng = HipEngine(spec, 128, 32, B, device="cuda:0", graph=False)
    torch.manual_seed(0)
    eng.params.normal_(0, 0.05)
    eng.set_dataset(0, torch.rand((B, 1, 16, 16), device="cuda:0"), torch.rand((B, 1, 256, 256), device="cuda:0"))
    rows = []
    for it in range(20):
        eng.forward_backward(0, None, 0, B, B)
        eng.sync()
        raw = eng.debug_read("scan", 0, count=3072, dtype=np.float64)
        rows.append(raw.view(np.int64)[: 256 * 8].reshape(256, 8).astype(np.float64))
    st = np.stack(rows[5:])[-1][:, :5]          # one launch (medians across launches would mix different dispatch orders)
    ok = (st[:, 0] > 0) & (np.diff(st, axis=1) >= 0).all(axis=1) & (st[:, 4] - st[:, 0] < 1e5)
    print(f"{int(ok.sum())} of {len(ok)} parity-0 workgroups with a complete stamp set")
    st = st[ok]
    d = np.diff(st, axis=1) / 100.0
    for i, name in enumerate(["BatchNorm constants", "operand loads + MFMA", "combine + stores", "statistics + atomics"]):
        print(f"{name:24s} median {np.median(d[:, i]):6.2f} us   max {d[:, i].max():6.2f}")
    print(f"workgroup lifetime: median {np.median(st[:, 4] - st[:, 0]) / 100:.2f} us; first start .. last end {(st[:, 4].max() - st[:, 0].min()) / 100:.2f} us; "
          f"start spread {(st[:, 0].max() - st[:, 0].min()) / 100:.2f} us")


def igb():
    """k_ig_bwd_pair of decoder layer 2 (CAE_HEAD_DBG=2): stamps of the first 512 weight-gradient and 256 input-gradient
    workgroups"""
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
    t0 = None
    for (name, lo, n) in (("weight gradient", 0, 512), ("input gradient", 512, 256)):
        st = raw[lo * 4:(lo + n) * 4].reshape(n, 4)
        ok = (st[:, 0] > 0) & (np.diff(st, axis=1) >= 0).all(axis=1) & (st[:, 3] - st[:, 0] < 1e5)
        st = st[ok]
        t0 = st[:, 0].min() if t0 is None else min(t0, st[:, 0].min())
        d = np.diff(st, axis=1) / 100.0
        print(f"{name}: {len(st)} workgroups; constants {np.median(d[:, 0]):.2f}  operand fetch + MFMA {np.median(d[:, 1]):.2f}  "
              f"epilogue {np.median(d[:, 2]):.2f} us; lifetime {np.median(st[:, 3] - st[:, 0]) / 100:.2f} us; "
              f"starts {(st[:, 0].min() - t0) / 100:.2f}..{(st[:, 0].max() - t0) / 100:.2f}, last end {(st[:, 3].max() - t0) / 100:.2f} us")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "igb":
        os.environ["CAE_HEAD_DBG"] = "2"
        igb()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "ig":
        ig()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "tail":
        tail()
        sys.exit(0)
    main()
