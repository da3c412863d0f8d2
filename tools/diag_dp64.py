#!/usr/bin/env python3
"""tests/test_timed_path_gpu.py::test_data_parallel_graph_at_64_rows_per_rank, with per-element errors of one tensor for the
DP engine's gradient arena, the plain forward_backward arena and the fused step's recovered gradient.
python tools/diag_dp64.py [tensor substring]"""
import os
import socket
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    import test_timed_path_gpu as T
    from cae_tools_amd.dp import DataParallel
    focus = sys.argv[1] if len(sys.argv) > 1 else "encoder_cnn.3.weight"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    torch.set_num_threads(8)
    spec, enc, dec = T._model(9)
    x, t = T._data(64, 10)
    (a, b, c) = (T._engine(spec, enc, dec, x, t), T._engine(spec, enc, dec, x, t), T._engine(spec, enc, dec, x, t))
    dp = DataParallel(b, dist, sync_bn=False, overlap=True)
    dp.broadcast_parameters(0)
    w0 = a.params.cpu().numpy().astype(np.float64)
    a.train_step(0, None, 0, 64)
    a.sync()
    g_fused = a.exp_avg.cpu().numpy().astype(np.float64) / 0.1 - T.WD * w0
    dp.train_step(0, None, 0, 64)
    b.sync()
    c.forward_backward(0, None, 0, 64, 64)
    c.sync()
    o32, o64 = T._oracles(spec, enc, dec)
    tr64 = {}
    o32.loss_and_grads(x, t)
    o64.loss_and_grads(x.double(), t.double(), trace=tr64)
    (g32, g64) = (o32.grads(), o64.grads())
    np.set_printoptions(linewidth=220, precision=2)
    for k in g32:
        (arena, off, numel, shape) = a.tensors[k]
        e64 = g64[k].numpy().reshape(-1)
        sc = np.abs(e64).max()
        row = [np.abs(v - e64).max() / sc for v in (g_fused[off:off + numel], b.grads[off:off + numel].cpu().numpy().astype(np.float64),
                                                    c.grads[off:off + numel].cpu().numpy().astype(np.float64),
                                                    g32[k].numpy().astype(np.float64).reshape(-1))]
        print(f"{k:34s} fused {row[0]:.1e}  dp {row[1]:.1e}  fwdbwd {row[2]:.1e}  o32 {row[3]:.1e}")
        if focus in k:
            for (nm, v) in (("fused", g_fused[off:off + numel]), ("dp", b.grads[off:off + numel].cpu().numpy().astype(np.float64)),
                            ("o32", g32[k].numpy().astype(np.float64).reshape(-1))):
                print("   ", nm, "error / max|g|:")
                print(((v - e64) / sc).reshape(shape[0], -1))
    # ReLU masks: where does the engine's masked gradient (zero where the BatchNorm output is <= 0) disagree with the fp64
    # oracle's mask, and how close to zero is the fp64 BatchNorm output there?
    n_enc = len(c.enc_layers)
    for l in range(len(c.dec_layers) - 1):
        y = tr64[f"dec_conv{l}"]
        pre = f"decoder_conv.{3 * l + 1}"
        (gam, bet) = (o64.dec[pre + ".weight"].detach(), o64.dec[pre + ".bias"].detach())
        z = (y - y.mean((0, 2, 3), keepdim=True)) / torch.sqrt(y.var((0, 2, 3), unbiased=False, keepdim=True) + 1e-5) * gam[None, :, None, None] + bet[None, :, None, None]
        gh = c.debug_read("grad", n_enc + l, count=y.numel()).reshape(y.shape)
        mism = torch.from_numpy(gh != 0) != (z > 0)
        idx = mism.nonzero()
        print(f"dec conv {l}: {y.numel()} outputs, mask mismatches {int(mism.sum())}; |z64| there: {[f'{float(z[tuple(i)]):.2e}' for i in idx[:6]]}; "
              f"smallest |z64| overall {float(z.abs().min()):.2e}")
    y = tr64["enc_conv1"]
    bnout = (y - y.mean((0, 2, 3), keepdim=True)) / y.std((0, 2, 3), unbiased=False, keepdim=True)
    print("enc conv1: smallest |x_hat| per channel:", bnout.abs().amin((0, 2, 3)).numpy())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
