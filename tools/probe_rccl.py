"""Probe (GPU box): can two ranks share ONE device under RCCL, and does an RCCL all-reduce survive hipGraph capture?

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/probe_rccl.py

Both ranks use cuda:0 (the box has one GPU).  Prints one line per finding; never raises (a refusal is a finding).
"""
import os
import sys
import time

import torch
import torch.distributed as dist


def main():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        x = torch.full((1024,), float(rank + 1), device=dev)
        dist.all_reduce(x)
        torch.cuda.synchronize()
        print(f"[rank {rank}] eager all_reduce ok: {float(x[0])} (expect {world * (world + 1) / 2})", flush=True)
    except Exception as ex:  # noqa: BLE001
        print(f"[rank {rank}] eager all_reduce FAILED: {type(ex).__name__}: {str(ex)[:300]}", flush=True)
        return
    try:
        y = torch.full((112272,), 1.0, device=dev)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            dist.all_reduce(y)   # warm-up outside capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        y.fill_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            dist.all_reduce(y)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print(f"[rank {rank}] captured all_reduce x3 ok: {float(y[0])} (expect {world ** 3})", flush=True)
        t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        print(f"[rank {rank}] graph replay of a 449 KB all_reduce: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us each", flush=True)
        y.fill_(0.0)
        t0 = time.perf_counter()
        for _ in range(200):
            dist.all_reduce(y)
        torch.cuda.synchronize()
        print(f"[rank {rank}] eager 449 KB all_reduce: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us each", flush=True)
    except Exception as ex:  # noqa: BLE001
        print(f"[rank {rank}] captured all_reduce FAILED: {type(ex).__name__}: {str(ex)[:300]}", flush=True)
    try:
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


if __name__ == "__main__":
    main()
    sys.exit(0)
