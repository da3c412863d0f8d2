#!/usr/bin/env python3
"""bench.py — train images/sec of the ConvAE hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "cfg2"): ConvAEModel 'conv', 16x16 -> 256x256, 1 channel,
fc_size 128 / latent 32 (API defaults, conv_ae_model.py:36), batch 64 per GPU, synthetic data = the restated gen.py "circle"
pattern (cae_tools_amd/data/datagen.py, 4096 training / 512 test cases, normalised and packed on the GPU by
DSDataset exactly as ConvAEModel.train does), random-init weights (torch.manual_seed(0)).  One step = one iteration of __train_epoch
(conv_ae_model.py:189-200): train-mode forward, MSE, backward, Adam.  Inputs are resident in
HBM before the timed region.  N > 1: one process per GPU, global batch 64*N (weak scaling: 64 samples
per GPU per step), rank r takes rows [64r, 64r+64) of every frozen global batch; the gradients are
all-reduced inside libcae_hip (RCCL, two buckets, second stream, captured in the step graph).

`value` times the hot path itself (the engine's epoch loop, K steps).  `train_api` (N = 1) is SURVEY §8(d)'s metric to the
letter: ConvAEModel.train() on the same data, 20 epochs, a test pass every 10, images/s of its epoch loop.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cae_tools_amd.engine import HipEngine  # noqa: E402
from cae_tools_amd.models.model_sizer import create_model_spec  # noqa: E402
from cae_tools_amd.models.encoder import Encoder  # noqa: E402
from cae_tools_amd.models.decoder import Decoder  # noqa: E402

IN_SIZE, OUT_SIZE = (16, 16), (256, 256)
FC, LATENT, BATCH = 128, 32, 64
N_TRAIN = 4096
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
# algorithmic bytes per image of a training step at this geometry, B = 64
ALGO_BYTES_PER_IMAGE = 2862766


def build_model(seed=0):
    spec = create_model_spec(input_size=IN_SIZE, input_channels=1, output_size=OUT_SIZE, output_channels=1)
    torch.manual_seed(seed)
    enc = Encoder(spec.get_input_layers(), encoded_space_dim=LATENT, fc_size=FC)
    dec = Decoder(spec.get_output_layers(), encoded_space_dim=LATENT, fc_size=FC)
    return spec, enc, dec


N_TEST = 512


def circle_data(n, seed):
    """the reference's test-data generator (test/datagen/gen.py 'circle'), restated with explicit seeds: an xarray-like data
    set with variables lowres (n,1,16,16) and hires (n,1,256,256), float32, values ~288..298"""
    from cae_tools_amd.data import datagen
    return datagen.generate("circle", n, seed=seed)


def packed(ds_train, order):
    """normalised, channel-packed device tensors (x, t) of the training set laid out in the frozen sample order - DSDataset's
    GPU path (cae_scan_f32, cae_normalise_pack_rows), exactly what ConvAEModel.train hands the engine"""
    from cae_tools_amd.models.ds_dataset import DSDataset
    return DSDataset(ds_train, ["lowres"], "hires").device_batches(order)


def train_api_leg(ds_train, ds_test, epochs=20, test_interval=10):
    """train images/s = N_train * epochs / seconds of ConvAEModel.train()'s epoch loop (conv_ae_model.py:328-334),
    the test pass every `test_interval` epochs included; data-set scan / normalise / upload and the final evaluate excluded"""
    import contextlib
    import io
    from cae_tools_amd.models.conv_ae_model import ConvAEModel
    torch.manual_seed(0)
    m = ConvAEModel(batch_size=BATCH, nr_epochs=epochs, test_interval=test_interval, fc_size=FC, encoded_dim_size=LATENT)
    with contextlib.redirect_stdout(io.StringIO()):
        m.train(["lowres"], "hires", ds_train, ds_test)
    t = m.timing
    return {"value": t["train_images"] / t["epoch_loop_seconds"], "unit": "images/s", "api": "ConvAEModel.train",
            "n_train": N_TRAIN, "n_test": N_TEST, "batch": BATCH, "epochs": epochs, "test_interval": test_interval,
            "epoch_loop_seconds": t["epoch_loop_seconds"],
            "train_loss_first_last": [m.history["train_loss"][0], m.history["train_loss"][-1]]}


def cpu_baseline(spec, enc, dec, steps=12, warm=2):
    """the CPU oracle (torch-CPU restatement pinned to the reference) timed on this box's host cores"""
    from oracle import cae_oracle as orc
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # a one-GPU box is given a 16-core share of the host; more intra-op threads than that only
    # oversubscribes (256 threads: 22 s per step).  Time 16 and 8 threads, report the faster.
    g = torch.Generator().manual_seed(7)
    x = torch.rand((BATCH, 1) + IN_SIZE, generator=g)
    t = torch.rand((BATCH, 1) + OUT_SIZE, generator=g)
    best = None
    for cores in sorted({min(avail, 16), min(avail, 8)}, reverse=True):
        torch.set_num_threads(cores)
        m = orc.OracleModel(spec.save(), enc.state_dict(), dec.state_dict(), lr=1e-3, weight_decay=1e-5)
        for _ in range(warm):
            m.train_step(x, t)
        times = []
        budget = time.perf_counter() + 12.0
        for _ in range(steps):
            t0 = time.perf_counter()
            m.train_step(x, t)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() > budget:
                break
        med = float(np.median(times))
        if best is None or med < best[0]:
            best = (med, cores, len(times))
    (med, cores, n) = best
    return {"value": BATCH / med, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} training steps at batch {BATCH} (median), {warm} warm-up, torch {torch.__version__} CPU, "
                      f"{cores} intra-op threads (host reports {os.cpu_count()} CPUs)"}


def roofline_from_profile(recs, steps, trace_us=None):
    """dominant kernel = the kernel SYMBOL with the largest total time over the profiled steps.

    Every launch is bracketed by a HIP event pair on its own stream.  A bracket reads longer than the kernel runs: an EMPTY
    bracket, recorded once per step, reads ~4.7 us, and with a kernel inside part of that overlaps the kernel.  The overlap is
    calibrated, not guessed: `trace_us` = rocprofv3 --kernel-trace averages of the same launches (profiles/
    kernel_trace_avg_us.json, written by tools/summarise_profile.py from the committed profile of this code state) gives
    f = median over the step's kernels of (bracket - trace) / empty, and a launch's duration is taken as bracket - f * empty
    (f = 0.5 when there is no profile yet: round 1's profiles gave 0.41).  `achieved` uses that duration; the raw bracket
    and the trace average of the dominant kernel are reported next to it."""
    empty = sorted(us for (name, _, us, _) in recs if name == "event_pair")
    overhead = empty[len(empty) // 2] if empty else 0.0
    agg = {}       # per (label, layer): the --table listing
    by_kernel = {}  # per kernel SYMBOL, as rocprofv3 --stats groups them: the s2_* kernels are template instances per layer
    for (name, layer, us, nbytes) in recs:  # shape (one symbol per layer), every other label is one symbol for all its layers
        if name == "event_pair":
            continue
        a = agg.setdefault((name, layer), [0.0, 0, nbytes])
        a[0] += us
        a[1] += 1
        k = by_kernel.setdefault((name, layer) if name.startswith("s2_") else (name, None), [0.0, 0, 0.0])
        k[0] += us
        k[1] += 1
        k[2] += nbytes
    overlap = 0.5
    if trace_us and overhead > 0:
        fs = sorted((v[0] / v[1] - trace_us[f"{k[0]}[layer {k[1]}]"]) / overhead for k, v in agg.items()
                    if f"{k[0]}[layer {k[1]}]" in trace_us)
        if fs:
            overlap = min(1.0, max(0.0, fs[len(fs) // 2]))
    total = sum(a[0] for a in agg.values())
    (key, (us_sum, count, bytes_sum)) = max(by_kernel.items(), key=lambda kv: kv[1][0])
    (raw_us, nbytes) = (us_sum / count, bytes_sum / count)
    avg_us = raw_us - overlap * overhead
    achieved = nbytes / (avg_us * 1e-6) / 1e9
    table = sorted(((k[0], k[1], v[0] / v[1], v[2], v[0] / total) for k, v in agg.items()), key=lambda r: -r[4])
    label = key[0] if key[1] is None else f"{key[0]}[layer {key[1]}]"
    prof = None
    if trace_us:
        vals = [v for k, v in trace_us.items() if k == label or k.startswith(label + "[")]
        prof = sum(vals) / len(vals) if vals else None
    corrected_step = (total - overlap * overhead * sum(v[1] for v in agg.values())) / steps
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            # the same figure from the raw event bracket (no calibration file involved): a lower bound on the fraction
            "achieved_bracketed": nbytes / (raw_us * 1e-6) / 1e9, "frac_bracketed": nbytes / (raw_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "kernel": label, "launches_per_step": count / steps, "avg_us": avg_us, "avg_us_bracketed": raw_us,
            "empty_event_pair_us": overhead, "bracket_overlap": overlap, "rocprofv3_avg_us": prof,
            "algorithmic_bytes_per_launch": nbytes, "share_of_step": us_sum / total}, table, corrected_step


def main():
    # stdout carries exactly ONE line (the JSON): libraries that print banners to fd 1 (RCCL does under
    # NCCL_DEBUG=VERSION) are diverted to stderr for the whole run
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-api", action="store_true", help="skip the ConvAEModel.train() leg (profiling passes)")
    ap.add_argument("--table", action="store_true", help="also print the per-kernel table to stderr")
    ap.add_argument("--force-dp", action="store_true", help="use the data-parallel code path even with one rank (rehearsal)")
    ap.add_argument("--sync-bn", action="store_true", help="data-parallel runs: BatchNorm over the global batch")
    ap.add_argument("--dp-host-allreduce", action="store_true",
                    help="data-parallel runs: torch.distributed all-reduce between two library calls instead of the in-library RCCL path")
    ap.add_argument("--launch-order", default=None, help="write the per-step launch sequence [label, algorithmic bytes] here")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    spec, enc, dec = build_model(0)
    eng = HipEngine(spec, FC, LATENT, max_batch=BATCH, device=device, graph=os.environ.get("CAE_GRAPH", "1") != "0")
    eng.load_state(enc.state_dict(), dec.state_dict())
    eng.set_hyper(lr=1e-3, weight_decay=1e-5)
    # every rank holds the same N_TRAIN samples and the same frozen shuffle (what ConvAEModel.train does under data
    # parallelism): a global batch is 64 * world consecutive rows of the permutation, rank r takes rows [64 r, 64 r + 64)
    ds_train = circle_data(N_TRAIN, 1234)
    # the frozen shuffle, materialised once as ConvAEModel.train does (the reference stacks its shuffled batches once and
    # reuses the list every epoch, conv_ae_model.py:315-325) - by the normalisation kernel, which writes every sample to its
    # row of the frozen order: batches are contiguous rows, no permutation look-up in the step, no second copy of the data
    x, t = packed(ds_train, np.random.default_rng(99).permutation(N_TRAIN))
    eng.set_dataset(0, x, t)
    perm = None
    global_batch = BATCH * world
    steps_per_epoch = N_TRAIN // global_batch
    dp_graph = None
    dp_times = {}

    if dist is None:
        def run(nsteps):
            done = 0
            while done < nsteps:
                n = min(steps_per_epoch, nsteps - done)
                eng.enqueue_train_steps(0, perm, n * BATCH, BATCH, 0)
                done += n
    else:
        from cae_tools_amd.dp import DataParallel, shard_bounds
        (lo, hi) = shard_bounds(global_batch, world, rank)
        native = not args.dp_host_allreduce
        try:
            if native:
                dp = DataParallel(eng, dist, sync_bn=args.sync_bn)      # joins the library's RCCL communicator (self-tested)
                dp.broadcast_parameters(0)
                dp_graph = eng.dp_graph_capture()
                # the two launch structures of the exchange, timed on the live communicator; the faster is kept (untimed region)
                dp_times = eng.dp_calibrate(dist, 0, perm, lo, hi - lo, global_batch, args.sync_bn)
        except Exception as ex:      # the in-library communicator could not be set up on this rank
            print(f"[bench rank {rank}] in-library data parallelism unavailable ({ex}); torch.distributed all-reduce instead",
                  file=sys.stderr)
            native = False
        ok = torch.tensor([1 if native else 0], device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)        # every rank takes the same path
        native = bool(ok.item())

        if native:
            def run(nsteps):
                done = 0
                while done < nsteps:
                    n = min(steps_per_epoch, nsteps - done)      # one epoch's steps: ONE call = one graph replay
                    eng.set_cursor(lo, 0)
                    eng.dp_train_steps(0, perm, hi - lo, global_batch, args.sync_bn, n)
                    done += n
        else:
            # fallback (round 1's protocol, also --dp-host-allreduce): per step cae_forward_backward, the gradient arena
            # all-reduced by torch.distributed (RCCL) on the engine's stream, cae_adam_step; per-rank BatchNorm statistics
            class _HalfSteps:        # hides the in-library entry points from DataParallel
                def __init__(self, e):
                    (self.e, self.grads, self.params, self.buffers, self.exp_avg, self.exp_avg_sq) = \
                        (e, e.grads, e.params, e.buffers, e.exp_avg, e.exp_avg_sq)
                    (self.stream, self.device, self.forward_backward, self.adam_step) = (e.stream, e.device, e.forward_backward, e.adam_step)
            dp = DataParallel(_HalfSteps(eng), dist, sync_bn=False)
            dp.broadcast_parameters(0)
            (dp_graph, dp_times) = (False, {})

            def run(nsteps):
                for k in range(nsteps):
                    dp.train_step(0, perm, (k % steps_per_epoch) * global_batch + lo, hi - lo, global_batch)

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])

    # every hipGraph shape the warm-up and the timed steps will replay is captured here, launching nothing: the timed
    # region then holds exactly K steps of work and no graph instantiation (a K-step graph is a different shape than a W-step one)
    if hasattr(eng, "capture_graphs") and (dist is None or native):
        with eng.capture_graphs():
            run(args.warmup)
            run(args.steps)
    run(args.warmup)
    eng.sync()
    torch.cuda.synchronize(device)
    barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run(args.steps)
    eng.sync()
    torch.cuda.synchronize(device)
    barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    eng._read_losses(0, min(steps_per_epoch, eng.loss_slots))  # drain loss slots (outside the timed region)

    result = None
    dp_info = (None, None, None)
    if dist is not None and native:
        import ctypes
        (w_, r_, g_) = (ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0))
        eng.lib.cae_dp_info(eng.handle, ctypes.byref(w_), ctypes.byref(r_), ctypes.byref(g_))
        dp_info = (int(w_.value), int(r_.value), bool(g_.value))
    if rank == 0:
        value = BATCH * world * args.steps / elapsed
        result = {
            "metric": "train images/sec (16x16->256x256, batch 64)", "value": value, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic (gen.py circle pattern, seeds 1234 / 4321)",
            "config": {"workload": "cfg2: ConvAEModel 'conv' 16x16->256x256 1-ch, fc128/latent32, batch 64 per GPU, "
                                   "train step = fwd+MSE+bwd+Adam, BatchNorm batch stats per GPU",
                       "global_batch": BATCH * world, "n_train": N_TRAIN, "params": 112271,
                       "parallelism": (f"dp{world}" + ("+syncbn" if args.sync_bn else "")) if dist is not None else "single",
                       "dp_collectives": None if dist is None else
                       ("torch.distributed all-reduce of the gradient arena between cae_forward_backward and cae_adam_step, plain launches"
                        if not native else
                        ("RCCL all-reduce of 2 gradient buckets, first one on a second stream beside the tail of backward"
                         if eng.dp_overlap and not args.sync_bn else
                         "one RCCL all-reduce of the whole gradient arena on the main stream after backward" if not args.sync_bn else
                         "RCCL all-reduce of 2 gradient buckets and of every BatchNorm sum table on the main stream") +
                        (", captured in the step graph" if dp_graph else ", plain launches")),
                       "dp_calibration_us_per_step": None if dist is None else
                       {k: round(v * 1e6, 1) for k, v in dp_times.items()},
                       # what the library's own communicator reports (cae_dp_info): did RCCL see N ranks, are the
                       # collectives inside the captured step graph, which exchange structure the calibration kept
                       "dp_world": dp_info[0], "dp_rank0": dp_info[1], "dp_graph_capture": dp_info[2],
                       "dp_structure": None if dist is None or not native else
                       ("syncbn" if args.sync_bn else ("overlap: bucket 0 on the second stream" if eng.dp_overlap else
                                                       "serial: one all-reduce after backward")),
                       "steps_per_graph_replay": min(steps_per_epoch, args.steps)},
            "step_roofline": {"algorithmic_bytes_per_image": ALGO_BYTES_PER_IMAGE,
                              "achieved_GBs": value / world * ALGO_BYTES_PER_IMAGE / 1e9,
                              "frac_of_8TBs": value / world * ALGO_BYTES_PER_IMAGE / 1e9 / HBM_PEAK_GBS},
        }
    # per-kernel timing with HIP events on the engine's stream (rank 0, plain launches)
    if rank == 0:
        prof_steps = 20
        eng.profile_begin()
        eng.enqueue_train_steps(0, perm, prof_steps * BATCH, BATCH, 0)
        recs = eng.profile_end()
        eng._read_losses(0, prof_steps)
        trace_path = os.path.join(ROOT, "profiles", "kernel_trace_avg_us.json")
        trace_us = None
        if os.path.exists(trace_path):
            with open(trace_path) as f:
                trace_us = json.load(f)
        roof, table, step_us = roofline_from_profile(recs, prof_steps, trace_us)
        if args.launch_order:
            kernels = [r for r in recs if r[0] != "event_pair"]
            per = len(kernels) // prof_steps
            with open(args.launch_order, "w") as f:
                json.dump([[f"{n}[layer {l}]", b] for (n, l, _, b) in kernels[:per]], f)
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)
            if roof["kernel"] in pmc:
                roof["traffic"] = pmc[roof["kernel"]]
            else:   # one symbol launched for several layers: the mean over its launches, like `achieved`
                vals = [v for k, v in pmc.items() if k.startswith(roof["kernel"] + "[")]
                roof["traffic"] = sum(vals) / len(vals) if vals else None
        result["roofline"] = roof
        result["eager_step_us_sum_of_kernels"] = step_us
        if args.table:
            for (name, layer, avg, nbytes, share) in table:
                print(f"{name:28s} L{layer:<2d} {avg:9.2f} us  {nbytes / 1e6:9.2f} MB  {nbytes / avg / 1e3:8.1f} GB/s"
                      f"  {100 * share:5.1f}%", file=sys.stderr)
        if world == 1 and not args.no_train_api:
            del eng         # ConvAEModel builds its own engine on the same data
            result["train_api"] = train_api_leg(ds_train, circle_data(N_TEST, 4321))
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is timed at N = 1 only
            result["cpu_baseline"] = cpu_baseline(spec, enc, dec)
    barrier()
    if rank == 0:
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
