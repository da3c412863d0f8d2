#!/usr/bin/env python3
"""Turn gpurun_out/profile/* (written by tools/profile_on_gpu.sh) into the tracked summaries under
profiles/: per-kernel stats (rocprofv3 --kernel-trace --stats), per-launch HBM traffic from the
FETCH_SIZE / WRITE_SIZE passes, and profiles/pmc_traffic.json that bench.py reads for roofline.traffic.

FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3 derived counters); on gfx950 FETCH_SIZE reports half the
bytes of a wide coalesced read stream (MI355X_MICROARCH.md §HBM) - both the raw and the doubled figure
are written, the doubled one is used as `traffic`.
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profile")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "round1"


def short(name):
    name = name.replace("cae::", "").replace("void ", "")
    return name.split("(")[0]


def per_step_sequences(rows):
    """split the dispatch list into steps at k_adam; returns list of lists of rows"""
    rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]))
    steps, cur = [], []
    for r in rows:
        cur.append(r)
        if "k_adam" in r["Kernel_Name"]:
            steps.append(cur)
            cur = []
    return steps


def counter_per_launch(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    steps = per_step_sequences(rows)
    n = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == n][2:]  # full training steps, warm
    out = []
    for i in range(n):
        vals = [float(s[i]["Counter_Value"]) for s in steps]
        out.append((short(steps[0][i]["Kernel_Name"]), sum(vals) / len(vals)))
    return out


def main():
    os.makedirs(DST, exist_ok=True)
    stats = list(csv.DictReader(open(os.path.join(SRC, "trace", "t_kernel_stats.csv"))))
    with open(os.path.join(DST, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent", "min_us", "max_us"])
        for r in stats:
            if float(r["Percentage"]) < 0.01:
                continue
            w.writerow([short(r["Name"]), r["Calls"], f'{float(r["TotalDurationNs"]) / 1e3:.1f}',
                        f'{float(r["AverageNs"]) / 1e3:.2f}', r["Percentage"], f'{float(r["MinNs"]) / 1e3:.2f}',
                        f'{float(r["MaxNs"]) / 1e3:.2f}'])
    order = json.load(open(os.path.join(SRC, "launch_order.json")))
    # per-launch kernel durations of the --kernel-trace pass, by bench.py's labels (bench.py calibrates its event brackets on them)
    trace_rows = list(csv.DictReader(open(os.path.join(SRC, "trace", "t_kernel_trace.csv"))))
    tsteps = per_step_sequences([r for r in trace_rows if "cae::" in r["Kernel_Name"]])
    tsteps = [st for st in tsteps if len(st) == len(order)][2:]
    if tsteps:
        avg = {}
        for i, (label, _) in enumerate(order):
            d = [(int(st[i]["End_Timestamp"]) - int(st[i]["Start_Timestamp"])) / 1e3 for st in tsteps]
            avg[label] = sum(d) / len(d)
        with open(os.path.join(DST, "kernel_trace_avg_us.json"), "w") as f:
            json.dump(avg, f, indent=1)
    fetch = counter_per_launch(os.path.join(SRC, "fetch", "f_counter_collection.csv"), "FETCH_SIZE")
    write = counter_per_launch(os.path.join(SRC, "write", "w_counter_collection.csv"), "WRITE_SIZE")
    assert len(fetch) == len(write) == len(order), (len(fetch), len(write), len(order))
    traffic = {}
    with open(os.path.join(DST, f"{tag}_hbm_traffic.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["launch", "kernel", "label", "algorithmic_bytes", "FETCH_SIZE_KB", "WRITE_SIZE_KB",
                    "traffic_bytes(2*fetch+write)"])
        for i, ((kf, fv), (kw, wv), (label, nbytes)) in enumerate(zip(fetch, write, order)):
            t = (2.0 * fv + wv) * 1024.0
            traffic[label] = t
            w.writerow([i, kf, label, int(nbytes), f"{fv:.1f}", f"{wv:.1f}", int(t)])
    with open(os.path.join(DST, "pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    # SQ counters per launch
    sqp = os.path.join(SRC, "sq", "s_counter_collection.csv")
    if os.path.exists(sqp):
        names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                 "SQ_INSTS_VALU", "SQ_INSTS_MFMA"]
        cols = {n: counter_per_launch(sqp, n) for n in names}
        with open(os.path.join(DST, f"{tag}_sq_counters.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["launch", "kernel", "label"] + names)
            for i, (label, _) in enumerate(order):
                w.writerow([i, cols[names[0]][i][0], label] + [f"{cols[n][i][1]:.4g}" for n in names])
    mp = os.path.join(SRC, "mfma", "m_counter_collection.csv")
    if os.path.exists(mp):
        names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32"]
        try:
            cols = {n: counter_per_launch(mp, n) for n in names}
            with open(os.path.join(DST, f"{tag}_mfma_counters.csv"), "w") as f:
                w = csv.writer(f)
                # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD: 4.0 = all four matrix pipes of every busy CU busy all the time
                w.writerow(["launch", "kernel", "label"] + names + ["mfma_busy_over_busy_cu(4=all four SIMDs)"])
                for i, (label, _) in enumerate(order):
                    (b, c) = (cols[names[0]][i][1], cols[names[1]][i][1])
                    w.writerow([i, cols[names[0]][i][0], label] + [f"{cols[n][i][1]:.4g}" for n in names] + [f"{b / c:.3f}" if c else ""])
        except Exception as ex:      # a counter the pass could not collect
            print("mfma counters:", ex)
    bench = open(os.path.join(SRC, "bench.json")).read().strip()
    with open(os.path.join(DST, f"{tag}_bench_under_rocprof.json"), "w") as f:
        f.write(bench + "\n")
    print("wrote", sorted(os.listdir(DST)))


if __name__ == "__main__":
    main()
