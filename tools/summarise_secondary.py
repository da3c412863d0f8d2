#!/usr/bin/env python3
"""gpurun_out/secondary/* (tools/profile_secondary.sh) -> profiles/<tag>_{unet,vae}_kernel_stats.csv, _hbm_traffic.csv and
_bench.json: the bench line with roofline.traffic = HBM bytes per training step from the FETCH_SIZE / WRITE_SIZE passes
(2 * FETCH_SIZE + WRITE_SIZE, KB -> bytes: the gfx950 correction of MI355X_MICROARCH.md, as tools/summarise_profile.py) and,
for the var path, an HBM roofline object on SURVEY.md §8(d)'s algorithmic bytes of the cfg5 trunk.

    python tools/summarise_secondary.py [tag]
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "secondary")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "round2"
VAE_TRUNK_BYTES_PER_IMAGE = 11927104      # SURVEY.md §8(d), cfg5 trunk at batch 16 (MS-SSIM passes not included)


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("unet::", "").replace("cae::", "").replace("void ", "").split("(")[0]


def per_kernel_counter(path, counter, steps_total):
    tot = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        tot[short(r["Kernel_Name"])] += float(r["Counter_Value"])
        n[short(r["Kernel_Name"])] += 1
    return {k: (v / steps_total, n[k] / steps_total) for k, v in tot.items()}


def main():
    for m in ("unet", "vae"):
        bench = json.loads(open(os.path.join(SRC, f"{m}_bench.json")).read().strip().splitlines()[-1])
        stats = list(csv.DictReader(open(os.path.join(SRC, f"{m}_trace", "t_kernel_stats.csv"))))
        steps = 23      # 20 timed + 3 warm-up steps in the trace pass
        with open(os.path.join(DST, f"{tag}_{m}_kernel_stats.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls_per_step", "us_per_step", "avg_us", "percent", "min_us", "max_us"])
            for r in stats:
                if float(r["Percentage"]) < 0.05:
                    continue
                w.writerow([short(r["Name"]), f'{int(r["Calls"]) / steps:.2f}', f'{float(r["TotalDurationNs"]) / 1e3 / steps:.1f}',
                            f'{float(r["AverageNs"]) / 1e3:.2f}', r["Percentage"], f'{float(r["MinNs"]) / 1e3:.2f}', f'{float(r["MaxNs"]) / 1e3:.2f}'])
        psteps = 12     # 10 + 2 in the counter passes
        fetch = per_kernel_counter(os.path.join(SRC, f"{m}_fetch", "f_counter_collection.csv"), "FETCH_SIZE", psteps)
        write = per_kernel_counter(os.path.join(SRC, f"{m}_write", "w_counter_collection.csv"), "WRITE_SIZE", psteps)
        total = 0.0
        with open(os.path.join(DST, f"{tag}_{m}_hbm_traffic.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "launches_per_step", "FETCH_SIZE_KB_per_step", "WRITE_SIZE_KB_per_step", "traffic_MB_per_step(2*fetch+write)"])
            rows = []
            for k in sorted(set(fetch) | set(write)):
                (fv, n) = fetch.get(k, (0.0, 0.0))
                (wv, _) = write.get(k, (0.0, 0.0))
                t = (2.0 * fv + wv) * 1024.0
                total += t
                rows.append((t, k, n, fv, wv))
            for (t, k, n, fv, wv) in sorted(rows, reverse=True):
                if t > 0:
                    w.writerow([k, f"{n:.2f}", f"{fv:.1f}", f"{wv:.1f}", f"{t / 1e6:.2f}"])
        dt = bench["ms_per_step"] * 1e-3
        if m == "unet":
            bench["roofline"]["traffic"] = total
            bench["hbm_traffic_GBs"] = total / dt / 1e9
        else:
            B = 16
            algo = VAE_TRUNK_BYTES_PER_IMAGE * B
            bench["roofline"] = {"bound": "hbm", "achieved": algo / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                                 "frac": algo / dt / 1e9 / 8000.0, "traffic": total,
                                 "algorithmic_bytes_per_step": algo, "note": "trunk bytes only (SURVEY 8d); MS-SSIM passes add to traffic, not to the algorithmic figure",
                                 "hbm_traffic_GBs": total / dt / 1e9}
        with open(os.path.join(DST, f"{tag}_{m}_bench.json"), "w") as f:
            f.write(json.dumps(bench) + "\n")
        print(m, f"{bench['ms_per_step']:.3f} ms/step, traffic {total / 1e9:.2f} GB/step = {total / dt / 1e12:.2f} TB/s")


if __name__ == "__main__":
    main()
