"""Summary of tools/profile_ctbwd.sh: k_ct_bwd_lds (CAE_CTBWD=7) against k_ig_bwd_pair (default), per launch.
Writes profiles/round2_ctbwd_compare.csv.  FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes
(KiB units; FETCH_SIZE doubled for wide coalesced reads is NOT applied here: both kernels read mostly 4..16-byte pieces)."""
import csv
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "ctbwd")


def kernel_avgs(path, key):
    rows = list(csv.DictReader(open(path)))
    acc = defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        if key in name:
            acc["all"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
    return acc["all"]


def counters(path, key):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for r in csv.DictReader(open(path)):
        if key in r["Kernel_Name"]:
            acc[r["Counter_Name"]]["sum"] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
    return {k: v["sum"] / max(1, n[k]) for k, v in acc.items()}


def main():
    lines = [["kernel", "launches_per_step", "avg_us", "VALU_per_launch", "MFMA_per_launch", "VALU_per_MFMA", "VMEM_RD_per_launch",
              "LDS_per_launch", "fetch_MB_per_launch", "write_MB_per_launch"]]
    for mode, key in (("0", "k_ig_bwd_pair"), ("7", "k_ct_bwd")):   # k_ct_bwd_lds (whole images) and k_ct_bwd_band (the 16->8 layer)
        d = kernel_avgs(os.path.join(OUT, f"trace{mode}", "t_kernel_trace.csv"), key)
        c = counters(os.path.join(OUT, f"sq{mode}", "s_counter_collection.csv"), key)
        f = counters(os.path.join(OUT, f"fetch{mode}", "f_counter_collection.csv"), key)
        w = counters(os.path.join(OUT, f"write{mode}", "w_counter_collection.csv"), key)
        valu, mfma = c.get("SQ_INSTS_VALU", 0.0), c.get("SQ_INSTS_MFMA", 0.0)
        lines.append([key, 3, round(sum(d) / max(1, len(d)), 2), round(valu), round(mfma), round(valu / max(1.0, mfma), 1),
                      round(c.get("SQ_INSTS_VMEM_RD", 0.0)), round(c.get("SQ_INSTS_LDS", 0.0)),
                      round(f.get("FETCH_SIZE", 0.0) * 1024 / 1e6, 3), round(w.get("WRITE_SIZE", 0.0) * 1024 / 1e6, 3)])
    for mode in ("0", "7"):
        import json
        b = json.loads(open(os.path.join(OUT, f"bench{mode}.json")).read().strip().splitlines()[-1])
        lines.append([f"bench CAE_CTBWD={mode} us/step (under rocprofv3)", "", round(b["ms_per_step"] * 1000, 2)])
    dst = os.path.join(ROOT, "profiles", "round2_ctbwd_compare.csv")
    with open(dst, "w", newline="") as fh:
        csv.writer(fh).writerows(lines)
    for l in lines:
        print(l)


if __name__ == "__main__":
    main()
