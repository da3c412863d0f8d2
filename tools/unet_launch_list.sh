#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 kernel trace of tools/bench_unet.py, then the launches of the LAST training step in
# order (kernel, grid, workgroup, LDS, duration) into gpurun_out/<tag>_unet_launches.txt.
set -e
export TMPDIR=/tmp
TAG=${1:-probe}
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 tools/bench_unet.py --steps 6 --warmup 2 > /dev/null 2> $OUT/trace.err
python3 - <<PY > gpurun_out/${TAG}_unet_launches.txt
import csv
rows = list(csv.DictReader(open("$OUT/trace/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("unet::", "").replace("void ", "").split("(")[0]
# the last step starts at the last k_gather
idx = max(i for i, r in enumerate(rows) if short(r["Kernel_Name"]).endswith("k_gather"))
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {d:8.1f} {short(r["Kernel_Name"]):40s} grid {g} wg {r["Workgroup_Size_X"]} lds {r["LDS_Block_Size"]}')
PY
rm -rf $OUT/trace
