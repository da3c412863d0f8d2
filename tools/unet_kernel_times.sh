#!/bin/bash
# Runs ON THE GPU BOX: tools/bench_unet.py under rocprofv3 --kernel-trace --stats, then the per-kernel table (us per step)
# into gpurun_out/<tag>_unet_kernels.txt.   usage: bash tools/unet_kernel_times.sh <tag>
set -e
export TMPDIR=/tmp
TAG=${1:-probe}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 tools/bench_unet.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 tools/bench_unet.py --steps 20 --warmup 3 > /dev/null 2> $OUT/trace.err
python3 - <<PY > gpurun_out/${TAG}_unet_kernels.txt
import csv, json
rows = list(csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")))
print(open("$OUT/bench.json").read().strip().splitlines()[-1][:160])
for r in rows:
    if float(r["Percentage"]) < 0.3: continue
    n = r["Name"].replace("(anonymous namespace)::", "").replace("unet::", "").replace("void ", "").split("(")[0]
    print(f'{n:45s} {int(r["Calls"])/23:6.2f} {float(r["TotalDurationNs"])/1e3/23:8.1f} {float(r["AverageNs"])/1e3:8.2f}')
PY
rm -rf $OUT/trace
cat gpurun_out/${TAG}_unet_kernels.txt
