#!/bin/bash
# Runs ON THE GPU BOX: instruction-mix and HBM counters of the LDS-staged backward alternative (CAE_CTBWD=7) next to the default
# gather pair, kernel-trace + PMC passes only (as the pool requires).  Summaries: tools/summarise_ctbwd.py.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/ctbwd
mkdir -p $OUT
for mode in 0 7; do
  export CAE_CTBWD=$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$mode -o t -- python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-train-api > $OUT/bench$mode.json 2> $OUT/bench$mode.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq$mode -o s -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-api > /dev/null 2> $OUT/sq$mode.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch$mode -o f -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-api > /dev/null 2> $OUT/fetch$mode.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write$mode -o w -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-api > /dev/null 2> $OUT/write$mode.err
done
ls $OUT
