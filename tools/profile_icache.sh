#!/bin/bash
# Runs ON THE GPU BOX: instruction-cache counters per kernel of the bench step (kernel-trace + PMC only).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/icache
mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $OUT/a -o a -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-api > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/b -o b -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-api > /dev/null 2> $OUT/b.err
ls $OUT/a $OUT/b
