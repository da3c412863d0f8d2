#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel-trace stats of the bench command, then the two HBM
# counter passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Counter passes use --kernel-trace only, as the pool requires.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/profile
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --steps 1000 --warmup 100 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-api --launch-order $OUT/launch_order.json > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-api > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/sq -o s -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-api > /dev/null 2> $OUT/sq.err
# matrix-core utilisation of the MFMA kernels (north_star): busy cycles of the MFMA pipes over busy cycles of the CUs
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $OUT/mfma -o m -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-api > /dev/null 2> $OUT/mfma.err
ls $OUT $OUT/trace | head -30
cat $OUT/bench.json | cut -c1-300
