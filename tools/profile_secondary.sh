#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the two secondary benches (UNET cfg3, var cfg5) under rocprofv3 - kernel-trace stats,
# then the FETCH_SIZE and WRITE_SIZE passes (separate passes: MI355X_MICROARCH.md "rocprofv3 PMC slots") - and the bench lines
# themselves with the CPU baseline.  tools/summarise_secondary.py turns the result into profiles/<tag>_{unet,vae}_*.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/secondary
mkdir -p $OUT
for m in unet vae; do
  python3 tools/bench_$m.py --steps 20 --warmup 3 --cpu > $OUT/${m}_bench.json 2> $OUT/${m}_bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${m}_trace -o t -- python3 tools/bench_$m.py --steps 20 --warmup 3 > /dev/null 2> $OUT/${m}_trace.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${m}_fetch -o f -- python3 tools/bench_$m.py --steps 10 --warmup 2 > /dev/null 2> $OUT/${m}_fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${m}_write -o w -- python3 tools/bench_$m.py --steps 10 --warmup 2 > /dev/null 2> $OUT/${m}_write.err
  echo "$m done"
done
