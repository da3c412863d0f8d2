#!/bin/bash
# AddressSanitizer / LeakSanitizer run of the HOST side of libcae_hip (SURVEY.md §5 "sanitizers"; GPU ASAN is not available on
# this pool, so the device code is compiled as usual and never launched): every .hip source is compiled with
# -fsanitize=address -fno-gpu-sanitize into cae_tools_amd/csrc/_obj_asan/, linked with tests/asan/plan_check.cpp, and the
# executable - engine plans, tensor tables, error paths, destruction, for the ConvAE engine, the var engine (a trunk-mode
# ConvAE engine inside) and the UNET engine - runs on the CPU.  ~4 minutes (engine.hip is one translation unit).
#     bash tools/asan_host_check.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/cae_tools_amd/csrc/_obj_asan
mkdir -p "$OBJ"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize -Wno-cuda-compat -Wno-pass-failed -I$ROOT/include -I$ROOT/cae_tools_amd/csrc"
pids=()
for f in engine ctbwd vae_engine unet_engine linear_engine; do
  if [ ! -f "$OBJ/$f.o" ] || [ -n "$(find "$ROOT/cae_tools_amd/csrc" "$ROOT/include" -newer "$OBJ/$f.o" \( -name '*.h' -o -name "$f.hip" \) | head -1)" ]; then
    $HIPCC $FLAGS -c "$ROOT/cae_tools_amd/csrc/$f.hip" -o "$OBJ/$f.o" > "$OBJ/$f.log" 2>&1 &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -g -fsanitize=address -I"$ROOT/include" -c "$ROOT/tests/asan/plan_check.cpp" -o "$OBJ/plan_check.o"
$HIPCC --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -o "$OBJ/plan_check" "$OBJ/plan_check.o" "$OBJ/engine.o" "$OBJ/ctbwd.o" "$OBJ/vae_engine.o" "$OBJ/unet_engine.o" "$OBJ/linear_engine.o" -ldl > "$OBJ/link.log" 2>&1
ASAN_OPTIONS=detect_leaks=1:protect_shadow_gap=0 "$OBJ/plan_check"
