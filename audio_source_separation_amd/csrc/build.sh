#!/bin/bash
# Build libassx.so (HIP kernels + C-ABI) for gfx950, in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function"
pids=()
for src in assx_api assx_bss assx_nmf assx_stft assx_generic; do
  stale=0
  [ -f "$src.o" ] || stale=1
  for dep in "$src.hip" *.hpp ../../include/assx.h build.sh; do
    [ "$dep" -nt "$src.o" ] && stale=1
  done
  if [ "$stale" = 1 ]; then
    $HIPCC $FLAGS -c "$src.hip" -o "$src.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libassx.so assx_api.o assx_bss.o assx_nmf.o assx_stft.o assx_generic.o
echo "built $(pwd)/libassx.so"
