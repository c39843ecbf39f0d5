#!/bin/bash
# Build libassx.so (HIP kernels + C-ABI) for gfx950, in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function"
# ASSX_DEV=1: kernel-tuning build (M = 4, float64 instantiations of the BSS kernels only; objects kept apart in dev/)
OBJ=.
if [ "${ASSX_DEV:-0}" = 1 ]; then FLAGS="$FLAGS -DASSX_DEV_ONLY_M4_F64"; OBJ=dev; mkdir -p dev; fi
# ASSX_DEV=32: the same for the float32 instantiations
if [ "${ASSX_DEV:-0}" = 32 ]; then FLAGS="$FLAGS -DASSX_DEV_ONLY_M4_F32"; OBJ=dev32; mkdir -p dev32; fi
# ASSX_EXTRA_FLAGS / ASSX_OBJ / ASSX_OUT: A/B builds (e.g. -DASSX_PACKED_F32=0 into another object directory and library)
FLAGS="$FLAGS ${ASSX_EXTRA_FLAGS:-}"
OBJ=${ASSX_OBJ:-$OBJ}; mkdir -p "$OBJ"
OUT=${ASSX_OUT:-libassx.so}
pids=()
for src in assx_api assx_bss assx_nmf assx_stft assx_generic assx_widem assx_xfer; do
  stale=0
  [ -f "$OBJ/$src.o" ] || stale=1
  for dep in "$src.hip" *.hpp ../../include/assx.h build.sh; do
    [ "$dep" -nt "$OBJ/$src.o" ] && stale=1
  done
  if [ "$stale" = 1 ]; then
    # the matrix-core NMF kernels: accumulators in VGPRs.  Left to its heuristics the compiler keeps the loop-carried
    # accumulators in VGPRs but runs the MFMA chains on AGPRs: 16 v_accvgpr_write + 16 v_accvgpr_read + ~30 wait states per
    # 16 x 16 sub-tile (config 2: 65 -> 61 us per update).  Not for assx_bss.hip: two cov_mfma_kernel variants spill with it.
    SRCFLAGS=""
    [ "$src" = assx_nmf ] && SRCFLAGS="-mllvm -amdgpu-mfma-vgpr-form"
    $HIPCC $FLAGS $SRCFLAGS -c "$src.hip" -o "$OBJ/$src.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJ/assx_api.o $OBJ/assx_bss.o $OBJ/assx_nmf.o $OBJ/assx_stft.o $OBJ/assx_generic.o $OBJ/assx_widem.o $OBJ/assx_xfer.o -lpthread
echo "built $(pwd)/$OUT"
