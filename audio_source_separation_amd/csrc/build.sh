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
# ASSX_CHECK (default 1): every translation unit is compiled with -save-temps and tools/asm_wait_check.py walks its DEVICE
# assembly -- no instruction may touch the destination of an inline-asm load that no s_waitcnt has covered yet (the
# round-3 bug: registers of an asm LDS read copied before their wait, wrong only with two workgroups per CU at full
# size) -- and at every inline-asm LDS read the LDS-direct loads (buffer_load ... lds) in flight are counted against the
# largest counted wait of the kernel + one trip's loads, three times round each loop: a ring that has lost a wait.  A
# report fails the build.  ASSX_CHECK=0 skips it (kernel-tuning loops); an object built that way is rebuilt
# and checked by the next checking build ($OBJ/<src>.checked is the stamp).
CHECK=${ASSX_CHECK:-1}
CHECKER="$(cd ../.. && pwd)/tools/asm_wait_check.py"
pids=()
built=()
# ASSX_SRCS (tests): compile + check these sources (paths relative to csrc/, without .hip) instead of the library's own and
# stop before the link -- tests/test_asm_waits.py feeds the build a kernel with the round-3 bug and expects it to fail.
SRCS="${ASSX_SRCS:-assx_api assx_bss assx_nmf assx_stft assx_generic assx_widem assx_xfer assx_iterate assx_comm}"
for srcpath in $SRCS; do
  src=$(basename "$srcpath"); dir=$(dirname "$srcpath")
  stale=0
  [ -f "$OBJ/$src.o" ] || stale=1
  [ "$CHECK" = 1 ] && [ ! -f "$OBJ/$src.checked" ] && stale=1
  for dep in "$dir/$src.hip" *.hpp ../../include/assx.h build.sh; do
    [ "$dep" -nt "$OBJ/$src.o" ] && stale=1
  done
  if [ "$stale" = 1 ]; then
    # the matrix-core NMF kernels: accumulators in VGPRs.  Left to its heuristics the compiler keeps the loop-carried
    # accumulators in VGPRs but runs the MFMA chains on AGPRs: 16 v_accvgpr_write + 16 v_accvgpr_read + ~30 wait states per
    # 16 x 16 sub-tile (config 2: 65 -> 61 us per update).  Not for assx_bss.hip: two cov_mfma_kernel variants spill with it.
    SRCFLAGS=""
    [ "$src" = assx_nmf ] && SRCFLAGS="-mllvm -amdgpu-mfma-vgpr-form"
    # the streaming kernels: no SLP vectorisation.  In the float32 instantiations the vectoriser pairs independent
    # multiply-adds of two sources into v_pk_fma_f32 and pays four v_mov_b32 per pair to line the operands up (5
    # instructions for 2): basis_stream_vd_kernel<float> 34.7 -> 29.7 us, the float32 bench line +8.5 % (profiles/
    # r04_f32_slp_ab.txt).  float64 code and results are untouched (same digests, same times); float32 results move in the
    # last bit on two covariance shapes (packed multiply + add where the scalar form contracts to a fused multiply-add).  Not
    # for assx_nmf (config 1 in float32 is 4 % faster WITH it) and not needed for assx_widem (neutral).
    [ "$src" = assx_bss ] && SRCFLAGS="-fno-slp-vectorize"
    rm -f "$OBJ/$src.checked"
    if [ "$CHECK" = 1 ]; then
      rm -rf "$OBJ/temps_$src"; mkdir -p "$OBJ/temps_$src"
      $HIPCC $FLAGS $SRCFLAGS -save-temps=obj -c "$dir/$src.hip" -o "$OBJ/temps_$src/$src.o" &
    else
      $HIPCC $FLAGS $SRCFLAGS -c "$dir/$src.hip" -o "$OBJ/$src.o" &
    fi
    pids+=($!)
    built+=($src)
  fi
done
fail=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || fail=1; }; done
[ "$fail" = 0 ] || { echo "build.sh: compilation failed" >&2; exit 1; }
if [ "$CHECK" = 1 ]; then
  for src in "${built[@]:-}"; do
    [ -n "$src" ] || continue
    asm="$OBJ/temps_$src/$src-hip-amdgcn-amd-amdhsa-gfx950.s"
    [ -f "$asm" ] || { echo "build.sh: no device assembly for $src ($asm)" >&2; exit 1; }
    if ! python3 "$CHECKER" "$asm" > "$OBJ/temps_$src/check.log" 2>&1; then
      echo "build.sh: asm_wait_check FAILED for $src.hip (a pending inline-asm load touched before its wait, a ring without its counted wait, or inline asm that writes M0):" >&2
      tail -n 40 "$OBJ/temps_$src/check.log" >&2
      exit 1
    fi
    echo "asm_wait_check $src: $(tail -n 1 "$OBJ/temps_$src/check.log") ($(grep -c ASMSTART "$asm") asm blocks)"
    mv "$OBJ/temps_$src/$src.o" "$OBJ/$src.o"
    rm -rf "$OBJ/temps_$src"
    touch "$OBJ/$src.checked"
  done
fi
[ -n "${ASSX_SRCS:-}" ] && { echo "ASSX_SRCS given: compiled and checked, not linked"; exit 0; }
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJ/assx_api.o $OBJ/assx_bss.o $OBJ/assx_nmf.o $OBJ/assx_stft.o $OBJ/assx_generic.o $OBJ/assx_widem.o $OBJ/assx_xfer.o $OBJ/assx_iterate.o $OBJ/assx_comm.o -lpthread -ldl
echo "built $(pwd)/$OUT"
