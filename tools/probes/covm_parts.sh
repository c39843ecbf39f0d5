# Which part of cov_mfma_kernel costs what: builds with one part compiled out (COVM_SKIP bits, assx_cov_mfma.hpp; results
# are wrong by construction, only the time means something) copied over libassx.so one after the other.
#   for v in 0 1 2 4 8 16 32 3 5 63; do ASSX_DEV=1 ASSX_OBJ=ab/o$v ASSX_OUT=ab/libassx_skip$v.so ASSX_EXTRA_FLAGS="-DASSX_PROBE_BUILD -DCOVM_SKIP=$v" bash build.sh; done
C=audio_source_separation_amd/csrc
cp $C/libassx.so /tmp/full.so
for v in 0 1 2 4 8 16 32 3 5 63 0; do
  cp $C/ab/libassx_skip$v.so $C/libassx.so
  echo "== skip $v: $(python tools/microbench.py --K ${1:-10} --only 'cov TV' --reps 50 2>/dev/null | grep cov)"
done
cp /tmp/full.so $C/libassx.so
