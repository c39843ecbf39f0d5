#!/bin/bash
# A/B of two builds of widem::pair_cov_kernel (run on the GPU box): the library in the tree (a) against $OLD_LIB (b, default
# csrc/ab/libassx_nospread.so: assx_widem.hip of the reference build linked with the tree's other objects, see
# paircov_knockout.sh for the recipe) -- digests, kernel times under rocprofv3, wide-channel tests.  Used for the two
# re-orderings of the weight chain that profiles/r04_paircov_spread_ab.txt records (both removed again).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pcspread; mkdir -p $OUT
cd $ROOT
OLD=${OLD_LIB:-$ROOT/audio_source_separation_amd/csrc/ab/libassx_nospread.so}
python tools/probes/ip_dpp_ab.py > $OUT/a.txt 2>&1
ASSX_LIB_PATH=$OLD python tools/probes/ip_dpp_ab.py > $OUT/b.txt 2>&1
diff <(sed 's/ *#.*//' $OUT/a.txt) <(sed 's/ *#.*//' $OUT/b.txt) > $OUT/diff.txt && echo "all digests equal" >> $OUT/diff.txt
for rep in 1 2; do
for v in a b; do
  lib=$ROOT/audio_source_separation_amd/csrc/libassx.so; [ $v = b ] && lib=$OLD
  for cfg in 8:4 7:4 6:4 5:4 8:10; do
    ASSX_LIB_PATH=$lib rocprofv3 --kernel-trace --stats -d $OUT/prof -o p -- python tools/widem_bench.py $cfg > $OUT/bench.txt 2>&1
    echo "$v $cfg $(grep 'ms/iteration' $OUT/bench.txt | head -1) $(python tools/rocprof_summary.py $OUT/prof 2>/dev/null | grep 'pair_cov' | head -1)" | tee -a $OUT/summary.txt
    rm -rf $OUT/prof
  done
done
done
timeout 900 python -m pytest tests/test_gpu_widem.py tests/test_gpu_coresidency.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 > $OUT/tests.log
cat $OUT/diff.txt $OUT/tests.log
