#!/bin/bash
# A/B of the DPP data movement in the per-bin sweeps (run on the GPU box): digests + times with the library in the tree
# and with ab/libassx_nodpp.so (ASSX_EXTRA_FLAGS=-DASSX_GROUP_DPP=0 ASSX_OBJ=ab ASSX_OUT=ab/libassx_nodpp.so build.sh).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ipdpp; mkdir -p $OUT
cd $ROOT
NODPP=$ROOT/audio_source_separation_amd/csrc/ab/libassx_nodpp.so
python tools/probes/ip_dpp_ab.py > $OUT/a.txt 2>&1
ASSX_LIB_PATH=$NODPP python tools/probes/ip_dpp_ab.py > $OUT/b.txt 2>&1
diff <(sed 's/ *#.*//' $OUT/a.txt) <(sed 's/ *#.*//' $OUT/b.txt) > $OUT/diff.txt && echo "all digests equal" >> $OUT/diff.txt
python tools/covw_ab.py all > $OUT/covw_ab.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_a -o p -- python tools/probes/ip_dpp_ab.py > /dev/null 2>&1
ASSX_LIB_PATH=$NODPP rocprofv3 --kernel-trace --stats -d $OUT/prof_b -o p -- python tools/probes/ip_dpp_ab.py > /dev/null 2>&1
for t in a b; do python tools/rocprof_summary.py $OUT/prof_$t 2>&1 | grep -i "ip_group\|iss_group\|ip2\|kernel |" > $OUT/${t}_kernels.md; done
rm -rf $OUT/prof_*
[ "$1" = notests ] || timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $OUT/tests.log
cat $OUT/diff.txt; cat $OUT/a_kernels.md; cat $OUT/b_kernels.md; grep "#" $OUT/a.txt $OUT/b.txt; cat $OUT/tests.log 2>/dev/null; tail -3 $OUT/covw_ab.txt
