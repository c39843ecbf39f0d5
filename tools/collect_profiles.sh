#!/bin/bash
# Everything profiles/<tag>_* is made of, in one run on the GPU box:  bash tools/collect_profiles.sh r06
# (bench lines, rocprofv3 kernel tables, SQ / traffic PMC passes -- PMC only ever with --kernel-trace, as gpurun wants)
TAG="${1:-r06}"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
py() { python "$@"; }
B="$ROOT/bench.py"
SIDE="--measure-traffic 0 --with-b8 0 --with-loss 0"
py $B > $OUT/bench_f64.json 2> $OUT/bench_f64.err
( cd $ROOT && bash tools/driver_cmd_runs.sh $TAG 5 > $OUT/bench_driver_cmd_summary.txt 2>&1 )   # the driver's exact command, 5 runs + the 500/50 line
py $B --dtype float32 --cpu-iters 0 $SIDE > $OUT/bench_f32.json 2>/dev/null
py $B --cpu-iters 0 --utterances-per-gpu 8 --steps 100 --warmup 10 --roofline-b8 0 $SIDE > $OUT/bench_f64_8utt.json 2>/dev/null
py $B --cpu-iters 0 --basis 10 --steps 200 --warmup 20 --with-b8 0 --with-loss 0 > $OUT/bench_f64_k10.json 2>/dev/null
py $B --cpu-iters 0 --basis 10 --utterances-per-gpu 8 --steps 50 --warmup 5 --roofline-b8 0 $SIDE > $OUT/bench_f64_k10_8utt.json 2>/dev/null
py $B --cpu-iters 0 --config5 on --config5-utterances 64 --config5-iterations 100 --roofline-b8 0 $SIDE > $OUT/bench_f64_config5_1gpu_64utt.json 2>/dev/null
py $ROOT/tools/bench_configs.py > $OUT/bench_configs.json 2>/dev/null
py $ROOT/tools/nmf_bench.py float64 > $OUT/nmf_bench_f64.txt 2>/dev/null
py $ROOT/tools/nmf_bench.py float32 > $OUT/nmf_bench_f32.txt 2>/dev/null
py $ROOT/tools/widem_bench.py 5:4 6:4 7:4 8:4 8:10 > $OUT/widem_bench.txt 2>/dev/null
py $ROOT/tools/widem_bench.py 5:4 8:4 --dtype float32 > $OUT/widem_bench_f32.txt 2>/dev/null
py $ROOT/tools/fshard_bench.py > $OUT/fshard_bench_k4.json 2>/dev/null
for c in cfg1 cfg3; do for d in float64 float32; do py $ROOT/tools/probes/small_cfg_probe.py $c $d 2000 2>/dev/null >> $OUT/small_cfgs.txt; done; done
for d in float64 float32; do py $ROOT/tools/probes/call_cfgs.py $d 2>/dev/null >> $OUT/call_cfgs.txt; done
# rocprofv3 kernel tables: the driver's own command line, the K=10 line, NMF config 2, the wide-channel path
RP="--cpu-iters 0 --measure-traffic 0 --with-b8 0 --with-loss 0 --with-other-configs 0"
rocprofv3 --kernel-trace --stats -d $OUT/prof_cfg4 -o p -- python $B --steps 20 --warmup 5 $RP --with-f32 0 --with-default-basis 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_cfg5_8utt -o p -- python $B --steps 20 --warmup 5 $RP --utterances-per-gpu 8 --roofline-b8 0 --with-f32 0 --with-default-basis 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_f32 -o p -- python $B --steps 20 --warmup 5 $RP --dtype float32 --roofline-b8 0 --with-default-basis 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_k10 -o p -- python $B --steps 20 --warmup 5 $RP --basis 10 --with-f32 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_nmf -o p -- python $ROOT/tools/nmf_bench.py float64 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_m8 -o p -- python $ROOT/tools/widem_bench.py 8:4 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_m5 -o p -- python $ROOT/tools/widem_bench.py 5:4 > /dev/null 2>&1
for t in cfg4 cfg5_8utt f32 k10 nmf m8 m5; do py $ROOT/tools/rocprof_summary.py $OUT/prof_$t > $OUT/${t}_kernel_stats.md 2>&1; done
# conditioning of the IP sweep's matrices over a run of the headline input
py $ROOT/tools/probes/ip_cond_hist.py > $OUT/ip_cond_hist.txt 2>/dev/null
# SQ counters
bash $ROOT/tools/pmc_kernel.sh "cov TV partial" cov_stream $TAG/sq_cov_k4 > $OUT/sq_cov_k4.txt 2>&1
bash $ROOT/tools/pmc_kernel.sh "cov TV partial" cov_mfma $TAG/sq_cov_k10 --K 10 > $OUT/sq_cov_k10.txt 2>&1
bash $ROOT/tools/pmc_kernel.sh "ilrma_spatial_update" pair_cov $TAG/sq_pair_cov_m8 --M 8 > $OUT/sq_pair_cov_m8.txt 2>&1
# HBM traffic (the covariance kernels and, since round 6, the two passes of the source model)
ASSX_ROUND="round 6" py $ROOT/tools/pmc_traffic.py collect > /dev/null 2>&1
ASSX_ROUND="round 6" py $ROOT/tools/pmc_traffic.py report > $OUT/cov_traffic.json 2>&1
cp $ROOT/profiles/cov_traffic.json $OUT/cov_traffic.json 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -mllvm -amdgpu-mfma-vgpr-form $ROOT/tools/probes/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe > $OUT/clock_probe.txt; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $ROOT/tools/probes/mfma_valu_share_probe.hip -o /tmp/share_probe && /tmp/share_probe > $OUT/mfma_valu_share_probe.txt
cd $ROOT && timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/gpu_tests.log
# soak: the suite twice more (timing-dependent errors show as a run that differs)
for i in 2 3 4 5; do timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -1 >> $OUT/soak.txt; done
rm -rf $OUT/prof_*/*.db $OUT/sq_*_[abc] $ROOT/gpurun_out/pmc_fetch_* $ROOT/gpurun_out/pmc_write_* 2>/dev/null
ls -la $OUT
