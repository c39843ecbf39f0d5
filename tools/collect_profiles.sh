#!/bin/bash
# Collect the evidence committed under profiles/ (run on the GPU box through gpurun):
#   bash tools/collect_profiles.sh TAG      -> gpurun_out/TAG/*
set -u
TAG="${1:-r01}"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py --with-loss > $OUT/bench_f64.json 2> $OUT/bench_f64.err
python bench.py --dtype float32 --cpu-iters 0 > $OUT/bench_f32.json 2> $OUT/bench_f32.err
python tools/bench_configs.py > $OUT/bench_configs.json 2> $OUT/bench_configs.err
python bench.py --steps 30 --warmup 5 --utterances-per-gpu 8 --cpu-iters 0 > $OUT/bench_f64_b8.json 2> $OUT/bench_f64_b8.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof -o p -- python $ROOT/bench.py --steps 20 --warmup 3 --cpu-iters 0 > $OUT/prof_bench.log 2>&1)
python tools/pmc_traffic.py collect
python tools/pmc_traffic.py report > $OUT/cov_traffic.log 2>&1
cp profiles/cov_traffic.json $OUT/cov_traffic.json
tail -n 1 $OUT/bench_f64.json | cut -c1-300
