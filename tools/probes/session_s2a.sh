set -x
mkdir -p gpurun_out/s2a
O=gpurun_out/s2a
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gpu_tests.log
python bench.py > $O/bench_f64.json 2> $O/bench_f64.err
python bench.py --cpu-iters 0 --basis 10 --steps 200 --warmup 20 > $O/bench_f64_k10.json 2>/dev/null
python tools/microbench.py --K 10 --reps 50 > $O/micro_k10.txt 2>&1
python tools/microbench.py --K 4 --reps 50 > $O/micro_k4.txt 2>&1
python tools/call_breakdown.py > $O/call_breakdown.json 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probes/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe > $O/clock_probe.txt 2>&1
python tools/widem_bench.py > $O/widem_bench.txt 2>&1
