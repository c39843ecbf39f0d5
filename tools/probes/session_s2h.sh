O=gpurun_out/s2h; mkdir -p $O
python tools/covw_ab.py > $O/covw_ab.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "wide or basis or long_ranges or cov" 2>&1 | tail -5 > $O/ops_tests.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "default_basis" 2>&1 | tail -3 > $O/fullsize.log
python tools/microbench.py --K 10 --reps 50 --only "cov TV" > $O/micro_k10.txt 2>&1
python tools/microbench.py --K 10 --B 8 --reps 10 --only "cov TV" >> $O/micro_k10.txt 2>&1
python tools/microbench.py --K 16 --reps 50 --only "cov TV" >> $O/micro_k10.txt 2>&1
python tools/microbench.py --K 6 --reps 50 --only "cov TV" >> $O/micro_k10.txt 2>&1
python bench.py --cpu-iters 0 --basis 10 --steps 200 --warmup 20 > $O/bench_f64_k10.json 2>/dev/null
