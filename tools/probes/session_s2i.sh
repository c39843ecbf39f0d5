O=gpurun_out/s2i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -k "nmf" 2>&1 | tail -5 > $O/nmf_tests.log
python tools/nmf_bench.py float64 32 > $O/nmf_f64_k32.txt 2>&1
python tools/nmf_bench.py float32 32 > $O/nmf_f32_k32.txt 2>&1
python tools/nmf_bench.py float64 10 | head -2 > $O/nmf_f64_k10.txt 2>&1
python tools/microbench.py --K 10 --only 'ilrma_source_update' --reps 30 > $O/micro_k10.txt 2>&1
python tools/widem_bench.py 8:4 5:4 > $O/widem.txt 2>&1
