O=gpurun_out/s2l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "streaming_source_model or ilrma_source_update" 2>&1 | tail -15 > $O/ops_tests.log
timeout 900 python -m pytest tests/test_gpu_widem.py -x -q 2>&1 | tail -15 > $O/widem_tests.log
python tools/microbench.py --K 10 --only 'ilrma_source_update' --reps 30 > $O/micro.txt 2>&1
ASSX_SRC_NMF=0 python tools/microbench.py --K 10 --only 'ilrma_source_update' --reps 30 >> $O/micro.txt 2>&1
python tools/microbench.py --K 16 --only 'ilrma_source_update' --reps 30 >> $O/micro.txt 2>&1
python tools/microbench.py --K 6 --only 'ilrma_source_update' --reps 30 >> $O/micro.txt 2>&1
python tools/widem_bench.py 5:4 8:4 8:10 > $O/widem.txt 2>&1
ASSX_SRC_NMF=0 python tools/widem_bench.py 8:4 >> $O/widem.txt 2>&1
python bench.py --cpu-iters 0 --basis 10 --steps 200 --warmup 20 --roofline-b8 0 > $O/bench_k10.json 2>/dev/null
