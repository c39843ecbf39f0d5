O=gpurun_out/s2r; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_widem.py tests/test_gpu_manychan.py tests/test_gpu_models.py tests/test_gpu_multi.py -x -q 2>&1 | tail -6 > $O/tests.log
python tools/widem_bench.py 5:4 6:4 8:4 > $O/widem.txt 2>&1
ASSX_NMF_SMALL=0 python tools/widem_bench.py 8:4 >> $O/widem.txt 2>&1
python tools/widem_bench.py 5:4 8:4 --dtype float32 >> $O/widem.txt 2>&1
python tools/nmf_bench.py float64 4 | head -1 > $O/nmf_k4.txt 2>&1
ASSX_NMF_SMALL=0 python tools/nmf_bench.py float64 4 | head -1 >> $O/nmf_k4.txt 2>&1
python tools/fshard_bench.py > $O/fshard.json 2>/dev/null
