O=gpurun_out/s3p; mkdir -p $O; rm -f $O/*
timeout 2400 python -m pytest tests/test_gpu_widem.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4 > $O/tests.log
timeout 600 python tools/widem_bench.py 5:4 6:4 7:4 8:4 8:10 2>/dev/null > $O/bench.txt
timeout 600 python tools/widem_bench.py --dtype float32 5:4 8:4 2>/dev/null >> $O/bench.txt
ASSX_WIDEM_PAIRS_M5=1 ASSX_PAIR_LDS_PAD=20000 timeout 600 python tools/widem_bench.py 5:4 2>/dev/null >> $O/bench.txt
