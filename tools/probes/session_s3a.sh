O=gpurun_out/s3a; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_widem.py -x -q 2>&1 | tail -5 > $O/tests.log
echo "pairs=1" > $O/bench.txt; timeout 600 python tools/widem_bench.py 5:4 6:4 7:4 8:4 8:10 2>/dev/null >> $O/bench.txt
echo "pairs=0" >> $O/bench.txt; ASSX_WIDEM_PAIRS=0 timeout 600 python tools/widem_bench.py 5:4 6:4 7:4 8:4 8:10 2>/dev/null >> $O/bench.txt
echo "f32 pairs=1" >> $O/bench.txt; timeout 600 python tools/widem_bench.py --dtype float32 5:4 8:4 2>/dev/null >> $O/bench.txt
echo "f32 pairs=0" >> $O/bench.txt; ASSX_WIDEM_PAIRS=0 timeout 600 python tools/widem_bench.py --dtype float32 5:4 8:4 2>/dev/null >> $O/bench.txt
