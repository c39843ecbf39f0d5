O=gpurun_out/s2d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_widem.py -x -q 2>&1 | tail -5 > $O/widem_tests.log
python tools/widem_bench.py 5:4 6:4 7:4 8:4 8:10 > $O/widem_bench.txt 2>&1
python tools/widem_bench.py 8:4 5:4 --dtype float32 > $O/f32.txt 2>&1
