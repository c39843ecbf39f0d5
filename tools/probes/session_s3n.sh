O=gpurun_out/s3t; mkdir -p $O; rm -f $O/*
for D in float64 float32; do
ASSX_WIDEM_PAIRS=0 python tools/probes/paircov_check.py run /tmp/p0.npz $D 2>/dev/null
python tools/probes/paircov_check.py run /tmp/p1.npz $D 2>/dev/null
echo "== $D" >> $O/cmp.txt
python tools/probes/paircov_check.py cmp /tmp/p1.npz /tmp/p0.npz 2>&1 | grep -v "^   " >> $O/cmp.txt
done
timeout 2400 python -m pytest tests/test_gpu_widem.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3 > $O/tests.log
timeout 600 python tools/widem_bench.py 5:4 6:4 7:4 8:4 8:10 2>/dev/null > $O/bench.txt
timeout 600 python tools/widem_bench.py --dtype float32 5:4 8:4 2>/dev/null >> $O/bench.txt
