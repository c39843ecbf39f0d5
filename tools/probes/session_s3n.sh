O=gpurun_out/s3x; mkdir -p $O; rm -f $O/*
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1 >> $O/soak.log; done
for D in float64 float32; do for i in 1 2 3; do
ASSX_WIDEM_PAIRS=0 python tools/probes/paircov_check.py run /tmp/p0.npz $D 2>/dev/null
python tools/probes/paircov_check.py run /tmp/p1.npz $D 2>/dev/null
python tools/probes/paircov_check.py cmp /tmp/p1.npz /tmp/p0.npz 2>&1 | grep -c "count 0 of" >> $O/soak.log
done; done
