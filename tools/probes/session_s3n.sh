O=gpurun_out/s3q; mkdir -p $O; rm -f $O/*
C=audio_source_separation_amd/csrc
ASSX_WIDEM_PAIRS=0 python tools/probes/paircov_check.py run /tmp/p0.npz 2>/dev/null
cp $C/libassx.so /tmp/main.so; cp $C/libassx_wait0.so $C/libassx.so
ASSX_WIDEM_PAIRS_M5=1 python tools/probes/paircov_check.py run /tmp/p1.npz 2>/dev/null
echo "== vmcnt(0) every trip + s_sleep in the prologue" >> $O/cmp.txt
python tools/probes/paircov_check.py cmp /tmp/p1.npz /tmp/p0.npz 2>&1 | grep -v "^   " >> $O/cmp.txt
cp /tmp/main.so $C/libassx.so
