O=gpurun_out/s3w; mkdir -p $O; rm -f $O/*
C=audio_source_separation_amd/csrc
cp $C/libassx.so /tmp/main.so
for v in main m6 main m6; do
  if [ $v = main ]; then cp /tmp/main.so $C/libassx.so; else cp $C/libassx_$v.so $C/libassx.so; fi
  echo "== $v" >> $O/bench.txt
  timeout 300 python tools/widem_bench.py 6:4 2>/dev/null >> $O/bench.txt
done
cp /tmp/main.so $C/libassx.so
