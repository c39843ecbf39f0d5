C=audio_source_separation_amd/csrc
cp $C/libassx.so /tmp/packed.so
for v in scalar packed scalar packed; do
  if [ $v = scalar ]; then cp $C/ab/libassx_scalar.so $C/libassx.so; else cp /tmp/packed.so $C/libassx.so; fi
  echo "== $v"; python tools/microbench.py --dtype float32 --only "$1" --reps 50 2>/dev/null | grep -v "^$"
  python tools/microbench.py --dtype float32 --B 8 --only "$1" --reps 20 2>/dev/null
done
cp /tmp/packed.so $C/libassx.so
