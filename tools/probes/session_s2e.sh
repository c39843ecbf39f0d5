O=gpurun_out/s2e; mkdir -p $O
for g in 256 512 768 1024; do echo "G=$g" >> $O/g_sweep.txt; ASSX_G=$g python tools/widem_bench.py 8:4 6:4 5:4 2>/dev/null >> $O/g_sweep.txt; done
