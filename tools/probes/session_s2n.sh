O=gpurun_out/s2n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_widem.py -x -q -k "source_model" 2>&1 | tail -3 > $O/tests.log
for bw in 256 512 768 1024 1536 2048; do for aw in 1024 2048 4096; do
echo "basis_wgs=$bw act_wgs=$aw $(ASSX_NMF_BASIS_WGS=$bw ASSX_NMF_ACT_WGS=$aw python tools/microbench.py --K 10 --only 'ilrma_source_update' --reps 30 2>/dev/null | grep source)" >> $O/sweep_k10.txt
done; done
for bw in 256 512 1024 2048; do for aw in 1024 2048 4096; do
echo "M8 basis_wgs=$bw act_wgs=$aw $(ASSX_NMF_BASIS_WGS=$bw ASSX_NMF_ACT_WGS=$aw python tools/widem_bench.py 8:4 2>/dev/null | grep M=8)" >> $O/sweep_m8.txt
done; done
