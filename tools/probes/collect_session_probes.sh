# Outputs of the probes of the last session of round 2 -> gpurun_out/r02s5/ (copied to profiles/r02_*.txt by hand).
# The two .hip probes are compiled HERE when their binaries did not travel (hipcc takes minutes on a fresh box:
# build them before calling gpurun: hipcc --offload-arch=gfx950 -O3 -w tools/probes/X.hip -o tools/probes/X.bin).
[ -x tools/probes/barrier_probe.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probes/barrier_cost_probe.hip -o tools/probes/barrier_probe.bin
[ -x tools/probes/vgpr_bank_probe.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probes/vgpr_bank_probe.hip -o tools/probes/vgpr_bank_probe.bin
O=gpurun_out/r02s5; mkdir -p $O
tools/probes/barrier_probe.bin > $O/barrier_cost_probe.txt 2>&1
tools/probes/vgpr_bank_probe.bin > $O/vgpr_bank_probe.txt 2>&1
bash tools/probes/g_b8.sh > $O/cov_workgroup_target_sweep.txt 2>&1
bash tools/probes/t_stride.sh > $O/cov_row_stride_sweep.txt 2>&1
( for b in 1 2 4 8 16 32; do python bench.py --cpu-iters 0 --utterances-per-gpu $b --steps $((400/b+10)) --warmup 10 --roofline-b8 0 --kernel-reps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('utterances/GPU %2d: %8.1f utterance-it/s, %.4f ms/step, cov kernel %.4f ms = %.3f of 8 TB/s' % (d['config']['utterances_per_gpu'], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"; done ) > $O/batch_size_sweep.txt 2>&1
python tools/covw_ab.py > $O/covw_ab.txt 2>&1
python tools/auxiva_bench.py > $O/auxiva_bench.txt 2>/dev/null
python bench.py > $O/bench_f64.json 2> $O/bench_f64.err
ls -la $O
