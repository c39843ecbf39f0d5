O=gpurun_out/s3k; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -mllvm -amdgpu-mfma-vgpr-form tools/probes/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe > $O/clock_probe.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -k "nmf" 2>&1 | tail -3 > $O/tests_nmf.log
