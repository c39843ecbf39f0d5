O=gpurun_out/s3i; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probes/mfma_valu_share_probe.hip -o /tmp/share_probe && /tmp/share_probe > $O/mfma_valu_share_probe.txt 2>&1
