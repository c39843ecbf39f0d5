O=gpurun_out/s2f; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "wide_channel or frequency_sharded" 2>&1 | tail -15 > $O/fullsize_new.log
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -k "rccl" 2>&1 | tail -15 > $O/multi_new.log
python tools/fshard_bench.py > $O/fshard_bench_k4.json 2> $O/fshard_err.txt
python tools/fshard_bench.py --basis 10 > $O/fshard_bench_k10.json 2>> $O/fshard_err.txt
