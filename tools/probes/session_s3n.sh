O=gpurun_out/s3s; mkdir -p $O; rm -f $O/*
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/tests.log
python bench.py > $O/bench.json 2> $O/bench.err
