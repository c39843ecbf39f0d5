#!/usr/bin/env python3
"""Why does the float32 NumPy-in / NumPy-out call take 38 ms in tools/bench_configs.py and 22 ms in tools/call_breakdown.py?
Times the call in the order bench_configs.py reaches it (float64 section first) and on its own."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio_source_separation_amd.bss.ilrma import GaussILRMA
rng = np.random.default_rng(0)
M, F, T = 4, 1025, 4096
Xh = (rng.standard_normal((M, F, T)) + 1j * rng.standard_normal((M, F, T))) * rng.random((M, 1, T)) ** 2
order = sys.argv[1:] or ["float32", "float64", "float32"]
for dtype in order:
    GaussILRMA(n_basis=4, recordable_loss=False, dtype=dtype)(Xh, iteration=2)
    walls = []
    for _ in range(4):
        m2 = GaussILRMA(n_basis=4, recordable_loss=False, dtype=dtype)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Y = m2(Xh, iteration=100)
        walls.append(round((time.perf_counter() - t0) * 1e3, 2))
        del Y
    print(dtype, walls, flush=True)
