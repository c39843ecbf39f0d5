#!/usr/bin/env python3
"""Per-step durations inside the driver's 20-step region (HIP events between the steps): where do the ~45 us by which the
20-step figure exceeds 20 x the 500-step figure sit -- in the first steps after the idle moment of the bracket, or spread?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from audio_source_separation_amd.bss.ilrma import GaussILRMA
dev = torch.device("cuda", 0)
M, F, T, K = 4, 1025, 4096, 4
X = bench.synth_mixture(torch, dev, 1, M, F, T, seed=1000).to(torch.complex128).contiguous()
np.random.seed(111)
m = GaussILRMA(n_basis=K, recordable_loss=False, device=dev)
m.input = X
m._reset()
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    for _ in range(20):
        m.update_once()
    torch.cuda.synchronize()
for rep in range(4):
    for _ in range(5):
        m.update_once()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(20):
        m.update_once()
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(20)]
    print("region %d: wall %.1f us = %.2f per step; events: first %.1f  steps 2-5 %s  mean of 6-20 %.1f us" %
          (rep, wall * 1e6, wall * 1e6 / 20, d[0], ["%.1f" % x for x in d[1:5]], sum(d[5:]) / 15))
