#!/usr/bin/env python3
"""Why is a SECOND loss-recording model in one process slow?  (bench.py --with-loss: value_k10_with_loss 1300-1700 it/s
where the same leg alone gives 4200.)  Times enqueue and completion of update_once() + _record_loss() loops."""
import gc
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio_source_separation_amd.bss.ilrma import GaussILRMA  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
X = (torch.randn((1, 4, 1025, 4096), dtype=torch.float64, device=dev, generator=g) + 1j * torch.randn((1, 4, 1025, 4096), dtype=torch.float64, device=dev, generator=g)).contiguous()


def leg(K, loss, steps=100, tag=""):
    np.random.seed(1)
    m = GaussILRMA(n_basis=K, recordable_loss=loss)
    m.input = X
    m._reset()
    for _ in range(10):
        m.update_once()
        if loss:
            m._record_loss()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.update_once()
        if loss:
            m._record_loss()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-28s K=%2d loss=%d: %.1f us/iteration (enqueue %.1f us)" % (tag, K, loss, (t2 - t0) / steps * 1e6, (t1 - t0) / steps * 1e6), flush=True)
    return m


mode = sys.argv[1] if len(sys.argv) > 1 else "all"
leg(10, True, tag="first loss model")
leg(10, True, tag="second loss model")
m = leg(4, True, tag="third (K=4)")
del m
gc.collect()
leg(10, True, tag="after gc.collect()")
leg(10, False, tag="loss off")
leg(10, True, tag="again with loss")
