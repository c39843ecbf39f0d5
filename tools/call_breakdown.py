#!/usr/bin/env python3
"""Where the time of one NumPy-in / NumPy-out GaussILRMA call goes (config 4, 100 iterations, loss off): the phases of
__call__ timed one by one with a device synchronisation after each (so the sum exceeds the un-instrumented wall)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_source_separation_amd._device import to_numpy  # noqa: E402
from audio_source_separation_amd.bss.ilrma import GaussILRMA  # noqa: E402

rng = np.random.default_rng(0)
M, F, T = 4, 1025, 4096
X = (rng.standard_normal((M, F, T)) + 1j * rng.standard_normal((M, F, T))) * rng.random((M, 1, T)) ** 2
out = {}
for dtype in ("float64", "float32"):
    GaussILRMA(n_basis=4, recordable_loss=False, dtype=dtype)(X, iteration=2)  # warm-up
    res = {}

    def lap(name, t0):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res[name] = round((t1 - t0) * 1e3, 3)
        return t1

    m = GaussILRMA(n_basis=4, recordable_loss=False, dtype=dtype)
    torch.cuda.synchronize()
    t = time.perf_counter()
    m.input = X
    m._reset()
    t = lap("reset_ms (upload X, initial state)", t)
    m.update_once()
    t = lap("first_update_once_ms (uploads T, V; plain covariance)", t)
    for _ in range(99):
        m.update_once()
    t = lap("99_update_once_ms", t)
    eng = m._engine
    scale = eng.projection_back_scale(m._X, m._Wd, m.reference_id, m._status)
    Y = eng.demix(m._X, m._Wd, scale=scale)
    m._check_status()
    t = lap("projection_back_ms", t)
    Yh = to_numpy(Y, np.complex128)
    t = lap("download_ms", t)
    walls = []
    for _ in range(3):
        m2 = GaussILRMA(n_basis=4, recordable_loss=False, dtype=dtype)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m2(X, iteration=100)
        walls.append(round((time.perf_counter() - t0) * 1e3, 2))
    res["uninstrumented_call_ms"] = walls
    out[dtype] = res
print(json.dumps(out, indent=1))
