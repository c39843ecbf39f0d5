#!/usr/bin/env python3
"""Host <-> HBM staging rates (assx_upload / assx_download) on a config-4 sized array (268.7 MB complex128), against
torch's pageable copy; sweep of the host thread count and the chunk size.  Each setting runs in a fresh host thread =
a fresh assx context = a fresh staging ring (ASSX_XFER_* are read when the ring is created)."""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_source_separation_amd._device import to_device, to_numpy  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
X = rng.standard_normal((4, 1025, 4096)) + 1j * rng.standard_normal((4, 1025, 4096))
NB = X.nbytes


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        del r
    return best


def measure(dev_dtype):
    t = to_device(X, dev_dtype, dev)  # creates the ring
    up = timed(lambda: to_device(X, dev_dtype, dev))
    down = timed(lambda: to_numpy(t, np.complex128))
    return {"upload_ms": round(up * 1e3, 2), "upload_GBps_host_bytes": round(NB / up / 1e9, 1),
            "download_ms": round(down * 1e3, 2), "download_GBps_host_bytes": round(NB / down / 1e9, 1)}


def in_thread(fn, env):
    box = {}
    os.environ.update(env)
    th = threading.Thread(target=lambda: box.setdefault("r", fn()))
    th.start()
    th.join()
    return box["r"]


out = {"bytes": NB, "host_cpus": os.cpu_count(), "torch_pageable": {}, "sweep_c128": [], "default": {}}
Xt = torch.from_numpy(X)
up = timed(lambda: Xt.to(dev))
Xd = Xt.to(dev)
down = timed(lambda: Xd.cpu())
out["torch_pageable"]["c128"] = {"upload_ms": round(up * 1e3, 2), "download_ms": round(down * 1e3, 2),
                                 "upload_GBps": round(NB / up / 1e9, 1), "download_GBps": round(NB / down / 1e9, 1)}
up = timed(lambda: Xt.to(torch.complex64).to(dev))  # round 2's float32 path: host-side conversion, then the copy
out["torch_pageable"]["c128_host_to_c64_dev"] = {"upload_ms": round(up * 1e3, 2)}
for threads in (1, 2, 4, 8, 16, 32):
    for chunk in (4, 16, 64):
        r = in_thread(lambda: measure(torch.complex128), {"ASSX_XFER_THREADS": str(threads), "ASSX_XFER_CHUNK_MB": str(chunk)})
        r.update(threads=threads, chunk_mb=chunk)
        out["sweep_c128"].append(r)
os.environ.pop("ASSX_XFER_THREADS")
os.environ.pop("ASSX_XFER_CHUNK_MB")
out["default"]["c128_dev"] = in_thread(lambda: measure(torch.complex128), {})
out["default"]["c64_dev"] = in_thread(lambda: measure(torch.complex64), {})
print(json.dumps(out, indent=1))
