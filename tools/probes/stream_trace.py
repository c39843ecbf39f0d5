#!/usr/bin/env python3
"""Per-workgroup timeline of cov_stream_kernel from a -DASSX_PROBE_BUILD -DSTREAM_TRACE=1 build (csrc/assx_stream.hpp): entry,
end of the first trip and exit of every workgroup on the 100 MHz clock, its XCD and its launch slot.
   ASSX_LIB_PATH=<probe lib> python tools/probes/stream_trace.py [B] [K]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio_source_separation_amd import _lib  # noqa: E402
from audio_source_separation_amd.ops import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
eng = Engine("float64")
M, F, T = 4, 1025, 4096
g = torch.Generator(device=eng.dev).manual_seed(0)
X = (torch.randn((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g) +
     1j * torch.randn((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g)).contiguous()
Tb = torch.rand((B, M, F, K), dtype=torch.float64, device=eng.dev, generator=g) + 0.1
V = torch.rand((B, M, K, T), dtype=torch.float64, device=eng.dev, generator=g) + 0.1
lib = ctypes.CDLL(_lib.LIB_PATH)
N = 8 * 4096
buf = (ctypes.c_ulonglong * N)()
for _ in range(5):
    eng.ilrma_cov_partials(X, Tb, V)
torch.cuda.synchronize()
for rep in range(2):
    assert lib.assx_debug_stream_trace(buf, 1) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.ilrma_cov_partials(X, Tb, V)
    e1.record()
    torch.cuda.synchronize()
    print("kernel (event) %.1f us" % (e0.elapsed_time(e1) * 1e3))
    assert lib.assx_debug_stream_trace(buf, 0) == 0
    a = np.frombuffer(buf, dtype=np.uint64).copy().reshape(4096, 8).astype(np.int64)
    live = a[:, 0] != 0
    gidx = np.nonzero(live)[0]
    a = a[live]
    t0 = a[:, 0].min()
    ent, ex, first = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, (a[:, 3] - t0) / 100.0
    xcc, nblk, slot = a[:, 2] & 0xf, (a[:, 2] >> 8) & 0xffffff, a[:, 2] >> 32
    print("workgroups %d, blocks per workgroup %d..%d | entry us: median %.2f p90 %.2f max %.2f | first trip done: median %.2f | exit: min %.2f p10 %.2f median %.2f p90 %.2f max %.2f" %
          (live.sum(), nblk.min(), nblk.max(), np.median(ent), np.percentile(ent, 90), ent.max(), np.median(first[first > 0]) if (first > 0).any() else -1,
           ex.min(), np.percentile(ex, 10), np.median(ex), np.percentile(ex, 90), ex.max()))
    res = ex - ent
    print("resident us: min %.2f median %.2f p90 %.2f max %.2f | launch slot %% 8 == xcc for %d of %d" %
          (res.min(), np.median(res), np.percentile(res, 90), res.max(), int(((slot % 8) == xcc).sum()), len(slot)))
    for x in range(8):
        m = xcc == x
        if m.any():
            print("  xcc %d: %4d workgroups | entry median %.2f max %.2f | resident median %.2f | exit median %.2f max %.2f" %
                  (x, m.sum(), np.median(ent[m]), ent[m].max(), np.median(res[m]), np.median(ex[m]), ex[m].max()))
    hw = a[:, 4]
    wave_id, simd, cu, sh, se = hw & 0xf, (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    ghz = (a[:, 6] - a[:, 5]) / np.maximum(res * 1e3, 1)
    print("shader clock GHz: median %.3f min %.3f max %.3f" % (np.median(ghz), ghz.min(), ghz.max()))
    late = res > np.median(res)
    def by(name, key):
        ks = sorted(set(key.tolist()))
        print("  by %-8s" % name, " ".join("%s:%.1f(%d%%late)" % (k, np.median(res[key == k]), 100 * late[key == k].mean()) for k in ks[:20]))
    by("simd", simd); by("wave_id", wave_id); by("se", se); by("sh", sh); by("cu", cu); by("xcc", xcc)
    by("g/250", gidx // 250); by("slot/256", slot // 256); by("nblk", nblk)
    cuid = cu | (sh << 4) | (se << 5) | (xcc << 8)
    per = {}
    for c, r in zip(cuid.tolist(), res.tolist()):
        per.setdefault(c, []).append(r)
    cnt = np.array([len(v) for v in per.values()])
    print("  workgroups per CU:", " ".join("%d:%d" % (k, int((cnt == k).sum())) for k in sorted(set(cnt.tolist()))))
    spread = np.array([max(v) - min(v) for v in per.values()])
    print("  within a CU: max - min resident median %.2f us; CU median resident: min %.2f max %.2f" %
          (np.median(spread), min(np.median(v) for v in per.values()), max(np.median(v) for v in per.values())))
    # bin boundaries crossed by a range: blocks of a bin = T / 64
    hist, edges = np.histogram(ex, bins=14)
    print("exit histogram (us):", " ".join("%.1f:%d" % (edges[i], hist[i]) for i in range(len(hist))))
    hist, edges = np.histogram(ent, bins=10)
    print("entry histogram (us):", " ".join("%.1f:%d" % (edges[i], hist[i]) for i in range(len(hist))))
