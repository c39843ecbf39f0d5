#!/usr/bin/env python3
"""Timeline of nmf_basis_mfma_kernel from a -DASSX_PROBE_BUILD -DNMF_TRACE=1 build (csrc/assx_nmf_mfma.hpp): entry / exit of
every workgroup on the 100 MHz clock, and shader-clock stamps of the four waves of one workgroup (1 entry, 2 block
prologue done, 3 each step after staging, 4 loop done, 5 cross-wave combine done, 6 records stored, 7 ticket taken, 8 / 9
block done without / with the finalize).   ASSX_LIB_PATH=<probe lib> python tools/probes/nmf_trace.py [dtype] [K]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio_source_separation_amd import _lib  # noqa: E402
from audio_source_separation_amd.ops import Engine  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "float64"
F, T, K = 1025, 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 32
eng = Engine(dtype)
g = torch.Generator(device=eng.dev).manual_seed(0)
X = (torch.rand((1, F, T), dtype=torch.float64, device=eng.dev, generator=g) ** 2).to(eng.prec.real)
Tb = torch.rand((1, F, K), dtype=torch.float64, device=eng.dev, generator=g).to(eng.prec.real)
V = torch.rand((1, K, T), dtype=torch.float64, device=eng.dev, generator=g).to(eng.prec.real)
lib = ctypes.CDLL(_lib.LIB_PATH)
N = 8192 + 4 * 512
buf = (ctypes.c_ulonglong * N)()
for _ in range(5):
    eng.nmf_update(_lib.NMF_IS_MM, X, Tb, V)
torch.cuda.synchronize()
assert lib.assx_debug_nmf_trace(buf, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
eng.nmf_update(_lib.NMF_IS_MM, X, Tb, V)
e1.record()
torch.cuda.synchronize()
print("update (event) %.1f us" % (e0.elapsed_time(e1) * 1e3))
assert lib.assx_debug_nmf_trace(buf, 0) == 0
a = np.frombuffer(buf, dtype=np.uint64).copy()
wg = a[:8192].reshape(1024, 8).astype(np.int64)
live = wg[:, 0] != 0
wg = wg[live]
t0 = wg[:, 0].min()
ent, ex, tk = (wg[:, 0] - t0) / 100.0, (wg[:, 1] - t0) / 100.0, (wg[:, 2] - t0) / 100.0  # us
fin = wg[:, 3] > 0
print("workgroups %d (%d of them finalize a block) | entry us: min %.2f median %.2f max %.2f | exit us: min %.2f median %.2f p90 %.2f max %.2f" %
      (live.sum(), fin.sum(), ent.min(), np.median(ent), ent.max(), ex.min(), np.median(ex), np.percentile(ex, 90), ex.max()))
print("resident us: median %.2f max %.2f | last ticket taken at us: median %.2f max %.2f" %
      (np.median(ex - ent), (ex - ent).max(), np.median(tk), tk.max()))
print("not finalizing: ticket->exit us median %.2f | finalizing: ticket->exit median %.2f max %.2f, exit median %.2f max %.2f" %
      (np.median((ex - tk)[~fin]), np.median((ex - tk)[fin]), (ex - tk)[fin].max(), np.median(ex[fin]), ex[fin].max()))
ghz = (wg[:, 5] - wg[:, 4]) / ((ex - ent) * 1e3)
xcc, hw = wg[:, 7] & 0xf, wg[:, 6]
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)  # cu_id, sh_id, se_id, xcc
print("shader clock per workgroup GHz: min %.3f median %.3f max %.3f" % (ghz.min(), np.median(ghz), ghz.max()))
gi = np.nonzero(live)[0]
print("workgroup index %% 8 == xcc id for %d of %d" % (int(((gi % 8) == xcc).sum()), len(gi)))
for x in range(8):
    m = xcc == x
    if m.any():
        d = (tk - ent)[m]
        print("  xcc %d: %3d workgroups, %3d distinct CUs, clock %.3f GHz, entry->ticket us min %.2f median %.2f p90 %.2f max %.2f, exit median %.2f max %.2f" %
              (x, m.sum(), len(set(cu[m].tolist())), np.median(ghz[m]), d.min(), np.median(d), np.percentile(d, 90), d.max(), np.median(ex[m]), ex[m].max()))
per_cu = {}
for c, r in zip(cu.tolist(), (ex - ent).tolist()):
    per_cu.setdefault(c, []).append(r)
cnt = np.array([len(v) for v in per_cu.values()])
print("workgroups per CU: " + " ".join("%d:%d" % (k, int((cnt == k).sum())) for k in sorted(set(cnt.tolist()))))
for k in sorted(set(cnt.tolist())):
    rs = [r for v in per_cu.values() if len(v) == k for r in v]
    print("  CUs with %d workgroups: resident median %.2f us" % (k, np.median(rs)))
d = tk - ent
order = np.argsort(d)
print("entry->ticket us: min %.2f p10 %.2f median %.2f p90 %.2f max %.2f" % (d.min(), np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max()))
print("slowest 12 workgroups (index, xcc, cu, entry, entry->ticket):", [(int(gi[i]), int(xcc[i]), int(cu[i] & 0xff), round(float(ent[i]), 2), round(float(d[i]), 2)) for i in order[-12:]])
print("fastest 6 workgroups:", [(int(gi[i]), int(xcc[i]), int(cu[i] & 0xff), round(float(ent[i]), 2), round(float(d[i]), 2)) for i in order[:6]])
# do the two workgroups of a CU finish together?
pairs = {}
for i in range(len(gi)):
    pairs.setdefault(int(cu[i]), []).append(float(d[i]))
dd = np.array([abs(v[0] - v[1]) for v in pairs.values() if len(v) == 2])
if len(dd):
    print("CUs with two workgroups: |difference of entry->ticket| median %.2f max %.2f us; CU mean spread: min %.2f max %.2f" %
          (np.median(dd), dd.max(), min(np.mean(v) for v in pairs.values()), max(np.mean(v) for v in pairs.values())))
hist, edges = np.histogram(ex, bins=12)
print("exit histogram (us):", " ".join("%.1f:%d" % (edges[i], hist[i]) for i in range(len(hist))))
for wv in range(4):
    s = a[8192 + wv * 512:8192 + (wv + 1) * 512]
    s = s[s != 0]
    ids, c = (s >> np.uint64(56)).astype(int), (s & np.uint64((1 << 56) - 1)).astype(np.int64)
    c = c - c[0]
    steps = c[ids == 3]
    d = np.diff(steps)
    line = "wave %d: stamps %d | " % (wv, len(s))
    for i in range(1, len(ids)):
        if ids[i] != 3 or ids[i - 1] != 3:
            line += "%d->%d %d  " % (ids[i - 1], ids[i], c[i] - c[i - 1])
    print(line)
    if len(d):
        print("        steps %d, cycles per step: mean %.0f min %d max %d | total %d cycles" % (len(steps), d.mean(), d.min(), d.max(), c[-1]))
