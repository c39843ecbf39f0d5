#!/usr/bin/env python3
"""One GaussILRMA.update_once() (7 launches, config 4) captured into a HIP graph with torch.cuda.CUDAGraph and replayed,
against the plain launches: capture cost, time per iteration, bit identity."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from audio_source_separation_amd.bss.ilrma import GaussILRMA
dev = torch.device("cuda", 0)
M, F, T, K = 4, 1025, 4096, 4
g = torch.Generator(device=dev).manual_seed(0)
X = torch.view_as_complex(torch.randn((M, F, T, 2), dtype=torch.float64, device=dev, generator=g)).contiguous()
np.random.seed(1)
m = GaussILRMA(n_basis=K, recordable_loss=False)
m.input = X
m._reset()
for _ in range(5):
    m.update_once()
torch.cuda.synchronize()
def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("eager      %.1f us/iter" % timeit(m.update_once))
W0 = m._Wd.clone(); T0 = m._Td.clone(); V0 = m._Vd.clone()
t0 = time.perf_counter()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    m.update_once()
torch.cuda.synchronize()
print("capture + instantiate %.2f ms" % ((time.perf_counter() - t0) * 1e3))
print("graph      %.1f us/iter" % timeit(gr.replay))
# same results? run k steps eager vs graph from same state
def run(fn, k=5):
    m._Wd.copy_(W0); m._Td.copy_(T0); m._Vd.copy_(V0)
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return m._Wd.clone(), m._Td.clone(), m._Vd.clone()
a = run(m.update_once); b = run(gr.replay)
print("bit-identical:", all(torch.equal(x, y) for x, y in zip(a, b)))
# host cost of a replay alone
t0 = time.perf_counter()
for _ in range(200): gr.replay()
print("host time per replay %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6)); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): m.update_once()
print("host time per eager update_once %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6)); torch.cuda.synchronize()
