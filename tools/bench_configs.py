#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md: BASELINE.json configs 1-4 through the drop-in classes (device-resident
inputs unless stated), plus the PCIe-inclusive ILRMA rate when the boundary is handed host NumPy buffers.
One PROCESS per dtype (the parent only merges): measured in one process, the second dtype's section inherited the
allocator / staging state of the first and its small configs and NumPy call read 10-100 % slow
(profiles/r03_f32_call_probe.txt)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_source_separation_amd.algorithm.nmf import EUCNMF, ISNMF  # noqa: E402
from audio_source_separation_amd.bss.ilrma import GaussILRMA  # noqa: E402
from audio_source_separation_amd.bss.iva import AuxGaussIVA, AuxLaplaceIVA  # noqa: E402

if "--dtype" not in sys.argv:  # parent: one child per dtype
    import subprocess
    merged = {}
    for d in ("float64", "float32"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dtype", d], capture_output=True, text=True, check=True)
        merged.update(json.loads(r.stdout[r.stdout.index("{"):]))
    print(json.dumps(merged, indent=1))
    sys.exit(0)

dev = torch.device("cuda", 0)
out = {}


def mix(M, F, T, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    S = torch.randn((M, F, T), dtype=torch.float64, device=dev, generator=g) + \
        1j * torch.randn((M, F, T), dtype=torch.float64, device=dev, generator=g)
    env = torch.rand((M, 1, T), dtype=torch.float64, device=dev, generator=g) ** 2
    A = torch.randn((F, M, M), dtype=torch.complex128, device=dev, generator=g)
    return torch.einsum("fmn,nft->mft", A, S * env).contiguous()


def time_updates(model, steps, warmup=3):
    for _ in range(warmup):
        model.update_once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.update_once()
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


for dtype in (sys.argv[sys.argv.index("--dtype") + 1],):
    res = {}
    # cfg1: EUC-NMF F=513 T=256 K=8 ; cfg2: IS-NMF F=1025 T=4096 K=32
    for name, cls, (F, T, K) in (("cfg1_eucnmf_513x256_k8", EUCNMF, (513, 256, 8)),
                                 ("cfg2_isnmf_1025x4096_k32", ISNMF, (1025, 4096, 32))):
        X = torch.rand((F, T), dtype=torch.float64, device=dev) ** 2
        np.random.seed(0)
        m = cls(n_basis=K, dtype=dtype)
        m.target = X
        m._reset()
        ups = time_updates(m, 300 if F < 1000 else 60, warmup=20)
        r = 8 if dtype == "float64" else 4
        res[name] = {"update_once_per_s": round(ups, 1), "gflops": round(12 * F * T * K * ups / 1e9, 1),
                     "algorithmic_GBps": round(2 * F * T * r * ups / 1e9, 1)}
    # cfg3: AuxLaplaceIVA / AuxGaussIVA  M=2 F=1025 T=2048
    for name, cls in (("cfg3_auxlaplaceiva_m2_1025x2048", AuxLaplaceIVA), ("cfg3_auxgaussiva_m2_1025x2048", AuxGaussIVA)):
        X = mix(2, 1025, 2048)
        m = cls(recordable_loss=False, dtype=dtype)
        m.input = X.to(torch.complex128 if dtype == "float64" else torch.complex64)
        m._reset()
        its = time_updates(m, 300, warmup=20)
        c = 16 if dtype == "float64" else 8
        res[name] = {"iterations_per_s": round(its, 1), "algorithmic_GBps": round(2 * 2 * 1025 * 2048 * c * its / 1e9, 1)}
    # cfg4 via the class, device-resident
    X = mix(4, 1025, 4096)
    np.random.seed(111)
    m = GaussILRMA(n_basis=4, recordable_loss=False, dtype=dtype)
    m.input = X.to(torch.complex128 if dtype == "float64" else torch.complex64)
    m._reset()
    its = time_updates(m, 50)
    c = 16 if dtype == "float64" else 8
    res["cfg4_gaussilrma_m4_1025x4096_k4"] = {"iterations_per_s": round(its, 1),
                                             "algorithmic_GBps_3passes": round(3 * 4 * 1025 * 4096 * c * its / 1e9, 1)}
    # PCIe-inclusive: NumPy in, NumPy out, 100 iterations, loss off (upload X, download Y included); best of 3 calls
    # and the split of one call into its upload / loop / download parts
    from audio_source_separation_amd._device import to_device, to_numpy
    Xh = X.cpu().numpy()
    np.random.seed(111)
    m = GaussILRMA(n_basis=4, recordable_loss=False, dtype=dtype)
    m(Xh, iteration=2)  # warm-up (allocations, staging ring)
    walls = []
    for _ in range(7):  # the first calls of a section carry allocator / staging-ring warm-up of this dtype: report them all
        m2 = GaussILRMA(n_basis=4, recordable_loss=False, dtype=dtype)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Y = m2(Xh, iteration=100)
        walls.append(time.perf_counter() - t0)
        del Y
    dt = min(walls)
    cplx = torch.complex128 if dtype == "float64" else torch.complex64
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Xd = to_device(Xh, cplx, dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    Yh = to_numpy(Xd, np.complex128)
    t2 = time.perf_counter()
    res["cfg4_numpy_in_numpy_out_100it"] = {"wall_s": round(dt, 4), "iterations_per_s_incl_pcie": round(100 / dt, 1),
                                            "walls_s": [round(w, 4) for w in walls],
                                            "upload_ms": round((t1 - t0) * 1e3, 2),
                                            "download_ms": round((t2 - t1) * 1e3, 2)}
    del Xd, Yh
    out[dtype] = res
print(json.dumps(out, indent=1))
