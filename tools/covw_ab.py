#!/usr/bin/env python3
"""Regression check for edits of the n_basis > 4 covariance kernel (cov_wide_kernel): a digest of (U, W) after one
spatial update per shape -- instruction-level edits must leave every bit alone -- and the kernel time.

    python tools/covw_ab.py            # float64, M = 4 shapes, compared with the recorded digests
    python tools/covw_ab.py all        # + M = 2, 3 and float32 (digests printed only)
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [  # B, M, F, T, K, domain
    (1, 4, 1025, 4096, 10, 2), (1, 4, 257, 1000, 10, 2), (2, 4, 70, 333, 5, 2), (1, 4, 33, 64, 7, 1),
    (1, 4, 129, 700, 16, 2), (3, 4, 19, 150, 12, 2), (1, 4, 9, 130, 30, 2), (1, 4, 40, 65, 6, 2),
]
EXTRA = [(1, 2, 65, 300, 5, 2), (1, 3, 21, 333, 17, 1), (2, 3, 40, 200, 10, 2), (1, 2, 33, 129, 9, 2)]


def child(dtype, all_m):
    import torch
    sys.path.insert(0, ROOT)
    from audio_source_separation_amd.ops import Engine
    eng = Engine(dtype)
    out = []
    for (B, M, F, T, K, dom) in SHAPES + (EXTRA if all_m else []):
        g = torch.Generator(device=eng.dev).manual_seed(F * 7 + K)
        X = (torch.randn((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g) +
             1j * torch.randn((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g)).to(eng.prec.cplx).contiguous()
        W = (torch.eye(M, dtype=torch.complex128, device=eng.dev).repeat(B, F, 1, 1) +
             0.1 * torch.randn((B, F, M, M), dtype=torch.complex128, device=eng.dev, generator=g)).to(eng.prec.cplx).contiguous()
        Tb = (torch.rand((B, M, F, K), dtype=torch.float64, device=eng.dev, generator=g) + 0.1).to(eng.prec.real)
        V = (torch.rand((B, M, K, T), dtype=torch.float64, device=eng.dev, generator=g) + 0.1).to(eng.prec.real)
        U = eng.empty((B, M, F, M, M), complex_=True)
        st = eng.new_status(B)
        eng.ilrma_spatial_update(X, W, Tb, V, domain=dom, status=st, U_out=U)
        torch.cuda.synchronize()
        h = hashlib.sha1(torch.view_as_real(U).cpu().numpy().tobytes() + torch.view_as_real(W).cpu().numpy().tobytes()).hexdigest()[:16]
        ok = bool(torch.isfinite(torch.view_as_real(U)).all().item())
        ms = 0.0
        if F * T >= 250000:
            for _ in range(3):
                eng.ilrma_cov_partials(X, Tb, V, domain=dom)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                eng.ilrma_cov_partials(X, Tb, V, domain=dom)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / 30
        out.append("%s B%d M%d F%d T%d K%d d%d %s finite=%s %.1f us" % (dtype, B, M, F, T, K, dom, h, ok, ms * 1e3))
    print("\n".join(out))


# digests of (U, W) from the build of commit b9b69be (cov_wide_kernel before the instruction trims), float64, M = 4
EXPECT = {"F1025 T4096 K10": "54c6c5a1dcc9a038", "F257 T1000 K10": "2081bfc55bff83b4", "F70 T333 K5": "2c2f2ea377788b89",
          "F33 T64 K7": "a5ea0cecbdd03b45", "F129 T700 K16": "b307aa1b405da8df", "F19 T150 K12": "aaf1d485c7b9cfef",
          "F9 T130 K30": "11388518ff9ecbe9", "F40 T65 K6": "af09664937449c6a"}

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], sys.argv[3] == "1")
        sys.exit(0)
    all_m = "1" if (len(sys.argv) > 1 and sys.argv[1] == "all") else "0"
    bad = 0
    for dt in (["float64", "float32"] if all_m == "1" else ["float64"]):
        out = subprocess.run([sys.executable, __file__, "child", dt, all_m], capture_output=True, text=True).stdout
        for line in out.strip().splitlines():
            f = line.split()
            key = " ".join(f[3:6])
            verdict = ""
            if dt == "float64" and f[2] == "M4" and key in EXPECT:
                verdict = "same bits" if EXPECT[key] == f[7] else "DIFFERENT BITS"
                bad += verdict != "same bits"
            print(line, verdict)
    sys.exit(1 if bad else 0)
