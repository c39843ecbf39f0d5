#!/usr/bin/env python3
"""A/B of the in-group data movement of the per-bin sweeps (csrc/assx_group_linalg.hpp: DPP / v_readlane against
ds_bpermute): digests of (U, W, status) after one spatial update per shape -- pure data movement, every bit must stay --
and the time of the spatial update at benchmark size.  Run once per library and diff the outputs:

    python tools/probes/ip_dpp_ab.py                                  > a.txt      # the library in the tree
    ASSX_LIB_PATH=.../ab/libassx_nodpp.so python tools/probes/ip_dpp_ab.py > b.txt  # built with -DASSX_GROUP_DPP=0
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from audio_source_separation_amd import _lib  # noqa: E402
from audio_source_separation_amd.ops import Engine  # noqa: E402

SHAPES = [  # B, M, F, T, K, algorithm
    (1, 4, 1025, 4096, 4, "IP"), (1, 4, 257, 1000, 10, "IP"), (2, 4, 70, 333, 3, "IP"), (1, 2, 65, 300, 2, "IP"),
    (1, 3, 21, 333, 4, "IP"), (2, 3, 40, 200, 10, "IP"), (1, 2, 33, 129, 9, "IP"), (1, 5, 40, 300, 4, "IP"),
    (1, 6, 33, 200, 3, "IP"), (1, 7, 20, 150, 4, "IP"), (1, 8, 1025, 4096, 4, "IP"), (1, 8, 30, 200, 10, "IP"),
    (1, 4, 129, 500, 4, "ISS"), (1, 3, 65, 300, 4, "ISS"), (1, 4, 129, 500, 4, "IP2"), (1, 2, 65, 300, 2, "IP2"),
    (1, 6, 40, 256, 4, "ISS"),
]


def main():
    for dtype in ("float64", "float32"):
        eng = Engine(dtype)
        for (B, M, F, T, K, alg) in SHAPES:
            if alg == "IP2" and M != 2:
                continue
            g = torch.Generator(device=eng.dev).manual_seed(F * 7 + K + M)
            X = (torch.randn((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g) +
                 1j * torch.randn((B, M, F, T), dtype=torch.float64, device=eng.dev, generator=g)).to(eng.prec.cplx).contiguous()
            W = (torch.eye(M, dtype=torch.complex128, device=eng.dev).repeat(B, F, 1, 1) +
                 0.3 * torch.randn((B, F, M, M), dtype=torch.complex128, device=eng.dev, generator=g)).to(eng.prec.cplx).contiguous()
            Tb = (torch.rand((B, M, F, K), dtype=torch.float64, device=eng.dev, generator=g) + 0.1).to(eng.prec.real)
            V = (torch.rand((B, M, K, T), dtype=torch.float64, device=eng.dev, generator=g) + 0.1).to(eng.prec.real)
            st = eng.new_status(B)
            spatial = {"IP": _lib.SPATIAL_IP, "ISS": _lib.SPATIAL_ISS, "IP2": _lib.SPATIAL_IP2}[alg]
            for _ in range(2):  # two sweeps: the second starts from a W that is no longer near the identity
                eng.ilrma_spatial_update(X, W, Tb, V, domain=2, status=st, spatial=spatial)
            torch.cuda.synchronize()
            h = hashlib.sha1(torch.view_as_real(W).cpu().numpy().tobytes() + st.cpu().numpy().tobytes()).hexdigest()[:16]
            ok = bool(torch.isfinite(torch.view_as_real(W)).all().item())
            line = "%s %s B%d M%d F%d T%d K%d %s finite=%s" % (dtype, alg, B, M, F, T, K, h, ok)
            if F * T >= 250000 and alg == "IP":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    eng.ilrma_spatial_update(X, W, Tb, V, domain=2, status=st)
                e0.record()
                for _ in range(30):
                    eng.ilrma_spatial_update(X, W, Tb, V, domain=2, status=st)
                e1.record()
                e1.synchronize()
                line += "   # spatial update %.1f us" % (e0.elapsed_time(e1) / 30 * 1e3)
            print(line, flush=True)


if __name__ == "__main__":
    main()
