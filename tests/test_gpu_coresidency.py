"""Full-size oracle steps under OTHER occupancies than the default partitions give.

Round 3's wrong-answer bug (registers of an inline-asm LDS read consumed before their wait) showed only where several
workgroups shared a CU at benchmark size.  The static checker now runs in the build (csrc/build.sh); these tests vary
the residency on purpose: ASSX_G (read on every call) forces the number of ranges of every flat partition -- the
streaming kernels of the M <= 4 path (cov_stream / basis_stream_vd / act_stream_vd), cov_mfma_kernel (n_basis 10) and the
wide-channel covariance (pair_cov_kernel, M = 5 and 8) -- to 1024 (half a round: every CU under-filled, long ranges),
3072 (a round and a half) and 4096 workgroups (two rounds, late workgroups join CUs whose first ones are mid-range).
Each case runs the update TWICE from the same state: the two results must be bit-identical (a timing-dependent read
shows as run-to-run differences) and agree with ONE oracle step (1025 x 4096, ilrma.py:286-338).
Plus one oracle step of the run-time-M path at M = 12 and full size (round 3 had it at toy sizes only)."""
import numpy as np
import pytest

from conftest import rel_err
from oracle import oracle_np as orc
from test_gpu_fullsize import _cfg4_state

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

_ORACLE = {}


def _oracle_step(M, K):
    if (M, K) not in _ORACLE:
        X, T0, V0 = _cfg4_state(40 + M, K=K, M=M)
        W = np.tile(np.eye(M, dtype=np.complex128), (1025, 1, 1))
        W, Tr, Vr, mask = orc.ilrma_update_once(X, W, T0, V0)
        assert mask.all()
        _ORACLE[(M, K)] = (X, T0, V0, W, Tr, Vr)
    return _ORACLE[(M, K)]


@pytest.mark.parametrize("G", ["1024", "3072", "4096"])
@pytest.mark.parametrize("M,K", [(4, 4), (4, 10), (5, 4), (8, 4)])
def test_full_size_oracle_step_with_forced_partitions(M, K, G, monkeypatch):
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    X, T0, V0, W, Tr, Vr = _oracle_step(M, K)
    monkeypatch.setenv("ASSX_G", G)
    runs = []
    for rep in range(2):
        m = GaussILRMA(n_basis=K)
        m.basis, m.activation = T0, V0
        m.input = X
        m._reset()
        m.update_once()
        runs.append((m.demix_filter.copy(), m.basis.copy(), m.activation.copy()))
    for a, b in zip(*runs):
        assert np.array_equal(a, b)  # bit-stable run to run
    Wg, Tg, Vg = runs[0]
    assert rel_err(Tg, Tr) < 1e-10 and rel_err(Vg, Vr) < 1e-10 and rel_err(Wg, W) < 1e-8
    # per bin, so that garbage in a few bins cannot hide in a norm over all of them
    per_bin = np.abs(Wg - W).max(axis=(1, 2)) / np.abs(W).max(axis=(1, 2))
    assert per_bin.max() < 1e-6, (int(per_bin.argmax()), float(per_bin.max()))


def test_many_channel_full_size_oracle_step():
    """M = 12 on the run-time-M kernels (csrc/assx_widem_rt.hpp) at 1025 bins x 2048 frames against one oracle step."""
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    M, K, F, T = 12, 4, 1025, 2048
    X, T0, V0 = _cfg4_state(52, M=M, F=F, T=T, K=K)
    m = GaussILRMA(n_basis=K)
    m.basis, m.activation = T0, V0
    m.input = X
    m._reset()
    m.update_once()
    W = np.tile(np.eye(M, dtype=np.complex128), (F, 1, 1))
    W, Tr, Vr, mask = orc.ilrma_update_once(X, W, T0, V0)
    assert mask.all()
    assert rel_err(m.basis, Tr) < 1e-10 and rel_err(m.activation, Vr) < 1e-10 and rel_err(m.demix_filter, W) < 1e-8
