"""Stage-level GPU parity: every C-ABI entry point against the CPU oracle on the same seeded inputs.

Tolerances (relative Frobenius error unless stated):
  float64 storage: 1e-11 for single reductions/contractions (summation order differs), 1e-9 after an IP sweep.
  float32 storage: 2e-5 for single reductions, 2e-3 after an IP sweep (cond(WU) amplifies rounding).
"""
import os

import numpy as np
import pytest

from conftest import load_golden, need_lab, rel_err
from oracle import oracle_np as orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", params=["float64", "float32"])
def eng(request):
    from audio_source_separation_amd.ops import Engine
    return Engine(dtype=request.param)


def tol(eng, t64, t32):
    return t64 if eng.prec.name == "float64" else t32


def dev_c(eng, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(eng.dev, eng.prec.cplx).contiguous()


def dev_r(eng, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(eng.dev, eng.prec.real).contiguous()


def host(t):
    a = t.detach().cpu().numpy()
    return a.astype(np.complex128) if np.iscomplexobj(a) else a.astype(np.float64)


def mixture(M, F, T, seed):
    rng = np.random.default_rng(seed)
    S = (rng.standard_normal((M, F, T)) + 1j * rng.standard_normal((M, F, T))) * rng.random((M, 1, T)) ** 2
    A = rng.standard_normal((F, M, M)) + 1j * rng.standard_normal((F, M, M))
    return np.einsum("fmn,nft->mft", A, S)


def rand_filters(M, F, seed):
    rng = np.random.default_rng(seed)
    return np.eye(M)[None] + 0.3 * (rng.standard_normal((F, M, M)) + 1j * rng.standard_normal((F, M, M)))


SHAPES = [(2, 9, 70), (3, 17, 200), (4, 33, 257), (4, 5, 1030)]


@pytest.mark.parametrize("M,F,T", SHAPES)
def test_demix(eng, M, F, T):
    X, W = mixture(M, F, T, 1), rand_filters(M, F, 2)
    Y = host(eng.demix(dev_c(eng, X[None]), dev_c(eng, W[None])))[0]
    assert rel_err(Y, orc.separate(X, W)) < tol(eng, 1e-14, 1e-6)
    rng = np.random.default_rng(3)
    s = rng.standard_normal((M, F)) + 1j * rng.standard_normal((M, F))
    Ys = host(eng.demix(dev_c(eng, X[None]), dev_c(eng, W[None]), scale=dev_c(eng, s[None])))[0]
    assert rel_err(Ys, orc.separate(X, W) * s[..., None]) < tol(eng, 1e-14, 1e-6)


@pytest.mark.parametrize("M,F,T", SHAPES)
def test_cov_accumulate(eng, M, F, T):
    X = mixture(M, F, T, 4)
    rng = np.random.default_rng(5)
    r_nt = rng.random((M, T)) ** 3
    r_nt[:, :3] = 0.0  # floored at eps
    r_nft = rng.random((M, F, T)) ** 3
    Xd = dev_c(eng, X[None])
    U = host(eng.cov_accumulate(Xd, dev_r(eng, r_nt[None]), eps=1e-3))[0]
    assert rel_err(U, orc.weighted_covariance(X, r_nt, 1e-3)) < tol(eng, 1e-12, 2e-5)
    U = host(eng.cov_accumulate(Xd, dev_r(eng, r_nft[None]), eps=1e-3))[0]
    assert rel_err(U, orc.weighted_covariance(X, r_nft, 1e-3)) < tol(eng, 1e-12, 2e-5)
    C = host(eng.cov_accumulate(Xd))[0]
    assert rel_err(C, orc.weighted_covariance(X, np.ones((1, T)), 1e-3)) < tol(eng, 1e-12, 2e-5)
    # Hermitian by construction
    assert np.array_equal(U, U.conj().transpose(0, 1, 3, 2))


def test_cov_batched_matches_single(eng):
    """Utterances are independent: a batched launch is bit-identical to per-utterance launches."""
    M, F, T = 4, 9, 300
    Xs = np.stack([mixture(M, F, T, 10 + b) for b in range(3)])
    rs = np.random.default_rng(6).random((3, M, T)) + 0.1
    Ub = host(eng.cov_accumulate(dev_c(eng, Xs), dev_r(eng, rs)))
    for b in range(3):
        U1 = host(eng.cov_accumulate(dev_c(eng, Xs[b:b + 1]), dev_r(eng, rs[b:b + 1])))[0]
        assert np.array_equal(Ub[b], U1)


@pytest.mark.parametrize("ip_par", ["0", "1"])
@pytest.mark.parametrize("M,F,T", SHAPES[:3])
def test_ip_update(eng, M, F, T, ip_par, monkeypatch):
    if ip_par != "0":
        need_lab("ASSX_IP_PAR")
    monkeypatch.setenv("ASSX_IP_PAR", ip_par)
    X, W = mixture(M, F, T, 7), rand_filters(M, F, 8)
    r = np.random.default_rng(9).random((M, T)) + 0.05
    U = orc.weighted_covariance(X, r)
    Wd = dev_c(eng, W[None])
    st = eng.new_status(1)
    eng.ip_update(dev_c(eng, U[None]), Wd, 1e12, st)
    Wref, mask = orc.ip_update(W.copy(), U)
    assert mask.all() and int(st.item()) == 0
    assert rel_err(host(Wd)[0], Wref) < tol(eng, 1e-10, 1e-3)


@pytest.mark.parametrize("ip_par", ["0", "1"])
def test_ip_cond_guard_and_singular(ip_par, monkeypatch):
    """cond(WU) >= threshold keeps the row (ilrma.py:520-528); an exactly singular WU flags LinAlgError.
    ip_par = 1 (laboratory builds): the sources-side-by-side form of the sweep (ASSX_IP_PAR, csrc/assx_group_linalg.hpp:
    ip_par_kernel)."""
    if ip_par != "0":
        need_lab("ASSX_IP_PAR")
    monkeypatch.setenv("ASSX_IP_PAR", ip_par)
    from audio_source_separation_amd import _lib
    from audio_source_separation_amd.ops import Engine
    eng = Engine("float64")
    g = load_golden("edge_cond_ilrma")
    X = g["X"]
    M, F, T = X.shape
    Tb, V = g["T0"], g["V0"]
    P = np.abs(X) ** 2
    T1, V1 = orc.ilrma_source_update(P, Tb, V)
    U = orc.weighted_covariance(X, orc.ilrma_variance(T1, V1))
    W0 = np.tile(np.eye(M, dtype=np.complex128), (F, 1, 1))
    Wref, mask = orc.ip_update(W0.copy(), U)
    Wd = dev_c(eng, W0[None])
    st = eng.new_status(1)
    eng.ip_update(dev_c(eng, U[None]), Wd, 1e12, st)
    Wg = host(Wd)[0]
    kept = ~mask.all(axis=0)
    assert kept[[2, 5]].all() and kept.sum() == 2
    assert np.array_equal(Wg[kept], W0[kept])  # bit-exact: rows untouched
    assert rel_err(Wg, Wref) < 1e-9
    assert int(st.item()) & _lib.STATUS_COND_REJECT and not int(st.item()) & _lib.STATUS_SINGULAR
    # moderately ill-conditioned bins straddling a custom threshold: decisions must match numpy's cond_2
    rng = np.random.default_rng(11)
    Ub = np.empty((M, 64, M, M), dtype=np.complex128)
    for f in range(64):
        Q1, _ = np.linalg.qr(rng.standard_normal((M, M)) + 1j * rng.standard_normal((M, M)))
        s = np.logspace(0, -rng.uniform(0, 8), M)
        Ub[:, f] = (Q1 * s) @ Q1.conj().T
    Wb = np.tile(np.eye(M, dtype=np.complex128), (64, 1, 1))
    for thr in (1e3, 1e4, 1e5):
        Wr, mk = orc.ip_update(Wb.copy(), Ub, thr)
        Wd = dev_c(eng, Wb[None])
        eng.ip_update(dev_c(eng, Ub[None]), Wd, thr, eng.new_status(1))
        got_kept = np.all(host(Wd)[0] == Wb, axis=2).T  # (N,F): row n of bin f untouched
        # identity rows of an updated W can only coincide with the old row if kept
        assert np.array_equal(got_kept, ~mk), thr
    # exactly singular
    Uz = np.zeros((M, 3, M, M), dtype=np.complex128)
    st = eng.new_status(1)
    eng.ip_update(dev_c(eng, Uz[None]), dev_c(eng, Wb[None, :3]), 1e12, st)
    assert int(st.item()) & _lib.STATUS_SINGULAR


@pytest.mark.parametrize("M,K,domain", [(2, 2, 2), (4, 4, 2), (3, 5, 2), (4, 10, 2), (2, 4, 1), (3, 3, 1.5)])
def test_ilrma_source_update(eng, M, K, domain):
    F, T = 19, 150
    X, W = mixture(M, F, T, 20 + M), rand_filters(M, F, 21)
    rng = np.random.default_rng(22)
    Tb, V = rng.random((M, F, K)), rng.random((M, K, T))
    Td, Vd = dev_r(eng, Tb[None]), dev_r(eng, V[None])
    eng.ilrma_source_update(dev_c(eng, X[None]), dev_c(eng, W[None]), Td, Vd, domain=domain)
    T1, V1 = orc.ilrma_source_update(np.abs(orc.separate(X, W)) ** 2, Tb, V, domain)
    assert rel_err(host(Td)[0], T1) < tol(eng, 1e-11, 5e-5)
    assert rel_err(host(Vd)[0], V1) < tol(eng, 1e-11, 5e-5)


@pytest.mark.parametrize("M,F,T,K", [(4, 1025, 660, 10), (2, 1025, 657, 10), (3, 513, 330, 12)])
def test_xfed_source_model_stays_inside_its_workspace(eng, M, F, T, K):
    """Round 4's advisor finding, on the device: at F = 1025, T = 657..664, n_basis = 10 twelve workgroups of the X-fed basis
    half meet one block while the workspace held eleven slabs -- the twelfth was written past the slab area, for some T
    past the end of the buffer.  The call runs on a workspace of exactly assx_workspace_bytes followed by a guard band:
    the band must come back untouched and the update must equal the oracle's."""
    from audio_source_separation_amd._device import ptr, stream_ptr
    X, W = mixture(M, F, T, 700 + M), rand_filters(M, F, 701)
    rng = np.random.default_rng(702)
    Tb, V = rng.random((M, F, K)) + 0.05, rng.random((M, K, T)) + 0.05
    Xd, Wd, Td, Vd = dev_c(eng, X[None]), dev_c(eng, W[None]), dev_r(eng, Tb[None]), dev_r(eng, V[None])
    n = int(eng._L.assx_workspace_bytes(1, M, F, T, K, eng.prec.code))
    guard = 1 << 22
    buf = torch.full((n + guard,), 0x5A, dtype=torch.uint8, device=eng.dev)
    rc = eng._L.assx_ilrma_source_update(eng.ctx, ptr(Xd), ptr(Wd), ptr(Td), ptr(Vd), 2.0, 1e-12, (1 << M) - 1, ptr(None),
                                         ptr(buf), 1, M, F, T, K, eng.prec.code, stream_ptr(eng.dev))
    assert rc == 0
    torch.cuda.synchronize()
    assert bool((buf[n:] == 0x5A).all()), "the source update wrote past assx_workspace_bytes"
    T1, V1 = orc.ilrma_source_update(np.abs(orc.separate(X, W)) ** 2, Tb, V, 2)
    assert rel_err(host(Td)[0], T1) < tol(eng, 1e-11, 5e-5)
    assert rel_err(host(Vd)[0], V1) < tol(eng, 1e-11, 5e-5)


@pytest.mark.parametrize("budget", [16, 150, 768, 4000])
@pytest.mark.parametrize("M,F,T,K", [(4, 70, 700, 10), (2, 33, 2100, 16), (3, 300, 130, 7)])
def test_xfed_source_model_partition_forms(eng, M, F, T, K, budget, monkeypatch):
    """The X-fed halves under every form of the work partition (block-aligned with 1 ... 16 workgroups per block, the flat
    fallback, more blocks than budget), against the oracle; two utterances in one call == one at a time, bit for bit."""
    monkeypatch.setenv("ASSX_NMF_XFED_WGS", str(budget))  # read on every call
    rng = np.random.default_rng(800 + budget + K)
    Xs = np.stack([mixture(M, F, T, 801 + M), mixture(M, F, T, 802 + M)])
    W = np.stack([rand_filters(M, F, 803), rand_filters(M, F, 804)])
    Tb, V = rng.random((2, M, F, K)) + 0.05, rng.random((2, M, K, T)) + 0.05
    Xb, Wb, Td, Vd = dev_c(eng, Xs), dev_c(eng, W), dev_r(eng, Tb), dev_r(eng, V)
    eng.ilrma_source_update(Xb, Wb, Td, Vd)
    for b in range(2):
        T1, V1 = orc.ilrma_source_update(np.abs(orc.separate(Xs[b], W[b])) ** 2, Tb[b], V[b], 2)
        assert rel_err(host(Td)[b], T1) < tol(eng, 1e-11, 5e-5)
        assert rel_err(host(Vd)[b], V1) < tol(eng, 1e-11, 5e-5)
        t1, v1 = dev_r(eng, Tb[b:b + 1]), dev_r(eng, V[b:b + 1])
        eng.ilrma_source_update(Xb[b:b + 1], Wb[b:b + 1], t1, v1)
        assert torch.equal(t1[0], Td[b]) and torch.equal(v1[0], Vd[b])


def test_ticket_kernels_on_two_streams_of_one_context(eng):
    """The "last workgroup done" tickets are device words owned by the context -- one buffer PER STREAM (round 4's advisor:
    with one shared buffer two NMF updates issued from one thread on two torch streams counted on the same words: a
    partial sum applied, counters left non-zero for every later call).  Two different problems alternate on two streams
    without any synchronisation between them; each must equal its own serial run bit for bit, and so must a run afterwards."""
    F, T, K = 257, 1200, 16
    rng = np.random.default_rng(710)
    probs = []
    for i in range(2):
        X = rng.random((1, F + 16 * i, T)) ** 2 + 1e-3
        probs.append((X, rng.random((1, F + 16 * i, K)) + 0.1, rng.random((1, K, T)) + 0.1))
    from audio_source_separation_amd import _lib
    from audio_source_separation_amd.ops import Engine
    kind = _lib.NMF_IS_MM
    engs = [eng, Engine(dtype=eng.prec.name)]  # a workspace each (as two models have), ONE context: same thread, same device
    assert engs[0].ctx is engs[1].ctx or engs[0].ctx.value == engs[1].ctx.value

    def serial(p, n):
        Xd, Td, Vd = dev_r(eng, p[0]), dev_r(eng, p[1]), dev_r(eng, p[2])
        for _ in range(n):
            eng.nmf_update(kind, Xd, Td, Vd)
        torch.cuda.synchronize()
        return Td.clone(), Vd.clone()

    n_it = 12
    want = [serial(p, n_it) for p in probs]
    streams = [torch.cuda.Stream(device=eng.dev), torch.cuda.Stream(device=eng.dev)]
    state = [(dev_r(eng, p[0]), dev_r(eng, p[1]), dev_r(eng, p[2])) for p in probs]
    torch.cuda.synchronize()
    for _ in range(n_it):
        for e, st, (Xd, Td, Vd) in zip(engs, streams, state):
            with torch.cuda.stream(st):
                e.nmf_update(kind, Xd, Td, Vd)
    torch.cuda.synchronize()
    for (Tw, Vw), (_, Td, Vd) in zip(want, state):
        assert torch.equal(Tw, Td) and torch.equal(Vw, Vd)
    again = serial(probs[0], n_it)
    assert torch.equal(again[0], want[0][0]) and torch.equal(again[1], want[0][1])


def test_ticket_kernels_inside_a_stream_capture_on_a_fresh_stream(eng):
    """Round 5's advisor (medium): torch.cuda.graph captures on a side stream the context has never seen; the first
    ticket kernel there used to hipMalloc the stream's ticket words -- illegal inside a capture.  The words of a new
    stream now come out of a pool the context allocated when it was created: a matrix-core NMF update (two ticket
    kernels) is captured WITHOUT a warm-up on the capture stream, replayed, and must equal the plain launches bit for bit;
    and a thread that ends right after queueing ticket kernels (its context is destroyed by the collector) is harmless."""
    import gc
    import threading
    from audio_source_separation_amd import _lib
    F, T, K = 257, 1200, 16
    rng = np.random.default_rng(711)
    X, Tb, V = rng.random((1, F, T)) ** 2 + 1e-3, rng.random((1, F, K)) + 0.1, rng.random((1, K, T)) + 0.1
    kind = _lib.NMF_IS_MM
    Xd, Td, Vd = dev_r(eng, X), dev_r(eng, Tb), dev_r(eng, V)
    for _ in range(3):
        eng.nmf_update(kind, Xd, Td, Vd)
    torch.cuda.synchronize()
    want = (Td.clone(), Vd.clone())
    Td.copy_(dev_r(eng, Tb)); Vd.copy_(dev_r(eng, V))
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=eng.dev)  # never used with this context before
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        eng.nmf_update(kind, Xd, Td, Vd)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(Td, want[0]) and torch.equal(Vd, want[1])

    # a worker thread queues updates on its own context and ends at once
    res = {}

    def worker():
        from audio_source_separation_amd.ops import Engine
        e2 = Engine(dtype=eng.prec.name)
        t2, v2 = dev_r(eng, Tb), dev_r(eng, V)
        for _ in range(3):
            e2.nmf_update(kind, Xd, t2, v2)
        res["t"], res["v"] = t2, v2  # no synchronisation: the launches may still be running when the thread ends

    th = threading.Thread(target=worker)
    th.start()
    th.join()
    gc.collect()  # drops the thread's context: assx_ctx_destroy waits for its device before it frees the ticket words
    torch.cuda.synchronize()
    assert torch.equal(res["t"], want[0]) and torch.equal(res["v"], want[1])


@pytest.mark.parametrize("M,K,G", [(4, 10, 0), (4, 10, 3), (2, 5, 2), (3, 8, 5), (4, 12, 1), (4, 16, 7), (3, 13, 4), (2, 7, 0)])
def test_source_model_wide_basis(eng, M, K, G):
    """The n_basis 5..16 source model (demixed-power map + matrix-core NMF halves) without a loss request, ragged T, zero
    entries in the model (floors), with the flat partitions forced down (ASSX_G; 0 = default) as the other wide-basis
    tests do; two utterances in one call == one at a time, bit for bit.  (Written for the streaming one-wave-per-source
    source model of round 3, which passed it and was not kept: profiles/r03_src_nmf_experiment.txt.)"""
    import os
    F, T = 11, 461
    rng = np.random.default_rng(500 + 10 * M + K)
    Xs = np.stack([mixture(M, F, T, 501 + M), mixture(M, F, T, 502 + M)])
    W = np.stack([rand_filters(M, F, 503), rand_filters(M, F, 504)])
    Tb, V = rng.random((2, M, F, K)) + 0.02, rng.random((2, M, K, T)) + 0.02
    Tb[0, 0, 2, :] = 0.0
    V[1, M - 1, :, 70:75] = 0.0  # variance 0 -> floored at eps
    if G:
        os.environ["ASSX_G"] = str(G)
    try:
        Xb, Wb = dev_c(eng, Xs), dev_c(eng, W)
        Td, Vd = dev_r(eng, Tb), dev_r(eng, V)
        eng.ilrma_source_update(Xb, Wb, Td, Vd)
        for b in range(2):
            T1, V1 = orc.ilrma_source_update(np.abs(orc.separate(Xs[b], W[b])) ** 2, Tb[b], V[b], 2)
            assert rel_err(host(Td)[b], T1) < tol(eng, 1e-11, 5e-5)
            assert rel_err(host(Vd)[b], V1) < tol(eng, 1e-11, 5e-5)
            t1, v1 = dev_r(eng, Tb[b:b + 1]), dev_r(eng, V[b:b + 1])
            eng.ilrma_source_update(Xb[b:b + 1], Wb[b:b + 1], t1, v1)
            assert torch.equal(t1[0], Td[b]) and torch.equal(v1[0], Vd[b])
    finally:
        os.environ.pop("ASSX_G", None)


@pytest.mark.parametrize("M,K,domain", [(2, 2, 2), (4, 4, 2), (3, 5, 2), (4, 2, 1)])
def test_ilrma_spatial_update(eng, M, K, domain):
    F, T = 19, 330
    X, W = mixture(M, F, T, 30 + M), rand_filters(M, F, 31)
    rng = np.random.default_rng(32)
    Tb, V = rng.random((M, F, K)) + 0.05, rng.random((M, K, T)) + 0.05
    Wd = dev_c(eng, W[None])
    Ud = eng.empty((1, M, F, M, M), complex_=True)
    st = eng.new_status(1)
    eng.ilrma_spatial_update(dev_c(eng, X[None]), Wd, dev_r(eng, Tb[None]), dev_r(eng, V[None]), domain=domain,
                             status=st, U_out=Ud)
    Wref, Uref, mask = orc.ilrma_spatial_update_ip(X, W.copy(), Tb, V, domain)
    assert mask.all() and int(st.item()) == 0
    assert rel_err(host(Ud)[0], Uref) < tol(eng, 1e-12, 2e-5)
    assert rel_err(host(Wd)[0], Wref) < tol(eng, 1e-9, 2e-3)


def test_ilrma_stage_fixture(eng):
    """The reference's own stage outputs (tests/golden/ilrma_stages.npz)."""
    g = load_golden("ilrma_stages")
    X, W0, T0, V0 = g["X"], g["W0"], g["T0"], g["V0"]
    Xd, Wd, Td, Vd = dev_c(eng, X[None]), dev_c(eng, W0[None]), dev_r(eng, T0[None]), dev_r(eng, V0[None])
    l0 = eng.ilrma_loss(Xd, Wd, Td, Vd).item()
    np.testing.assert_allclose(l0, g["loss0"], rtol=tol(eng, 1e-12, 1e-5))
    eng.ilrma_source_update(Xd, Wd, Td, Vd)
    assert rel_err(host(Td)[0], g["T1"]) < tol(eng, 1e-11, 5e-5)
    assert rel_err(host(Vd)[0], g["V1"]) < tol(eng, 1e-11, 5e-5)
    Ud = eng.empty((1,) + g["U"].shape, complex_=True)
    eng.ilrma_spatial_update(Xd, Wd, Td, Vd, U_out=Ud)
    assert rel_err(host(Ud)[0], g["U"]) < tol(eng, 1e-11, 1e-4)
    assert rel_err(host(Wd)[0], g["W1"]) < tol(eng, 1e-9, 2e-3)
    l1 = eng.ilrma_loss(Xd, Wd, Td, Vd).item()
    np.testing.assert_allclose(l1, g["loss1"], rtol=tol(eng, 1e-10, 1e-4))


@pytest.mark.parametrize("M,F,T", SHAPES)
def test_power_and_normalize(eng, M, F, T):
    X, W = mixture(M, F, T, 40), rand_filters(M, F, 41)
    Xd, Wd = dev_c(eng, X[None]), dev_c(eng, W[None])
    p = host(eng.demix_power(Xd, Wd))[0]
    pref = (np.abs(orc.separate(X, W)) ** 2).mean(axis=(1, 2))
    np.testing.assert_allclose(p, pref, rtol=tol(eng, 1e-12, 2e-5))
    C = eng.cov_accumulate(Xd)
    p2 = host(eng.power_from_cov(C.reshape(1, F, M, M), Wd, T))[0]
    np.testing.assert_allclose(p2, pref, rtol=tol(eng, 1e-12, 2e-5))
    K = 3
    Tb = np.random.default_rng(42).random((M, F, K))
    for domain in (2, 1, 1.5):
        Wn, Tn = dev_c(eng, W[None]), dev_r(eng, Tb[None])
        eng.ilrma_normalize_power(Wn, Tn, dev_r(eng, pref[None]), domain=domain)
        Wr, Tr = orc.ilrma_normalize(X, W, Tb, "power", domain)
        assert rel_err(host(Wn)[0], Wr) < tol(eng, 1e-14, 1e-6)
        assert rel_err(host(Tn)[0], Tr) < tol(eng, 1e-13, 1e-6)


@pytest.mark.parametrize("M,F,T", SHAPES[:3])
def test_projection_back(eng, M, F, T):
    X, W = mixture(M, F, T, 50), rand_filters(M, F, 51)
    Xd, Wd = dev_c(eng, X[None]), dev_c(eng, W[None])
    Y = orc.separate(X, W)
    for ref in (0, M - 1):
        s = host(eng.projection_back_scale(Xd, Wd, ref))[0]
        assert rel_err(s, orc.projection_back(Y, X[ref])) < tol(eng, 1e-10, 2e-3)
    s2 = host(eng.projection_back(dev_c(eng, Y[None]), dev_c(eng, X[0][None])))[0]
    assert rel_err(s2, orc.projection_back(Y, X[0])) < tol(eng, 1e-10, 2e-3)
    # 'projection-back' normalisation
    K = 2
    Tb = np.random.default_rng(52).random((M, F, K))
    sc = orc.projection_back(Y, X[0])
    for domain in (2, 1):
        Wn, Tn = dev_c(eng, W[None]), dev_r(eng, Tb[None])
        eng.ilrma_normalize_pb(Wn, Tn, dev_c(eng, sc[None]), domain=domain)
        Wr, Tr = orc.ilrma_normalize(X, W, Tb, "projection-back", domain)
        assert rel_err(host(Wn)[0], Wr) < tol(eng, 1e-13, 1e-6)
        assert rel_err(host(Tn)[0], Tr) < tol(eng, 1e-13, 1e-6)


def test_projection_back_golden(eng):
    g = load_golden("projection_back")
    for N in (2, 3, 4):
        s = host(eng.projection_back(dev_c(eng, g["Y_n%d" % N][None]), dev_c(eng, g["ref_n%d" % N][None])))[0]
        assert rel_err(s, g["scale_n%d" % N]) < tol(eng, 1e-11, 1e-3)


@pytest.mark.parametrize("fold", ["0", "1"])
@pytest.mark.parametrize("kind", ["laplace", "gauss"])
@pytest.mark.parametrize("M,F,T", SHAPES)
def test_auxiva_weights_and_loss(eng, kind, M, F, T, fold, monkeypatch):
    """fold = 1: the statistic's finalize, the log-det terms and the loss sum inside the pass behind tickets
    (ASSX_AUX_FOLD, off by default: measured slower than the separate launches, profiles/r04_auxiva_fold.txt)."""
    from audio_source_separation_amd import _lib
    if fold != "0":
        need_lab("ASSX_AUX_FOLD")
    monkeypatch.setenv("ASSX_AUX_FOLD", fold)
    X, W = mixture(M, F, T, 60), rand_filters(M, F, 61)
    code = _lib.IVA_LAPLACE if kind == "laplace" else _lib.IVA_GAUSS
    r, loss = eng.auxiva_weights(dev_c(eng, X[None]), dev_c(eng, W[None]), code, with_loss=True)
    assert rel_err(host(r)[0], orc.auxiva_weights(orc.separate(X, W), kind)) < tol(eng, 1e-12, 2e-5)
    np.testing.assert_allclose(loss.item(), orc.auxiva_loss(X, W, kind), rtol=tol(eng, 1e-11, 1e-4))


@pytest.mark.parametrize("kind", ["laplace", "gauss"])
def test_auxiva_update_once(eng, kind):
    from audio_source_separation_amd import _lib
    M, F, T = 3, 17, 300
    X, W = mixture(M, F, T, 70), rand_filters(M, F, 71)
    code = _lib.IVA_LAPLACE if kind == "laplace" else _lib.IVA_GAUSS
    Xd, Wd = dev_c(eng, X[None]), dev_c(eng, W[None])
    r, _ = eng.auxiva_weights(Xd, Wd, code)
    eng.auxiva_spatial_update(Xd, Wd, r)
    Wref, _, mask = orc.auxiva_update_once_ip(X, W.copy(), orc.separate(X, W), kind)
    assert mask.all()
    assert rel_err(host(Wd)[0], Wref) < tol(eng, 1e-9, 2e-3)


NMF_CASES = ["euc_d2", "euc_d1", "euc_d15", "kl_d2", "kl_d1", "kl_d15", "is_mm_d2", "is_mm_d1", "is_mm_d15",
             "is_me_d2", "is_k32"]


@pytest.mark.parametrize("name", NMF_CASES)
def test_nmf_golden(eng, name):
    from audio_source_separation_amd import _lib
    g = load_golden("nmf_" + name)
    kind = {"EUC": _lib.NMF_EUC, "KL": _lib.NMF_KL, "IS": _lib.NMF_IS_MM}[str(g["kind"])]
    if str(g["algorithm"]) == "me":
        kind = _lib.NMF_IS_ME
    domain = float(g["domain"])
    Xd = dev_r(eng, g["X"][None])
    Td, Vd = dev_r(eng, g["T0"][None]), dev_r(eng, g["V0"][None])
    losses = []
    done = 0
    for k in g["iters"]:
        while done < int(k):
            eng.nmf_update(kind, Xd, Td, Vd, domain=domain)
            losses.append(eng.nmf_loss(kind, Xd, Td, Vd, domain=domain).item())
            done += 1
        scale = 1 if k <= 5 else 10
        assert rel_err(host(Td)[0], g["T_%d" % k]) < scale * tol(eng, 1e-11, 1e-4), k
        assert rel_err(host(Vd)[0], g["V_%d" % k]) < scale * tol(eng, 1e-11, 1e-4), k
    np.testing.assert_allclose(losses, g["loss_%d" % int(g["iters"][-1])], rtol=tol(eng, 1e-10, 2e-4))


@pytest.mark.parametrize("K", [8, 32])
def test_nmf_loss_edges_eps_zero_and_extreme_ratios(K):
    """Round 5's advisor (low): with eps = 0 and a zero row of the basis the model is 0 in a whole bin; the reference divides by
    it -- KL: +inf, IS: inf - log(inf) = nan -- and the domain-2 criterion kernels must say the same (their reciprocal + Newton
    step used to turn the inf into nan for KL).  And ratios of 1e+-80 in four consecutive frames must not overflow the
    mantissa product of the IS form (values that the per-element logarithm of the reference handles)."""
    import warnings
    from audio_source_separation_amd import _lib
    from audio_source_separation_amd.ops import Engine
    eng = Engine("float64")
    F, T = 37, 80
    rng = np.random.default_rng(97)
    X, Tb, V = rng.random((F, T)) + 0.1, rng.random((F, K)) + 0.1, rng.random((K, T)) + 0.1
    Tz = Tb.copy()
    Tz[5, :] = 0.0
    for kind, code in (("KL", _lib.NMF_KL), ("IS", _lib.NMF_IS_MM)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with np.errstate(all="ignore"):
                want = orc.nmf_loss(kind, X, Tz, V, eps=0.0)
        got = eng.nmf_loss(code, dev_r(eng, X[None]), dev_r(eng, Tz[None]), dev_r(eng, V[None]), eps=0.0).item()
        assert (np.isnan(want) and np.isnan(got)) or want == got, (kind, want, got)
        assert np.isposinf(want) or np.isnan(want)
    # extreme but finite ratios: the model 1e80 times too small in one bin, 1e80 times too large in another
    Te = Tb.copy()
    Te[3, :] *= 1e-80
    Te[9, :] *= 1e80
    tiny = 1e-300  # a floor that never bites: the ratios really are 1e+-80
    want = orc.nmf_loss("IS", X, Te, V, eps=tiny)
    got = eng.nmf_loss(_lib.NMF_IS_MM, dev_r(eng, X[None]), dev_r(eng, Te[None]), dev_r(eng, V[None]), eps=tiny).item()
    assert np.isfinite(want) and np.isfinite(got) and want > 1e80
    np.testing.assert_allclose(got, want, rtol=1e-10)
    # only the logarithms (the part that travels as a mantissa product): a model 1e80 too LARGE in every bin
    want = orc.nmf_loss("IS", X, Tb * 1e80, V, eps=tiny)
    got = eng.nmf_loss(_lib.NMF_IS_MM, dev_r(eng, X[None]), dev_r(eng, (Tb * 1e80)[None]), dev_r(eng, V[None]), eps=tiny).item()
    np.testing.assert_allclose(got, want, rtol=1e-10)
    # the same criterion riding on the next update's basis half (assx_nmf_iterate: the loss of update i inside update i + 1)
    Td, Vd = dev_r(eng, Te[None]), dev_r(eng, V[None])
    loss = torch.zeros((2, 1), dtype=torch.float64, device=eng.dev)
    eng.nmf_iterate(2, _lib.NMF_IS_MM, dev_r(eng, X[None]), Td, Vd, eps=tiny, loss=loss)
    To, Vo = orc.nmf_update_once("IS", X, Te, V, eps=tiny)
    np.testing.assert_allclose(loss[0, 0].item(), orc.nmf_loss("IS", X, To, Vo, eps=tiny), rtol=1e-9)  # fused into update 2's basis half
    To, Vo = orc.nmf_update_once("IS", X, To, Vo, eps=tiny)
    np.testing.assert_allclose(loss[1, 0].item(), orc.nmf_loss("IS", X, To, Vo, eps=tiny), rtol=1e-9)  # the stand-alone pass


XNMF_CASES = ["t_nu1", "t_nu1000", "t_k20", "cauchy_naive", "cauchy_mm", "cauchy_me", "cauchy_mm_fast", "cauchy_mm_k20"]


@pytest.mark.parametrize("name", XNMF_CASES)
def test_tnmf_cauchy_nmf_golden(eng, name):
    """tNMF / CauchyNMF (nmf.py:358-600) through assx_nmf_update_ex / assx_nmf_loss_ex vs the reference's outputs."""
    from audio_source_separation_amd import _lib
    g = load_golden("xnmf_" + name)
    if str(g["kind"]) == "t":
        kind, param = _lib.NMF_T, float(g["nu"])
    else:
        kind = {"naive-multipricative": _lib.NMF_CAUCHY_NAIVE, "mm": _lib.NMF_CAUCHY_MM, "me": _lib.NMF_CAUCHY_ME,
                "mm_fast": _lib.NMF_CAUCHY_MM_FAST}[str(g["algorithm"])]
        param = 0.0
    Xd = dev_r(eng, g["X"][None])
    Td, Vd = dev_r(eng, g["T0"][None]), dev_r(eng, g["V0"][None])
    losses, done = [], 0
    for k in g["iters"]:
        while done < int(k):
            eng.nmf_update(kind, Xd, Td, Vd, param=param)
            losses.append(eng.nmf_loss(kind, Xd, Td, Vd, param=param).item())
            done += 1
        scale = 1 if k <= 5 else 10
        assert rel_err(host(Td)[0], g["T_%d" % k]) < scale * tol(eng, 1e-11, 1e-4), k
        assert rel_err(host(Vd)[0], g["V_%d" % k]) < scale * tol(eng, 1e-11, 1e-4), k
    np.testing.assert_allclose(losses, g["loss_%d" % int(g["iters"][-1])], rtol=tol(eng, 1e-10, 2e-4))


@pytest.mark.parametrize("M,F,T", SHAPES[:3])
def test_iss_update(eng, M, F, T):
    """ISS on the covariances (assx_iss_update) == the reference's Y-based sweep (oracle.iss_update)."""
    X, W = mixture(M, F, T, 80), rand_filters(M, F, 81)
    R = np.random.default_rng(82).random((M, F, T)) + 0.05
    U = orc.weighted_covariance(X, R)
    Wd = dev_c(eng, W[None])
    eng.iss_update(dev_c(eng, U[None]), Wd, T)
    Y = orc.iss_update(orc.separate(X, W), R)
    assert rel_err(orc.separate(X, host(Wd)[0]), Y) < tol(eng, 1e-10, 2e-3)


@pytest.mark.parametrize("M,F,T", SHAPES[:3])
def test_ip2_update(eng, M, F, T):
    """Pairwise update (assx_ip2_update) vs the oracle restatement (np.linalg.eig, argsort, parallel_sort)."""
    X, W = mixture(M, F, T, 90), rand_filters(M, F, 91)
    R = np.random.default_rng(92).random((M, T)) + 0.05
    U = orc.weighted_covariance(X, R)
    for pair in ((0, 1), (M - 1, 0)):
        Wd = dev_c(eng, W[None])
        st = eng.new_status(1)
        eng.ip2_update(dev_c(eng, U[None]), Wd, pair, 1e12, st)
        Wref, cm, cn = orc.ip2_update(W.copy(), U[pair[0]], U[pair[1]], pair[0], pair[1])
        assert cm.all() and cn.all() and int(st.item()) == 0
        got = host(Wd)[0]
        # rows other than the pair are untouched, bit for bit
        others = [n for n in range(M) if n not in pair]
        stored = host(dev_c(eng, W[None]))[0]  # W as the kernel saw it (rounded to the storage dtype)
        assert np.array_equal(got[:, others], stored[:, others])
        assert rel_err(got, Wref) < tol(eng, 1e-9, 5e-3)


@pytest.mark.parametrize("M,K", [(2, 3), (3, 4), (4, 4), (4, 7), (3, 10), (2, 5), (4, 20)])
def test_partitioned_source_update(eng, M, K):
    """Z, T, V updates of the partitioning branch (ilrma.py:368-408) vs the oracle; B = 2 utterances."""
    F, T = 21, 150
    rng = np.random.default_rng(70 + M + K)
    Xs = [mixture(M, F, T, 71 + b) for b in range(2)]
    Ws = [rand_filters(M, F, 73 + b) for b in range(2)]
    Z = rng.random((2, M, K)) + 0.1
    Z /= Z.sum(axis=1, keepdims=True)
    Tb, V = rng.random((2, F, K)) + 0.05, rng.random((2, K, T)) + 0.05
    Zd, Td, Vd = dev_r(eng, Z), dev_r(eng, Tb), dev_r(eng, V)
    Teff, Veff = eng.empty((2, M, F, K)), eng.empty((2, M, K, T))
    eng.ilrma_source_update_partitioned(dev_c(eng, np.stack(Xs)), dev_c(eng, np.stack(Ws)), Zd, Td, Vd, Teff, Veff)
    for b in range(2):
        Z1, T1, V1 = orc.part_source_update(np.abs(orc.separate(Xs[b], Ws[b])) ** 2, Z[b], Tb[b], V[b])
        assert rel_err(host(Zd)[b], Z1) < tol(eng, 1e-11, 5e-5)
        assert rel_err(host(Td)[b], T1) < tol(eng, 1e-11, 5e-5)
        assert rel_err(host(Vd)[b], V1) < tol(eng, 1e-11, 5e-5)
        # the expansion left behind describes the updated model
        assert rel_err(host(Teff)[b], Z1[:, None, :] * T1[None]) < tol(eng, 1e-11, 5e-5)
        assert rel_err(host(Veff)[b], np.broadcast_to(V1, (M,) + V1.shape)) < tol(eng, 1e-11, 5e-5)


def test_partitioned_expand_and_normalize(eng):
    M, F, K, T = 3, 11, 5, 20
    rng = np.random.default_rng(80)
    Z, Tb, V = rng.random((2, M, K)), rng.random((2, F, K)), rng.random((2, K, T))
    W = rng.standard_normal((2, F, M, M)) + 1j * rng.standard_normal((2, F, M, M))
    pb = rng.random((2, M, F)) + 0.1
    Teff, Veff = eng.empty((2, M, F, K)), eng.empty((2, M, K, T))
    eng.ilrma_expand_partitioned(dev_r(eng, Z), dev_r(eng, Tb), dev_r(eng, V), Teff, Veff)
    sr = 0 if eng.prec.name == "float64" else 1e-6
    assert rel_err(host(Teff), Z[:, :, None, :] * Tb[:, None]) <= sr
    assert rel_err(host(Veff), np.broadcast_to(V[:, None], (2, M, K, T))) <= sr
    Wd, Zd, Td = dev_c(eng, W), dev_r(eng, Z), dev_r(eng, Tb)
    eng.ilrma_normalize_power_bins_partitioned(Wd, Zd, Td, torch.from_numpy(pb).to(eng.dev), T)
    aux = np.sqrt(pb.mean(axis=2))  # (B, N)
    Zaux = Z / aux[:, :, None] ** 2
    Zs = Zaux.sum(axis=1, keepdims=True)
    assert rel_err(host(Zd), Zaux / Zs) < tol(eng, 1e-13, 1e-5)
    assert rel_err(host(Td), Tb * Zs) < tol(eng, 1e-13, 1e-5)
    assert rel_err(host(Wd), W / aux[:, None, :, None]) < tol(eng, 1e-13, 1e-5)


@pytest.mark.parametrize("M,K,nu", [(2, 2, 1.0), (3, 4, 5.0), (4, 4, 100.0), (4, 7, 2.5), (3, 10, 5.0), (4, 18, 1.0)])
def test_tilrma_stages(eng, M, K, nu):
    """t-ILRMA source model, spatial model and loss, one stage at a time, vs the oracle (ilrma.py:880-1018)."""
    F, T = 19, 150
    X, W = mixture(M, F, T, 90 + M), rand_filters(M, F, 91)
    rng = np.random.default_rng(92)
    Tb, V = rng.random((M, F, K)) + 0.05, rng.random((M, K, T)) + 0.05
    Xd, Wd, Td, Vd = dev_c(eng, X[None]), dev_c(eng, W[None]), dev_r(eng, Tb[None]), dev_r(eng, V[None])
    got = float(eng.tilrma_loss(Xd, Wd, Td, Vd, nu).item())
    np.testing.assert_allclose(got, orc.tilrma_loss(X, W, Tb, V, nu), rtol=tol(eng, 1e-12, 1e-5))
    eng.tilrma_source_update(Xd, Wd, Td, Vd, nu)
    T1, V1 = orc.tilrma_source_update(np.abs(orc.separate(X, W)) ** 2, Tb, V, nu)
    assert rel_err(host(Td)[0], T1) < tol(eng, 1e-11, 5e-5)
    assert rel_err(host(Vd)[0], V1) < tol(eng, 1e-11, 5e-5)
    Xi = eng.empty((1, M, F, T))
    st = eng.new_status(1)
    Td, Vd = dev_r(eng, T1[None]), dev_r(eng, V1[None])
    eng.tilrma_spatial_update(Xd, Wd, Td, Vd, nu, Xi, status=st)
    W1, _ = orc.tilrma_spatial_update(X, W, T1, V1, nu)
    assert int(st.item()) == 0
    assert rel_err(host(Wd)[0], W1) < tol(eng, 1e-9, 2e-3)


@pytest.fixture
def few_workgroups():
    """Force the flat partitions down to a handful of workgroups (ASSX_G), so that a small input gives every workgroup
    a long range: several trips of the steady-state loop, the first-trip and drain code, ranges that start and end in
    the middle of a bin and flush several partial records -- the paths a full-size utterance exercises."""
    def _set(g):
        os.environ["ASSX_G"] = str(g)
    yield _set
    os.environ.pop("ASSX_G", None)


@pytest.mark.parametrize("G", [5, 16])
@pytest.mark.parametrize("M,K,T", [(4, 4, 1030), (3, 2, 1500), (2, 3, 1024), (4, 6, 700)])
def test_streaming_kernels_long_ranges(eng, few_workgroups, G, M, K, T):
    few_workgroups(G)
    F = 19
    X, W = mixture(M, F, T, 100 + M), rand_filters(M, F, 101)
    rng = np.random.default_rng(102 + K)
    Tb, V = rng.random((M, F, K)) + 0.05, rng.random((M, K, T)) + 0.05
    Xd, Wd = dev_c(eng, X[None]), dev_c(eng, W[None])
    # loss
    got = float(eng.ilrma_loss(Xd, Wd, dev_r(eng, Tb[None]), dev_r(eng, V[None])).item())
    ref = orc.ilrma_loss(X, W, Tb, V, 2)
    np.testing.assert_allclose(got, ref, rtol=tol(eng, 1e-12, 1e-5))
    # source model
    Td, Vd = dev_r(eng, Tb[None]), dev_r(eng, V[None])
    eng.ilrma_source_update(Xd, Wd, Td, Vd)
    T1, V1 = orc.ilrma_source_update(np.abs(orc.separate(X, W)) ** 2, Tb, V, 2)
    assert rel_err(host(Td)[0], T1) < tol(eng, 1e-11, 5e-5)
    assert rel_err(host(Vd)[0], V1) < tol(eng, 1e-11, 5e-5)
    # spatial model
    Ud = eng.empty((1, M, F, M, M), complex_=True)
    st = eng.new_status(1)
    Wd2 = dev_c(eng, W[None])
    eng.ilrma_spatial_update(Xd, Wd2, dev_r(eng, T1[None]), dev_r(eng, V1[None]), status=st, U_out=Ud)
    Wref, Uref, mask = orc.ilrma_spatial_update_ip(X, W.copy(), T1, V1, 2)
    assert mask.all() and int(st.item()) == 0
    assert rel_err(host(Ud)[0], Uref) < tol(eng, 1e-12, 2e-5)
    assert rel_err(host(Wd2)[0], Wref) < tol(eng, 1e-9, 2e-3)
    # AuxIVA-style weights given per (n, t) and plain covariance take the same kernel with other weight kinds
    r = rng.random((M, T)) + 0.1
    U2 = eng.cov_accumulate(Xd, dev_r(eng, r[None]))
    assert rel_err(host(U2)[0], orc.weighted_covariance(X, r)) < tol(eng, 1e-12, 2e-5)


@pytest.mark.parametrize("G", [3, 16])
def test_streaming_kernels_long_ranges_two_utterances(eng, few_workgroups, G):
    """Ranges that cross an utterance boundary (descriptor rebasing) with a ragged frame count."""
    few_workgroups(G)
    M, K, F, T = 4, 4, 11, 700
    rng = np.random.default_rng(120)
    Xs = [mixture(M, F, T, 121 + b) for b in range(2)]
    Ws = [rand_filters(M, F, 123 + b) for b in range(2)]
    Tb, V = rng.random((2, M, F, K)) + 0.05, rng.random((2, M, K, T)) + 0.05
    Xd, Wd = dev_c(eng, np.stack(Xs)), dev_c(eng, np.stack(Ws))
    Td, Vd = dev_r(eng, Tb), dev_r(eng, V)
    loss = host(eng.ilrma_loss(Xd, Wd, Td, Vd))
    eng.ilrma_source_update(Xd, Wd, Td, Vd)
    st = eng.new_status(2)
    Wd2 = Wd.clone()
    eng.ilrma_spatial_update(Xd, Wd2, Td, Vd, status=st)
    for b in range(2):
        np.testing.assert_allclose(loss[b], orc.ilrma_loss(Xs[b], Ws[b], Tb[b], V[b], 2), rtol=tol(eng, 1e-12, 1e-5))
        T1, V1 = orc.ilrma_source_update(np.abs(orc.separate(Xs[b], Ws[b])) ** 2, Tb[b], V[b], 2)
        assert rel_err(host(Td)[b], T1) < tol(eng, 1e-11, 5e-5)
        assert rel_err(host(Vd)[b], V1) < tol(eng, 1e-11, 5e-5)
        Wref, _, _ = orc.ilrma_spatial_update_ip(Xs[b], Ws[b].copy(), T1, V1, 2)
        assert rel_err(host(Wd2)[b], Wref) < tol(eng, 1e-9, 2e-3)


@pytest.mark.parametrize("G", [0, 5])
@pytest.mark.parametrize("M,K,domain,T", [(4, 4, 2, 1030), (2, 3, 2, 700), (3, 6, 2, 300), (4, 2, 1, 300)])
def test_source_update_loss_of_entry_state(eng, few_workgroups, G, M, K, domain, T):
    """`loss_prev` of assx_ilrma_source_update = compute_negative_loglikelihood of the model at entry (ilrma.py:648-677):
    fused into the basis pass for domain 2 / K <= 4, a pass of its own otherwise; two utterances, ragged T."""
    if G:
        few_workgroups(G)
    F = 13
    rng = np.random.default_rng(140 + M + K)
    Xs = [mixture(M, F, T, 141 + b) for b in range(2)]
    Ws = [rand_filters(M, F, 143 + b) for b in range(2)]
    Tb, V = rng.random((2, M, F, K)) + 0.05, rng.random((2, M, K, T)) + 0.05
    Xd, Wd, Td, Vd = dev_c(eng, np.stack(Xs)), dev_c(eng, np.stack(Ws)), dev_r(eng, Tb), dev_r(eng, V)
    lp = eng.empty((2,), dtype=torch.float64)
    eng.ilrma_source_update(Xd, Wd, Td, Vd, domain=domain, loss_prev=lp)
    for b in range(2):
        np.testing.assert_allclose(host(lp)[b], orc.ilrma_loss(Xs[b], Ws[b], Tb[b], V[b], domain),
                                   rtol=tol(eng, 1e-12, 1e-5))
        T1, V1 = orc.ilrma_source_update(np.abs(orc.separate(Xs[b], Ws[b])) ** 2, Tb[b], V[b], domain)
        assert rel_err(host(Td)[b], T1) < tol(eng, 1e-11, 5e-5)
        assert rel_err(host(Vd)[b], V1) < tol(eng, 1e-11, 5e-5)


@pytest.mark.parametrize("F,T", [(65, 2000), (65, 3800), (256, 600)])
def test_fused_loss_on_aligned_partitions_with_short_groups(eng, F, T):
    """Default (un-shrunk) partitions whose workgroup budget is a whole number of parts per bin, but where the parts
    round up to fewer non-empty ones (F = 65, T = 2000: 31 parts of 2 frame blocks over 32 blocks = 16 used): every
    loss partial the finish kernel sums must have been written.  The scratch is poisoned with NaN first."""
    M, K = 2, 2
    rng = np.random.default_rng(190)
    X, W = mixture(M, F, T, 191), rand_filters(M, F, 192)
    Tb, V = rng.random((M, F, K)) + 0.05, rng.random((M, K, T)) + 0.05
    Xd, Wd, Td, Vd = dev_c(eng, X[None]), dev_c(eng, W[None]), dev_r(eng, Tb[None]), dev_r(eng, V[None])
    ref = host(eng.ilrma_loss(Xd, Wd, Td, Vd))[0]
    eng._scratch(1, M, F, T, K).view(torch.uint8).fill_(0xFF)  # all-ones bytes = NaN in both precisions
    lp = eng.empty((1,), dtype=torch.float64)
    eng.ilrma_source_update(Xd, Wd, Td, Vd, loss_prev=lp)
    np.testing.assert_allclose(host(lp)[0], ref, rtol=tol(eng, 1e-13, 1e-6))
    np.testing.assert_allclose(ref, orc.ilrma_loss(X, W, Tb, V, 2), rtol=tol(eng, 1e-12, 1e-5))


@pytest.mark.parametrize("G", [0, 2, 5])
@pytest.mark.parametrize("M,K,domain,F,T", [(4, 10, 2, 19, 150), (2, 5, 2, 8, 64), (3, 17, 1, 21, 333), (4, 70, 2, 9, 130),
                                            (2, 6, 1.5, 17, 257)])
def test_wide_basis_path(eng, few_workgroups, G, M, K, domain, F, T):
    """n_basis > 4 (the reference's default is 10): source model through the materialised demixed power + the batched
    IS-NMF update on the matrix cores, covariance through the materialised source variance, bin-batched loss; two
    utterances, ragged F (bins are taken 8 at a time) and ragged T.  G > 0 squeezes the flat partitions into a few
    workgroups, so that a range crosses bin groups and utterances (record flush, basis-row reload, slot arithmetic)."""
    if G:
        few_workgroups(G)
    rng = np.random.default_rng(150 + M + K)
    Xs = [mixture(M, F, T, 151 + b) for b in range(2)]
    Ws = [rand_filters(M, F, 153 + b) for b in range(2)]
    Tb, V = rng.random((2, M, F, K)) + 0.05, rng.random((2, M, K, T)) + 0.05
    Xd, Td, Vd = dev_c(eng, np.stack(Xs)), dev_r(eng, Tb), dev_r(eng, V)
    # loss
    Wd = dev_c(eng, np.stack(Ws))
    l0 = host(eng.ilrma_loss(Xd, Wd, Td, Vd, domain=domain))
    for b in range(2):
        np.testing.assert_allclose(l0[b], orc.ilrma_loss(Xs[b], Ws[b], Tb[b], V[b], domain), rtol=tol(eng, 1e-12, 1e-5))
    # spatial model
    Ud = eng.empty((2, M, F, M, M), complex_=True)
    st = eng.new_status(2)
    eng.ilrma_spatial_update(Xd, Wd, Td, Vd, domain=domain, status=st, U_out=Ud)
    W1 = host(Wd)
    for b in range(2):
        Wref, Uref, mask = orc.ilrma_spatial_update_ip(Xs[b], Ws[b].copy(), Tb[b], V[b], domain)
        assert mask.all()
        assert rel_err(host(Ud)[b], Uref) < tol(eng, 1e-12, 2e-5)
        assert rel_err(W1[b], Wref) < tol(eng, 1e-9, 2e-3)
    # source model (with the loss of the entry state), from the original filters
    Wd = dev_c(eng, np.stack(Ws))
    lp = eng.empty((2,), dtype=torch.float64)
    eng.ilrma_source_update(Xd, Wd, Td, Vd, domain=domain, loss_prev=lp)
    for b in range(2):
        np.testing.assert_allclose(host(lp)[b], l0[b], rtol=tol(eng, 1e-13, 1e-6))
        T1, V1 = orc.ilrma_source_update(np.abs(orc.separate(Xs[b], Ws[b])) ** 2, Tb[b], V[b], domain)
        assert rel_err(host(Td)[b], T1) < tol(eng, 1e-11, 5e-5)
        assert rel_err(host(Vd)[b], V1) < tol(eng, 1e-11, 5e-5)


@pytest.mark.parametrize("M,K,pair", [(4, 3, (1, 2)), (4, 10, (3, 0)), (3, 6, (0, 2)), (2, 12, (1,))])
def test_source_update_of_selected_sources(eng, M, K, pair):
    """Pairwise update (ilrma.py:432-481): only the selected sources' models move, and they move exactly as in the
    full update; with n_basis > 4 the update runs on copies of the whole model and the selection is copied back.
    The loss of the entry state rides along."""
    F, T = 19, 150
    rng = np.random.default_rng(170 + M + K)
    Xs = [mixture(M, F, T, 171 + b) for b in range(2)]
    Ws = [rand_filters(M, F, 173 + b) for b in range(2)]
    Tb, V = rng.random((2, M, F, K)) + 0.05, rng.random((2, M, K, T)) + 0.05
    Xd, Wd, Td, Vd = dev_c(eng, np.stack(Xs)), dev_c(eng, np.stack(Ws)), dev_r(eng, Tb), dev_r(eng, V)
    lp = eng.empty((2,), dtype=torch.float64)
    eng.ilrma_source_update(Xd, Wd, Td, Vd, sources=pair, loss_prev=lp)
    for b in range(2):
        np.testing.assert_allclose(host(lp)[b], orc.ilrma_loss(Xs[b], Ws[b], Tb[b], V[b], 2), rtol=tol(eng, 1e-12, 1e-5))
        T1, V1 = orc.ilrma_source_update(np.abs(orc.separate(Xs[b], Ws[b])) ** 2, Tb[b], V[b], 2)
        for n in range(M):
            if n in pair:
                assert rel_err(host(Td)[b, n], T1[n]) < tol(eng, 1e-11, 5e-5)
                assert rel_err(host(Vd)[b, n], V1[n]) < tol(eng, 1e-11, 5e-5)
            else:  # untouched, bit for bit
                assert np.array_equal(host(Td)[b, n], host(dev_r(eng, Tb))[b, n])
                assert np.array_equal(host(Vd)[b, n], host(dev_r(eng, V))[b, n])


def test_oversize_utterance_is_rejected(eng):
    """Buffer offsets are 32-bit: an utterance of 2^28 or more complex samples is refused, not mis-addressed."""
    from audio_source_separation_amd import _lib as L
    from audio_source_separation_amd._device import ptr
    t = eng.empty((16,))
    rc = L.lib.assx_ilrma_loss(eng.ctx, ptr(t), ptr(t), ptr(t), ptr(t), 2.0, 1e-12, ptr(t), ptr(t), 1, 4, 1 << 13, 1 << 13, 4,
                               eng.prec.code, eng._st())
    assert rc == -2  # ASSX_E_UNSUPPORTED
    assert b"4 GiB" in L.lib.assx_last_error(eng.ctx)


F4_IDLMA = ["f4_idlma_m2_d2", "f4_idlma_m3_d1", "f4_idlma_m4_d2", "f4_idlma_m4_d15", "f4_idlma_m5_d15",
            "f4_idlma_m6_d2"]
F4_FASTMNMF = ["f4_fastmnmf_m2_n2", "f4_fastmnmf_m3_n2", "f4_fastmnmf_m4_n3", "f4_fastmnmf_m4_n5_part",
               "f4_fastmnmf_m5_n3", "f4_fastmnmf_m6_n4_part"]


@pytest.mark.parametrize("name", F4_IDLMA)
def test_f4_idlma_update_space_model(eng, name):
    """GaussIDLMA.update_space_model (sss/idlma.py:175-210) vs the reference's own output, NumPy and batched tensors."""
    from audio_source_separation_amd.sss.idlma import update_space_model
    g = load_golden(name)
    W1 = update_space_model(g["X"], g["W0"], g["dnn_output"], domain=float(g["domain"]), dtype=eng.prec.name)
    assert W1.dtype == np.complex128 and rel_err(W1, g["W1"]) < tol(eng, 1e-10, 2e-3)
    Xt, Wt, Rt = (torch.from_numpy(np.stack([g[k], g[k]])).cuda() for k in ("X", "W0", "dnn_output"))
    W2 = update_space_model(Xt, Wt, Rt, domain=float(g["domain"]), dtype=eng.prec.name)
    assert isinstance(W2, torch.Tensor) and rel_err(W2[1].cpu().numpy(), g["W1"]) < tol(eng, 1e-10, 2e-3)
    assert torch.equal(W2[0], W2[1])


@pytest.mark.parametrize("name", F4_FASTMNMF)
def test_f4_fastmnmf_update_diagonalizer(eng, name):
    """FastMultichannelISNMF.update_diagonalizer (bss/mnmf.py:848-888) vs the reference's own output."""
    from audio_source_separation_amd.bss.mnmf import update_diagonalizer
    g = load_golden(name)
    Q1 = update_diagonalizer(g["X"], g["Q0"], g["g"], variance=g["variance"], dtype=eng.prec.name)
    assert rel_err(Q1, g["Q1"]) < tol(eng, 1e-10, 2e-3)
    lat = g["latent"] if bool(g["partitioning"]) else None
    Q2 = update_diagonalizer(g["X"], g["Q0"], g["g"], basis=g["basis"], activation=g["activation"], latent=lat,
                             dtype=eng.prec.name)
    assert rel_err(Q2, g["Q1"]) < tol(eng, 1e-10, 2e-3)
    with pytest.raises(ValueError):
        update_diagonalizer(g["X"], g["Q0"], g["g"])
