"""Wide-channel path (5 <= M <= 8, csrc/assx_widem.hip): every C-ABI entry point it serves against the oracle on
seeded inputs, float64 and float32, ragged sizes, batched == single.  The reference is generic in M
(src/bss/ilrma.py:61-62); model-level parity on the reference's own outputs is in test_gpu_models.py
(fixtures ilrma_m5 / m6 / m8, auxiva_*_m5 / m6, iss_*_m5, ip2_*_m5 / m6, part_ilrma_m5 / m6, tilrma_m5 / m6)."""
import numpy as np
import pytest

from conftest import rel_err
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
    S = (rng.standard_normal((M, F, T)) + 1j * rng.standard_normal((M, F, T))) * (0.2 + rng.random((M, 1, T)) ** 2)
    A = rng.standard_normal((F, M, M)) + 1j * rng.standard_normal((F, M, M))
    return np.einsum("fmn,nft->mft", A, S)


def rand_filters(M, F, seed):
    rng = np.random.default_rng(seed)
    return np.eye(M)[None] + 0.2 * (rng.standard_normal((F, M, M)) + 1j * rng.standard_normal((F, M, M)))


SHAPES = [(5, 7, 300), (6, 9, 257), (7, 5, 520), (8, 11, 400)]


@pytest.mark.parametrize("M,F,T", SHAPES)
def test_demix_cov_ip(eng, M, F, T):
    X, W = mixture(M, F, T, 1), rand_filters(M, F, 2)
    Xd = dev_c(eng, X[None])
    Y = eng.demix(Xd, dev_c(eng, W[None]))
    assert rel_err(host(Y)[0], orc.separate(X, W)) < tol(eng, 1e-14, 2e-6)
    rng = np.random.default_rng(3)
    r_nt, r_nft = rng.random((M, T)) + 0.05, rng.random((M, F, T)) + 0.05
    r_nt[0, :3] = 0.0  # eps floor
    for r in (r_nt, r_nft):
        U = eng.cov_accumulate(Xd, dev_r(eng, r[None]))
        assert rel_err(host(U)[0], orc.weighted_covariance(X, r)) < tol(eng, 1e-12, 2e-5)
    C = eng.cov_accumulate(Xd)
    assert rel_err(host(C)[0, 0], orc.weighted_covariance(X, np.ones((1, T)))[0]) < tol(eng, 1e-12, 2e-5)
    Uh = host(U)[0]
    assert np.array_equal(Uh, Uh.conj().transpose(0, 1, 3, 2))  # Hermitian bit-exact
    # IP sweep on the oracle's covariance
    Uo = orc.weighted_covariance(X, r_nft)
    Wd = dev_c(eng, W[None])
    st = eng.new_status(1)
    eng.ip_update(dev_c(eng, Uo[None]), Wd, 1e12, st)
    Wref, mask = orc.ip_update(W.copy(), Uo)
    assert mask.all() and int(st.item()) == 0
    assert rel_err(host(Wd)[0], Wref) < tol(eng, 1e-9, 2e-3)


@pytest.mark.parametrize("M,K,domain", [(5, 3, 2), (6, 10, 1), (8, 4, 2), (7, 2, 1.5)])
def test_ilrma_stages(eng, M, K, domain):
    F, T = 9, 333
    X, W = mixture(M, F, T, 20 + M), rand_filters(M, F, 21)
    rng = np.random.default_rng(22)
    Tb, V = rng.random((M, F, K)) + 0.05, rng.random((M, K, T)) + 0.05
    Xd, Wd = dev_c(eng, X[None]), dev_c(eng, W[None])
    # loss
    got = float(eng.ilrma_loss(Xd, Wd, dev_r(eng, Tb[None]), dev_r(eng, V[None]), domain=domain).item())
    np.testing.assert_allclose(got, orc.ilrma_loss(X, W, Tb, V, domain), rtol=tol(eng, 1e-12, 1e-5))
    # source model, with the loss of the entry state riding along
    Td, Vd = dev_r(eng, Tb[None]), dev_r(eng, V[None])
    lp = eng.empty((1,), dtype=torch.float64)
    eng.ilrma_source_update(Xd, Wd, Td, Vd, domain=domain, loss_prev=lp)
    T1, V1 = orc.ilrma_source_update(np.abs(orc.separate(X, W)) ** 2, Tb, V, domain)
    assert rel_err(host(Td)[0], T1) < tol(eng, 1e-11, 1e-4) and rel_err(host(Vd)[0], V1) < tol(eng, 1e-11, 1e-4)
    np.testing.assert_allclose(lp.item(), orc.ilrma_loss(X, W, Tb, V, domain), rtol=tol(eng, 1e-12, 1e-5))
    # pairwise source update: only the selected sources move
    Td2, Vd2 = dev_r(eng, Tb[None]), dev_r(eng, V[None])
    eng.ilrma_source_update(Xd, Wd, Td2, Vd2, domain=domain, sources=(1, M - 1))
    Tg, Vg = host(Td2)[0], host(Vd2)[0]
    keep = [n for n in range(M) if n not in (1, M - 1)]
    assert np.array_equal(Tg[keep], dev_r(eng, Tb).cpu().numpy().astype(np.float64)[keep])
    assert rel_err(Tg[[1, M - 1]], T1[[1, M - 1]]) < tol(eng, 1e-11, 1e-4)
    assert rel_err(Vg[[1, M - 1]], V1[[1, M - 1]]) < tol(eng, 1e-11, 1e-4)
    # spatial model + the per-bin power statistic of the updated filters
    Ud = eng.empty((1, M, F, M, M), complex_=True)
    C = eng.cov_accumulate(Xd).reshape(1, F, M, M)
    pb = eng.empty((1, M, F), dtype=torch.float64)
    st = eng.new_status(1)
    Wd2 = dev_c(eng, W[None])
    eng.ilrma_spatial_update(Xd, Wd2, dev_r(eng, T1[None]), dev_r(eng, V1[None]), domain=domain, status=st, U_out=Ud,
                             C=C, power_bins=pb)
    Wref, Uref, mask = orc.ilrma_spatial_update_ip(X, W.copy(), T1, V1, domain)
    assert mask.all() and int(st.item()) == 0
    assert rel_err(host(Ud)[0], Uref) < tol(eng, 1e-12, 2e-5)
    assert rel_err(host(Wd2)[0], Wref) < tol(eng, 1e-9, 2e-3)
    P = np.abs(orc.separate(X, Wref)) ** 2
    np.testing.assert_allclose(host(pb)[0], P.mean(axis=2), rtol=tol(eng, 1e-9, 2e-3))
    # power statistics and normalisation
    p_direct = eng.demix_power(Xd, Wd2)
    p_cov = eng.power_from_cov(C, Wd2, T)
    np.testing.assert_allclose(host(p_direct)[0], P.mean(axis=(1, 2)), rtol=tol(eng, 1e-10, 2e-3))
    np.testing.assert_allclose(host(p_cov)[0], P.mean(axis=(1, 2)), rtol=tol(eng, 1e-9, 2e-3))
    # projection back
    sc = eng.projection_back_scale(Xd, Wd2, 1, st)
    assert rel_err(host(sc)[0], orc.projection_back(orc.separate(X, Wref), X[1])) < tol(eng, 1e-9, 5e-3)
    Yd = dev_c(eng, orc.separate(X, Wref)[None])
    sc2 = eng.projection_back(Yd, dev_c(eng, X[1][None]), st)
    assert rel_err(host(sc2)[0], orc.projection_back(orc.separate(X, Wref), X[1])) < tol(eng, 1e-9, 5e-3)


@pytest.mark.parametrize("kind", ["laplace", "gauss"])
@pytest.mark.parametrize("M", [5, 8])
def test_auxiva_stages(eng, kind, M):
    from audio_source_separation_amd import _lib
    F, T = 9, 300
    X, W = mixture(M, F, T, 40 + M), rand_filters(M, F, 41)
    Xd, Wd = dev_c(eng, X[None]), dev_c(eng, W[None])
    code = _lib.IVA_LAPLACE if kind == "laplace" else _lib.IVA_GAUSS
    r, loss = eng.auxiva_weights(Xd, Wd, code, with_loss=True)
    Y = orc.separate(X, W)
    assert rel_err(host(r)[0], orc.auxiva_weights(Y, kind)) < tol(eng, 1e-12, 2e-5)
    np.testing.assert_allclose(loss.item(), orc.auxiva_loss(X, W, kind), rtol=tol(eng, 1e-12, 1e-5))
    st = eng.new_status(1)
    eng.auxiva_spatial_update(Xd, Wd, r, status=st)
    Wref, _, mask = orc.auxiva_update_once_ip(X, W.copy(), Y, kind)
    assert mask.all() and int(st.item()) == 0
    assert rel_err(host(Wd)[0], Wref) < tol(eng, 1e-9, 2e-3)


@pytest.mark.parametrize("M,K,nu", [(5, 3, 5.0), (7, 10, 1.0)])
def test_tilrma_stages(eng, M, K, nu):
    """t-ILRMA beyond 4 channels (ilrma.py:899-1018): source update, spatial update (returns Xi), loss -- each against
    the oracle from the same state; two utterances in one call == one at a time."""
    F, T = 6, 260
    rng = np.random.default_rng(70 + M)
    Xs = np.stack([mixture(M, F, T, 71 + M), mixture(M, F, T, 72 + M)])
    W = np.stack([rand_filters(M, F, 73), rand_filters(M, F, 74)])
    Tb, V = rng.random((2, M, F, K)) + 0.05, rng.random((2, M, K, T)) + 0.05
    Xb, Wb, Tbd, Vbd = dev_c(eng, Xs), dev_c(eng, W), dev_r(eng, Tb), dev_r(eng, V)
    loss = host(eng.tilrma_loss(Xb, Wb, Tbd, Vbd, nu))
    eng.tilrma_source_update(Xb, Wb, Tbd, Vbd, nu)
    Xi = eng.empty((2, M, F, T))
    st = eng.new_status(2)
    Wn = eng.tilrma_spatial_update(Xb, Wb.clone(), Tbd, Vbd, nu, Xi, status=st)
    assert int(st.sum().item()) == 0
    for b in range(2):
        assert abs(loss[b] - orc.tilrma_loss(Xs[b], W[b], Tb[b], V[b], nu)) < tol(eng, 1e-10, 1e-4) * abs(loss[b])
        P = np.abs(orc.separate(Xs[b], W[b])) ** 2
        Tr, Vr = orc.tilrma_source_update(P, Tb[b], V[b], nu)
        assert rel_err(host(Tbd)[b], Tr) < tol(eng, 1e-10, 2e-4)
        assert rel_err(host(Vbd)[b], Vr) < tol(eng, 1e-10, 2e-4)
        Wr, _ = orc.tilrma_spatial_update(Xs[b], W[b], host(Tbd)[b], host(Vbd)[b], nu)
        assert rel_err(host(Wn)[b], Wr) < tol(eng, 1e-9, 5e-3)
        # one at a time: bit-identical
        W1, T1, V1 = dev_c(eng, W[b:b + 1]), dev_r(eng, Tb[b:b + 1]), dev_r(eng, V[b:b + 1])
        eng.tilrma_source_update(Xb[b:b + 1], W1, T1, V1, nu)
        W1n = eng.tilrma_spatial_update(Xb[b:b + 1], W1, T1, V1, nu, eng.empty((1, M, F, T)), status=eng.new_status(1))
        assert torch.equal(T1[0], Tbd[b]) and torch.equal(V1[0], Vbd[b]) and torch.equal(W1n[0], Wn[b])


@pytest.mark.parametrize("M,K", [(5, 3), (8, 10)])
def test_partitioned_source_update(eng, M, K):
    """partitioning=True beyond 4 channels (ilrma.py:368-408): Z, T, V after one source update against the oracle."""
    F, T = 7, 200
    rng = np.random.default_rng(80 + M)
    X = mixture(M, F, T, 81 + M)
    W = rand_filters(M, F, 82)
    Z = rng.random((M, K)) * 1e-2 + 1 / M
    Z = Z / Z.sum(axis=0)
    Tb, V = rng.random((F, K)) + 0.05, rng.random((K, T)) + 0.05
    Zd, Td, Vd = dev_r(eng, Z[None]), dev_r(eng, Tb[None]), dev_r(eng, V[None])
    Teff, Veff = eng.empty((1, M, F, K)), eng.empty((1, M, K, T))
    eng.ilrma_source_update_partitioned(dev_c(eng, X[None]), dev_c(eng, W[None]), Zd, Td, Vd, Teff, Veff)
    P = np.abs(orc.separate(X, W)) ** 2
    Zr, Tr, Vr = orc.part_source_update(P, Z, Tb, V)
    assert rel_err(host(Zd)[0], Zr) < tol(eng, 1e-10, 2e-4)
    assert rel_err(host(Td)[0], Tr) < tol(eng, 1e-10, 2e-4)
    assert rel_err(host(Vd)[0], Vr) < tol(eng, 1e-10, 2e-4)
    assert rel_err(host(Teff)[0], Zr[:, None, :] * Tr[None]) < tol(eng, 1e-10, 2e-4)
    assert rel_err(host(Veff)[0], np.broadcast_to(Vr, (M, K, T))) < tol(eng, 1e-10, 2e-4)


def test_batched_equals_single_and_unsupported_entry_points(eng):
    from audio_source_separation_amd._lib import AssxError
    M, F, T, K = 6, 7, 200, 3
    Xs = np.stack([mixture(M, F, T, 60), mixture(M, F, T, 61)])
    W = np.stack([rand_filters(M, F, 62), rand_filters(M, F, 63)])
    rng = np.random.default_rng(64)
    Tb, V = rng.random((2, M, F, K)) + 0.05, rng.random((2, M, K, T)) + 0.05
    Wb, Tbd, Vbd = dev_c(eng, W), dev_r(eng, Tb), dev_r(eng, V)
    Xb = dev_c(eng, Xs)
    eng.ilrma_source_update(Xb, Wb, Tbd, Vbd)
    eng.ilrma_spatial_update(Xb, Wb, Tbd, Vbd, status=eng.new_status(2))
    for b in range(2):
        W1, T1, V1 = dev_c(eng, W[b:b + 1]), dev_r(eng, Tb[b:b + 1]), dev_r(eng, V[b:b + 1])
        X1 = dev_c(eng, Xs[b:b + 1])
        eng.ilrma_source_update(X1, W1, T1, V1)
        eng.ilrma_spatial_update(X1, W1, T1, V1, status=eng.new_status(1))
        assert torch.equal(W1[0], Wb[b]) and torch.equal(T1[0], Tbd[b]) and torch.equal(V1[0], Vbd[b])
    with pytest.raises(AssxError):
        eng.demix(dev_c(eng, np.zeros((1, 33, 3, 70), dtype=np.complex128)),
                  dev_c(eng, np.zeros((1, 3, 33, 33), dtype=np.complex128)))  # M = 33 (9 <= M <= 32: test_gpu_manychan.py)


@pytest.mark.parametrize("M,K,domain,G", [(5, 3, 2, 1), (6, 10, 1, 3), (7, 4, 2, 7), (8, 4, 2, 2), (8, 10, 2, 5), (8, 16, 1.5, 3),
                                          (8, 20, 2, 3), (8, 4, 2, 234), (5, 3, 2, 150), (7, 4, 2, 117), (6, 4, 2, 78)])
def test_src_cov_long_ranges(eng, M, K, domain, G):
    """pair_cov_kernel (csrc/assx_widem_cov.hpp) on partitions forced down to a handful of workgroups (ASSX_G), so that a
    small input drives what a full-size utterance does: prologue, steady trips of the ring, ranges that start / end inside
    a bin and flush several records, basis-row reloads at bin boundaries, ragged T, every weight form (rebuilt from
    (Tb, V) for n_basis <= 4, map given beyond / for t-ILRMA and IDLMA, (N,T) for AuxIVA) -- and forced UP to ranges of
    one, two or three items (the pipeline runs two items ahead: its prologue and its last trips are then all there is);
    two utterances in one call == one at a time, bit for bit."""
    import os
    F, T = 9, 777
    rng = np.random.default_rng(300 + M + K)
    Xs = np.stack([mixture(M, F, T, 301 + M), mixture(M, F, T, 302 + M)])
    W = np.stack([rand_filters(M, F, 303), rand_filters(M, F, 304)])
    Tb, V = rng.random((2, M, F, K)) + 0.05, rng.random((2, M, K, T)) + 0.05
    V[0, 1, :, 5:9] = 0.0  # variance below eps: floored
    r_nt, r_nft = rng.random((2, M, T)) + 0.05, rng.random((2, M, F, T)) + 0.05
    os.environ["ASSX_G"] = str(G)
    try:
        Xb = dev_c(eng, Xs)
        Ub = eng.empty((2, M, F, M, M), complex_=True)
        Wb = dev_c(eng, W)
        eng.ilrma_spatial_update(Xb, Wb, dev_r(eng, Tb), dev_r(eng, V), domain=domain, status=eng.new_status(2), U_out=Ub)
        U_nt = eng.cov_accumulate(Xb, dev_r(eng, r_nt))
        U_nft = eng.cov_accumulate(Xb, dev_r(eng, r_nft))
        for b in range(2):
            Wref, Uref, mask = orc.ilrma_spatial_update_ip(Xs[b], W[b].copy(), Tb[b], V[b], domain)
            assert rel_err(host(Ub)[b], Uref) < tol(eng, 1e-12, 2e-5)
            if b == 1:  # utterance 0 has floored variances: weights of 1e12, a sweep that amplifies rounding
                assert rel_err(host(Wb)[b], Wref) < tol(eng, 1e-9, 2e-3)
            assert rel_err(host(U_nt)[b], orc.weighted_covariance(Xs[b], r_nt[b])) < tol(eng, 1e-12, 2e-5)
            assert rel_err(host(U_nft)[b], orc.weighted_covariance(Xs[b], r_nft[b])) < tol(eng, 1e-12, 2e-5)
            U1 = eng.empty((1, M, F, M, M), complex_=True)
            eng.ilrma_spatial_update(Xb[b:b + 1], dev_c(eng, W[b:b + 1]), dev_r(eng, Tb[b:b + 1]), dev_r(eng, V[b:b + 1]),
                                     domain=domain, status=eng.new_status(1), U_out=U1)
            assert torch.equal(U1[0], Ub[b])
            assert torch.equal(eng.cov_accumulate(Xb[b:b + 1], dev_r(eng, r_nft[b:b + 1]))[0], U_nft[b])
        Uh = host(Ub)
        assert np.array_equal(Uh, Uh.conj().swapaxes(-1, -2))  # Hermitian bit-exact
    finally:
        os.environ.pop("ASSX_G", None)


@pytest.mark.parametrize("M,K,G", [(5, 3, 1), (6, 4, 3), (7, 10, 2), (8, 4, 0), (8, 4, 5), (8, 8, 3), (5, 16, 2), (8, 16, 2)])
def test_source_model_without_loss(eng, M, K, G):
    """The wide-channel source model without a loss request (the route the benchmark loop takes), n_basis up to 16, forced
    partitions, against the oracle; batched == single bit for bit.  (Written for the streaming source model of round 3,
    which passed it and was not kept: profiles/r03_src_nmf_experiment.txt.)"""
    import os
    F, T = 9, 400
    rng = np.random.default_rng(600 + 10 * M + K)
    Xs = np.stack([mixture(M, F, T, 601 + M), mixture(M, F, T, 602 + M)])
    W = np.stack([rand_filters(M, F, 603), rand_filters(M, F, 604)])
    Tb, V = rng.random((2, M, F, K)) + 0.02, rng.random((2, M, K, T)) + 0.02
    V[0, 1, :, 3:6] = 0.0
    if G:
        os.environ["ASSX_G"] = str(G)
    try:
        Xb, Wb = dev_c(eng, Xs), dev_c(eng, W)
        Td, Vd = dev_r(eng, Tb), dev_r(eng, V)
        eng.ilrma_source_update(Xb, Wb, Td, Vd)
        for b in range(2):
            T1, V1 = orc.ilrma_source_update(np.abs(orc.separate(Xs[b], W[b])) ** 2, Tb[b], V[b], 2)
            assert rel_err(host(Td)[b], T1) < tol(eng, 1e-11, 1e-4)
            assert rel_err(host(Vd)[b], V1) < tol(eng, 1e-11, 1e-4)
            t1, v1 = dev_r(eng, Tb[b:b + 1]), dev_r(eng, V[b:b + 1])
            eng.ilrma_source_update(Xb[b:b + 1], Wb[b:b + 1], t1, v1)
            assert torch.equal(t1[0], Td[b]) and torch.equal(v1[0], Vd[b])
    finally:
        os.environ.pop("ASSX_G", None)



def test_pair_and_source_forms_of_the_covariance_agree():
    """The two streaming forms of the wide-channel covariance (pairs split over the waves: default; one wave per source:
    ASSX_WIDEM_PAIRS=0, read once per process; laboratory builds only) give the same U up to the rounding of their different
    product order."""
    from conftest import need_lab
    need_lab("ASSX_WIDEM_PAIRS")
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from audio_source_separation_amd.ops import Engine\n"
        "eng = Engine(dtype='float64')\n"
        "g = torch.Generator(device=eng.dev).manual_seed(5)\n"
        "out = {}\n"
        "for M in (5, 6, 7, 8):\n"
        "    F, T, K = 7, 500, 3\n"
        "    X = torch.view_as_complex(torch.randn((1, M, F, T, 2), dtype=torch.float64, device=eng.dev, generator=g)).contiguous()\n"
        "    W = (torch.eye(M, dtype=torch.complex128, device=eng.dev).expand(1, F, M, M) + 0).contiguous()\n"
        "    Tb = torch.rand((1, M, F, K), dtype=torch.float64, device=eng.dev, generator=g) + 0.1\n"
        "    V = torch.rand((1, M, K, T), dtype=torch.float64, device=eng.dev, generator=g) + 0.1\n"
        "    U = eng.empty((1, M, F, M, M), complex_=True)\n"
        "    eng.ilrma_spatial_update(X, W, Tb, V, domain=2, status=eng.new_status(1), U_out=U)\n"
        "    out['U%%d' %% M] = U.cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n" % root)
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for mode in ("1", "0"):
            env = dict(os.environ, ASSX_WIDEM_PAIRS=mode)
            path = os.path.join(d, "u%s.npz" % mode)
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=600)
            res[mode] = dict(np.load(path))
    for k in res["1"]:
        assert rel_err(res["1"][k], res["0"][k]) < 1e-13
