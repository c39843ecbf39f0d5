"""More than 8 channels (9 <= M <= 32): the run-time channel-count kernels of csrc/assx_widem_rt.hpp behind the same C-ABI
entry points and classes -- every stage against the oracle on seeded inputs (the oracle is generic in M, like the
reference: src/bss/ilrma.py:61-62, src/bss/iva.py:39-59), float64 and float32, batched == single; a reference-generated
fixture at M = 9 is in test_gpu_models.py (`ilrma_m9`).  IP and, since round 6, ISS, IP2 and the partitioning function."""
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
    return np.eye(M)[None] + 0.1 * (rng.standard_normal((F, M, M)) + 1j * rng.standard_normal((F, M, M)))


SHAPES = [(9, 5, 300), (12, 4, 257), (17, 3, 150)]


@pytest.mark.parametrize("M,F,T", SHAPES)
def test_demix_cov_ip(eng, M, F, T):
    X, W = mixture(M, F, T, 1), rand_filters(M, F, 2)
    Xd = dev_c(eng, X[None])
    Y = eng.demix(Xd, dev_c(eng, W[None]))
    assert rel_err(host(Y)[0], orc.separate(X, W)) < tol(eng, 1e-14, 3e-6)
    rng = np.random.default_rng(3)
    r_nt, r_nft = rng.random((M, T)) + 0.05, rng.random((M, F, T)) + 0.05
    r_nt[0, :3] = 0.0  # eps floor
    for r in (r_nt, r_nft):
        U = eng.cov_accumulate(Xd, dev_r(eng, r[None]))
        assert rel_err(host(U)[0], orc.weighted_covariance(X, r)) < tol(eng, 1e-12, 3e-5)
    C = eng.cov_accumulate(Xd)
    assert rel_err(host(C)[0, 0], orc.weighted_covariance(X, np.ones((1, T)))[0]) < tol(eng, 1e-12, 3e-5)
    Uh = host(U)[0]
    assert np.array_equal(Uh, Uh.conj().transpose(0, 1, 3, 2))  # Hermitian bit-exact
    Uo = orc.weighted_covariance(X, r_nft)
    Wd = dev_c(eng, W[None])
    st = eng.new_status(1)
    eng.ip_update(dev_c(eng, Uo[None]), Wd, 1e12, st)
    Wref, mask = orc.ip_update(W.copy(), Uo)
    assert mask.all() and int(st.item()) == 0
    assert rel_err(host(Wd)[0], Wref) < tol(eng, 1e-8, 5e-3)


@pytest.mark.parametrize("M,K,domain", [(9, 3, 2), (12, 10, 1), (10, 4, 2)])
def test_ilrma_stages(eng, M, K, domain):
    F, T = 5, 333
    X, W = mixture(M, F, T, 20 + M), rand_filters(M, F, 21)
    rng = np.random.default_rng(22)
    Tb, V = rng.random((M, F, K)) + 0.05, rng.random((M, K, T)) + 0.05
    Xd, Wd = dev_c(eng, X[None]), dev_c(eng, W[None])
    got = float(eng.ilrma_loss(Xd, Wd, dev_r(eng, Tb[None]), dev_r(eng, V[None]), domain=domain).item())
    np.testing.assert_allclose(got, orc.ilrma_loss(X, W, Tb, V, domain), rtol=tol(eng, 1e-12, 2e-5))
    Td, Vd = dev_r(eng, Tb[None]), dev_r(eng, V[None])
    lp = eng.empty((1,), dtype=torch.float64)
    eng.ilrma_source_update(Xd, Wd, Td, Vd, domain=domain, loss_prev=lp)
    T1, V1 = orc.ilrma_source_update(np.abs(orc.separate(X, W)) ** 2, Tb, V, domain)
    assert rel_err(host(Td)[0], T1) < tol(eng, 1e-11, 2e-4) and rel_err(host(Vd)[0], V1) < tol(eng, 1e-11, 2e-4)
    np.testing.assert_allclose(lp.item(), orc.ilrma_loss(X, W, Tb, V, domain), rtol=tol(eng, 1e-12, 2e-5))
    Ud = eng.empty((1, M, F, M, M), complex_=True)
    C = eng.cov_accumulate(Xd).reshape(1, F, M, M)
    pb = eng.empty((1, M, F), dtype=torch.float64)
    st = eng.new_status(1)
    Wd2 = dev_c(eng, W[None])
    eng.ilrma_spatial_update(Xd, Wd2, dev_r(eng, T1[None]), dev_r(eng, V1[None]), domain=domain, status=st, U_out=Ud,
                             C=C, power_bins=pb)
    Wref, Uref, mask = orc.ilrma_spatial_update_ip(X, W.copy(), T1, V1, domain)
    assert mask.all() and int(st.item()) == 0
    assert rel_err(host(Ud)[0], Uref) < tol(eng, 1e-12, 3e-5)
    assert rel_err(host(Wd2)[0], Wref) < tol(eng, 1e-8, 5e-3)
    P = np.abs(orc.separate(X, Wref)) ** 2
    np.testing.assert_allclose(host(pb)[0], P.mean(axis=2), rtol=tol(eng, 1e-8, 5e-3))
    p_direct = eng.demix_power(Xd, Wd2)
    p_cov = eng.power_from_cov(C, Wd2, T)
    np.testing.assert_allclose(host(p_direct)[0], P.mean(axis=(1, 2)), rtol=tol(eng, 1e-9, 5e-3))
    np.testing.assert_allclose(host(p_cov)[0], P.mean(axis=(1, 2)), rtol=tol(eng, 1e-8, 5e-3))
    sc = eng.projection_back_scale(Xd, Wd2, 1, st)
    assert rel_err(host(sc)[0], orc.projection_back(orc.separate(X, Wref), X[1])) < tol(eng, 1e-8, 1e-2)


@pytest.mark.parametrize("kind", ["laplace", "gauss"])
def test_auxiva_stages(eng, kind):
    from audio_source_separation_amd import _lib
    M, F, T = 9, 6, 300
    X, W = mixture(M, F, T, 40 + M), rand_filters(M, F, 41)
    Xd, Wd = dev_c(eng, X[None]), dev_c(eng, W[None])
    code = _lib.IVA_LAPLACE if kind == "laplace" else _lib.IVA_GAUSS
    r, loss = eng.auxiva_weights(Xd, Wd, code, with_loss=True)
    Y = orc.separate(X, W)
    assert rel_err(host(r)[0], orc.auxiva_weights(Y, kind)) < tol(eng, 1e-12, 3e-5)
    np.testing.assert_allclose(loss.item(), orc.auxiva_loss(X, W, kind), rtol=tol(eng, 1e-12, 2e-5))
    st = eng.new_status(1)
    eng.auxiva_spatial_update(Xd, Wd, r, status=st)
    Wref, _, mask = orc.auxiva_update_once_ip(X, W.copy(), Y, kind)
    assert mask.all() and int(st.item()) == 0
    assert rel_err(host(Wd)[0], Wref) < tol(eng, 1e-8, 5e-3)


def test_classes_batched_and_unsupported():
    """GaussILRMA and AuxLaplaceIVA at M = 9 through the classes against the oracle's whole loop; two utterances in one
    call == one at a time, bit for bit; what the run-time path does not serve raises."""
    from audio_source_separation_amd._lib import AssxError
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    from audio_source_separation_amd.bss.iva import AuxLaplaceIVA
    M, F, T, K = 9, 6, 260, 3
    Xs = np.stack([mixture(M, F, T, 60), mixture(M, F, T, 61)])
    st = [np.random.RandomState(70 + u) for u in range(2)]
    T0 = np.stack([s.rand(M, F, K) for s in st])
    V0 = np.stack([s.rand(M, K, T) for s in st])
    m = GaussILRMA(n_basis=K)
    m.basis, m.activation = T0, V0
    Yb = m(Xs, iteration=4)
    for b in range(2):
        ref = orc.gauss_ilrma(Xs[b], 4, T0[b], V0[b])
        assert rel_err(Yb[b], ref["Y"]) < 1e-7 and rel_err(np.asarray(m.basis)[b], ref["T"]) < 1e-7
        np.testing.assert_allclose(np.asarray(m.loss)[:, b], ref["loss"], rtol=1e-9)
        m1 = GaussILRMA(n_basis=K)
        m1.basis, m1.activation = T0[b], V0[b]
        assert np.array_equal(m1(Xs[b], iteration=4), Yb[b])
    a = AuxLaplaceIVA()
    Ya = a(Xs[0], iteration=3)
    refa = orc.auxiva(Xs[0], 3, "laplace")
    assert rel_err(Ya, refa["Y"]) < 1e-7
    np.testing.assert_allclose(np.asarray(a.loss), refa["loss"], rtol=1e-9)
    W = a.compute_demix_filter(orc.separate(Xs[0], refa["W"]), Xs[0])
    assert rel_err(np.asarray(W), refa["W"]) < 1e-7
    with pytest.raises((AssxError, NotImplementedError, ValueError)):
        GaussILRMA(n_basis=K)(np.zeros((33, 4, 70), dtype=np.complex128), iteration=1)  # M = 33


@pytest.mark.parametrize("M,F,T", [(9, 6, 260), (12, 4, 257), (17, 3, 150)])
def test_iss_sweep_more_than_8_channels(eng, M, F, T):
    """The ISS sweep with a run-time channel count (csrc/assx_widem_rt.hpp: iss_rt_kernel; round 5's review, missing #2):
    one sweep against the oracle's rank-one updates of Y (ilrma.py:557-562: sums, not means) restated on W, from a dense
    weighted covariance; then the classes' whole loops."""
    X, W = mixture(M, F, T, 31), rand_filters(M, F, 32)
    rng = np.random.default_rng(33)
    r = rng.random((M, F, T)) + 0.05
    U = orc.weighted_covariance(X, r)
    Wd = dev_c(eng, W[None])
    eng.iss_update(dev_c(eng, U[None]), Wd, T)
    Y = orc.iss_update(orc.separate(X, W), r)
    Wref = orc.compute_demix_filter(Y, X)
    assert rel_err(host(Wd)[0], Wref) < tol(eng, 1e-9, 2e-3)


def test_iss_classes_more_than_8_channels():
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    from audio_source_separation_amd.bss.iva import AuxLaplaceIVA
    M, F, T, K = 9, 6, 260, 3
    X = mixture(M, F, T, 60)
    st = np.random.RandomState(70)
    T0, V0 = st.rand(M, F, K), st.rand(M, K, T)
    m = GaussILRMA(n_basis=K, algorithm_spatial="ISS")
    m.basis, m.activation = T0, V0
    Y = m(X, iteration=3)
    ref = orc.gauss_ilrma_iss(X, 3, T0, V0)
    assert rel_err(Y, ref["Y"]) < 1e-7 and rel_err(np.asarray(m.basis), ref["T"]) < 1e-7
    np.testing.assert_allclose(np.asarray(m.loss), ref["loss"], rtol=1e-8)
    a = AuxLaplaceIVA(algorithm_spatial="ISS")
    Ya = a(X, iteration=3)
    refa = orc.auxiva_iss(X, 3, "laplace")
    assert rel_err(Ya, refa["Y"]) < 1e-7
    np.testing.assert_allclose(np.asarray(a.loss), refa["loss"], rtol=1e-8)


@pytest.mark.parametrize("M,F,T,pair", [(9, 6, 260, (0, 1)), (12, 4, 257, (11, 0)), (17, 3, 150, (5, 6))])
def test_ip2_sweep_more_than_8_channels(eng, M, F, T, pair):
    """The pairwise (IP2) update with a run-time channel count (csrc/assx_widem_rt.hpp: ip2_rt_kernel): rows pm / pn against the
    oracle's update (generalised 2 x 2 eigenproblem, LAPACK's eigenvector convention, descending order), every other row
    untouched bit for bit."""
    X, W = mixture(M, F, T, 41), rand_filters(M, F, 42)
    rng = np.random.default_rng(43)
    r = rng.random((M, F, T)) + 0.05
    U = orc.weighted_covariance(X, r)
    Wd = dev_c(eng, W[None])
    st = eng.new_status(1)
    eng.ip2_update(dev_c(eng, U[None]), Wd, pair, 1e12, st)
    Wref, _, _ = orc.ip2_update(W.copy(), U[pair[0]], U[pair[1]], pair[0], pair[1])
    got = host(Wd)[0]
    assert int(st.item()) == 0
    assert rel_err(got[:, list(pair), :], Wref[:, list(pair), :]) < tol(eng, 1e-9, 5e-3)
    others = [n for n in range(M) if n not in pair]
    assert np.array_equal(got[:, others, :], host(dev_c(eng, W[None]))[0][:, others, :])


def test_ip2_and_partitioning_classes_more_than_8_channels():
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    M, F, T, K = 9, 6, 260, 3
    X = mixture(M, F, T, 60)
    st = np.random.RandomState(70)
    T0, V0 = st.rand(M, F, K), st.rand(M, K, T)
    m = GaussILRMA(n_basis=K, algorithm_spatial="IP2")
    m.basis, m.activation = T0, V0
    Y = m(X, iteration=4)
    ref = orc.gauss_ilrma_ip2(X, 4, T0, V0)
    assert rel_err(Y, ref["Y"]) < 1e-7 and rel_err(np.asarray(m.basis), ref["T"]) < 1e-7
    np.testing.assert_allclose(np.asarray(m.loss), ref["loss"], rtol=1e-8)
    assert tuple(m.update_pair) == tuple(ref["update_pair"])
    # the partitioning function (shared bases + latent variables), IP and ISS
    for alg in ("IP", "ISS"):
        st = np.random.RandomState(71)
        Z0 = st.rand(M, K)
        Z0 = Z0 / Z0.sum(axis=0)
        Tp, Vp = st.rand(F, K), st.rand(K, T)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mp = GaussILRMA(n_basis=K, partitioning=True, algorithm_spatial=alg)
        mp.latent, mp.basis, mp.activation = Z0, Tp, Vp
        Yp = mp(X, iteration=3)
        refp = orc.gauss_ilrma_partitioned(X, 3, Z0, Tp, Vp, algorithm_spatial=alg)
        assert rel_err(Yp, refp["Y"]) < 1e-7, alg
        assert rel_err(np.asarray(mp.latent), refp["Z"]) < 1e-7 and rel_err(np.asarray(mp.basis), refp["T"]) < 1e-7, alg
        np.testing.assert_allclose(np.asarray(mp.loss), refp["loss"], rtol=1e-8)
