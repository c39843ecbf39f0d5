"""STFT / iSTFT on the device against the reference's own outputs (tests/golden/stft.npz,
scipy.signal.stft / istft as src/transform/stft.py calls them) and against the numpy.fft oracle at larger sizes.

Tolerances: relative Frobenius error 1e-12 in float64 (FFT butterflies vs pocketfft differ by rounding only),
2e-5 in float32.
"""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import oracle_np as orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def stft_perturb(X):
    """Same rule as tests/golden/make_golden.py:stft_perturb."""
    return X * (1.0 + 0.1 * np.cos(np.arange(X.size, dtype=np.float64)).reshape(X.shape)) + 0.01j


@pytest.fixture(scope="module", params=["float64", "float32"])
def dtype(request):
    return request.param


def tol(dtype, t64, t32):
    return t64 if dtype == "float64" else t32


def test_stft_istft_golden(dtype):
    from audio_source_separation_amd.transform import stft, istft
    g = load_golden("stft")
    for i, (L, N, hop, hamming) in enumerate(g["cases"]):
        L, N, hop, wf = int(L), int(N), int(hop), ("hamming" if hamming else "hann")
        X = stft(g["x%d" % i], fft_size=N, hop_size=hop, window_fn=wf, dtype=dtype)
        assert X.shape == g["X%d" % i].shape and np.iscomplexobj(X)
        assert rel_err(X, g["X%d" % i]) < tol(dtype, 1e-12, 2e-5), (L, N, hop)
        y = istft(stft_perturb(g["X%d" % i]), fft_size=N, hop_size=hop, window_fn=wf, dtype=dtype)
        assert y.shape == g["y%d" % i].shape
        assert rel_err(y, g["y%d" % i]) < tol(dtype, 1e-12, 2e-5), (L, N, hop)
        ycut = istft(g["X%d" % i], fft_size=N, hop_size=hop, window_fn=wf, length=L, dtype=dtype)
        assert ycut.shape == g["ycut%d" % i].shape
        assert rel_err(ycut, g["ycut%d" % i]) < tol(dtype, 1e-12, 2e-5), (L, N, hop)


@pytest.mark.parametrize("L,N,hop", [(40000, 2048, 512), (70001, 4096, 2048), (30000, 8192, 2048), (20000, 1000, 250),
                                     (9000, 2, 1), (5000, 16384, 4096)])
def test_stft_istft_oracle(dtype, L, N, hop):
    """The sizes the separation loop is benchmarked at (fft_size 2048 -> 1025 bins), the largest in-LDS FFT, the
    direct-DFT route (1000 and 16384 points) and the smallest frame; round trip = identity (Hann at 75 % / 50 %)."""
    from audio_source_separation_amd.transform import stft, istft
    if L < N:
        pytest.skip("signal shorter than the frame")
    rng = np.random.default_rng(L + N)
    x = rng.standard_normal((3, L))
    X = stft(x, fft_size=N, hop_size=hop, dtype=dtype)
    Xo = orc.stft(x, N, hop)
    assert X.shape == Xo.shape
    assert rel_err(X, Xo) < tol(dtype, 1e-12, 2e-5)
    y = istft(stft_perturb(Xo), fft_size=N, hop_size=hop, dtype=dtype)
    yo = orc.istft(stft_perturb(Xo), N, hop)
    assert y.shape == yo.shape
    assert rel_err(y, yo) < tol(dtype, 1e-12, 2e-5)
    back = istft(X, fft_size=N, hop_size=hop, length=L, dtype=dtype)
    assert rel_err(back, x) < tol(dtype, 1e-12, 2e-5)


def test_stft_device_tensors_and_batch_axes(dtype):
    """Device tensor in -> device tensor out, leading axes kept (a (B, M, L) batch of multichannel recordings)."""
    from audio_source_separation_amd.transform import stft, istft
    from audio_source_separation_amd._device import require_gpu
    dev = require_gpu(None)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 3, 3000))
    xd = torch.from_numpy(x).to(dev)
    X = stft(xd, fft_size=256, hop_size=64, dtype=dtype)
    assert isinstance(X, torch.Tensor) and X.is_cuda and tuple(X.shape) == (2, 3, 129, orc.stft(x, 256, 64).shape[-1])
    assert rel_err(X.cpu().numpy().astype(np.complex128), orc.stft(x, 256, 64)) < tol(dtype, 1e-12, 2e-5)
    y = istft(X, fft_size=256, hop_size=64, length=3000, dtype=dtype)
    assert isinstance(y, torch.Tensor) and tuple(y.shape) == (2, 3, 3000)
    assert rel_err(y.cpu().numpy().astype(np.float64), x) < tol(dtype, 1e-12, 2e-5)


def test_stft_errors(dtype):
    from audio_source_separation_amd.transform import stft
    from audio_source_separation_amd._lib import AssxError
    with pytest.raises(ValueError):
        stft(np.zeros((1, 100)), fft_size=32, hop_size=8, window_fn="blackman", dtype=dtype)
    with pytest.raises(AssxError, match="shorter than fft_size"):
        stft(np.zeros((1, 10)), fft_size=32, hop_size=8, dtype=dtype)


def test_wav_to_wav(dtype):
    """stft -> GaussILRMA -> istft: the end-to-end chain of the reference's `_test` drivers (ilrma.py:1289-1299) on a
    synthetic instantaneous mixture."""
    from audio_source_separation_amd.transform import stft, istft
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    rng = np.random.default_rng(11)
    L = 16000
    s = rng.standard_normal((2, L)) * (1.0 + 0.9 * np.sin(np.arange(L) / 400.0 + np.array([[0.0], [1.5]])))
    x = np.array([[1.0, 0.6], [0.5, 1.0]]) @ s + 1e-3 * rng.standard_normal((2, L))
    X = stft(x, fft_size=512, hop_size=128, dtype=dtype)
    np.random.seed(111)
    Y = GaussILRMA(n_basis=2, dtype=dtype)(X, iteration=20)
    y = istft(Y, fft_size=512, hop_size=128, length=L, dtype=dtype)
    assert y.shape == (2, L) and np.isfinite(y).all()
    assert rel_err(y, orc.istft(Y, 512, 128, length=L)) < tol(dtype, 1e-12, 2e-5)


def test_wav_file_to_wav_file(dtype, tmp_path):
    """The whole chain through FILES: write_wav -> read_wav -> stft -> GaussILRMA -> istft -> write_wav -> read_wav
    (ref: utils_audio.py:4-18 either side of ilrma.py:1289-1299), device tensors between the two file ends."""
    import torch
    from audio_source_separation_amd.transform import stft, istft
    from audio_source_separation_amd.bss.ilrma import GaussILRMA
    from audio_source_separation_amd.utils.utils_audio import read_wav, write_wav
    rng = np.random.default_rng(12)
    L, sr = 8000, 16000
    s = 0.2 * rng.standard_normal((2, L)) * (1.0 + 0.9 * np.sin(np.arange(L) / 300.0 + np.array([[0.0], [1.5]])))
    x = np.array([[1.0, 0.6], [0.5, 1.0]]) @ s
    src, dst = str(tmp_path / "mix.wav"), str(tmp_path / "sep.wav")
    write_wav(src, x, sr, channel_last=False)
    mix, sr2 = read_wav(src)                       # (L, 2)
    X = stft(torch.from_numpy(mix.T.copy()).cuda(), fft_size=256, hop_size=64, dtype=dtype)
    np.random.seed(5)
    Y = GaussILRMA(n_basis=2, dtype=dtype)(X, iteration=10)
    y = istft(Y, fft_size=256, hop_size=64, length=L, dtype=dtype)
    assert isinstance(y, torch.Tensor) and tuple(y.shape) == (2, L)
    write_wav(dst, y, sr2, channel_last=False)
    out, sr3 = read_wav(dst)
    assert sr3 == sr and out.shape == (L, 2) and np.isfinite(out).all()
    q = np.clip(y.cpu().numpy().astype(np.float64) * 32768, -32768, 32767).astype(np.int16).T / 32768
    assert np.array_equal(out, q)
