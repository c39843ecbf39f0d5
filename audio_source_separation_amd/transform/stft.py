"""stft / istft on MI355X -- drop-in for `transform.stft.stft / istft`
(/root/reference/src/transform/stft.py:4-17, i.e. scipy.signal.stft / istft with nperseg=fft_size,
noverlap=fft_size-hop_size and scipy's defaults: zero boundary, padded, 'spectrum' scaling, one-sided).

NumPy in -> NumPy out, device tensor in -> device tensor out (so wav -> separation -> wav never leaves HBM).
The transforms run in assx_stft / assx_istft (hand-written in-LDS FFT); there is no CPU path.
"""
import numpy as np

from .._device import to_device, to_numpy, torch
from ..ops import Engine

_ENGINES = {}


def _engine(dtype, device):
    key = (dtype, str(device))
    if key not in _ENGINES:
        _ENGINES[key] = Engine(dtype=dtype, device=device)
    return _ENGINES[key]


def build_window(fft_size, window_fn='hann'):
    """Periodic (DFT-even) window, as scipy.signal.get_window(window_fn, fft_size) returns it (stft.py:19-27)."""
    n = np.arange(fft_size)
    if window_fn == 'hann':
        return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / fft_size)
    if window_fn == 'hamming':
        return 0.54 - 0.46 * np.cos(2.0 * np.pi * n / fft_size)
    raise ValueError("Not support {} window.".format(window_fn))


def _prepare(input, real, dtype, device):
    eng = _engine(dtype, device)
    lead = tuple(input.shape[:-1 if real else -2])
    x = to_device(input, eng.prec.real if real else eng.prec.cplx, eng.dev)
    return eng, lead, x


def stft(input, fft_size, hop_size=None, window_fn='hann', normalize=False, *, dtype='float64', device=None):
    """
    Args:
        input: (..., n_samples) real
    Returns:
        output: (..., fft_size//2+1, n_frames) complex
    """
    if hop_size is None:
        raise TypeError("hop_size is required")  # the reference computes fft_size - None
    if fft_size - hop_size >= fft_size:
        raise ValueError('noverlap must be less than nperseg.')
    eng, lead, x = _prepare(input, True, dtype, device)
    win = to_device(build_window(fft_size, window_fn), eng.prec.real, eng.dev)
    X = eng.stft(x.reshape(-1, x.shape[-1]).contiguous(), win, int(fft_size), int(hop_size))
    X = X.reshape(lead + tuple(X.shape[1:]))
    if isinstance(input, torch.Tensor):
        return X
    return to_numpy(X, np.complex128 if dtype == 'float64' else np.complex64)


def istft(input, fft_size, hop_size=None, window_fn='hann', normalize=False, length=None, *, dtype='float64',
          device=None):
    """
    Args:
        input: (..., fft_size//2+1, n_frames) complex
    Returns:
        output: (..., n_samples) real, cut to `length` if given
    """
    if hop_size is None:
        raise TypeError("hop_size is required")
    eng, lead, X = _prepare(input, False, dtype, device)
    win = to_device(build_window(fft_size, window_fn), eng.prec.real, eng.dev)
    y = eng.istft(X.reshape((-1,) + tuple(X.shape[-2:])).contiguous(), win, int(fft_size), int(hop_size))
    y = y.reshape(lead + (y.shape[-1],))
    if length is not None:
        y = y[..., :length]
    if isinstance(input, torch.Tensor):
        return y
    return to_numpy(y, np.float64 if dtype == 'float64' else np.float32)
