"""wav file IO either side of the device path -- drop-in for the reference's `utils.utils_audio.read_wav / write_wav`
(/root/reference/src/utils/utils_audio.py:4-18): 16-bit PCM through scipy.io.wavfile, samples scaled by 2**15.

Host-side by nature (file IO); everything between `read_wav` and `write_wav` -- stft, the separation loop, istft --
runs on the GPU (audio_source_separation_amd.transform, .bss).
"""
import numpy as np
from scipy.io import wavfile


def read_wav(path):
    """Returns (signal, sr): int16 samples divided by 32768 -- (n_samples,) or (n_samples, n_channels) float64."""
    sr, signal = wavfile.read(path)
    signal = signal / 32768

    return signal, sr


def write_wav(path, signal, sr, channel_last=True):
    """signal * 32768 clipped to int16; 1-D or 2-D; `channel_last=False` means (n_channels, n_samples)."""
    if hasattr(signal, "detach"):  # a device tensor straight from istft
        signal = signal.detach().cpu().numpy()
    signal = signal * 32768
    signal = np.clip(signal, -32768, 32767).astype(np.int16)

    if signal.ndim not in [1, 2]:
        raise ValueError("Only support 1D or 2D input.")
    if signal.ndim == 2 and not channel_last:
        signal = signal.transpose()
    wavfile.write(path, sr, signal)
