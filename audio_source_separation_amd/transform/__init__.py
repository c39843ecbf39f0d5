from .stft import stft, istft, build_window  # noqa: F401
