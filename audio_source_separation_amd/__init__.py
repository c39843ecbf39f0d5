"""MI355X-native iterative source-separation hot path (NMF / AuxIVA / Gauss-ILRMA).

Drop-in class surface of tky823/audio_source_separation for this path, backed by hand-written
HIP kernels behind a C-ABI (include/assx.h).  Importing the package loads libassx.so and fails
loudly if it has not been built; there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when csrc/libassx.so is missing)

__version__ = "0.1.0"


def version():
    return _lib.version()
