import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def lab_build():
    """True when the loaded library is a laboratory build (-DASSX_LAB=1: assx_version() ends in "+lab").  The variants that
    were measured and not kept (ASSX_IP_PAR, ASSX_AUX_FOLD, ASSX_WIDEM_PAIRS=0, ASSX_UTT_ORDER=0, ...) exist only there;
    in the shipped library their environment switches are not read at all, so the tests of those variants skip."""
    from audio_source_separation_amd import _lib
    return "+lab" in _lib.version()


def need_lab(switch):
    if not lab_build():
        pytest.skip("%s is a laboratory-build switch (csrc/build.sh with ASSX_EXTRA_FLAGS=-DASSX_LAB=1); the shipped "
                    "library does not read it" % switch)
