import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "music-synthesis_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fatal signal during the GPU tests must leave NATIVE frames (a fault inside hipGraphLaunch shows only interpreter
    # frames in Python's faulthandler): one occurrence then suffices for a diagnosis -- faults are never re-provoked
    import faulthandler
    faulthandler.enable()
    try:
        from featuresynth._ops import lib as L
        if os.path.exists(L.LIB_PATH):
            L.load().ms_debug_install_crash_handler()
    except Exception as e:          # (CPU-only runs without the library: the ABI tests report that themselves)
        sys.stderr.write("conftest: native crash handler not installed: %r\n" % (e,))


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def rel_l2(a, b):
    import numpy as np
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def stable_seed(name):
    """Seed from a test-case name that does not change between processes (str hash() is randomised)."""
    import zlib
    return zlib.crc32(name.encode()) & 0x7FFFFFFF
