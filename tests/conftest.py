import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """Builds (if stale) and loads libblah2hip.so; never falls back to anything else."""
    from blah2_amd import build
    build.build_all(verbose=False)
    import blah2_amd
    return blah2_amd.load()


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


def load_golden(name):
    import numpy as np
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: z[k] for k in z.files}
    iq = g["iq"]
    g["x"] = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    g["y"] = iq[:, 2].astype(np.float64) + 1j * iq[:, 3].astype(np.float64)
    return g
