"""GPU parity tests for the WienerHopf clutter filter through the C ABI.

Tolerance: the filtered surveillance channel is compared with the compiled
reference's output (tests/golden, fp64) as max|dy| / max|y_ref| <= 1e-4.  The
filter removes a component ~10x stronger than what remains, so the fp32 error
relative to the *input* level is ~10x smaller than this figure.
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import blah2_oracle as O

pytestmark = pytest.mark.gpu
Y_TOL = 1e-4


@pytest.fixture(scope="module")
def b2(built_lib):
    import blah2_amd
    assert blah2_amd.device_count() > 0
    return blah2_amd


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("path", ["c64", "c32"])
def test_clutter_golden(b2, name, path):
    g = load_golden(name)
    n = int(g["params"][1])
    dmin, dmax = (int(v) for v in g["clutter_params"])
    wh = b2.WienerHopf(dmin, dmax, n)
    x, y = g["x"], g["y"]
    if path == "c32":
        x, y = x.astype(np.complex64), y.astype(np.complex64)
    ok, yf = wh.process(x, y)
    assert ok == bool(g["clutter_ok"])
    ref = g["clutter_y"]
    err = np.max(np.abs(yf.astype(np.complex128) - ref)) / np.max(np.abs(ref))
    assert err <= Y_TOL, f"filtered-channel error {err:.3e}"
    assert np.linalg.norm(yf) < 0.5 * np.linalg.norm(g["y"])


@pytest.mark.parametrize("fft_len", [1024, 2048, 4096])
def test_clutter_every_transform_length(b2, fft_len, monkeypatch):
    monkeypatch.setenv("BLAH2HIP_CLUTTER_FFT_LEN", str(fft_len))
    g = load_golden("medium")
    n = int(g["params"][1])
    dmin, dmax = (int(v) for v in g["clutter_params"])
    ok, yf = b2.WienerHopf(dmin, dmax, n).process(g["x"], g["y"])
    assert ok
    assert np.max(np.abs(yf - g["clutter_y"])) / np.max(np.abs(g["clutter_y"])) <= Y_TOL


def test_clutter_failure_contract(b2):
    # all-zero reference channel: the Toeplitz matrix is singular, the reference's
    # chol() fails, process() returns false and y is untouched (WienerHopf.cpp:111-115)
    n = 20000
    y = (np.arange(n) % 7 + 1j).astype(np.complex128)
    ok, yf = b2.WienerHopf(-3, 20, n).process(np.zeros(n, dtype=np.complex128), y)
    assert not ok and np.array_equal(yf, y)
