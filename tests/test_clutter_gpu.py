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
    if dmin <= 0:  # the taps cover the direct path at delay 0
        assert np.linalg.norm(yf) < 0.5 * np.linalg.norm(g["y"])


@pytest.mark.parametrize("fft_len", [1024, 2048, 4096])
def test_clutter_every_transform_length(b2, fft_len):
    g = load_golden("medium")
    n = int(g["params"][1])
    dmin, dmax = (int(v) for v in g["clutter_params"])
    wh = b2.WienerHopf(dmin, dmax, n)
    wh.set_fft_len(fft_len)
    assert wh.fft_len == fft_len
    ok, yf = wh.process(g["x"], g["y"])
    assert ok
    assert np.max(np.abs(yf - g["clutter_y"])) / np.max(np.abs(g["clutter_y"])) <= Y_TOL


def test_clutter_failure_contract(b2):
    # all-zero reference channel: the Toeplitz matrix is singular, the reference's
    # chol() fails, process() returns false and y is untouched (WienerHopf.cpp:111-115)
    n = 20000
    y = (np.arange(n) % 7 + 1j).astype(np.complex128)
    ok, yf = b2.WienerHopf(-3, 20, n).process(np.zeros(n, dtype=np.complex128), y)
    assert not ok and np.array_equal(yf, y)


@pytest.mark.parametrize("name", golden_names())
def test_clutter_golden_half_window_correlation(b2, name):
    """The half-window form of the correlations (2 transforms per F/2 samples, chosen by the planner for
    filters with many taps) forced on the small fixtures, where a workgroup's run is a few segments and
    the wrap-around product is a large share of the lags."""
    g = load_golden(name)
    n = int(g["params"][1])
    dmin, dmax = (int(v) for v in g["clutter_params"])
    if dmax - dmin > 4081:
        pytest.skip("a long filter (chunks of 2048 taps on child handles) has one plan: tests/test_clutter_long_gpu.py")
    wh = b2.WienerHopf(dmin, dmax, n)
    wh.set_corr_form("half")
    ok, yf = wh.process(g["x"], g["y"])
    assert ok == bool(g["clutter_ok"])
    assert np.max(np.abs(yf - g["clutter_y"])) / np.max(np.abs(g["clutter_y"])) <= Y_TOL


@pytest.mark.parametrize("mode,n,taps", [("auto", 300_000, 700), ("window", 300_000, 700), ("half", 300_000, 700),
                                         ("window", 1_000_000, 2047), ("half", 50_000, 300), ("half", 4_099, 1025)])
def test_clutter_correlation_forms_agree_with_the_oracle(b2, mode, n, taps):
    """r, b and the filtered channel from both correlation forms against the fp64 oracle: 700 taps (the
    planner itself picks the half-window form there), 2047 taps on the windowed form (cfg 3 runs the
    other one), a short CPI, and a CPI of two segments and three samples with the widest filter F = 2048 takes."""
    x, y = O.synth_iq(n, seed=n % 97 + taps, fs=1_000_000, targets=((20, 40.0, 0.05),))
    dmin, dmax = -7, taps - 7
    ok_ref, y_ref, w_ref, r_ref, b_ref = O.wiener_hopf(x, y, dmin, dmax, return_filter=True)
    wh = b2.WienerHopf(dmin, dmax, n)
    wh.set_corr_form(mode)
    ok, yf = wh.process(x.astype(np.complex64), y.astype(np.complex64))
    assert ok and ok_ref
    _, w, r, b = wh.read_last(0)
    assert np.max(np.abs(r - r_ref)) / np.abs(r_ref[0]) <= 1e-5
    assert np.max(np.abs(b - b_ref)) / np.max(np.abs(b_ref)) <= 1e-5
    assert np.max(np.abs(yf.astype(np.complex128) - y_ref)) / np.max(np.abs(y_ref)) <= Y_TOL


@pytest.mark.parametrize("n,taps", [(1_000_000, 2047),   # cfg 3's filter: F = 4096, history 2047 -> blocks of 2048, sixteen-points-per-thread kernel
                                    (400_003, 1015),     # F = 2048, history 1015 -> blocks of 1024; a CPI that ends inside a block
                                    (20_000, 2040),      # ten blocks: a workgroup's run is one or two blocks long, most of the grid idle
                                    (2_500, 2047)])      # a CPI shorter than two blocks: every window runs over an end of the CPI
def test_fir_carries_the_window_overlap(b2, n, taps):
    """clutter_fir_kernel with blocks of exactly F/2 samples (filters whose history is just under half the transform): the
    upper half of a block's window is kept in registers as the lower half of the next block's.  The taps (the correlation
    and solve kernels do not change) and the filtered channel within fp32 rounding of the whole-window form
    (BLAH2HIP_CLUTTER_OPT_FIR_CARRY = 0) and of the oracle; three launches on one handle."""
    x, y = O.synth_iq(n, seed=n % 89 + taps, fs=1_000_000, targets=((20, 40.0, 0.05),))
    dmin, dmax = -7, taps - 7
    ok_ref, y_ref = O.wiener_hopf(x, y, dmin, dmax)[:2]
    assert ok_ref
    out = {}
    for carry in (True, False):
        wh = b2.WienerHopf(dmin, dmax, n)
        wh.set_fir_carry(carry)
        for rep in range(3 if carry else 1):
            ok, yf = wh.process(x.astype(np.complex64), y.astype(np.complex64))
            assert ok
            if rep:
                assert np.array_equal(yf, out[carry][0])
            else:
                out[carry] = (yf, wh.read_last(0)[1])
        assert np.max(np.abs(yf.astype(np.complex128) - y_ref)) / np.max(np.abs(y_ref)) <= Y_TOL
    # the taps: the same kernels, but the number of partial correlations per CPI follows the block count (fp32 rounding)
    assert np.max(np.abs(out[True][1] - out[False][1])) <= 1e-5 * np.max(np.abs(out[False][1]))
    assert np.max(np.abs(out[True][0] - out[False][0])) / np.max(np.abs(y_ref)) <= 2e-5


@pytest.mark.parametrize("n,taps,corr", [(300_000, 410, "auto"), (300_000, 700, "half"), (1_000_000, 2047, "auto"), (100_000, 60, "auto")])
def test_clutter_int16_wire_format_equals_fp32_planes(b2, n, taps, corr):
    """blah2hip_clutter_process_dev_fmt(FMT_I16): the correlation and FIR kernels read the .rspduo words (I1 Q1 I2 Q2,
    RspDuo.cpp:512-526) directly.  int16 -> fp32 is exact, so r, b, the taps and the filtered channel must equal the
    fp32-plane path's on the same values bit for bit (same kernels, same order of operations), and the oracle within
    the usual gate; every transform length and both correlation forms are reached by the parameter sets."""
    import torch
    B = 2
    xs, ys = zip(*(O.synth_iq(n, seed=900 + c, fs=1_000_000, targets=((20, 40.0, 0.05),)) for c in range(B)))
    dmin, dmax = -7, taps - 7
    iq = np.stack([np.stack([x.real, x.imag, y.real, y.imag], axis=-1) for x, y in zip(xs, ys)]).astype(np.int16)
    d_iq = torch.from_numpy(iq).cuda()
    dx = torch.from_numpy(np.stack(xs).astype(np.complex64)).cuda()
    dy = torch.from_numpy(np.stack(ys).astype(np.complex64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for fmt in ("c32", "i16"):
        wh = b2.WienerHopf(dmin, dmax, n, max_batch=B)
        wh.set_corr_form(corr)
        yo = torch.zeros((B, n + 3), dtype=torch.complex64, device="cuda")  # an output stride that differs from the input's
        ok = torch.zeros(B, dtype=torch.int32, device="cuda")
        if fmt == "c32":
            wh.process_dev_fmt(b2.FMT_C32, dx.data_ptr(), dy.data_ptr(), B, n, yo.data_ptr(), n + 3, ok.data_ptr(), st)
        else:
            wh.process_dev_fmt(b2.FMT_I16, d_iq.data_ptr(), None, B, n, yo.data_ptr(), n + 3, ok.data_ptr(), st)
        torch.cuda.synchronize()
        assert ok.cpu().tolist() == [1] * B
        outs[fmt] = (yo.cpu().numpy()[:, :n], [wh.read_last(c) for c in range(B)])
    for c in range(B):
        assert np.array_equal(outs["c32"][0][c], outs["i16"][0][c])
        for u, v in zip(outs["c32"][1][c][1:], outs["i16"][1][c][1:]):
            assert np.array_equal(u, v)
        ok_ref, y_ref = O.wiener_hopf(xs[c], ys[c], dmin, dmax)[:2]
        assert ok_ref
        assert np.max(np.abs(outs["i16"][0][c].astype(np.complex128) - y_ref)) / np.max(np.abs(y_ref)) <= Y_TOL
