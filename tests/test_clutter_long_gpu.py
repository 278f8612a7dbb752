"""GPU: clutter filters of more than 4081 taps (csrc/clutter.hip, "LONG filters"; WienerHopf.cpp:58-163 takes any nBins).

One on-chip transform holds 4081 taps.  Beyond, the engine runs the same kernels chunk by chunk of 2048 lags / taps on
rotated and shifted copies of the channels and solves the normal equations in one workgroup on vectors in global memory.
Checked against the oracle's fp64 chain (taps, normal equations, filtered channel), the compiled reference's fixture
`long_filter` (4610 taps; in tests/test_clutter_gpu.py with every other fixture), batches with a stride, a failed solve,
a positive first lag (the reference's unsigned index arithmetic), the solve on its own against LAPACK, and the refusals.
"""
import numpy as np
import pytest

from oracle import blah2_oracle as O

pytestmark = pytest.mark.gpu
Y_TOL = 1e-4


@pytest.fixture(scope="module")
def b2(built_lib):
    import blah2_amd
    assert blah2_amd.device_count() > 0
    return blah2_amd


def channels(n, seed, taps_at=((0, 0.8), (700, 0.2), (3000, 0.1), (5000, 0.05)), noise=30.0, dmin=0):
    """x white; y = sum_k a_k xs[n - k] + noise with xs the filter's own shifted reference: clutter across the whole window."""
    rng = np.random.default_rng(seed)
    x = np.round((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 300.0)
    y = noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    for k, a in taps_at:
        y = y + a * np.roll(x, k + dmin)
    return x.astype(np.complex64), np.round(y).astype(np.complex64)


@pytest.mark.parametrize("dmin,dmax,n", [(-10, 4600, 60_000), (0, 9000, 40_000), (-3, 4082 - 3, 50_000), (2, 6002, 64_000)])
def test_long_filter_against_the_oracle(b2, dmin, dmax, n):
    nb = dmax - dmin
    x, y = channels(n, seed=nb, dmin=dmin)
    wh = b2.WienerHopf(dmin, dmax, n)
    assert wh.nBins == nb and wh.fft_len == 4096
    ok, yf = wh.process(x, y)
    okr, yfr, w_ref, r_ref, b_ref = O.wiener_hopf(x.astype(np.complex128), y.astype(np.complex128), dmin, dmax, return_filter=True)
    assert ok and okr
    _, w, r, b = wh.read_last()
    # the normal equations (fp32 correlations, fp64 reduction) and the taps
    assert np.max(np.abs(r - r_ref)) <= 2e-6 * abs(r_ref[0]) and np.max(np.abs(b - b_ref)) <= 2e-6 * abs(r_ref[0])
    assert O.toeplitz_residual(r, w.astype(np.complex128), b) <= 1e-5
    assert np.max(np.abs(w - w_ref)) <= 2e-5 * np.max(np.abs(w_ref))
    err = np.max(np.abs(yf.astype(np.complex128) - yfr)) / np.max(np.abs(yfr))
    print(f"\n[long] {nb} taps from lag {dmin}, {n} samples: filtered-channel error {err:.2e}, |y'| / |y| = "
          f"{np.linalg.norm(yfr) / np.linalg.norm(y):.3f}")
    assert err <= Y_TOL
    wh.close()


def test_batch_with_a_stride_in_place_and_a_failed_solve(b2):
    import torch
    dmin, dmax, n, B, stride = -10, 4600, 60_000, 3, 60_000 + 4096
    data = [channels(n, seed=40 + c, dmin=dmin) for c in range(B)]
    xb = np.zeros((B, stride), dtype=np.complex64)
    yb = np.zeros((B, stride), dtype=np.complex64)
    for c in range(B):
        xb[c, :n], yb[c, :n] = data[c]
    xb[1] = 0  # all-zero reference: chol() fails (WienerHopf.cpp:111-115), the surveillance channel passes through
    x, y = torch.from_numpy(xb).cuda(), torch.from_numpy(yb).cuda()
    ok = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    wh = b2.WienerHopf(dmin, dmax, n, max_batch=B)
    st = torch.cuda.current_stream().cuda_stream
    wh.process_dev(x.data_ptr(), y.data_ptr(), B, stride, y.data_ptr(), ok.data_ptr(), st)  # in place
    torch.cuda.synchronize()
    assert ok.cpu().tolist() == [1, 0, 1]
    out = y.cpu().numpy()
    assert np.array_equal(out[1], yb[1]) and not out[:, n:].any()
    for c in (0, 2):
        ref = O.wiener_hopf(data[c][0].astype(np.complex128), data[c][1].astype(np.complex128), dmin, dmax)[1]
        assert np.max(np.abs(out[c, :n] - ref)) / np.max(np.abs(ref)) <= Y_TOL
    # the estimate alone leaves the same taps (the fused range kernel does not take 4610 of them; a caller may)
    _, w_full, _, _ = wh.read_last(0)
    wh.estimate_dev_fmt(b2.FMT_C32, x.data_ptr(), torch.from_numpy(yb).cuda().data_ptr(), B, stride, ok.data_ptr(), st)
    torch.cuda.synchronize()
    _, w_est, _, _ = wh.read_last(0)
    assert np.array_equal(w_full, w_est)
    wh.close()


def test_the_solve_on_its_own_against_lapack(b2):
    import scipy.linalg as sla
    n = 6000
    rng = np.random.default_rng(6)
    # a positive definite Hermitian Toeplitz matrix: the autocorrelation of a short random sequence + a ridge
    h = rng.standard_normal(40) + 1j * rng.standard_normal(40)
    r = np.zeros(n, dtype=np.complex128)
    ac = np.correlate(h, h, mode="full")[39:]
    r[:40] = ac
    r[0] += 0.5 * abs(ac[0])
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    wh = b2.WienerHopf(0, n, 20_000)
    ok, w = wh.solve(r[None], b[None])
    assert ok[0]
    # A[i][j] = conj(r[i-j]) for i >= j (WienerHopf.cpp:85-97: a = conj(r), then the strict lower triangle conjugated again)
    assert O.toeplitz_residual(r, w[0].astype(np.complex128), b) <= 1e-5
    a = np.conj(r)
    A = sla.toeplitz(a, a)
    low = np.tril_indices(n, -1)
    A[low] = np.conj(A[low])
    w_ref = np.linalg.solve(A, b)
    assert np.max(np.abs(w[0] - w_ref)) <= 1e-5 * np.max(np.abs(w_ref))
    # not positive definite: ok = 0, taps zero
    r_bad = r.copy()
    r_bad[1] = 2.0 * r_bad[0]
    ok, w = wh.solve(r_bad[None], b[None])
    assert not ok[0] and not w.any()
    wh.close()


def test_refusals(b2):
    with pytest.raises(b2.Blah2HipError) as e:
        b2.WienerHopf(0, 5000, 4000)  # more taps than samples: the reference reads its correlation lags out of bounds
    assert e.value.code == b2._lib.ERR_UNSUPPORTED
    import torch
    wh = b2.WienerHopf(-10, 4600, 60_000)
    iq = torch.zeros((60_000, 4), dtype=torch.int16, device="cuda")
    yf = torch.zeros(60_000, dtype=torch.complex64, device="cuda")
    with pytest.raises(b2.Blah2HipError) as e:  # the int16 words: fp32 planes only on this path
        wh.process_dev_fmt(b2.FMT_I16, iq.data_ptr(), None, 1, 60_000, yf.data_ptr(), 60_000)
    assert e.value.code == b2._lib.ERR_UNSUPPORTED
    with pytest.raises(b2.Blah2HipError):
        wh.set_fft_len(2048)
    wh.close()
