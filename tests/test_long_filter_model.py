"""CPU: the identities the engine's LONG clutter filters rest on (csrc/clutter.hip long_process), in NumPy against the oracle
(WienerHopf.cpp:58-163): a filter of more taps than one transform holds is run chunk by chunk of C lags / taps --
  b[cC + j] = (the C-lag correlation of xs with y ROTATED by cC)[j]                                   (circular, :100-108)
  r[cC + j] = (the same with xs rotated by cC in y's place)[j]                                          (:76-84)
  y - (w * xs)[0..N) = y - sum_c (w[cC ...] * (xs DELAYED by cC, zeros shifted in))[0..N)               (linear, :125-160)
with xs[i] = x[(uint32(i) - uint32(delayMin)) mod N] (:61-70) -- including a positive first lag, where the reference's
unsigned arithmetic is not a plain rotation."""
import numpy as np
import pytest

from oracle import blah2_oracle as O


def xs_of(x, dmin):
    n = x.size
    i = np.arange(n, dtype=np.uint64)
    idx = ((i - np.uint64(dmin & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)) % np.uint64(n)
    return x[idx.astype(np.int64)]


def corr_lags(y, xs, C):
    """The child handle's b: sum_n y[n] conj(xs[(n - k) mod N]) for k < C."""
    n = y.size
    return np.array([np.sum(y * np.conj(np.roll(xs, k))) for k in range(C)]) / 1.0


@pytest.mark.parametrize("dmin,nbins,n,C", [(-3, 50, 211, 16), (0, 37, 150, 16), (2, 45, 187, 8), (-7, 64, 256, 32)])
def test_chunked_correlations_and_fir(dmin, nbins, n, C):
    rng = np.random.default_rng(nbins)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    y = 0.7 * np.roll(x, 3) + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    ok, yf_ref, w_ref, r_ref, b_ref = O.wiener_hopf(x, y, dmin, dmin + nbins, return_filter=True)
    assert ok
    xs = xs_of(x, dmin)
    r = np.zeros(nbins, dtype=np.complex128)
    b = np.zeros(nbins, dtype=np.complex128)
    chunks = -(-nbins // C)
    for c in range(chunks):
        cnt = min(C, nbins - c * C)
        b[c * C:c * C + cnt] = corr_lags(np.roll(y, -c * C), xs, C)[:cnt]          # long_plane_kernel<0>
        r[c * C:c * C + cnt] = corr_lags(np.roll(xs, -c * C), xs, C)[:cnt]         # long_plane_kernel<1>
    # the oracle's r, b are the unnormalised sums divided by nothing here: compare up to its own scale
    scale = r_ref[0] / r[0]
    assert abs(scale.imag) < 1e-12 * abs(scale)
    assert np.max(np.abs(r * scale - r_ref)) <= 1e-10 * abs(r_ref[0])
    assert np.max(np.abs(b * scale - b_ref)) <= 1e-10 * abs(r_ref[0])
    # the FIR as a sum over chunks of taps on xs delayed by cC (long_plane_kernel<2>: zeros shifted in, no wrap)
    out = y.copy()
    for c in range(chunks):
        wc = np.zeros(C, dtype=np.complex128)
        cnt = min(C, nbins - c * C)
        wc[:cnt] = w_ref[c * C:c * C + cnt]
        xd = np.zeros(n, dtype=np.complex128)
        xd[c * C:] = xs[:n - c * C]
        out -= np.convolve(wc, xd)[:n]
    assert np.max(np.abs(out - yf_ref)) <= 1e-10 * np.max(np.abs(yf_ref))
