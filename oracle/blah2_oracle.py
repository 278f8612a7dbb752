"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the blah2 cross-ambiguity hot path.

fp64 NumPy restatement of the reference's algorithm.  Nothing in the product
(`blah2_amd/`) may import this module; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s ``cpu_baseline`` leg do, and there
only as the checker.

Parity pin: this restatement is checked in ``tests/test_oracle.py`` against
  * the reference's own known answers
      - test/unit/process/ambiguity/TestAmbiguity.cpp:87-92, 110-115
      - test/unit/process/meta/TestHammingNumber.cpp:15-17
  * outputs of the reference's OWN sources compiled here (``oracle/_ref``,
    built by ``oracle/Makefile`` from /root/reference/src), committed as
    fixtures under ``tests/golden/`` by ``tests/golden/make_golden.py``.
The only map-value known answer the reference ships
(TestAmbiguity.cpp:176-177) needs a capture file that is not in the
repository, so map values are pinned by the compiled reference, not by a
reference-shipped vector.

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

try:  # scipy's pocketfft accepts any length and is faster than numpy.fft
    from scipy import fft as _fft
except Exception:  # pragma: no cover
    from numpy import fft as _fft

C_LIGHT = 299792458  # src/data/meta/Constants.h:13 (uint32_t)


# --------------------------------------------------------------------------
# src/process/meta/HammingNumber.cpp:38-48
def next_hamming(value: int) -> int:
    """Smallest 5-smooth integer STRICTLY greater than ``value``."""
    best = None
    p2 = 1
    limit = max(2 * value + 2, 2)
    while p2 <= limit:
        p23 = p2
        while p23 <= limit:
            p235 = p23
            while p235 <= limit:
                if p235 > value and (best is None or p235 < best):
                    best = p235
                p235 *= 5
            p23 *= 3
        p2 *= 2
    return int(best)


# --------------------------------------------------------------------------
@dataclass
class AmbiguityDims:
    """Derived sizes of ``Ambiguity::Ambiguity`` (Ambiguity.cpp:11-82)."""

    delay_min: int
    delay_max: int
    doppler_min: int
    doppler_max: int
    fs: int
    n_samples: int
    round_hamming: bool
    n_delay_bins: int = 0
    n_doppler_bins: int = 0
    n_corr: int = 0
    nfft: int = 0
    cpi: float = 0.0
    doppler_middle: float = 0.0
    delay: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int64))
    doppler: np.ndarray = field(default_factory=lambda: np.zeros(0))


def ambiguity_dims(delay_min, delay_max, doppler_min, doppler_max, fs, n, round_hamming=False, n_doppler_bins=0):
    d = AmbiguityDims(int(delay_min), int(delay_max), int(doppler_min), int(doppler_max),
                      int(fs), int(n), bool(round_hamming))
    # :22 uint16 narrowing
    d.n_delay_bins = (d.delay_max - d.delay_min + 1) & 0xFFFF
    d.doppler_middle = (d.doppler_min + d.doppler_max) / 2.0  # :23
    # :26-36 count bins with the nominal resolution fs/n; same fp64 expression
    res = 1.0 / (float(d.n_samples) / float(d.fs))
    i = 1
    while d.doppler_middle + (i * res) <= d.doppler_max:
        i += 1
    d.n_doppler_bins = (2 * (i - 1) + 1) & 0xFFFF
    if n_doppler_bins:
        # extension (not reachable through the reference's constructor): an explicit
        # bin count with the reference's arithmetic below unchanged; for an even count the axis
        # gets its extra bin at the end, where (j + nD//2 + 1) % nD of :165 puts Nyquist
        d.n_doppler_bins = int(n_doppler_bins) & 0xFFFF
    d.n_corr = (d.n_samples // d.n_doppler_bins) & 0xFFFF  # :39
    d.cpi = (float(d.n_corr) * d.n_doppler_bins) / d.fs  # :40
    res = 1.0 / d.cpi  # :43
    d.delay = np.arange(d.delay_min, d.delay_min + d.n_delay_bins, dtype=np.int64)  # :49-50
    h = (d.n_doppler_bins - 1) // 2
    # :52-59  mid -/+ i*res, evaluated exactly as the reference does (i*res, then add)
    k = np.arange(-h, d.n_doppler_bins - h, dtype=np.float64)
    d.doppler = d.doppler_middle + k * res
    d.nfft = 2 * d.n_corr - 1  # :62
    if d.round_hamming:
        d.nfft = next_hamming(d.nfft)  # :63-65
    return d


# --------------------------------------------------------------------------
def ambiguity_process(dims: AmbiguityDims, x, y, workers=None):
    """``Ambiguity::process`` (Ambiguity.cpp:92-172).

    x = reference, y = surveillance, complex, at least n_corr*n_doppler_bins
    samples.  Returns the complex128 map [n_doppler_bins, n_delay_bins].
    ``workers``: pocketfft threads over the batch of pulses (the reference plans FFTW
    with 4 threads, blah2.cpp:115-120); results do not depend on it.
    """
    x = np.asarray(x, dtype=np.complex128)
    y = np.asarray(y, dtype=np.complex128)
    nD, nC, nfft = dims.n_doppler_bins, dims.n_corr, dims.nfft
    if dims.doppler_middle != 0:  # :95-102, applied to the reference channel, sign +
        i = np.arange(x.shape[0], dtype=np.float64)
        x = x * np.exp(1j * 2.0 * np.pi * dims.doppler_middle * (i / dims.fs))
    used = nD * nC  # :105
    X = x[:used].reshape(nD, nC)
    Y = y[:used].reshape(nD, nC)
    # :108-129 zero-pad each pulse to nfft, Y*conj(X)/nfft, unnormalised backward FFT
    kw = {"workers": workers} if workers else {}
    FX = _fft.fft(X, n=nfft, axis=1, **kw)
    FY = _fft.fft(Y, n=nfft, axis=1, **kw)
    Z = _fft.ifft(FY * np.conj(FX), axis=1, **kw)  # = backward(FY*conj(FX)/nfft)
    # :132-146 nets to R[i][j] = z[(delayMin + j) mod nfft]
    lag = (dims.delay_min + np.arange(dims.n_delay_bins)) % nfft
    R = Z[:, lag]
    # :152-169 forward FFT over pulses, out[j] = D[(j + nD//2 + 1) % nD]
    D = _fft.fft(R, axis=0, **kw)
    sel = (np.arange(nD) + nD // 2 + 1) % nD
    return D[sel, :]


def ambiguity_process_direct(dims: AmbiguityDims, x, y):
    """Same quantity by its time-domain definition;
    O(nD*nCorr*nDelay), small cases only.  Independent of any FFT."""
    x = np.asarray(x, dtype=np.complex128)
    y = np.asarray(y, dtype=np.complex128)
    nD, nC = dims.n_doppler_bins, dims.n_corr
    R = np.zeros((nD, dims.n_delay_bins), dtype=np.complex128)
    for i in range(nD):
        xs = x[i * nC:(i + 1) * nC]
        ys = y[i * nC:(i + 1) * nC]
        for j, d in enumerate(range(dims.delay_min, dims.delay_min + dims.n_delay_bins)):
            if d >= 0:
                R[i, j] = np.sum(ys[d:] * np.conj(xs[:nC - d])) if d < nC else 0
            else:
                R[i, j] = np.sum(ys[:nC + d] * np.conj(xs[-d:])) if -d < nC else 0
    k = np.arange(nD)
    W = np.exp(-2j * np.pi * np.outer(k, k) / nD)
    D = W @ R
    sel = (np.arange(nD) + nD // 2 + 1) % nD
    return D[sel, :]


# --------------------------------------------------------------------------
def map_metrics(m):
    """``Map::set_metrics`` (src/data/Map.cpp:187-206) -> (noisePower, maxPower)."""
    v = 10.0 * np.log10(np.abs(np.asarray(m, dtype=np.complex128)))
    noise = float(np.sum(v) / v.size)
    peak = max(0.0, float(np.max(v)))  # the running max starts at 0 (:193)
    return noise, peak - noise


def map_db(m, noise_power):
    """cell values of ``Map::to_json`` (src/data/Map.cpp:126): 10log10|z| - noisePower."""
    return 10.0 * np.log10(np.abs(np.asarray(m, dtype=np.complex128))) - noise_power


# --------------------------------------------------------------------------
def cfar1d(m, delay_axis, doppler_axis, noise_power, pfa, n_guard, n_train, min_delay, min_doppler):
    """``CfarDetector1D::process`` (src/process/detection/CfarDetector1D.cpp:23-100).

    Returns (delay, doppler, snr) float64 arrays in the reference's row-major
    emission order.
    """
    m = np.asarray(m, dtype=np.complex128)
    nD, nC = m.shape
    out_delay, out_doppler, out_snr = [], [], []
    for i in range(nD):
        if abs(doppler_axis[i]) < min_doppler:  # :40
            continue
        row = m[i]
        sq = np.abs(row * row)  # :47
        snr = 10.0 * np.log10(np.abs(row)) - noise_power  # :48
        for j in range(nC):
            if delay_axis[j] < min_delay:  # :53
                continue
            idx = [k for k in range(j - n_guard - n_train, j - n_guard) if 0 < k < nC]  # :59-65 (k > 0)
            idx += [k for k in range(j + n_guard + 1, j + n_guard + n_train + 1) if 0 <= k < nC]  # :66-72
            n_cells = len(idx)
            if n_cells == 0:
                # :76-82: alpha = 0*(pfa^(-inf)-1) = 0*inf = nan; 0/0 = nan; nothing exceeds nan
                continue
            alpha = n_cells * (math.pow(pfa, -1.0 / n_cells) - 1)  # :76
            noise = 0.0
            for k in idx:  # :78-81 sequential fp64 sum in index order
                noise += sq[k]
            noise /= n_cells
            if sq[j] > alpha * noise:  # :86
                out_delay.append(float(j + delay_axis[0]))
                out_doppler.append(float(doppler_axis[i]))
                out_snr.append(float(snr[j]))
    return np.array(out_delay), np.array(out_doppler), np.array(out_snr)


def cfar1d_fast(m, delay_axis, doppler_axis, noise_power, pfa, n_guard, n_train, min_delay, min_doppler):
    """Vectorised form of :func:`cfar1d` for big maps (prefix sums instead of the
    sequential window sum, so thresholds agree to ~1e-13 relative, not bitwise)."""
    m = np.asarray(m, dtype=np.complex128)
    nD, nC = m.shape
    sq = np.abs(m * m)
    j = np.arange(nC)
    lo0 = np.clip(j - n_guard - n_train, 1, nC)  # k > 0
    lo1 = np.clip(j - n_guard, 1, nC)
    hi0 = np.clip(j + n_guard + 1, 0, nC)
    hi1 = np.clip(j + n_guard + n_train + 1, 0, nC)
    lo1 = np.maximum(lo1, lo0)
    hi1 = np.maximum(hi1, hi0)
    n_cells = (lo1 - lo0) + (hi1 - hi0)
    cs = np.concatenate([np.zeros((nD, 1)), np.cumsum(sq, axis=1)], axis=1)
    tot = (cs[:, lo1] - cs[:, lo0]) + (cs[:, hi1] - cs[:, hi0])
    with np.errstate(divide="ignore", invalid="ignore"):
        alpha = n_cells * (np.power(pfa, -1.0 / n_cells) - 1)
        thr = alpha * (tot / n_cells)
    hit = sq > thr
    hit &= (np.asarray(delay_axis) >= min_delay)[None, :]
    hit &= (np.abs(np.asarray(doppler_axis)) >= min_doppler)[:, None]
    ii, jj = np.nonzero(hit)
    snr = 10.0 * np.log10(np.abs(m[ii, jj])) - noise_power
    return (jj + delay_axis[0]).astype(np.float64), np.asarray(doppler_axis)[ii].astype(np.float64), snr


# --------------------------------------------------------------------------
# 2-D cell-averaging CFAR.  The reference has only the 1-D detector; BASELINE.json
# configs[2] asks for a 2-D one, defined in SURVEY.md section 8g as the direct
# extension of CfarDetector1D.cpp:23-100:
#   * training cells: the (2(nGd+nTd)+1) x (2(nGf+nTf)+1) rectangle centred on the
#     cell under test minus the (2nGd+1) x (2nGf+1) guard box, in-bounds cells only;
#   * delay column 0 never trains (the 1-D detector's `k > 0`, :61);
#   * statistic |z|^2, alpha = N (pfa^(-1/N) - 1) with N the in-bounds training count;
#   * the same minDelay / minDoppler skips and row-major emission order;
# so that with nGf = nTf = 0 it is exactly the 1-D detector.
def cfar2d_bruteforce(m, delay_axis, doppler_axis, noise_power, pfa, ng_d, nt_d, ng_f, nt_f, min_delay, min_doppler):
    m = np.asarray(m, dtype=np.complex128)
    nD, nC = m.shape
    sq = np.abs(m * m)
    out = []
    for i in range(nD):
        if abs(doppler_axis[i]) < min_doppler:
            continue
        for j in range(nC):
            if delay_axis[j] < min_delay:
                continue
            tot, n = 0.0, 0
            for ii in range(max(0, i - ng_f - nt_f), min(nD, i + ng_f + nt_f + 1)):
                for kk in range(max(1, j - ng_d - nt_d), min(nC, j + ng_d + nt_d + 1)):
                    if abs(ii - i) <= ng_f and abs(kk - j) <= ng_d:
                        continue
                    tot += sq[ii, kk]
                    n += 1
            if n == 0:
                continue
            alpha = n * (math.pow(pfa, -1.0 / n) - 1)
            if sq[i, j] > alpha * (tot / n):
                out.append((float(j + delay_axis[0]), float(doppler_axis[i]),
                            float(10.0 * np.log10(np.abs(m[i, j])) - noise_power)))
    a = np.array(out).reshape(-1, 3)
    return a[:, 0], a[:, 1], a[:, 2]


def cfar2d(m, delay_axis, doppler_axis, noise_power, pfa, ng_d, nt_d, ng_f, nt_f, min_delay, min_doppler,
           return_margin=False):
    """Summed-area-table form of :func:`cfar2d_bruteforce` (fp64)."""
    m = np.asarray(m, dtype=np.complex128)
    nD, nC = m.shape
    sq = np.abs(m * m)
    z = sq.copy()
    z[:, 0] = 0.0  # column 0 never trains
    sat = np.zeros((nD + 1, nC + 1))
    sat[1:, 1:] = np.cumsum(np.cumsum(z, axis=0), axis=1)

    def box(r0, r1, c0, c1):  # half-open [r0,r1) x [c0,c1), already clipped
        return sat[r1][:, c1] - sat[r0][:, c1] - sat[r1][:, c0] + sat[r0][:, c0]

    i = np.arange(nD)
    j = np.arange(nC)
    R0, R1 = np.clip(i - ng_f - nt_f, 0, nD), np.clip(i + ng_f + nt_f + 1, 0, nD)
    G0, G1 = np.clip(i - ng_f, 0, nD), np.clip(i + ng_f + 1, 0, nD)
    C0, C1 = np.clip(j - ng_d - nt_d, 0, nC), np.clip(j + ng_d + nt_d + 1, 0, nC)
    H0, H1 = np.clip(j - ng_d, 0, nC), np.clip(j + ng_d + 1, 0, nC)
    tot = box(R0, R1, C0, C1) - box(G0, G1, H0, H1)
    cols = lambda a, b: np.maximum(b, 1) - np.maximum(a, 1)  # columns >= 1 in [a, b)
    n = np.outer(R1 - R0, cols(C0, C1)) - np.outer(G1 - G0, cols(H0, H1))
    with np.errstate(divide="ignore", invalid="ignore"):
        alpha = n * (np.power(pfa, -1.0 / n) - 1)
        thr = alpha * (tot / n)
    hit = (sq > thr) & (n > 0)
    hit &= (np.asarray(delay_axis) >= min_delay)[None, :]
    hit &= (np.abs(np.asarray(doppler_axis)) >= min_doppler)[:, None]
    ii, jj = np.nonzero(hit)
    snr = 10.0 * np.log10(np.abs(m[ii, jj])) - noise_power
    res = ((jj + delay_axis[0]).astype(np.float64), np.asarray(doppler_axis)[ii].astype(np.float64), snr)
    if return_margin:
        with np.errstate(divide="ignore", invalid="ignore"):
            return res + (sq / thr,)
    return res


# --------------------------------------------------------------------------
def spectrum_dims(n, bandwidth):
    """``SpectrumAnalyser::SpectrumAnalyser`` (src/process/spectrum/SpectrumAnalyser.cpp:9-24):
    (decimation, nSpectrum, nfft); decimation is the double quotient truncated to uint32."""
    decimation = int(n / float(bandwidth))  # :16
    n_spectrum = n // decimation            # :17
    return decimation, n_spectrum, n_spectrum * decimation  # :18


def spectrum_process(x, n, bandwidth):
    """``SpectrumAnalyser::process`` (SpectrumAnalyser.cpp:31-71): returns
    (spectrum, frequency).  spectrum[k] = FFT(x[:nfft])[(k*decimation + nfft//2 + 1) % nfft]
    (:43-54).  The frequency loop's uint32 counter starts at (2^32 - nSpectrum)//2
    (``-nSpectrum/2`` on an unsigned), which is never < nSpectrum//2: the axis is empty (:64)."""
    decimation, n_spectrum, nfft = spectrum_dims(n, bandwidth)
    X = _fft.fft(np.asarray(x, dtype=np.complex128)[:nfft])
    i = np.arange(0, nfft, decimation, dtype=np.int64)
    spectrum = X[(i + nfft // 2 + 1) % nfft]
    frequency = np.empty(0, dtype=np.float64)
    start = ((1 << 32) - n_spectrum) // 2
    assert not start < n_spectrum // 2
    return spectrum, frequency


# --------------------------------------------------------------------------
def wiener_hopf(x, y, delay_min, delay_max, return_filter=False):
    """``WienerHopf::process`` (src/process/clutter/WienerHopf.cpp:58-163).

    Returns (ok, y_filtered).  x is not modified.  ``nBins = delayMax-delayMin``
    (no +1, :12).  With ``return_filter`` also the taps w, the matrix's first
    column r (A[i][j] = r[i-j]) and the right-hand side b of A w = b.
    """
    import scipy.linalg as sla

    x = np.asarray(x, dtype=np.complex128)
    y = np.asarray(y, dtype=np.complex128)
    n = x.shape[0]
    n_bins = delay_max - delay_min
    # :67  dataX[i] = x[(i - delayMin) mod N]; the reference evaluates
    # (i - delayMin) in uint32, which for delayMin <= 0 is the plain roll below.
    i = np.arange(n, dtype=np.uint64)
    idx = ((i - np.uint64(delay_min & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)) % np.uint64(n)
    xs = x[idx.astype(np.int64)]
    FX = _fft.fft(xs)
    FY = _fft.fft(y)
    # :76-84  a[k] = conj(backward(|X|^2)[k]) / N ;  backward = N * ifft
    r = _fft.ifft(FX * np.conj(FX))[:n_bins]
    a = np.conj(r)
    # :85-97  A = toeplitz(a) (symmetric), then conj of the strict lower triangle
    A = sla.toeplitz(a, a)
    low = np.tril_indices(n_bins, -1)
    A[low] = np.conj(A[low])
    # :100-108
    b = _fft.ifft(FY * np.conj(FX))[:n_bins]
    # :111-122  upper Cholesky, two triangular solves
    try:
        U = sla.cholesky(A, lower=False)
    except np.linalg.LinAlgError:
        return (False, y.copy(), None, r, b) if return_filter else (False, y.copy())
    t = sla.solve_triangular(U.conj().T, b, lower=True)
    w = sla.solve_triangular(U, t, lower=False)
    # :125-160  y - (w * xs)[0:N]  (linear convolution; the FFT length is immaterial)
    L = n_bins + n + 1
    filt = _fft.ifft(_fft.fft(xs, n=L) * _fft.fft(w, n=L))[:n]
    if return_filter:
        return True, y - filt, w, r, b
    return True, y - filt


def toeplitz_residual(r, w, b):
    """||A w - b|| / ||b|| for the Hermitian Toeplitz A[i][j] = r[i-j] (r[-k] = conj r[k]) of
    WienerHopf.cpp:85-97, evaluated with an FFT product in fp64."""
    r = np.asarray(r, dtype=np.complex128)
    w = np.asarray(w, dtype=np.complex128)
    n = r.shape[0]
    L = 1 << int(np.ceil(np.log2(2 * n)))
    c = np.zeros(L, dtype=np.complex128)
    c[:n] = r
    c[L - n + 1:] = np.conj(r[1:][::-1])
    Aw = _fft.ifft(_fft.fft(c) * _fft.fft(w, n=L))[:n]
    return float(np.linalg.norm(Aw - b) / np.linalg.norm(b))


# --------------------------------------------------------------------------
def synth_iq(n, seed=20240925, targets=((37, -63.0, 0.05),), fs=2_000_000, direct=0.8,
             ref_amp=300.0, noise_amp=30.0, quantise=True):
    """Seeded synthetic reference/surveillance pair.

    y = direct*x + sum_k a_k * x[n - d_k] * exp(j 2 pi f_k n / fs) + noise,
    rounded and clipped to int16 like the .rspduo wire format
    (src/capture/rspduo/RspDuo.cpp:512-526) when ``quantise``.
    Returns complex128 arrays (integer-valued when quantised).
    """
    rng = np.random.default_rng(seed)
    x = ref_amp * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    y = direct * x
    t = np.arange(n) / fs
    for d, f, a in targets:
        xd = np.roll(x, d)
        if d > 0:
            xd[:d] = 0
        y = y + a * xd * np.exp(2j * np.pi * f * t)
    y = y + noise_amp * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    if quantise:
        def q(v):
            return np.clip(np.rint(v.real), -32768, 32767) + 1j * np.clip(np.rint(v.imag), -32768, 32767)
        x, y = q(x), q(y)
    return x, y
