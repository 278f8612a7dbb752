"""CPU: randomized self-consistency of the oracle (hypothesis) -- the checker must be
right before it checks anything.  Fast forms against their definitional forms, and
the derived sizes against the invariants the reference's constructor guarantees."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import blah2_oracle as O
from oracle import ref_lib as R

FAST = settings(max_examples=60, deadline=None)


def is_hamming(v):
    for p in (2, 3, 5):
        while v % p == 0:
            v //= p
    return v == 1


@FAST
@given(st.integers(min_value=0, max_value=200_000))
def test_next_hamming_is_the_smallest_5_smooth_number_above(v):
    # HammingNumber.cpp:38-48: strictly greater, 2^a 3^b 5^c, none in between
    h = O.next_hamming(v)
    assert h > v and is_hamming(h)
    assert not any(is_hamming(k) for k in range(v + 1, h))
    if R.available():
        assert R.next_hamming(v) == h


@FAST
@given(st.integers(-64, 1), st.integers(0, 200), st.integers(1, 300), st.integers(5_000, 400_000), st.booleans())
def test_ambiguity_dims_invariants(delay_min, span, doppler_max, n, round_hamming):
    # Ambiguity.cpp:11-82 for symmetric Doppler limits (every shipped configuration)
    fs = 200_000
    delay_max = max(delay_min + span, -1)
    n_doppler = 2 * (doppler_max * n // fs) + 1
    # nCorr is a uint16_t in the reference (Ambiguity.h:110): n / nDopplerBins wraps mod 65536,
    # and a wrapped value of 0 divides by zero at :43 -- outside any usable configuration
    if (n // n_doppler) & 0xFFFF == 0:
        return
    d = O.ambiguity_dims(delay_min, delay_max, -doppler_max, doppler_max, fs, n, round_hamming)
    assert d.n_delay_bins == delay_max - delay_min + 1
    assert d.n_doppler_bins % 2 == 1
    assert d.n_doppler_bins == n_doppler
    assert d.n_corr == (n // d.n_doppler_bins) & 0xFFFF and d.n_corr * d.n_doppler_bins <= n
    assert d.nfft == (O.next_hamming(2 * d.n_corr - 1) if round_hamming else 2 * d.n_corr - 1)
    assert d.delay[0] == delay_min and d.delay[-1] == delay_max
    assert abs(d.doppler[d.n_doppler_bins // 2]) < 1e-9  # centre bin is 0 Hz
    assert np.allclose(np.diff(d.doppler), 1.0 / d.cpi)


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2**32 - 1), st.integers(-6, 1), st.integers(2, 24), st.integers(1, 6))
def test_fft_form_equals_the_defining_sums(seed, delay_min, span, doppler_max):
    # the restatement used as the oracle (three FFTs per pulse) against the literal
    # double sum of Ambiguity.cpp:106-169 on small random problems
    rng = np.random.default_rng(seed)
    fs, n = 1000, int(rng.integers(300, 700))
    delay_max = max(delay_min + span, -1)
    d = O.ambiguity_dims(delay_min, delay_max, -doppler_max, doppler_max, fs, n, bool(seed & 1))
    if d.n_corr < 8 or d.n_delay_bins >= d.n_corr:
        return
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    y = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    a, b = O.ambiguity_process(d, x, y), O.ambiguity_process_direct(d, x, y)
    assert np.max(np.abs(a - b)) <= 1e-10 * np.max(np.abs(b))


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2**32 - 1), st.integers(0, 3), st.integers(1, 7), st.integers(0, 4))
def test_cfar_fast_forms_equal_their_definitions(seed, n_guard, n_train, min_delay):
    rng = np.random.default_rng(seed)
    nD, nC = int(rng.integers(3, 24)) | 1, int(rng.integers(4, 60))
    m = (rng.standard_normal((nD, nC)) + 1j * rng.standard_normal((nD, nC))) * rng.choice([1.0, 30.0], size=(nD, nC), p=[0.95, 0.05])
    delay = np.arange(-2, nC - 2)
    doppler = (np.arange(nD) - nD // 2) * 2.5
    noise, _ = O.map_metrics(m)
    pfa = float(rng.choice([1e-1, 1e-2, 1e-4]))
    a = O.cfar1d(m, delay, doppler, noise, pfa, n_guard, n_train, min_delay, 3.0)
    b = O.cfar1d_fast(m, delay, doppler, noise, pfa, n_guard, n_train, min_delay, 3.0)
    assert all(np.array_equal(u, v) for u, v in zip(a[:2], b[:2])) and np.allclose(a[2], b[2], rtol=0, atol=1e-12)
    ng_f, nt_f = int(rng.integers(0, 3)), int(rng.integers(0, 4))
    c = O.cfar2d_bruteforce(m, delay, doppler, noise, pfa, n_guard, n_train, ng_f, nt_f, min_delay, 3.0)
    e = O.cfar2d(m, delay, doppler, noise, pfa, n_guard, n_train, ng_f, nt_f, min_delay, 3.0)
    assert all(np.array_equal(u, v) for u, v in zip(c[:2], e[:2]))
    if ng_f == 0 and nt_f == 0:  # the 2-D extension collapses to CfarDetector1D.cpp:23-100
        assert all(np.array_equal(u, v) for u, v in zip(a[:2], e[:2]))


@FAST
@given(st.integers(2_000, 300_000), st.sampled_from([400.0, 1896.0, 2000.0, 2001.0]))
def test_spectrum_dims_invariants(n, bw):
    # SpectrumAnalyser.cpp:16-18 (n < bandwidth makes the reference divide by a zero decimation)
    if n < bw:
        return
    dec, ns, nfft = O.spectrum_dims(n, bw)
    assert dec == int(n / bw) and dec >= 1
    assert ns == n // dec and nfft == ns * dec <= n and n - nfft < dec
