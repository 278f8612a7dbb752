"""GPU: size-independent properties of the hot path at BASELINE.json's full
configs[1] size (2 MS/s, 1 s CPI, 513 x 411), where a per-cell comparison with
the fp64 oracle is already done once (test_ambiguity_gpu.py) and these checks add
what the algebra of the reference's algorithm guarantees for ANY input:

  * M is linear in the surveillance channel and conjugate-linear in the reference
    channel (Ambiguity.cpp:106-169 is a chain of linear maps on y and conj(x));
  * x correlated with itself puts the pulse energies on the zero-delay column:
    M[f = 0][d = 0] = sum |x|^2 over the samples the map uses (known answer in fp64);
  * a delayed, Doppler-shifted copy peaks at exactly that delay and Doppler bin with
    the amplitude the pulse sums give in closed form;
  * CA-CFAR detections are monotone in pfa (a smaller pfa raises every threshold,
    CfarDetector1D.cpp:76);
  * the clutter filter removes what lies in the span of its taps (y = h * xs comes
    out as ~0) and is idempotent (WienerHopf.cpp:58-163 is a least-squares projection).
"""
import numpy as np
import pytest

from oracle import blah2_oracle as O

pytestmark = pytest.mark.gpu

CFG2 = dict(dmin=-10, dmax=400, fmin=-256, fmax=256, fs=2_000_000, n=2_000_000)


@pytest.fixture(scope="module")
def b2(built_lib):
    import blah2_amd
    assert blah2_amd.device_count() > 0
    return blah2_amd


@pytest.fixture(scope="module")
def amb(b2):
    c = CFG2
    return b2.Ambiguity(c["dmin"], c["dmax"], c["fmin"], c["fmax"], c["fs"], c["n"], True)


def run(amb, x, y):
    return amb.process(x, y).data.astype(np.complex128)


def test_linear_in_surveillance_conj_linear_in_reference(amb):
    n, fs = CFG2["n"], CFG2["fs"]
    x, y1 = O.synth_iq(n, fs=fs, seed=101)
    _, y2 = O.synth_iq(n, fs=fs, seed=102, targets=((120, 100.0, 0.08),))
    a, b = 0.75 - 0.5j, -1.25 + 2.0j
    m1, m2 = run(amb, x, y1), run(amb, x, y2)
    m12 = run(amb, x, a * y1 + b * y2)
    peak = np.max(np.abs(m12))
    assert np.max(np.abs(m12 - (a * m1 + b * m2))) / peak <= 2e-6
    c = 0.6 + 0.8j
    mc = run(amb, c * x, y1)
    assert np.max(np.abs(mc - np.conj(c) * m1)) / np.max(np.abs(m1)) <= 2e-6


def test_autocorrelation_energy_known_answer(amb):
    n, fs = CFG2["n"], CFG2["fs"]
    x, _ = O.synth_iq(n, fs=fs, seed=103)
    m = run(amb, x, x)
    nD, nCorr = amb.get_n_doppler_bins(), amb.get_n_corr()
    i0 = int(np.argmin(np.abs(amb.doppler)))
    j0 = int(np.where(amb.delay == 0)[0][0])
    energy = float(np.sum(np.abs(x[: nD * nCorr]) ** 2))  # exact in fp64: int16-valued samples
    assert abs(m[i0, j0].real - energy) / energy <= 2e-6 and abs(m[i0, j0].imag) / energy <= 2e-6
    assert np.unravel_index(np.argmax(np.abs(m)), m.shape) == (i0, j0)


def test_delayed_doppler_shifted_copy_closed_form(amb):
    n, fs = CFG2["n"], CFG2["fs"]
    nD, nCorr = amb.get_n_doppler_bins(), amb.get_n_corr()
    d, k = 123, 40  # delay in samples, Doppler in bins of 1/cpi
    f = k / amb.get_cpi()
    rng = np.random.default_rng(7)
    x = np.round(300 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))
    t = np.arange(n)
    y = np.zeros(n, dtype=np.complex128)
    y[d:] = x[:-d]
    y *= np.exp(2j * np.pi * f * t / fs)
    m = run(amb, x, y)
    i0 = int(np.argmin(np.abs(amb.doppler - f)))
    j0 = int(np.where(amb.delay == d)[0][0])
    assert np.unravel_index(np.argmax(np.abs(m)), m.shape) == (i0, j0)
    # closed form of that cell: R[i][d] = sum_{n < nCorr - d} y[p + n + d] conj(x[p + n]), p = i*nCorr,
    # then the DFT over pulses at bin k (exp(-2 pi i i k / nD)), Ambiguity.cpp:106-169
    p = (np.arange(nD) * nCorr)[:, None] + np.arange(nCorr - d)[None, :]
    r = np.sum(y[p + d] * np.conj(x[p]), axis=1)
    want = np.sum(r * np.exp(-2j * np.pi * np.arange(nD) * k / nD))
    assert abs(m[i0, j0] - want) / abs(want) <= 5e-6


def test_cfar_detections_monotone_in_pfa(b2, amb):
    n, fs = CFG2["n"], CFG2["fs"]
    x, y = O.synth_iq(n, fs=fs, seed=104, targets=((37, -63.0, 0.05), (200, 150.0, 0.02), (320, -201.0, 0.01)))
    m = amb.process(x, y)
    prev = None
    for pfa in (1e-7, 1e-5, 1e-3, 1e-1):
        det = b2.CfarDetector1D(pfa, 2, 6, 5, 15.0).process(m)
        cells = set(zip(det.delay.tolist(), np.round(det.doppler, 6).tolist()))
        if prev is not None:
            assert prev <= cells, "a larger pfa lost detections"
        prev = cells
    assert len(prev) > 0


def test_clutter_filter_projects_out_its_own_span_and_is_idempotent(b2):
    n, fs = CFG2["n"], CFG2["fs"]
    dmin, dmax = -10, 400
    rng = np.random.default_rng(9)
    x = np.round(300 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))
    xs = np.roll(x, dmin)  # xs[i] = x[(i - delayMin) mod N], WienerHopf.cpp:67
    # y = a 5-tap FIR of xs inside the filter's span: the least-squares residual is ~0 (up to the
    # first few samples, where the linear convolution of the reference starts from zero history)
    taps = {0: 0.8, 3: -0.3 + 0.2j, 57: 0.1j, 200: 0.05, 409: -0.02 + 0.01j}
    y = np.zeros(n, dtype=np.complex128)
    for kk, h in taps.items():
        y[kk:] += h * xs[: n - kk]
    wh = b2.WienerHopf(dmin, dmax, n)
    ok, yf = wh.process(x, y)
    assert ok
    assert np.sqrt(np.mean(np.abs(yf[1000:]) ** 2)) <= 1e-4 * np.sqrt(np.mean(np.abs(y) ** 2))
    # idempotence on a channel that has content outside the span
    _, y2 = O.synth_iq(n, fs=fs, seed=105)
    ok1, f1 = wh.process(x, y2)
    ok2, f2 = wh.process(x, f1)
    assert ok1 and ok2
    assert np.sqrt(np.mean(np.abs(f2 - f1) ** 2)) <= 1e-4 * np.sqrt(np.mean(np.abs(f1) ** 2))
