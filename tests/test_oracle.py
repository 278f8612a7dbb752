"""CPU suite: pins the oracle (tests infrastructure) against the reference's
own known answers and against the compiled-reference fixtures in tests/golden.
No GPU, no product code."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import blah2_oracle as O
from oracle import ref_lib as R


# test/unit/process/meta/TestHammingNumber.cpp:15-17
@pytest.mark.parametrize("v,expect", [(104, 108), (3322, 3375), (19043, 19200)])
def test_next_hamming_reference_kat(v, expect):
    assert O.next_hamming(v) == expect


def test_next_hamming_is_strictly_greater():
    # HammingNumber.cpp:42 uses `>`: a 5-smooth input maps to the NEXT one
    assert O.next_hamming(108) == 120
    assert O.next_hamming(1) == 2
    assert O.next_hamming(0) == 1


# test/unit/process/ambiguity/TestAmbiguity.cpp:73-93 and :96-116
@pytest.mark.parametrize("round_hamming,nfft", [(False, 6643), (True, 6750)])
def test_constructor_reference_kat(round_hamming, nfft):
    d = O.ambiguity_dims(-10, 300, -300, 300, 2_000_000, int(0.5 * 2_000_000), round_hamming)
    assert abs(d.cpi - 0.5) <= 0.02
    assert d.doppler_middle == 0
    assert d.n_corr == 3322
    assert d.n_delay_bins == 300 + 10 + 1
    assert d.n_doppler_bins == 301
    assert d.nfft == nfft


# SURVEY.md section 8 size table (derived with the compiled reference)
@pytest.mark.parametrize("args,expect", [
    ((-10, 400, -200, 200, 2_000_000, 1_500_000), (301, 4983, 411, 10000)),
    ((-10, 400, -126, 126, 2_000_000, 1_000_000), (127, 7874, 411, 16000)),
    ((-10, 400, -256, 256, 2_000_000, 2_000_000), (513, 3898, 411, 8000)),
    ((-24, 2023, -512, 512, 10_000_000, 10_000_000), (1025, 9756, 2048, 19683)),
    ((-10, 400, -512, 512, 20_000_000, 40_000_000), (2049, 19521, 411, 39366)),
])
def test_baseline_config_sizes(args, expect):
    d = O.ambiguity_dims(*args, True)
    assert (d.n_doppler_bins, d.n_corr, d.n_delay_bins, d.nfft) == expect


@pytest.mark.parametrize("name", golden_names())
def test_numpy_restatement_matches_compiled_reference(name):
    g = load_golden(name)
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    assert [d.n_doppler_bins, d.n_delay_bins, d.n_corr, d.nfft] == list(g["dims"])
    assert d.cpi == float(g["cpi"]) and d.doppler_middle == float(g["doppler_middle"])
    assert np.array_equal(d.delay, g["delay"])
    assert np.allclose(d.doppler, g["doppler"], rtol=0, atol=1e-9)
    m = O.ambiguity_process(d, g["x"], g["y"])
    peak = np.max(np.abs(g["map"]))
    assert np.max(np.abs(m - g["map"])) / peak < 1e-12
    noise, mx = O.map_metrics(m)
    assert abs(noise - g["metrics"][0]) < 1e-9 and abs(mx - g["metrics"][1]) < 1e-9
    # samples left in the FIFOs after process() (it pops nCorr*nDoppler of them)
    assert list(g["leftover"]) == [n - d.n_corr * d.n_doppler_bins] * 2


@pytest.mark.parametrize("name", ["small_sym", "small_asym"])
def test_direct_definition_matches(name):
    g = load_golden(name)
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    x = g["x"]
    if d.doppler_middle != 0:
        i = np.arange(n)
        x = x * np.exp(2j * np.pi * d.doppler_middle * i / fs)
    m = O.ambiguity_process_direct(d, x, g["y"])
    assert np.max(np.abs(m - g["map"])) / np.max(np.abs(g["map"])) < 1e-11


@pytest.mark.parametrize("name", golden_names())
def test_cfar_restatement_matches_compiled_reference(name):
    g = load_golden(name)
    pfa, ng, nt, md, mdop = g["det_params"][:5]
    args = (g["map"], g["delay"], g["doppler"], g["metrics"][0], pfa, int(ng), int(nt), int(md), mdop)
    for fn in (O.cfar1d, O.cfar1d_fast):
        dl, dp, sn = fn(*args)
        assert np.array_equal(dl, g["cfar"][0])
        assert np.array_equal(dp, g["cfar"][1])
        assert np.allclose(sn, g["cfar"][2], rtol=0, atol=1e-9)


@pytest.mark.parametrize("name", golden_names())
def test_wiener_hopf_restatement_matches_compiled_reference(name):
    g = load_golden(name)
    ok, yf = O.wiener_hopf(g["x"], g["y"], int(g["clutter_params"][0]), int(g["clutter_params"][1]))
    assert ok == bool(g["clutter_ok"])
    assert np.max(np.abs(yf - g["clutter_y"])) / np.max(np.abs(g["clutter_y"])) < 1e-9
    # the filter removes the direct-path/clutter energy (where its taps cover the direct path at delay 0)
    if int(g["clutter_params"][0]) <= 0:
        assert np.linalg.norm(yf) < 0.5 * np.linalg.norm(g["y"])


def test_wiener_hopf_failure_contract():
    # an all-zero reference gives a singular (not positive definite) matrix:
    # WienerHopf.cpp:111-115 returns false and y is left alone
    n = 2000
    y = np.ones(n, dtype=np.complex128)
    ok, yf = O.wiener_hopf(np.zeros(n, dtype=np.complex128), y, -2, 10)
    assert not ok and np.array_equal(yf, y)


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_compiled_reference_live_matches_fixture_and_kats():
    assert [R.next_hamming(v) for v in (104, 3322, 19043)] == [108, 3375, 19200]
    a = R.RefAmbiguity(-10, 300, -300, 300, 2_000_000, 1_000_000, True)
    assert (a.n_corr, a.n_doppler_bins, a.n_delay_bins, a.nfft) == (3322, 301, 311, 6750)
    g = load_golden("small_sym")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    a = R.RefAmbiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    m, delay, doppler, noise, mx, left = a.process(g["x"], g["y"])
    assert np.array_equal(m, g["map"])
    assert noise == g["metrics"][0] and mx == g["metrics"][1]


# ---- SpectrumAnalyser (SpectrumAnalyser.cpp:9-71) ------------------------------------
def test_spectrum_restatement_matches_compiled_reference_fixtures():
    import os
    from conftest import GOLDEN
    for name in golden_names():
        g = load_golden(name)
        n = int(g["params"][1])
        spec, freq = O.spectrum_process(g["x"], n, 2000)
        assert freq.size == int(g["spectrum_n_frequency"]) == 0
        assert np.max(np.abs(spec - g["spectrum"])) <= 1e-12 * np.max(np.abs(g["spectrum"]))
    for f in sorted(os.listdir(os.path.join(GOLDEN, "spectrum"))):
        z = np.load(os.path.join(GOLDEN, "spectrum", f))
        n, bw = int(z["params"][0]), float(z["params"][1])
        x = z["iq"][:, 0].astype(np.float64) + 1j * z["iq"][:, 1].astype(np.float64)
        spec, freq = O.spectrum_process(x, n, bw)
        assert spec.size == O.spectrum_dims(n, bw)[1] == z["spectrum"].size
        assert np.max(np.abs(spec - z["spectrum"])) <= 1e-12 * np.max(np.abs(z["spectrum"]))


def test_spectrum_dims_of_the_shipped_configurations():
    # blah2.cpp:198-199: bandwidth 2000 on nSamples = fs * tCpi
    assert O.spectrum_dims(2_000_000, 2000) == (1000, 2000, 2_000_000)
    assert O.spectrum_dims(1_500_000, 2000) == (750, 2000, 1_500_000)
    assert O.spectrum_dims(999_999, 2000) == (499, 2004, 999_996)
