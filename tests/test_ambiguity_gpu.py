"""GPU parity tests for the ambiguity chain (range kernel -> Doppler kernel ->
metrics), through the C ABI.  Modelled on the reference's
test/unit/process/ambiguity/TestAmbiguity.cpp.

Tolerances (BASELINE.json: "map values within 1e-4 rel of FFTW reference",
SURVEY.md section 8d): the GPU computes in fp32, the oracle in fp64.
  * max |M_gpu - M_ref| / max |M_ref|            <= 1e-5  (gate is 1e-4)
  * element-wise relative error on cells whose magnitude is above the mean
    magnitude                                      <= 1e-4
  * |noisePower|, |maxPower| difference            <= 1e-3 dB (the reference's own
    test tolerance, TestAmbiguity.cpp:176-177)
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import blah2_oracle as O

pytestmark = pytest.mark.gpu

PEAK_TOL = 1e-5
CELL_TOL = 1e-4
DB_TOL = 1e-3


@pytest.fixture(scope="module")
def b2(built_lib):
    import blah2_amd
    assert blah2_amd.device_count() > 0, "no GPU visible: the HIP path cannot run"
    return blah2_amd


def assert_map_close(got, ref):
    ref = np.asarray(ref)
    got = np.asarray(got).astype(np.complex128)
    peak = np.max(np.abs(ref))
    err = np.abs(got - ref)
    assert np.max(err) / peak <= PEAK_TOL, f"peak-relative error {np.max(err) / peak:.3e}"
    strong = np.abs(ref) > np.mean(np.abs(ref))
    rel = err[strong] / np.abs(ref[strong])
    assert np.max(rel) <= CELL_TOL, f"element-wise relative error {np.max(rel):.3e}"


# TestAmbiguity.cpp:73-93 "Constructor" and :96-116 "Constructor_Round"
@pytest.mark.parametrize("round_hamming,nfft", [(False, 6643), (True, 6750)])
def test_constructor(b2, round_hamming, nfft):
    fs, tcpi = 2_000_000, 0.5
    amb = b2.Ambiguity(-10, 300, -300, 300, fs, int(tcpi * fs), round_hamming)
    assert abs(amb.get_cpi() - tcpi) <= 0.02
    assert amb.get_doppler_middle() == 0
    assert amb.get_n_corr() == 3322
    assert amb.get_n_delay_bins() == 300 + abs(-10) + 1
    assert amb.get_n_doppler_bins() == 301
    assert amb.get_nfft() == nfft
    d = O.ambiguity_dims(-10, 300, -300, 300, fs, int(tcpi * fs), round_hamming)
    assert np.array_equal(amb.delay, d.delay)
    assert np.array_equal(amb.doppler, d.doppler)


# TestAmbiguity.cpp:119-144 "Process_Simple": uniform(-100,100) IQ, positive metrics.
# Seeded here (the reference seeds from random_device) and additionally compared
# with the oracle on the same input.
@pytest.mark.parametrize("round_hamming", [True, False])
def test_process_simple(b2, round_hamming):
    fs, n = 2_000_000, 1_000_000
    rng = np.random.default_rng(42)
    x = rng.uniform(-100, 100, n) + 1j * rng.uniform(-100, 100, n)
    y = rng.uniform(-100, 100, n) + 1j * rng.uniform(-100, 100, n)
    amb = b2.Ambiguity(-10, 300, -300, 300, fs, n, round_hamming)
    m = amb.process(x, y)
    m.set_metrics()
    assert m.maxPower > 0.0
    assert m.noisePower > 0.0
    d = O.ambiguity_dims(-10, 300, -300, 300, fs, n, round_hamming)
    ref = O.ambiguity_process(d, x, y)
    assert_map_close(m.data, ref)
    noise, mx = O.map_metrics(ref)
    assert abs(m.noisePower - noise) <= DB_TOL and abs(m.maxPower - mx) <= DB_TOL
    assert amb.get_n_samples() == d.n_corr * d.n_doppler_bins  # Ambiguity.cpp:105


# The reference's "Process_File" needs a capture that is not shipped; these are
# the equivalent fixtures, produced by the reference's own sources (tests/golden).
@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("path", ["c64", "c32", "i16"])
def test_process_golden(b2, name, path):
    g = load_golden(name)
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    assert [amb.get_n_doppler_bins(), amb.get_n_delay_bins(), amb.get_n_corr(), amb.get_nfft()] == list(g["dims"])
    assert amb.get_cpi() == float(g["cpi"])
    assert amb.get_doppler_middle() == float(g["doppler_middle"])
    assert np.array_equal(amb.delay, g["delay"])
    assert np.array_equal(amb.doppler, g["doppler"])
    if path == "c64":
        m = amb.process(g["x"], g["y"])
    elif path == "c32":
        m = amb.process(g["x"].astype(np.complex64), g["y"].astype(np.complex64))
    else:
        m = amb.process_i16(g["iq"])
    assert_map_close(m.data, g["map"])
    assert abs(m.noisePower - g["metrics"][0]) <= DB_TOL
    assert abs(m.maxPower - g["metrics"][1]) <= DB_TOL


def test_underflow_raises_like_pop_front(b2):
    # IqData::pop_front throws on an empty deque (IqData.cpp:57-59)
    amb = b2.Ambiguity(-3, 20, -50, 50, 200_000, 20_000, True)
    x = np.zeros(1000, dtype=np.complex128)
    with pytest.raises(RuntimeError, match="empty deque"):
        amb.process(x, x)


@pytest.mark.parametrize("fft_len", [1024, 2048, 4096])
def test_every_transform_length(b2, fft_len):
    # the planner normally picks F by cost; force each kernel instantiation
    g = load_golden("medium")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    amb.set_fft_len(fft_len)
    assert amb.dims.fft_len == fft_len
    m = amb.process(g["x"], g["y"])
    assert_map_close(m.data, g["map"])


def test_direct_doppler_fallback(b2):
    # nDoppler > 2049 falls back to a direct DFT kernel; force it on a small case
    g = load_golden("medium")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    amb.set_doppler_kernel("direct")
    m = amb.process(g["x"], g["y"])
    assert amb.last_doppler_kernel() == "direct"
    assert_map_close(m.data, g["map"])
    assert abs(m.noisePower - g["metrics"][0]) <= DB_TOL


def cfg2():
    return dict(delayMin=-10, delayMax=400, dopplerMin=-256, dopplerMax=256, fs=2_000_000, n=2_000_000)


def test_baseline_cfg2_vs_oracle(b2):
    """BASELINE.json configs[1]: 2 MS/s, 1 s CPI, 513 x 411."""
    c = cfg2()
    x, y = O.synth_iq(c["n"], fs=c["fs"])
    amb = b2.Ambiguity(c["delayMin"], c["delayMax"], c["dopplerMin"], c["dopplerMax"], c["fs"], c["n"], True)
    assert (amb.get_n_doppler_bins(), amb.get_n_delay_bins(), amb.get_n_corr(), amb.get_nfft()) == (513, 411, 3898, 8000)
    m = amb.process(x, y)
    d = O.ambiguity_dims(c["delayMin"], c["delayMax"], c["dopplerMin"], c["dopplerMax"], c["fs"], c["n"], True)
    ref = O.ambiguity_process(d, x, y)
    assert_map_close(m.data, ref)
    noise, mx = O.map_metrics(ref)
    assert abs(m.noisePower - noise) <= DB_TOL and abs(m.maxPower - mx) <= DB_TOL
    # the injected target (delay 37 bins, -63 Hz) is the strongest off-zero-Doppler cell
    db = 10 * np.log10(np.abs(m.data.astype(np.complex128)))
    db[np.abs(amb.doppler) < 15, :] = -np.inf
    i, j = np.unravel_index(np.argmax(db), db.shape)
    assert amb.delay[j] == 37 and abs(amb.doppler[i] - (-63.0)) < 1.0


def test_batched_device_chain_matches_single(b2):
    """Several CPIs per launch (the replay/throughput path) give the same maps
    as one-at-a-time processing; size-independent property: linearity in y."""
    torch = pytest.importorskip("torch")
    g = load_golden("medium")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    B = 3
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh), max_batch=B)
    x = torch.from_numpy(np.stack([g["x"], g["x"], 2 * g["x"]]).astype(np.complex64)).cuda()
    y = torch.from_numpy(np.stack([g["y"], 3 * g["y"], g["y"]]).astype(np.complex64)).cuda()
    nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
    out = torch.zeros((B, nD, nC), dtype=torch.complex64, device="cuda")
    met = torch.zeros((B, 2), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    amb.process_dev(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert_map_close(o[0], g["map"])
    assert_map_close(o[1], 3 * g["map"])          # linear in the surveillance channel
    assert_map_close(o[2], 2 * g["map"])          # conj-linear in the reference channel (real factor)
    mt = met.cpu().numpy()
    assert abs(mt[0, 0] - g["metrics"][0]) <= DB_TOL
    assert abs(mt[1, 0] - (g["metrics"][0] + 10 * np.log10(3))) <= DB_TOL
    assert abs(mt[1, 1] - g["metrics"][1]) <= DB_TOL  # dynamic range is scale-free


# Map::to_json writes data[i][j] = 10*log10|M| - noisePower with two decimals (Map.cpp:115-185).  SURVEY.md 8d gate:
# |delta dB| <= 0.005 on the JSON map -- stated once in tests/gates.py (every cell down to 20 dB below the mean level,
# which is 20 dB below what html/js/plot_map.js:170 can show; below that the absolute error 0.005 dB means at that
# line).  At these two sizes (3e5 / 2e5 cells) no cell at all exceeds 0.005 dB (measured 0.0024 / 0.0009).
@pytest.mark.parametrize("args", [(-10, 300, -300, 300, 2_000_000, 1_000_000, True),
                                  (-10, 400, -256, 256, 2_000_000, 2_000_000, True)])
def test_json_db_map_within_half_a_hundredth(b2, args):
    from gates import db_map_gate
    fs, n = args[4], args[5]
    x, y = O.synth_iq(n, fs=fs, seed=7)
    m = b2.Ambiguity(*args).process(x, y)
    ref = O.ambiguity_process(O.ambiguity_dims(*args), x, y)
    g = db_map_gate(m.data, m.noisePower, ref)
    assert g["ok"], g
    assert g["db_max_all"] <= 0.005, g   # stronger than the gate, and true at these sizes


def test_device_db_map_matches_to_json_values(b2):
    """blah2hip_amb_db_dev: the fp32 dB map a front-end plots, for a batch, against the fp64
    oracle's Map::to_json values (same 0.005 dB gate)."""
    torch = pytest.importorskip("torch")
    g = load_golden("medium")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    B = 2
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh), max_batch=B)
    x = torch.from_numpy(np.stack([g["x"], g["x"]]).astype(np.complex64)).cuda()
    y = torch.from_numpy(np.stack([g["y"], 2 * g["y"]]).astype(np.complex64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    amb.process_dev(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, None, None, st)
    nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
    db = torch.zeros((B, nD, nC), dtype=torch.float32, device="cuda")
    amb.db_dev(None, None, B, db.data_ptr(), st)
    torch.cuda.synchronize()
    want = 10.0 * np.log10(np.abs(g["map"])) - g["metrics"][0]
    got = db.cpu().numpy()
    assert np.max(np.abs(got[0] - want)) <= 0.005
    assert np.max(np.abs(got[1] - want)) <= 0.005  # scaling y shifts |M| and noisePower alike
