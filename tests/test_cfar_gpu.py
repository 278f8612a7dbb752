"""GPU parity tests for CfarDetector1D through the C ABI.

The detector compares |z|^2 against alpha*mean; the GPU map is fp32, so a cell
whose margin |sq/threshold - 1| is below the map's own error may legitimately flip.
Parity rule (oracle/gates.py `detection_gate`): the lists are identical except at cells
whose margin lies within MARGIN_K (= 4) times the map error MEASURED on this very map
(element-wise above the mean level / largest error over the mean level, ~1e-6) of 1;
delay and Doppler of common detections are identical, snr within 1e-3 dB.
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from gates import cfar1d_margins, detection_gate, map_cell_gate, margin_eps
from oracle import blah2_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def b2(built_lib):
    import blah2_amd
    assert blah2_amd.device_count() > 0
    return blah2_amd


margins = cfar1d_margins  # |z|^2 / threshold per cell (fp64), CfarDetector1D.cpp:55-83


def measured_eps(m, m_ref, noise):
    """The map error the margin band is sized from (m: the device Map the detector ran on)."""
    return margin_eps(map_cell_gate(m.data, m_ref, noise))


def check_detections(amb, det, m_ref, noise, pfa, ng, nt, md, mdop, m=None):
    dl, dp, sn = O.cfar1d_fast(m_ref, amb.delay, amb.doppler, noise, pfa, ng, nt, md, mdop)
    ref = {(a, b): s for a, b, s in zip(dl, dp, sn)}
    got = {(a, b): s for a, b, s in zip(det.get_delay(), det.get_doppler(), det.get_snr())}
    mg = margins(np.asarray(m_ref, dtype=np.complex128), pfa, ng, nt)
    row = {f: i for i, f in enumerate(amb.doppler)}
    dg = detection_gate(ref, got, mg, amb.doppler, amb.delay[0], measured_eps(m, m_ref, noise))
    assert dg["ok"], f"non-borderline mismatch: {dg}"
    for key in set(ref) & set(got):
        assert abs(ref[key] - got[key]) < 1e-3
    # emission order is row-major like the reference's loops
    order = [(row[f], d) for d, f in zip(det.get_delay(), det.get_doppler())]
    assert order == sorted(order)


@pytest.mark.parametrize("name", golden_names())
def test_cfar_golden(b2, name):
    g = load_golden(name)
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    pfa, ng, nt, md, mdop = g["det_params"][:5]
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    m = amb.process(g["x"], g["y"])
    m.set_metrics()
    det = b2.CfarDetector1D(pfa, int(ng), int(nt), int(md), mdop).process(m)
    check_detections(amb, det, g["map"], g["metrics"][0], pfa, int(ng), int(nt), int(md), mdop, m)
    # the fixture's targets are far from the threshold: exact agreement expected
    assert np.array_equal(det.get_delay(), g["cfar"][0])
    assert np.array_equal(det.get_doppler(), g["cfar"][1])
    assert np.allclose(det.get_snr(), g["cfar"][2], rtol=0, atol=1e-3)


@pytest.mark.parametrize("params", [
    (1e-5, 2, 6, 5, 15.0),    # config/config.yml:36-40 defaults
    (1e-2, 0, 1, -10, 0.0),   # dense detections, windows clipped at both map edges
    (1e-3, 3, 20, 0, 0.0),    # windows wider than the distance to the edge (k > 0 quirk)
])
def test_cfar_dense_vs_oracle(b2, params):
    pfa, ng, nt, md, mdop = params
    g = load_golden("medium")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    m = amb.process(g["x"], g["y"])
    det = b2.CfarDetector1D(pfa, ng, nt, md, mdop).process(m)
    assert det.get_nDetections() > 0
    check_detections(amb, det, g["map"], g["metrics"][0], pfa, ng, nt, md, mdop, m)


def test_cfar_parameters_are_int8_like_the_reference(b2):
    with pytest.raises(ValueError):
        b2.CfarDetector1D(1e-5, 200, 6, 5, 15.0)


# ---- 2-D CA-CFAR (BASELINE configs[2]; defined in SURVEY.md 8g, oracle cfar2d) ----
# (nGd, nTd, nGf, nTf) the stream kernel is instantiated for: C2S_SHAPES in csrc/cfar_kernels.hpp, there as (nTd, nGd, nTf, nGf)
STREAM_SHAPES = {(2, 6, 1, 3), (2, 6, 0, 0), (2, 8, 1, 4), (1, 4, 1, 2), (1, 3, 1, 2), (0, 1, 0, 0), (0, 0, 0, 1), (0, 2, 0, 1), (2, 5, 2, 6)}
def check_2d(b2, amb, m, m_ref, noise, params):
    pfa, ngd, ntd, ngf, ntf, md, mdop = params
    det = b2.CfarDetector2D(pfa, ngd, ntd, ngf, ntf, md, mdop).process(m)
    dl, dp, sn, margin = O.cfar2d(m_ref, amb.delay, amb.doppler, noise, pfa, ngd, ntd, ngf, ntf, md, mdop,
                                  return_margin=True)
    ref = {(a, b): s for a, b, s in zip(dl, dp, sn)}
    got = {(a, b): s for a, b, s in zip(det.get_delay(), det.get_doppler(), det.get_snr())}
    row = {f: i for i, f in enumerate(amb.doppler)}
    dg = detection_gate(ref, got, margin, amb.doppler, amb.delay[0], measured_eps(m, m_ref, noise))
    assert dg["ok"], f"non-borderline mismatch: {dg}"
    for key in set(ref) & set(got):
        assert abs(ref[key] - got[key]) < 1e-3
    order = [(row[f], d) for d, f in zip(det.get_delay(), det.get_doppler())]
    assert order == sorted(order)
    return det


@pytest.mark.parametrize("params", [
    (1e-4, 2, 6, 1, 3, 5, 15.0),
    (1e-2, 1, 3, 1, 2, -10, 0.0),   # dense, windows clipped on all four map edges
    (1e-3, 0, 1, 0, 1, 0, 0.0),     # smallest possible annulus
])
def test_cfar2d_vs_oracle(b2, params):
    g = load_golden("medium")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    m = amb.process(g["x"], g["y"])
    det = check_2d(b2, amb, m, g["map"], g["metrics"][0], params)
    assert det.get_nDetections() > 0


def test_cfar2d_reduces_to_1d(b2):
    g = load_golden("medium")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    m = amb.process(g["x"], g["y"])
    for pfa, ng, nt, md, mdop in [(1e-5, 2, 6, 5, 15.0), (1e-2, 0, 1, -10, 0.0), (1e-3, 3, 20, 0, 0.0)]:
        d1 = b2.CfarDetector1D(pfa, ng, nt, md, mdop).process(m)
        d2 = b2.CfarDetector2D(pfa, ng, nt, 0, 0, md, mdop).process(m)
        assert np.array_equal(d1.get_delay(), d2.get_delay())
        assert np.array_equal(d1.get_doppler(), d2.get_doppler())
        assert np.allclose(d1.get_snr(), d2.get_snr(), rtol=0, atol=1e-9)


# ---- the one-pass tile kernel against the summed-area-table kernels and the brute-force oracle ----
@pytest.mark.parametrize("params", [
    (1e-4, 2, 6, 1, 3, 5, 15.0),      # the bench's window (17 x 9)
    (1e-3, 0, 1, 0, 0, 0, 0.0),       # one-dimensional, smallest annulus: no Doppler halo at all
    (1e-3, 0, 0, 0, 1, -10, 0.0),     # Doppler-only training (no delay halo)
    (1e-3, 5, 27, 3, 21, -10, 0.0),   # the largest window the tile kernel takes with two loads per row: halo 32 x 24
    (1e-3, 6, 34, 2, 4, 3, 7.0),      # three loads per row (halo 40 columns, the widest the tile kernel takes)
    (1e-2, 1, 3, 1, 2, -10, 0.0),
])
def test_cfar2d_tile_kernel_equals_sat_kernel(b2, params):
    """Forced kernels on the same device map: identical detection sets (all sum the same fp64 squares;
    only the summation order differs), on a 201 x 111 map whose tiles, strips and row segments are ragged in both
    directions.  The stream kernel takes part for the window shapes it is instantiated for and refuses the others."""
    g = load_golden("medium")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    m = amb.process(g["x"], g["y"])
    out = {}
    for which in ("tile", "sat", "stream"):
        amb.set_cfar2d_kernel(which)
        if which == "stream" and tuple(params[1:5]) not in STREAM_SHAPES:
            with pytest.raises(b2.Blah2HipError):
                b2.CfarDetector2D(*params).process(m)
            continue
        d = b2.CfarDetector2D(*params).process(m)
        out[which] = {(a, b): s for a, b, s in zip(d.get_delay(), d.get_doppler(), d.get_snr())}
    assert ("stream" in out) == (tuple(params[1:5]) in STREAM_SHAPES)
    got = m.data.astype(np.complex128)
    _, _, _, margin = O.cfar2d(got, amb.delay, amb.doppler, m.noisePower, *params, return_margin=True)
    row = {f: i for i, f in enumerate(amb.doppler)}
    for which in out:
        for key in set(out[which]) ^ set(out["sat"]):
            i, j = row[key[1]], int(key[0] - amb.delay[0])
            assert abs(margin[i, j] - 1) < 1e-9, (which, key, margin[i, j])
        for key in set(out[which]) & set(out["sat"]):
            assert out[which][key] == out["sat"][key]
    assert len(out["tile"]) > 0


@pytest.mark.parametrize("which", ["tile", "stream"])
def test_cfar2d_tile_kernel_vs_bruteforce_oracle(b2, which):
    """The literal four-loop definition (oracle cfar2d_bruteforce) on the device's own small map, incl. delay
    column 0 as a test cell (it never trains) and every clipped-window case."""
    g = load_golden("small_sym")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    m = amb.process(g["x"], g["y"])
    amb.set_cfar2d_kernel(which)
    got = m.data.astype(np.complex128)
    for params in [(1e-2, 1, 3, 1, 2, -100, 0.0), (0.2, 0, 2, 0, 1, -100, 0.0), (1e-3, 2, 5, 2, 6, 0, 0.0)]:
        d = b2.CfarDetector2D(*params).process(m)
        dl, dp, sn = O.cfar2d_bruteforce(got, amb.delay, amb.doppler, m.noisePower, *params)
        _, _, _, margin = O.cfar2d(got, amb.delay, amb.doppler, m.noisePower, *params, return_margin=True)
        ref, dev = set(zip(dl, dp)), set(zip(d.get_delay(), d.get_doppler()))
        row = {f: i for i, f in enumerate(amb.doppler)}
        for key in ref ^ dev:
            assert abs(margin[row[key[1]], int(key[0] - amb.delay[0])] - 1) < 1e-9, key
        assert len(ref) > 0


def test_cfar2d_window_beyond_the_tile_halo_takes_the_sat_kernels(b2):
    g = load_golden("medium")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    m = amb.process(g["x"], g["y"])
    big = (1e-3, 9, 40, 5, 20, -10, 0.0)  # halo 49 columns x 25 rows: beyond the tile kernel
    det = check_2d(b2, amb, m, g["map"], g["metrics"][0], big)  # automatic choice: summed-area table
    assert det.get_nDetections() > 0
    amb.set_cfar2d_kernel("tile")
    with pytest.raises(b2.Blah2HipError):
        b2.CfarDetector2D(*big).process(m)


def test_python_detector_follows_the_map_it_is_given(b2):
    """Like the C++ classes (Map::fingerprint): CfarDetector1D.process runs on the engine's device copy only while the Map
    still IS that copy.  A Map kept across a later process call, or whose cells the caller changed, is uploaded and
    evaluated as given (CfarDetector1D.cpp:23-100 reads x->data); the 2-D detector, which has no host-map entry, refuses."""
    g = load_golden("medium")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    pfa, ng, nt, md, mdop = g["det_params"][:5]
    det = b2.CfarDetector1D(pfa, int(ng), int(nt), int(md), mdop)
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    m1 = amb.process(g["x"], g["y"])
    assert m1.device_copy_is_current()
    d1 = det.process(m1)
    assert np.array_equal(d1.get_delay(), g["cfar"][0])
    m2 = amb.process(g["y"], g["x"])  # another CPI: the device now holds m2
    assert not m1.device_copy_is_current() and m2.device_copy_is_current()
    d1b = det.process(m1)             # evaluated on m1's own cells (uploaded), not on the device's m2
    assert np.array_equal(d1b.get_delay(), d1.get_delay()) and np.array_equal(d1b.get_doppler(), d1.get_doppler())
    assert np.allclose(d1b.get_snr(), d1.get_snr(), atol=1e-9)
    with pytest.raises(ValueError):
        b2.CfarDetector2D(pfa, 1, 3, 1, 2, int(md), mdop).process(m1)
    # cells changed by the caller: one cell lifted far above its neighbours must now be reported
    i, j = 7, 60
    m2.data[i, j] = 1e4 * np.abs(m2.data).max()
    assert not m2.device_copy_is_current()
    d2 = det.process(m2)
    assert (float(m2.delay[j]), float(m2.doppler[i])) in set(zip(d2.get_delay(), d2.get_doppler()))


def test_host_map_with_more_than_8192_delay_bins(b2):
    """blah2hip_cfar1d_map on a caller-held map of 9011 delay bins (round 6: it refused above 8192; CfarDetector1D.cpp:23-100
    has no such bound).  `wide_delay`: the reference's own list on its own map; the same map changed by the caller and
    handed back goes through the host-map entry, and so does a 20 000-bin row, which no longer fits the LDS as fp64 and is
    evaluated straight from memory -- against the oracle."""
    g = load_golden("wide_delay")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    pfa, ng, nt, md, mdop = g["det_params"][:5]
    det = b2.CfarDetector1D(pfa, int(ng), int(nt), int(md), mdop)
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    m = amb.process(g["x"], g["y"])
    assert m.data.shape == (5, 9011)
    d_dev = det.process(m)                       # the engine's device copy
    check_detections(amb, d_dev, g["map"], g["metrics"][0], pfa, int(ng), int(nt), int(md), mdop, m)
    assert np.array_equal(d_dev.get_delay(), g["cfar"][0]) and np.array_equal(d_dev.get_doppler(), g["cfar"][1])
    m.data[0, 0] *= 2                            # touched by the caller (column 0 never trains, delay -10 is below minDelay: the
    assert not m.device_copy_is_current()        # list cannot change): evaluated as given, through blah2hip_cfar1d_map
    d_host = det.process(m)
    assert np.array_equal(d_host.get_delay(), d_dev.get_delay()) and np.array_equal(d_host.get_doppler(), d_dev.get_doppler())
    # a row beyond the LDS: 20 000 bins, synthetic cells, the unstaged kernel
    rng = np.random.default_rng(8)
    nD, nC = 3, 20_000
    cells = (rng.standard_normal((nD, nC)) + 1j * rng.standard_normal((nD, nC))).astype(np.complex64)
    cells[1, 12_345] = 40.0
    delay = np.arange(-5, nC - 5)
    doppler = np.array([-1.0, 0.0, 1.0])
    noise, _ = O.map_metrics(cells.astype(np.complex128))
    big = b2.Map(None, cells, delay, doppler, noise, 0.0, 0)
    d_big = b2.CfarDetector1D(1e-4, 2, 8, 0, 0.5).process(big)
    dl, dp, _ = O.cfar1d_fast(cells.astype(np.complex128), delay, doppler, noise, 1e-4, 2, 8, 0, 0.5)
    assert set(zip(d_big.get_delay(), d_big.get_doppler())) == set(zip(dl, dp)) and (12_340.0, 1.0) not in set(zip(dl, dp))
    assert (float(delay[12_345]), 0.0) not in set(zip(dl, dp))  # the zero-Doppler row is below minDoppler
