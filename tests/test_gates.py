"""CPU: the parity gates of oracle/gates.py behave as stated (they are what GPUTEST and bench.py's parity blocks
assert, so they are tested themselves), and replay.LoopedCapture reads a capture cyclically."""
import numpy as np
import pytest

from conftest import load_golden
from gates import (CELL_TOL, MARGIN_K, cfar1d_margins, db_map_gate, detection_gate, map_cell_gate, margin_eps, mean_level,
                   notch_mask)
from oracle import blah2_oracle as O


def noise_map(shape=(65, 120), seed=1):
    rng = np.random.default_rng(seed)
    return 1e5 * (rng.standard_normal(shape) + 1j * rng.standard_normal(shape))


def test_cell_gate_is_relative_to_cells_above_the_maps_own_mean_level():
    ref = noise_map()
    noise, _ = O.map_metrics(ref)
    level = mean_level(noise)
    a = np.abs(ref)
    hi = np.unravel_index(np.argmax(a), a.shape)
    lo = np.unravel_index(np.argmin(a), a.shape)
    assert a[lo] < level < a[hi]
    got = ref.copy()
    got[lo] *= 1.01                      # a deep cell 1 % off: below the mean level, outside this gate (the dB gate holds it)
    g = map_cell_gate(got, ref, noise)
    assert g["ok"] and g["cell_rel_above_mean"] == 0.0
    mid = np.argwhere((a > level) & (a < 1.2 * level))[0]
    got = ref.copy()
    got[tuple(mid)] *= 1 + 2 * CELL_TOL   # 2e-4 on a cell just above the mean level: far below the peak, still caught
    g = map_cell_gate(got, ref, noise)
    assert not g["ok"] and abs(g["cell_rel_above_mean"] - 2 * CELL_TOL) < 1e-9 and g["peak_rel"] < CELL_TOL
    assert map_cell_gate(ref * (1 + 0.5 * CELL_TOL), ref, noise)["ok"]


def test_db_gate_with_a_notch():
    ref = noise_map((21, 111), seed=2)
    dop = np.arange(-10, 11) * 10.0
    dly = np.arange(-10, 101)
    nm = notch_mask(ref.shape, dop, dly, -10, 100)
    assert nm.sum() == 110 and nm[10, :110].all() and not nm[10, 110]
    ref[nm] *= 0.05                      # what the filter leaves at zero Doppler: ~13 dB under the floor
    noise, _ = O.map_metrics(ref)
    got = ref.copy()
    got[nm] += 2e-3 * mean_level(noise)  # a coherent tap-error residue: 4 % of those cells, 0.2 % of the mean level
    plain = db_map_gate(got, noise, ref, noise)
    assert not plain["ok"] and plain["cells_over_shown"] > 0          # the unexempted gate sees it ...
    g = db_map_gate(got, noise, ref, noise, notch=nm)
    assert g["ok"] and g["notch_cells"] == 110 and g["notch_db_max"] > 0.1   # ... the notch rule reports it and holds the absolute bound
    got[nm] += 2e-2 * mean_level(noise)
    assert not db_map_gate(got, noise, ref, noise, notch=nm)["ok"]    # 2.2 % of the mean level: over NOTCH_ABS
    assert notch_mask(ref.shape, dop + 3.0, dly, -10, 100).sum() == 0  # no zero-Doppler row: nothing is exempt


def test_margins_restated_and_the_margin_rule():
    g = load_golden("medium")
    pfa, ng, nt, md, mdop = g["det_params"][:5]
    m = g["map"]
    mg = cfar1d_margins(m, pfa, int(ng), int(nt))
    dl, dp, _ = O.cfar1d_fast(m, g["delay"], g["doppler"], g["metrics"][0], pfa, int(ng), int(nt), int(md), mdop)
    hit = (mg > 1) & (g["delay"] >= md)[None, :] & (np.abs(g["doppler"]) >= mdop)[:, None]
    ii, jj = np.nonzero(hit)
    assert set(zip(jj + g["delay"][0], g["doppler"][ii])) == set(zip(dl, dp)) and len(dl) > 0
    ref = list(zip(dl, dp))
    eps = 1e-6
    assert detection_gate(ref, ref, mg, g["doppler"], g["delay"][0], eps)["ok"]
    dropped = detection_gate(ref, ref[1:], mg, g["doppler"], g["delay"][0], eps)   # a clear detection missing
    assert not dropped["ok"] and dropped["n_differ"] == 1 and dropped["worst_margin_off_one"] > MARGIN_K * eps
    i, j = 3, 40                                                                      # a borderline cell may flip
    mg2 = mg.copy()
    mg2[i, j] = 1 + 2e-6
    extra = ref + [(float(j + g["delay"][0]), float(g["doppler"][i]))]
    assert detection_gate(ref, extra, mg2, g["doppler"], g["delay"][0], eps)["ok"]
    mg2[i, j] = 1 + 1e-5
    assert not detection_gate(ref, extra, mg2, g["doppler"], g["delay"][0], eps)["ok"]
    cell = map_cell_gate(m * (1 + 3e-6), m, g["metrics"][0])
    assert abs(margin_eps(cell) / 3e-6 - 1) < 1e-6 and cell["abs_err_over_mean_level"] > 1e-4  # the peak's own error does not widen the band


def test_looped_capture(tmp_path):
    from blah2_amd import replay as R
    n, k = 1000, 6
    iq = np.arange(k * n * 4, dtype=np.int64).astype(np.int16).reshape(k, n, 4)
    p = tmp_path / "c.rspduo"
    iq.tofile(p)
    cap = R.LoopedCapture(str(p), n, 3)
    assert cap.n_cpis == 18 and cap.n_file == 6
    assert np.array_equal(cap.cpi(13), iq[1])
    dst = np.zeros((2, n, 4), dtype=np.int16)
    cap.read_into(10, 2, dst)
    assert np.array_equal(dst, iq[4:6])
    with pytest.raises(ValueError):
        cap.read_into(11, 2, dst)        # would straddle the wrap
    seen = []
    out = R.replay(cap, lambda b: [{"s": int(v[0, 0])} for v in b], batch=2)
    seen = [r["s"] for r in out]
    assert seen == [int(iq[c % k, 0, 0]) for c in range(18)]
    cap.close()


def test_cell_gate_with_a_notch():
    """Behind a clutter filter the element-wise gate runs over the cells OUTSIDE the filter's notch; a coherent residue on
    a notch cell that stands above the mean level is reported (cell_rel_above_mean) and bounded by db_map_gate's absolute
    notch bound, not by 1e-4 of itself."""
    ref = noise_map((21, 111), seed=3)
    dop = np.arange(-10, 11) * 10.0
    dly = np.arange(-10, 101)
    nm = notch_mask(ref.shape, dop, dly, -10, 100)
    noise, _ = O.map_metrics(ref)
    lvl = mean_level(noise)
    ref[2, 30] = 200.0 * lvl                      # the map's peak: a target
    ref[10, 47] = 8.0 * lvl                       # its Doppler sidelobe inside the notch, above the mean level
    got = ref.copy()
    got[10, 47] += 4e-3 * lvl                     # the dominant tap's residue: 0.4 % of the mean level = 5e-4 of the cell
    plain, gated = map_cell_gate(got, ref, noise), map_cell_gate(got, ref, noise, notch=nm)
    assert not plain["ok"] and abs(plain["cell_rel_above_mean"] - 5e-4) < 1e-6
    assert gated["ok"] and gated["cell_rel_above_mean"] == plain["cell_rel_above_mean"] and gated["cell_rel_above_mean_outside_notch"] == 0.0
    assert gated["notch_cells_above_mean"] >= 1
    assert db_map_gate(got, noise, ref, noise, notch=nm)["ok"]            # 0.4 % <= NOTCH_ABS
    got[10, 47] += 2e-2 * lvl
    assert not db_map_gate(got, noise, ref, noise, notch=nm)["ok"]        # 2.4 %: the absolute bound catches it
    got = ref.copy()
    got[3, 20] *= 1 + 3e-4                        # the same size of error on an ordinary cell above the mean: caught by the cell gate
    if abs(ref[3, 20]) > lvl:
        assert not map_cell_gate(got, ref, noise, notch=nm)["ok"]
