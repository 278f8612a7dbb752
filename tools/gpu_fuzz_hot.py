#!/usr/bin/env python3
"""Random geometries with one strong echo (any Doppler, on or off the grid) through Ambiguity with the hot-column rewrite on and
off, against the oracle's cell-wise gate.   python tools/gpu_fuzz_hot.py [cases [seed]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import blah2_amd as b2  # noqa: E402
from gates import map_cell_gate  # noqa: E402
from oracle import blah2_oracle as O  # noqa: E402
from test_hot_columns_gpu import echo_cpi  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails_off = 0
for c in range(cases):
    half = int(rng.choice([25, 50, 100, 256, 300, 512, 700, 1024]))
    fs = int(rng.choice([200_000, 500_000, 1_000_000, 2_000_000]))
    n = int(rng.integers(fs // 4, fs + 1))
    dmin, dmax = -int(rng.integers(0, 12)), int(rng.integers(40, 420))
    args = (dmin, dmax, -half, half, fs, n)
    try:
        d = O.ambiguity_dims(*args, True)
    except Exception:
        continue
    if d.n_corr < 300:
        continue
    lag = int(rng.integers(max(1, dmin + 1), dmax))
    dop = float(rng.uniform(-0.9 * half, 0.9 * half))
    x, y = echo_cpi(n, fs, int(rng.integers(1, 1 << 30)), lag, dop, noise=float(rng.uniform(0.005, 0.1)))
    ref = O.ambiguity_process(d, x.astype(np.complex128), y.astype(np.complex128))
    res = {}
    for mode in ("off", "auto"):
        amb = b2.Ambiguity(*args, True)
        amb.set_hot_columns(mode)
        m = amb.process(x, y).data.copy()
        res[mode] = (map_cell_gate(m, ref), amb.hot_columns(), amb.last_doppler_kernel())
        amb.close()
    lvl = 10.0 ** (O.map_metrics(ref)[0] / 10.0)
    g0, g1 = res["off"][0], res["auto"][0]
    fails_off += 0 if g0["ok"] else 1
    print(f"case {c}: {d.n_doppler_bins} x {d.n_delay_bins}, {n} samples, echo at lag {lag} / {dop:.1f} Hz, {np.abs(ref).max() / lvl:.0f}x the mean level, "
          f"Doppler kernel {res['auto'][2]}: cell-rel off {g0['cell_rel_above_mean']:.2e} -> auto {g1['cell_rel_above_mean']:.2e} ({res['auto'][1]} hot)", flush=True)
    assert g1["ok"], g1
print(f"all {cases} inside the gate with the rewrite on; {fails_off} outside it with the rewrite off")
