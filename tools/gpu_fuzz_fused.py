#!/usr/bin/env python3
"""Random geometries through the fused FIR + range kernel against the two-stage chain (and random long clutter filters against the
oracle): a stress run beside the fixed cases of tests/test_fused_fir_gpu.py / test_clutter_long_gpu.py.
    python tools/gpu_fuzz_fused.py [cases [seed]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import blah2_amd as b2  # noqa: E402
from oracle import blah2_oracle as O  # noqa: E402
from test_fused_fir_gpu import run_chains, synth  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
done = 0
while done < cases:
    dmin = -int(rng.integers(0, 40))
    nbins = int(rng.integers(max(1, -dmin), 2049))
    dmax = dmin + nbins
    half = int(rng.integers(2, 12))
    nD = 2 * half + 1
    ncorr = int(rng.integers(2048 - dmin, 9000))
    slack = int(rng.integers(-dmin, -dmin + nD)) if nD > -dmin else -1
    if slack < -dmin or slack >= nD:
        continue
    n = ncorr * nD + slack
    args = (dmin, dmax, -half, half, n, n)
    try:
        d = O.ambiguity_dims(*args, True)
    except Exception:
        continue
    if d.n_corr != ncorr or d.n_doppler_bins != nD or d.n_delay_bins > 2049:
        continue
    fmt = "i16" if rng.integers(0, 2) else "c32"
    B = int(rng.integers(1, 4))
    data = [synth(n, n, int(rng.integers(1, 1 << 30))) for _ in range(B)]
    two, fus, ok2, okf = run_chains(b2, args, np.stack([v[0] for v in data]), np.stack([v[1] for v in data]), fmt)
    assert list(ok2) == [1] * B and list(okf) == [1] * B
    # against the UNFILTERED map's floor, sqrt(N) rms(x) rms(y): a short lag window can lie wholly inside the filter's, and what is
    # left of such a map is cancellation residue (its own peak is no yardstick)
    floor = np.sqrt(n) * np.sqrt(np.mean(np.abs(data[0][0]) ** 2) * np.mean(np.abs(data[0][1]) ** 2))
    dabs = float(np.abs(fus.astype(np.complex128) - two).max())
    err, err_peak = dabs / floor, dabs / float(np.abs(two).max())
    worst = max(worst, err)
    print(f"case {done}: lags {dmin}..{dmax} ({nbins} taps), {nD} pulses of {ncorr}, slack {slack}, {fmt}, batch {B}: fused - two-stage "
          f"{err:.2e} of the unfiltered floor ({err_peak:.2e} of the filtered map's peak)", flush=True)
    assert err <= 2e-5, err
    done += 1
print("fused: worst", worst)
# long filters
for k in range(max(2, cases // 10)):
    dmin = int(rng.integers(-30, 3))
    nbins = int(rng.integers(4082, 7000))
    n = int(rng.integers(nbins + 5000, 50000))
    x, y = synth(n, n, int(rng.integers(1, 1 << 30)))
    wh = b2.WienerHopf(dmin, dmin + nbins, n)
    ok, yf = wh.process(x, y)
    okr, yfr = O.wiener_hopf(x.astype(np.complex128), y.astype(np.complex128), dmin, dmin + nbins)[:2]
    err = float(np.max(np.abs(yf - yfr)) / np.max(np.abs(yfr)))
    print(f"long {k}: {nbins} taps from lag {dmin}, {n} samples: {err:.2e}", flush=True)
    assert ok and okr and err <= 1e-4
    wh.close()
print("all passed")
