#!/usr/bin/env python3
"""Where does the full chain differ from the compiled reference?  For a golden fixture: the ten cells with the largest
error relative to max(|cell|, mean level), with their Doppler row / lag / level, and the taps' own error.
    python tools/gpu_chain_diag.py [fixture ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import blah2_amd as b2  # noqa: E402
from conftest import load_golden  # noqa: E402
from oracle import blah2_oracle as O  # noqa: E402
from oracle import gates as G  # noqa: E402

for name in sys.argv[1:] or ["medium", "deep_cancel"]:
    g = load_golden(name)
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    cmin, cmax = (int(v) for v in g["clutter_params"])
    wh = b2.WienerHopf(cmin, cmax, n)
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    x = torch.from_numpy(g["x"].astype(np.complex64)).cuda()
    y = torch.from_numpy(g["y"].astype(np.complex64)).cuda()
    okf = torch.zeros(1, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    wh.process_dev(x.data_ptr(), y.data_ptr(), 1, n, y.data_ptr(), okf.data_ptr(), st)
    amb.process_dev(b2.FMT_C32, x.data_ptr(), y.data_ptr(), 1, n, None, None, st)
    torch.cuda.synchronize()
    m = amb.read_last(0)
    _, w, r, b = wh.read_last(0)
    ok, yref, w_ref, r_ref, b_ref = O.wiener_hopf(g["x"], g["y"], cmin, cmax, return_filter=True)
    yf = y.cpu().numpy().astype(np.complex128)
    ref = np.asarray(g["chain_map"], dtype=np.complex128)
    noise = float(g["chain_metrics"][0])
    level = G.mean_level(noise)
    err = np.abs(m.data.astype(np.complex128) - ref)
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, bool(rh))
    q = err / np.maximum(np.abs(ref), level)
    print(f"== {name}: F={wh.fft_len} taps {cmax - cmin}; w err max {np.abs(w - w_ref).max():.2e} (|w|max {np.abs(w_ref).max():.3f}); "
          f"w err at the largest tap {abs(w[np.argmax(np.abs(w_ref))] - w_ref[np.argmax(np.abs(w_ref))]):.2e}; "
          f"y err rms {np.sqrt(np.mean(np.abs(yf - g['clutter_y']) ** 2)):.2e} max {np.abs(yf - g['clutter_y']).max():.2e} "
          f"(|y_filtered| rms {np.sqrt(np.mean(np.abs(g['clutter_y']) ** 2)):.2f}, |y| rms {np.sqrt(np.mean(np.abs(g['y']) ** 2)):.1f})")
    # coherent part of the filtered channel's error: its correlation with the reference channel at the filter's lags
    e = yf - g["clutter_y"]
    xs = np.roll(g["x"], cmin)  # xs[i] = x[i - delayMin]
    coh = np.array([np.vdot(np.roll(xs, k), e) for k in range(0, min(cmax - cmin, 12))]) / np.vdot(xs, xs).real
    print("   <e, xs shifted by k> / <xs, xs>, k = 0..11 (an effective tap error):", " ".join(f"{abs(c):.1e}" for c in coh))
    for idx in np.argsort(q.ravel())[::-1][:10]:
        i, j = np.unravel_index(idx, q.shape)
        print(f"   row {i} (doppler {d.doppler[i]:+.1f} Hz) lag {d.delay[j]:4d}: |ref| {10 * np.log10(np.abs(ref[i, j]) / level):+6.1f} dB re mean level, "
              f"err/cell {err[i, j] / np.abs(ref[i, j]):.2e}, err/level {err[i, j] / level:.2e}")
