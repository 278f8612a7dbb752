#!/usr/bin/env python3
"""Repeats the tile-kernel 2-D CFAR on one fixture and reports detection sets that differ between runs
or from the SAT kernel (race hunting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import blah2_amd as b2
from conftest import load_golden

g = load_golden("medium")
fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, bool(rh))
m = amb.process(g["x"], g["y"])
for params in [(1e-2, 1, 3, 1, 2, -10, 0.0), (1e-4, 2, 6, 1, 3, 5, 15.0), (1e-3, 0, 1, 0, 1, 0, 0.0)]:
    amb.set_cfar2d_kernel("sat")
    d = b2.CfarDetector2D(*params).process(m)
    ref = set(zip(d.get_delay(), d.get_doppler()))
    amb.set_cfar2d_kernel("tile")
    bad = 0
    for it in range(200):
        d = b2.CfarDetector2D(*params).process(m)
        got = set(zip(d.get_delay(), d.get_doppler()))
        if got != ref:
            bad += 1
            if bad <= 5:
                print(params, "iter", it, "extra", sorted(got - ref)[:6], "missing", sorted(ref - got)[:6])
    print(params, "mismatching runs:", bad, "of 200; detections", len(ref))
