import numpy as np, sys
sys.path.insert(0, ".")
import blah2_amd
from oracle import blah2_oracle as O
for args in [(-10, 300, -300, 300, 2_000_000, 1_000_000, True), (-10, 400, -256, 256, 2_000_000, 2_000_000, True)]:
    dmin, dmax, fmin, fmax, fs, n, rh = args
    x, y = O.synth_iq(n, fs=fs)
    amb = blah2_amd.Ambiguity(*args)
    m = amb.process(x, y)
    d = O.ambiguity_dims(*args)
    ref = O.ambiguity_process(d, x, y)
    noise, mx = O.map_metrics(ref)
    dbr = 10 * np.log10(np.abs(ref)) - noise
    dbg = 10 * np.log10(np.abs(m.data.astype(np.complex128))) - m.noisePower
    dd = np.abs(dbr - dbg)
    t2 = lambda v: np.trunc(v * 100) / 100
    print(args[:2], "max dB diff", dd.max(), "cells > 0.005:", (dd > 0.005).sum(), "of", dd.size,
          "2-decimal strings differ:", (t2(dbr) != t2(dbg)).sum(), "min dB", dbr.min(), "p1", np.percentile(dbr, 1))
