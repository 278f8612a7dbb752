#!/usr/bin/env python3
"""Latency of the host-buffer entry points (what the drop-in classes use) at BASELINE configs[1]."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import blah2_amd as b2
from oracle import blah2_oracle as O  # synthetic IQ only

n, fs = 2_000_000, 2_000_000
x, y = O.synth_iq(n, fs=fs)
amb = b2.Ambiguity(-10, 400, -256, 256, fs, n, True)
for name, (xx, yy) in {"complex128 (IqData)": (x, y), "complex64": (x.astype(np.complex64), y.astype(np.complex64))}.items():
    amb.process(xx, yy)
    t = []
    for _ in range(10):
        t0 = time.perf_counter(); amb.process(xx, yy); t.append(time.perf_counter() - t0)
    print(f"Ambiguity.process, {name}: median {np.median(t)*1e3:.2f} ms, min {np.min(t)*1e3:.2f} ms")
iq = np.empty((n, 4), dtype=np.int16)
iq[:, 0], iq[:, 1], iq[:, 2], iq[:, 3] = x.real, x.imag, y.real, y.imag
amb.process_i16(iq) if hasattr(amb, "process_i16") else None
