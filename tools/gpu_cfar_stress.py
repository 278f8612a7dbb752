#!/usr/bin/env python3
"""Repeats the stream and the tile 2-D CFAR kernels on one fixture and on batches of synthetic cfg 3 maps and reports
detection sets that differ between runs or from the SAT kernel (race hunting)."""
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
for params in [(1e-2, 1, 3, 1, 2, -10, 0.0), (1e-4, 2, 6, 1, 3, 5, 15.0), (1e-3, 0, 2, 0, 1, 0, 0.0)]:
    amb.set_cfar2d_kernel("sat")
    d = b2.CfarDetector2D(*params).process(m)
    ref = set(zip(d.get_delay(), d.get_doppler()))
    for which in ("stream", "tile"):
        amb.set_cfar2d_kernel(which)
        bad = 0
        for it in range(200):
            d = b2.CfarDetector2D(*params).process(m)
            got = set(zip(d.get_delay(), d.get_doppler()))
            if got != ref:
                bad += 1
                if bad <= 5:
                    print(params, which, "iter", it, "extra", sorted(got - ref)[:6], "missing", sorted(ref - got)[:6])
        print(params, which, "mismatching runs:", bad, "of 200; detections", len(ref))

# batches of synthetic maps at the cfg 3 size: the same launch 100 times, hit cells compared with the first run's
import torch
import bench
(dmin, dmax, fmin, fmax, fs, n), _ = bench.CONFIGS["cfg3"]
B = 16
amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=B)
dev = torch.device("cuda", 0)
x, y = bench.synth_batch(torch, B, n, 4321, fs, dev)
nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
out = torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev)
met = torch.zeros((B, 2), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
amb.process_dev(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
CAP = 1 << 16
hits = torch.zeros((B, CAP, 2), dtype=torch.float64, device=dev)
cnt = torch.zeros(B, dtype=torch.int32, device=dev)
det = b2.CfarDetector2D(1e-5, 2, 6, 1, 3, 5, 15.0)
for which in ("stream", "tile"):
    amb.set_cfar2d_kernel(which)
    first, bad = None, 0
    for it in range(100):
        det.process_dev(amb, B, hits.data_ptr(), CAP, cnt.data_ptr(), out.data_ptr(), met.data_ptr(), st)
        torch.cuda.synchronize()
        c = cnt.cpu().numpy()
        h = hits.cpu().numpy().view(np.int32).reshape(B, CAP, 4)[:, :, :2]
        cells = [frozenset(map(tuple, h[b, :min(int(c[b]), CAP)])) for b in range(B)]
        if first is None:
            first = cells
        elif cells != first:
            bad += 1
    print("cfg3 x", B, which, "mismatching runs:", bad, "of 100; hits per CPI", [len(s_) for s_ in first][:4])
