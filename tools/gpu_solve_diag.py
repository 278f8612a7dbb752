#!/usr/bin/env python3
"""Toeplitz solve of the clutter filter (run through gpurun): time per launch and the residual
||A w - b|| / ||b|| of the device's OWN fp64 normal equations (isolates the solve from the fp32
correlations), for the indices-per-thread variants.  Not part of the product."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import blah2_amd
from oracle import blah2_oracle as O

dev = torch.device("cuda", 0)
for (dmin, dmax, n, B) in [(-10, 400, 2_000_000, 16), (-24, 2023, 10_000_000, 8), (-4, 124, 400_000, 4)]:
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    x = torch.view_as_complex(torch.round(300 * torch.randn((B, n, 2), generator=g, device=dev)))
    if dmax == 124:  # coloured reference: moving average of 16 samples
        xr = torch.view_as_real(x).cumsum(dim=1)
        xr = (xr[:, 16:] - xr[:, :-16]) / 4.0
        x = torch.view_as_complex(torch.round(torch.nn.functional.pad(xr, (0, 0, 0, 16))).contiguous())
    y = (0.8 * x + torch.view_as_complex(torch.round(30 * torch.randn((B, n, 2), generator=g, device=dev)))).contiguous()
    yo = torch.empty_like(y)
    okf = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for K in (0, 1, 2, 4):
        wh = blah2_amd.WienerHopf(dmin, dmax, n, max_batch=B)
        try:
            wh.set_solve_indices_per_thread(K)
        except blah2_amd.Blah2HipError as e:
            print(f"nBins={wh.nBins} K={K}: {e}")
            continue
        for _ in range(2):
            wh.process_dev(x.data_ptr(), y.data_ptr(), B, n, yo.data_ptr(), okf.data_ptr(), st)
        wh.set_timing(True)
        for _ in range(5):
            wh.process_dev(x.data_ptr(), y.data_ptr(), B, n, yo.data_ptr(), okf.data_ptr(), st)
        kt = wh.get_timing()
        ok, w, r, b = wh.read_last(B - 1)
        res = O.toeplitz_residual(r, w.astype(np.complex128), b)  # w is stored as fp32: floor ~6e-8
        print(f"nBins={wh.nBins} B={B} K={K}: solve {kt['clutter_solve'][0] / kt['clutter_solve'][1] * 1e3:8.1f} us/launch  "
              f"corr {kt['clutter_corr'][0] / kt['clutter_corr'][1] * 1e3 / B:6.1f} fir {kt['clutter_fir'][0] / kt['clutter_fir'][1] * 1e3 / B:6.1f} us/CPI  "
              f"ok={ok} residual {res:.2e}", flush=True)
        wh.close()
    del x, y, yo
