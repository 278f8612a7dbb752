#!/usr/bin/env python3
"""wall_clock64 buckets of the look-ahead Toeplitz solve's front and bulk waves (library built with -DSLA_TRACE:
tools/build_trace.sh; run with BLAH2HIP_LIBRARY=tools/ab/lib_sla_trace.so)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blah2_amd as b2  # noqa: E402
from tools.gpu_solve import normal_equations  # noqa: E402

st = torch.cuda.current_stream().cuda_stream
a = torch.randn(8192, 8192, device="cuda")
for n in (410, 2047):
    r, b = normal_equations(n, n)
    rb = torch.from_numpy(np.stack([r, b])[None].copy()).cuda()
    w = torch.zeros((1, n), dtype=torch.complex64, device="cuda")
    ok = torch.zeros(1, dtype=torch.int32, device="cuda")
    for E in (2, 3, 12):
        wh = b2.WienerHopf(-1, n - 1, 8192)
        wh.set_solve_form("lookahead", E)
        print(f"--- taps {n} E {E}", flush=True)
        for _ in range(40):   # ramp the clock
            a @ a
        for _ in range(2):
            wh.solve_dev(rb.data_ptr(), 1, w.data_ptr(), ok.data_ptr(), st)
        torch.cuda.synchronize()
        wh.close()
