#!/usr/bin/env python3
"""Range-kernel time vs batch size (reveals the resident-workgroup count)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blah2_amd
dmin, dmax, fmin, fmax, fs, n = (-10, 400, -256, 256, 2_000_000, 2_000_000)
dev = torch.device("cuda", 0)
BMAX = 18
x = torch.view_as_complex(torch.round(300 * torch.randn((BMAX, n, 2), device=dev)))
y = torch.view_as_complex(torch.round(300 * torch.randn((BMAX, n, 2), device=dev)))
amb = blah2_amd.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=BMAX)
nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
out = torch.zeros((BMAX, nD, nC), dtype=torch.complex64, device=dev)
met = torch.zeros((BMAX, 2), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
print("F", amb.dims.fft_len, "seg", amb.dims.n_seg, amb.dims.seg_len)
for B in range(1, BMAX + 1):
    for _ in range(2):
        amb.process_dev(0, x.data_ptr(), y.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
    amb.set_timing(True)
    for _ in range(8):
        amb.process_dev(0, x.data_ptr(), y.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
    kt = amb.get_timing()
    amb.set_timing(False)
    r = kt["range"][0] / kt["range"][1] * 1e3
    d = kt["doppler"][0] / kt["doppler"][1] * 1e3
    print(f"B={B:2d} pulses={513*B:5d} range={r:8.1f} us ({r/B:6.2f}/CPI) doppler={d:7.1f} us ({d/B:5.2f}/CPI)", flush=True)
