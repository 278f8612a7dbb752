#!/usr/bin/env python3
"""Range-kernel time against resident workgroups per CU (grid cap): is it latency-bound (time ~ 1/occupancy)
or throughput-bound (flat)?  Run through gpurun."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blah2_amd
dmin, dmax, fmin, fmax, fs, n = (-10, 400, -256, 256, 2_000_000, 2_000_000)
dev = torch.device("cuda", 0)
B = 64
x = torch.view_as_complex(torch.round(300 * torch.randn((2 * B, n, 2), device=dev)))
y = torch.view_as_complex(torch.round(300 * torch.randn((2 * B, n, 2), device=dev)))
amb = blah2_amd.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=B)
nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
out = torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev)
met = torch.zeros((B, 2), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for fmt, name in ((blah2_amd.FMT_C32, "c32"),):
    for grid in (256, 512, 768, 1024, 1280, 2048):
        amb.set_range_grid(grid)
        for i in range(2):
            amb.process_dev(fmt, x[(i % 2) * B].data_ptr(), y[(i % 2) * B].data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
        amb.set_timing(True)
        for i in range(6):
            amb.process_dev(fmt, x[(i % 2) * B].data_ptr(), y[(i % 2) * B].data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
        kt = amb.get_timing()
        amb.set_timing(False)
        r = kt["range"][0] / kt["range"][1] * 1e3
        print(f"{name} grid {grid:5d} ({grid / 256:.0f} workgroups per CU requested): range {r:8.1f} us per {B}-CPI launch = {r / B:6.2f} us/CPI", flush=True)
