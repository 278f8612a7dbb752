#!/usr/bin/env python3
"""Clock ramp after idle (run through gpurun): time of the ambiguity chain per step over the first
seconds of a run, per range kernel.  Not part of the product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blah2_amd
from bench import synth_batch

dev = torch.device("cuda", 0)
dmin, dmax, fmin, fmax, fs, n = -10, 400, -256, 256, 2_000_000, 2_000_000
B = 128
amb = blah2_amd.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=B)
parts = [synth_batch(torch, 16, n, 1000 + c0, fs, dev) for c0 in range(0, B, 16)]
x = torch.cat([p[0] for p in parts]); y = torch.cat([p[1] for p in parts]); del parts
nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
out = torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev)
met = torch.zeros((B, 2), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for kern in (0, blah2_amd._lib.RANGE_WAVE, 0, blah2_amd._lib.RANGE_WAVE):
    amb.set_range_kernel(kern)
    torch.cuda.synchronize(); time.sleep(2.0)  # back to idle clocks
    N = 1200
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        amb.process_dev(blah2_amd.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
    pts = [0, 1, 2, 5, 10, 20, 40, 80, 160, 320, 640, 1199]
    print("kernel", kern, " ".join(f"{i}:{ms[i]:.3f}" for i in pts), f"| mean 5..25: {sum(ms[5:25])/20:.3f}  mean 600..1200: {sum(ms[600:])/600:.3f}")
