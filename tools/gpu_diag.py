#!/usr/bin/env python3
"""GPU diagnostics (run through gpurun): accuracy statistics against the fp64
oracle and a timing sweep over the planner's knobs.  Not part of the product."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import blah2_amd
from oracle import blah2_oracle as O

CFG2 = (-10, 400, -256, 256, 2_000_000, 2_000_000)


def accuracy():
    dmin, dmax, fmin, fmax, fs, n = CFG2
    x, y = O.synth_iq(n, fs=fs)
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
    ref = O.ambiguity_process(d, x, y)
    for fft_len in (1024, 2048, 4096):
        amb = blah2_amd.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True)
        amb.set_fft_len(fft_len)
        m = amb.process(x, y).data.astype(np.complex128)
        err = np.abs(m - ref)
        peak = np.max(np.abs(ref))
        strong = np.abs(ref) > np.mean(np.abs(ref))
        rel = err / np.abs(ref)
        print(f"F={fft_len} seg={amb.dims.n_seg}x{amb.dims.seg_len}: peak-rel {err.max()/peak:.3e}  "
              f"cell-rel strong max {rel[strong].max():.3e} p99.9 {np.quantile(rel[strong], 0.999):.3e} "
              f"median {np.median(rel[strong]):.3e}; all cells max {rel.max():.3e}; "
              f"dB max {np.max(np.abs(10*np.log10(np.abs(m)) - 10*np.log10(np.abs(ref)))):.3e}", flush=True)
        amb.close()


def timing():
    dmin, dmax, fmin, fmax, fs, n = CFG2
    dev = torch.device("cuda", 0)
    for fft_len in (1024, 2048, 4096):
        for B in [int(b) for b in os.environ.get("DIAG_B", "1,8,16").split(",")]:
            amb = blah2_amd.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=B)
            amb.set_fft_len(fft_len)
            ring = max(2, int(600e6 // (16 * n * B)) + 1)
            xs = [torch.view_as_complex(torch.round(300 * torch.randn((B, n, 2), device=dev))) for _ in range(ring)]
            ys = [torch.view_as_complex(torch.round(300 * torch.randn((B, n, 2), device=dev))) for _ in range(ring)]
            nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
            out = torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev)
            met = torch.zeros((B, 2), dtype=torch.float64, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            steps = max(10, 80 // B)
            for i in range(3):
                amb.process_dev(0, xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                amb.process_dev(0, xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            amb.set_timing(True)
            for i in range(steps):
                amb.process_dev(0, xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
            kt = amb.get_timing()
            amb.set_timing(False)
            per = {k: v[0] / max(v[1], 1) * 1e3 / B for k, v in kt.items() if v[1]}
            print(f"F={fft_len} B={B}: {el/steps/B*1e6:8.2f} us/CPI wall  kernels us/CPI: " +
                  " ".join(f"{k}={v:.2f}" for k, v in per.items()), flush=True)
            amb.close()
            del xs, ys


if __name__ == "__main__":
    what = sys.argv[1:] or ["accuracy", "timing"]
    if "accuracy" in what:
        accuracy()
    if "timing" in what:
        timing()
