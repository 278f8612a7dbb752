#!/usr/bin/env python3
"""Run-to-run determinism of the chain's stages: the filter's normal equations (r, b), taps, filtered channel, the map (two-stage and
fused), at a geometry the fused kernel covers.   python tools/gpu_determinism_diag.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import blah2_amd as b2  # noqa: E402
from test_fused_fir_gpu import GEOMETRIES, synth  # noqa: E402

args = GEOMETRIES["pulse ends 5 samples past a boundary (nCorr 6149)"]
dmin, dmax, fmin, fmax, fs, n = args
x, y = synth(n, fs, 51, echo=(37, -3.0, 0.2), noise=3.0)
dev = torch.device("cuda", 0)
xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
st = torch.cuda.current_stream().cuda_stream
res = []
for rep in range(3):
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True)
    amb.set_fft_len(4096)
    amb.set_hot_columns(os.environ.get("HOT", "auto"))
    wh = b2.WienerHopf(dmin, dmax, n)
    if os.environ.get("SOLVE") == "stepwise":
        wh.set_solve_form("stepwise")
    nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
    yf = torch.empty_like(yd)
    ok = torch.zeros(1, dtype=torch.int32, device=dev)
    m2 = torch.zeros((nD, nC), dtype=torch.complex64, device=dev)
    mf = torch.zeros((nD, nC), dtype=torch.complex64, device=dev)
    met = torch.zeros(2, dtype=torch.float64, device=dev)
    wh.process_dev(xd.data_ptr(), yd.data_ptr(), 1, n, yf.data_ptr(), ok.data_ptr(), st)
    torch.cuda.synchronize()
    _, w, r, b = wh.read_last()
    amb.process_dev(b2.FMT_C32, xd.data_ptr(), yf.data_ptr(), 1, n, m2.data_ptr(), met.data_ptr(), st)
    amb.set_fir(wh)
    wh.estimate_dev_fmt(b2.FMT_C32, xd.data_ptr(), yd.data_ptr(), 1, n, ok.data_ptr(), st)
    amb.process_dev(b2.FMT_C32, xd.data_ptr(), yd.data_ptr(), 1, n, mf.data_ptr(), met.data_ptr(), st)
    torch.cuda.synchronize()
    _, w2, _, _ = wh.read_last()
    res.append(dict(r=r, b=b, w=w, w_est=w2, yf=yf.cpu().numpy(), two=m2.cpu().numpy(), fused=mf.cpu().numpy()))
    amb.close()
    wh.close()
for k in res[0]:
    same = [np.array_equal(res[0][k], res[i][k]) for i in (1, 2)]
    d = max(float(np.max(np.abs(res[0][k].astype(np.complex128) - res[i][k]))) for i in (1, 2))
    print(f"{k:6s} identical across runs: {same}   max |difference| {d:.3e} (max |value| {float(np.max(np.abs(res[0][k]))):.3e})")
