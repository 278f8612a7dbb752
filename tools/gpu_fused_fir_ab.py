#!/usr/bin/env python3
"""FIR -> range fusion, the prototype's A/B at configs[2]: the filter's FIR inside the range kernel (range_fir_kernel,
Ambiguity.set_fir) against the two-stage path (clutter_fir_kernel writes y', the range kernel reads it): maps against each
other and against the oracle chain, and the time of both per CPI.   python tools/gpu_fused_fir_ab.py [batch [reps]]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import blah2_amd as b2  # noqa: E402
from oracle import blah2_oracle as O  # noqa: E402
from oracle import gates as G  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg, _ = bench.CONFIGS["cfg3"]
dmin, dmax, fmin, fmax, fs, n = cfg
dev = torch.device("cuda", 0)
x, y = bench.synth_batch(torch, B, n, 7000, fs, dev)
st = torch.cuda.current_stream().cuda_stream
amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=B)
wh = b2.WienerHopf(dmin, dmax, n, max_batch=B)
nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
yf = torch.empty_like(y)
ok = torch.zeros(B, dtype=torch.int32, device=dev)
out2 = torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev)
outf = torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev)
met = torch.zeros((B, 2), dtype=torch.float64, device=dev)


def two_stage():
    wh.process_dev(x.data_ptr(), y.data_ptr(), B, n, yf.data_ptr(), ok.data_ptr(), st)
    amb.process_dev(b2.FMT_C32, x.data_ptr(), yf.data_ptr(), B, n, out2.data_ptr(), met.data_ptr(), st)


def fused():
    wh.estimate_dev_fmt(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, ok.data_ptr(), st)
    amb.process_dev(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, outf.data_ptr(), met.data_ptr(), st)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / B * 1e6


res = {"config": "cfg3", "batch": B, "reps": reps}
amb.set_fir(None)
res["two_stage_us_per_cpi"] = timed(two_stage)
amb.set_fir(wh)
res["fused_us_per_cpi"] = timed(fused)
res["fused_range_kernel"] = amb.info(b2._lib.INFO_LAST_RANGE_KERNEL)
a, f = out2.cpu().numpy(), outf.cpu().numpy()
xh, yh = x[0].cpu().numpy().astype(np.complex128), y[0].cpu().numpy().astype(np.complex128)
okr, yfr = O.wiener_hopf(xh, yh, dmin, dmax)[:2]
d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
ref = O.ambiguity_process(d, xh, yfr)
noise = O.map_metrics(ref)[0]
nm = G.notch_mask(ref.shape, d.doppler, d.delay, dmin, dmax)
for name, m in (("two_stage", a[0]), ("fused", f[0])):
    c = G.map_cell_gate(m, ref, noise, notch=nm)
    res[name + "_vs_oracle"] = {k: c[k] for k in ("ok", "cell_rel_above_mean_outside_notch", "peak_rel", "abs_err_over_mean_level")}
res["fused_vs_two_stage_max_abs_over_mean_level"] = float(np.abs(f.astype(np.complex128) - a).max() / G.mean_level(noise))
print(json.dumps(res, indent=1))
