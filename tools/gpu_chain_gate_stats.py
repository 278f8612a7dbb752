#!/usr/bin/env python3
"""The full chain's parity block (bench.parity_check: every gate of oracle/gates.py) over MANY CPIs of a configuration: the
spread of what the driver's run checks on one CPI per leg.   python tools/gpu_chain_gate_stats.py cfg2 12 1d   |  cfg3 4 2d"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import blah2_amd as b2  # noqa: E402
from oracle import blah2_oracle as O  # noqa: E402
from oracle import gates as G  # noqa: E402

config, n_cpi, cfar = sys.argv[1], int(sys.argv[2]), sys.argv[3]
fir = sys.argv[4] if len(sys.argv) > 4 else "auto"  # auto: fused into the range kernel where covered (what bench.py and the replay run)
cfg, _ = bench.CONFIGS[config]
dmin, dmax, fmin, fmax, fs, n = cfg
dev = torch.device("cuda", 0)
B = min(n_cpi, 8)
amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=B)
wh = b2.WienerHopf(dmin, dmax, n, max_batch=B)
det_params = (1e-5, 2, 6, 1, 3, 5, 15.0) if cfar == "2d" else (1e-5, 2, 6, 5, 15.0)
det = b2.CfarDetector2D(*det_params) if cfar == "2d" else b2.CfarDetector1D(*det_params)
nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
fused = fir != "two-stage" and amb.fir_fusable(wh, b2.FMT_C32) is None
if fused:
    amb.set_fir(wh)
print(f"{config}: FIR {'fused into the range kernel' if fused else 'two-stage'}")
CAP = 65536
st = torch.cuda.current_stream().cuda_stream
worst = {}
for b0 in range(0, n_cpi, B):
    x, y = bench.synth_batch(torch, B, n, 5000 + b0, fs, dev)
    yf = torch.empty_like(y)
    ok = torch.zeros(B, dtype=torch.int32, device=dev)
    out = torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev)
    met = torch.zeros((B, 2), dtype=torch.float64, device=dev)
    hits = torch.zeros((B, CAP, 2), dtype=torch.float64, device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    if fused:
        wh.estimate_dev_fmt(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, ok.data_ptr(), st)
        amb.process_dev(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
    else:
        wh.process_dev(x.data_ptr(), y.data_ptr(), B, n, yf.data_ptr(), ok.data_ptr(), st)
        amb.process_dev(b2.FMT_C32, x.data_ptr(), yf.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
    det.process_dev(amb, B, hits.data_ptr(), CAP, cnt.data_ptr(), out.data_ptr(), met.data_ptr(), st)
    torch.cuda.synchronize()
    o, m, okh, ch = out.cpu().numpy(), met.cpu().numpy(), ok.cpu().numpy(), cnt.cpu().numpy()
    for c in range(B):
        rec = hits[c, :max(int(ch[c]), 1)].cpu().numpy().view(b2.HIT_DTYPE).reshape(-1)
        dt = b2.hits_to_detection(amb, rec, int(ch[c]), CAP)
        r = bench.parity_check(np, O, G, cfg, "c32", "full", cfar, 0, x[c].cpu().numpy().astype(np.complex128), y[c].cpu().numpy().astype(np.complex128),
                               o[c], m[c], det_params, (dt.get_delay(), dt.get_doppler()), int(okh[c]))
        keys = ("cell_rel_above_mean", "cell_rel_above_mean_outside_notch", "db_max", "notch_abs_err_over_mean_level", "chain_err_over_direct_path", "metrics_db")
        print(f"cpi {b0 + c}: pass {r['pass']} detections {r['detections']['n_got']}/{r['detections']['n_ref']} differ {r['detections']['n_differ']}  " +
              "  ".join(f"{k} {r[k]:.2e}" for k in keys), flush=True)
        for k in keys:
            worst[k] = max(worst.get(k, 0.0), r[k])
        worst["fails"] = worst.get("fails", 0) + (0 if r["pass"] else 1)
print("worst:", worst)
