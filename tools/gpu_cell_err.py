#!/usr/bin/env python3
"""The element-wise map error of a configuration over several CPIs: cell-rel above the mean level (the 1e-4 gate), its
rms over those cells, the JSON-map figure.   python tools/gpu_cell_err.py cfg5 f16 6"""
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

config, fmt, n_cpi = sys.argv[1], sys.argv[2], int(sys.argv[3])
leak_mode = sys.argv[4] if len(sys.argv) > 4 else "auto"
(dmin, dmax, fmin, fmax, fs, n), _ = bench.CONFIGS[config]
dev = torch.device("cuda", 0)
B = min(n_cpi, 8)
amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=B)
amb.set_leak_compensation(leak_mode)
d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
out = torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev)
met = torch.zeros((B, 2), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
x, y = bench.synth_batch(torch, B, n, 777, fs, dev)
if fmt == "f16":
    xd = torch.view_as_real(x).to(torch.float16).contiguous()
    yd = torch.view_as_real(y).to(torch.float16).contiguous()
    amb.process_dev(b2.FMT_F16, xd.data_ptr(), yd.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
    xh = xd.cpu().numpy().astype(np.float64)
    yh = yd.cpu().numpy().astype(np.float64)
    xh, yh = xh[..., 0] + 1j * xh[..., 1], yh[..., 0] + 1j * yh[..., 1]
else:
    amb.process_dev(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
    xh, yh = x.cpu().numpy().astype(np.complex128), y.cpu().numpy().astype(np.complex128)
torch.cuda.synchronize()
o, mt = out.cpu().numpy(), met.cpu().numpy()
print(f"{config} {fmt}: {nD} x {nC}, F = {amb.dims.fft_len} x {amb.dims.n_seg} segments, doppler kernel {amb.last_doppler_kernel()}, "
      f"leak compensation {leak_mode}: (lags corrected, max|g|) = {amb.leak_info()}")
for c in range(B):
    ref = O.ambiguity_process(d, xh[c], yh[c])
    noise, _ = O.map_metrics(ref)
    cell = G.map_cell_gate(o[c], ref, noise)
    dbg = G.db_map_gate(o[c], mt[c][0], ref, noise)
    err = np.abs(o[c].astype(np.complex128) - ref)
    lvl = G.mean_level(noise)
    # by Doppler row: is the error even across the map or concentrated
    row_rms = np.sqrt(np.mean(err ** 2, axis=1)) / lvl
    print(f"cpi {c}: cell-rel above mean {cell['cell_rel_above_mean']:.3e}  err rms/level {np.sqrt(np.mean(err ** 2)) / lvl:.2e}  max/level {err.max() / lvl:.2e}  "
          f"peak_rel {cell['peak_rel']:.2e}  dB shown {dbg['db_max_shown']:.5f}  row rms/level min {row_rms.min():.2e} max {row_rms.max():.2e} (row {row_rms.argmax()})")

# Is the zero-Doppler row's error a FIXED leak of the lag-0 column (g_d = err[k][d] / M[k][d0], the same for every
# CPI and every Doppler row)?  Estimate g from CPI 0's zero-Doppler row, apply it to the other CPIs.
if B >= 2:
    d0 = int(np.argmin(np.abs(d.delay)))
    k0 = int(np.argmin(np.abs(d.doppler)))
    refs = [O.ambiguity_process(d, xh[c], yh[c]) for c in range(min(B, 3))]
    e0 = o[0].astype(np.complex128) - refs[0]
    g = e0[k0] / refs[0][k0, d0]
    g[d0] = 0
    print(f"leak hypothesis: |g| max {np.abs(g).max():.2e} rms {np.sqrt(np.mean(np.abs(g) ** 2)):.2e}; largest at lags {d.delay[np.argsort(np.abs(g))[::-1][:8]]}")
    for c in range(1, len(refs)):
        ec = o[c].astype(np.complex128) - refs[c]
        lvl = G.mean_level(O.map_metrics(refs[c])[0])
        corr = ec - g[None, :] * refs[c][:, d0:d0 + 1]
        corr[:, d0] = ec[:, d0]
        above = np.abs(refs[c]) > lvl
        print(f"cpi {c}: zero-Doppler row err rms/level {np.sqrt(np.mean(np.abs(ec[k0]) ** 2)) / lvl:.2e} -> after removing g x column d0: "
              f"{np.sqrt(np.mean(np.abs(corr[k0]) ** 2)) / lvl:.2e};  cell-rel above mean {np.max(np.abs(ec)[above] / np.abs(refs[c])[above]):.2e} -> "
              f"{np.max(np.abs(corr)[above] / np.abs(refs[c])[above]):.2e};  peak cell err/peak {abs(ec[k0, d0]) / abs(refs[c][k0, d0]):.2e}")
