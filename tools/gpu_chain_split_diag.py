"""Where the full chain's map error comes from: the filter's share (the engine's filtered channel through the ORACLE's map
stage) and the map stage's share, with the five worst cells of each."""
import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
import torch
import bench
import blah2_amd as b2
from oracle import blah2_oracle as O
from oracle import gates as G
# python tools/gpu_chain_split_diag.py [config [seed [cpi]]]   (tools/gpu_chain_gate_stats.py: seed = 5000 + 8 * (cpi // 8), cpi % 8)
cfg, _ = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
dmin, dmax, fmin, fmax, fs, n = cfg
dev = torch.device("cuda", 0)
B = 8
x, y = bench.synth_batch(torch, B, n, int(sys.argv[2]) if len(sys.argv) > 2 else 5008, fs, dev)
c = int(sys.argv[3]) if len(sys.argv) > 3 else 4
amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=1)
wh = b2.WienerHopf(dmin, dmax, n, max_batch=1)
xc, yc = x[c:c+1].contiguous(), y[c:c+1].contiguous()
yf = torch.empty_like(yc); ok = torch.zeros(1, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
wh.process_dev(xc.data_ptr(), yc.data_ptr(), 1, n, yf.data_ptr(), ok.data_ptr(), st)
amb.process_dev(b2.FMT_C32, xc.data_ptr(), yf.data_ptr(), 1, n, None, None, st)
torch.cuda.synchronize()
m = amb.read_last(0)
xh, yh = xc[0].cpu().numpy().astype(np.complex128), yc[0].cpu().numpy().astype(np.complex128)
okr, yfr, w, r, b = O.wiener_hopf(xh, yh, dmin, dmax, return_filter=True)
d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
ref = O.ambiguity_process(d, xh, yfr)
# the engine's filtered channel through the ORACLE's ambiguity: separates the filter's error from the map stage's
yfe = yf[0].cpu().numpy().astype(np.complex128)
ref_e = O.ambiguity_process(d, xh, yfe)
noise, _ = O.map_metrics(ref); lvl = G.mean_level(noise)
nm = G.notch_mask(ref.shape, d.doppler, d.delay, dmin, dmax)
got = m.data.astype(np.complex128)
for name, a_, b_ in (("engine map vs oracle chain", got, ref), ("oracle map of ENGINE's filtered y vs oracle chain (the filter's share)", ref_e, ref),
                     ("engine map vs oracle map of the engine's filtered y (the map stage's share)", got, ref_e)):
    err = np.abs(a_ - b_); above = (np.abs(ref) > lvl) & ~nm
    q = np.where(above, err / np.abs(ref), 0)
    print(name, "max cell-rel outside notch %.2e" % q.max(), " rms err/level %.2e" % (np.sqrt(np.mean(err[~nm]**2))/lvl))
    for idx in np.argsort(q.ravel())[::-1][:5]:
        i, j = np.unravel_index(idx, q.shape)
        print("    row %d (%.0f Hz) lag %d: |ref|/level %.2f err/level %.2e" % (i, d.doppler[i], d.delay[j], abs(ref[i,j])/lvl, err[i,j]/lvl))
print("y_f err rms %.2e of |y_f| rms %.2f (|y| rms %.1f)" % (np.sqrt(np.mean(np.abs(yfe-yfr)**2)), np.sqrt(np.mean(np.abs(yfr)**2)), np.sqrt(np.mean(np.abs(yh)**2))))
