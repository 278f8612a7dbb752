"""wave1k against the E8 kernel on the same inputs: where do the maps differ? (run through gpurun)"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import blah2_amd as b2
from blah2_amd import _lib
from oracle import blah2_oracle as O
args = (-10, 400, -256, 256, 2_000_000, 2_000_000)
n = args[5]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
xs, ys = zip(*(O.synth_iq(n, seed=s, fs=args[4], targets=((100, 40.0, 0.1), (17, -3.0, 0.05))) for s in range(2)))
dx = torch.from_numpy(np.stack([xs[i % 2] for i in range(B)]).astype(np.complex64)).cuda()
dy = torch.from_numpy(np.stack([ys[i % 2] for i in range(B)]).astype(np.complex64)).cuda()
res = {}
for name, k in (("e8", _lib.RANGE_E8), ("wave1k", _lib.RANGE_WAVE1K)):
    amb = b2.Ambiguity(*args, True, max_batch=B)
    amb.set_fft_len(1024)
    amb.set_range_kernel(k)
    out = torch.zeros((B, amb.get_n_doppler_bins(), amb.get_n_delay_bins()), dtype=torch.complex64, device="cuda")
    amb.process_dev(b2.FMT_C32, dx.data_ptr(), dy.data_ptr(), B, n, out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    res[name] = out.cpu().numpy()
    print(name, "range kernel", amb.info(_lib.INFO_LAST_RANGE_KERNEL), "seg", amb.dims.n_seg, amb.dims.seg_len)
d = np.abs(res["e8"] - res["wave1k"])
peak = np.abs(res["e8"]).max()
print("max rel", d.max() / peak)
print("per cpi", (d.reshape(B, -1).max(1) / peak).round(6))
print("per delay bin (cpi 0), worst 12:", np.argsort(-d[0].max(0))[:12], (np.sort(-d[0].max(0))[:12] / -peak).round(5))
# undo the Doppler transform to see which pulses are wrong: inverse DFT along the Doppler axis
r0 = np.fft.ifft(np.fft.ifftshift(res["e8"][0], axes=0), axis=0)
r1 = np.fft.ifft(np.fft.ifftshift(res["wave1k"][0], axes=0), axis=0)
dp = np.abs(r0 - r1)
print("per pulse (cpi 0), worst 12:", np.argsort(-dp.max(1))[:12], (np.sort(-dp.max(1))[:12] / -np.abs(r0).max()).round(5))
print("pulse x delay of the worst:", np.unravel_index(np.argmax(dp), dp.shape))
# which segment's contribution is the difference?  (scale: the engine divides by F, and the Doppler inverse above by nD)
pu = int(np.argsort(-dp.max(1))[0])
nCorr, segLen, nSeg = 3898, 557, 7
x = xs[0].astype(np.complex128)[pu * nCorr:(pu + 1) * nCorr]
y = ys[0].astype(np.complex128)[pu * nCorr:(pu + 1) * nCorr]
lags = np.arange(-10, 401)
def seg_corr(s):
    out = np.zeros(lags.size, complex)
    k = np.arange(s * segLen, min((s + 1) * segLen, nCorr))
    for j, l in enumerate(lags):
        ky = k + l
        ok = (ky >= 0) & (ky < nCorr)
        out[j] = np.sum(y[ky[ok]] * np.conj(x[k[ok]]))
    return out
diff = (r0 - r1)[pu]
tot = sum(seg_corr(s) for s in range(nSeg))
sc = np.vdot(tot, r0[pu]) / np.vdot(tot, tot)
print("scale engine/direct", sc)
for s in range(nSeg):
    c = seg_corr(s) * sc
    print("segment", s, "|diff - c_s| / |diff| =", np.linalg.norm(diff - c) / np.linalg.norm(diff), " |c_s|/|diff| =", np.linalg.norm(c) / np.linalg.norm(diff))
