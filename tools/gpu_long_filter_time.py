import sys, time; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import blah2_amd as b2
from oracle import blah2_oracle as O
from test_fused_fir_gpu import synth
n, B = 2_000_000, 2
dmin, dmax = -10, 4990
x, y = synth(n, n, 5)
xs = torch.from_numpy(np.stack([x, x])).cuda(); ys = torch.from_numpy(np.stack([y, y])).cuda()
yf = torch.empty_like(ys); ok = torch.zeros(B, dtype=torch.int32, device="cuda")
wh = b2.WienerHopf(dmin, dmax, n, max_batch=B)
st = torch.cuda.current_stream().cuda_stream
wh.set_timing(True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wh.process_dev(xs.data_ptr(), ys.data_ptr(), B, n, yf.data_ptr(), ok.data_ptr(), st)
    torch.cuda.synchronize(); print("5000 taps, 2 CPIs of 2M samples: %.1f ms" % ((time.perf_counter() - t0) * 1e3), ok.cpu().tolist())
print(wh.get_timing())
t0 = time.perf_counter(); okr, yfr = O.wiener_hopf(x.astype(np.complex128), y.astype(np.complex128), dmin, dmax)[:2]; print("oracle %.1f s" % (time.perf_counter() - t0))
out = yf[0].cpu().numpy()
print("error", np.max(np.abs(out - yfr)) / np.max(np.abs(yfr)))
