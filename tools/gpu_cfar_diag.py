#!/usr/bin/env python3
"""2-D CFAR kernel timing on a batch of synthetic maps (GPU box): stream vs tile vs SAT kernels, hit counts (and that
the forms report the same cells), and the same with a pfa so small that nothing fires (separates the hit-append
cost from the window sums).

    python tools/gpu_cfar_diag.py [cfg3|cfg2] [batch] [forms, e.g. stream,tile] [filter: 0|1|both]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import blah2_amd


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    forms = sys.argv[3].split(",") if len(sys.argv) > 3 else ["stream", "tile", "sat"]
    filts = {"0": (False,), "1": (True,), "both": (True, False)}[sys.argv[4] if len(sys.argv) > 4 else "0"]
    (dmin, dmax, fmin, fmax, fs, n), _ = bench.CONFIGS[cfgname]
    dev = torch.device("cuda", 0)
    amb = blah2_amd.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=B)
    wh = blah2_amd.WienerHopf(dmin, dmax, n, max_batch=B)
    x, y = bench.synth_batch(torch, B, n, 1234, fs, dev)
    nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
    out = torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev)
    met = torch.zeros((B, 2), dtype=torch.float64, device=dev)
    ok = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    CAP = 1 << 16
    hits = torch.zeros((B, CAP, 2), dtype=torch.float64, device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    for filt in filts:
        yy = y.clone()
        if filt:
            wh.process_dev(x.data_ptr(), yy.data_ptr(), B, n, yy.data_ptr(), ok.data_ptr(), st)
        amb.process_dev(blah2_amd.FMT_C32, x.data_ptr(), yy.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
        torch.cuda.synchronize()
        for pfa in (1e-5, 1e-300):
            cells = {}
            for which in forms:
                amb.set_cfar2d_kernel(which)
                det = blah2_amd.CfarDetector2D(pfa, 2, 6, 1, 3, 5, 15.0)
                t_pre = time.perf_counter()  # the shader clock needs ~0.3 s of load to ramp up
                while time.perf_counter() - t_pre < 0.5:
                    for _ in range(4):
                        det.process_dev(amb, B, hits.data_ptr(), CAP, cnt.data_ptr(), out.data_ptr(), met.data_ptr(), st)
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                R = 10
                for _ in range(R):
                    det.process_dev(amb, B, hits.data_ptr(), CAP, cnt.data_ptr(), out.data_ptr(), met.data_ptr(), st)
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / R / B * 1e6
                c = cnt.cpu().numpy()
                h = hits.cpu().numpy().view(np.int32).reshape(B, CAP, 4)[:, :, :2]  # blah2hip_hit_t: int32 row, col, double snr
                cells[which] = [set(map(tuple, h[b, :min(int(c[b]), CAP)])) for b in range(B)]
                same = all(cells[which][b] == cells[forms[0]][b] for b in range(B))
                print(f"{cfgname} x{B} clutter-filtered={filt} pfa={pfa:g} {which}: {us:7.2f} us/CPI, hits/CPI min {c.min()} max {c.max()}"
                      f"{'' if which == forms[0] else ', same cells as ' + forms[0] + ': ' + str(same)}", flush=True)


if __name__ == "__main__":
    main()
