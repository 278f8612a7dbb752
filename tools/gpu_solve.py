#!/usr/bin/env python3
"""Time per launch of the clutter filter's Toeplitz solve kernels alone (blah2hip_clutter_solve_dev, launches back to back
on device-resident normal equations after a warm-up that ramps the clock; HIP events around each launch):
the one-workgroup stepwise kernel against the look-ahead form at each slice width, by taps and batch.
    python tools/gpu_solve.py [--json out.json] [--quick]"""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blah2_amd as b2  # noqa: E402


def normal_equations(n, seed):
    rng = np.random.default_rng(seed)
    m = 8 * n
    sig = 300.0 * (rng.standard_normal(m) + 1j * rng.standard_normal(m))
    f = np.fft.fft(sig, 2 * m)
    r = np.fft.ifft(np.abs(f) ** 2)[:n].copy()
    r[0] = r[0].real
    return r, (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * abs(r[0])


def toeplitz(r):
    n = r.size
    i, j = np.indices((n, n))
    return np.where(i >= j, r[np.abs(i - j)], np.conj(r[np.abs(i - j)]))


def main():
    quick = "--quick" in sys.argv
    out = []
    st = torch.cuda.current_stream().cuda_stream
    for n in (410, 2047):
        r, b = normal_equations(n, n)
        ref = np.linalg.solve(toeplitz(r), b)
        for B in ((1, 32, 256) if quick else (1, 8, 32, 64, 128, 256)):
            wh = b2.WienerHopf(-1, n - 1, 8192, max_batch=B)
            rb = torch.from_numpy(np.repeat(np.stack([r, b])[None], B, 0).copy()).cuda()
            w = torch.zeros((B, n), dtype=torch.complex64, device="cuda")
            ok = torch.zeros(B, dtype=torch.int32, device="cuda")
            for form, E in (("stepwise", 0), ("lookahead", 0), ("lookahead", 2), ("lookahead", 3), ("lookahead", 6), ("lookahead", 12)):
                if E and E != 12 and B > 64:
                    continue
                wh.set_solve_form(form, E)
                reps = max(20, int(0.4 / (2e-3 if n > 1000 else 2e-4)))  # ~0.4 s of back-to-back launches ramps the clock
                for _ in range(reps):
                    wh.solve_dev(rb.data_ptr(), B, w.data_ptr(), ok.data_ptr(), st)
                torch.cuda.synchronize()
                wh.set_timing(True)
                for _ in range(50):
                    wh.solve_dev(rb.data_ptr(), B, w.data_ptr(), ok.data_ptr(), st)
                torch.cuda.synchronize()
                t = wh.get_timing()["clutter_solve"]
                wh.set_timing(False)
                info = wh.solve_info()
                wv = w.cpu().numpy()
                err = max(np.linalg.norm(wv[c] - ref) / np.linalg.norm(ref) for c in (0, B - 1))
                rec = {"taps": n, "batch": B, "form": form, "E": info["E"], "G": info["G"], "us_per_launch": round(1e3 * t[0] / t[1], 2),
                       "us_per_cpi": round(1e3 * t[0] / t[1] / B, 3), "fault": info["fault"], "ok": bool(ok.cpu().numpy().all()),
                       "err": float(err)}
                out.append(rec)
                print(json.dumps(rec), flush=True)
            wh.close()
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
