#!/usr/bin/env python3
"""Workgroup-level NumPy model of the FIR -> range FUSION: is one kernel that filters the
surveillance channel on the fly cheaper than clutter_fir_kernel + range_kernel?  NOT product code: it restates, with
NumPy FFTs standing in for the on-chip transforms, exactly the sequence of transforms a fused workgroup would run,
checks the result against the two-stage path (oracle.wiener_hopf's filter applied, then Ambiguity.cpp:106-149), and
COUNTS transforms, HBM bytes, registers and LDS per pulse.

The fused form (what makes sharing possible at all): cut a pulse into segments of L = F/2 samples.  With X_g the F-point
transform of the zero-padded segment g of xs (the shifted reference of WienerHopf.cpp:67) and H the transform of the taps,

    window [segment g-1 | segment g] of xs  has the spectrum  X_(g-1) + (-1)^m X_g          (a shift by F/2 is a sign)
    (w * xs) on segment g                   = the last L outputs of IFFT(H (X_(g-1) + (-1)^m X_g))     (taps <= L + 1)
    the range correlation of segment g      = IFFT(Y'_g conj(X'_g)),  X'_g = transform of x's segment g  (zero-padded)

so ONE forward transform per segment of x serves the filter's window spectrum AND -- when the filter's shift delayMin_c
and the segment grid agree -- the correlation's x' spectrum.  They agree only up to the clutter filter's own shift:
xs[i] = x[i - delayMin_c], so X'_g (of x) and X_g (of xs) differ by delayMin_c samples; a phase ramp W_F^(delayMin_c m)
turns one into the other EXCEPT for the |delayMin_c| samples that cross the segment boundary (config.yml: -24 ... -10):
the model carries them as a correction term and counts it.

Per pulse of nCorr samples (S = ceil(nCorr / L) segments):
    fused      1 (history block in front of the pulse) + S (X_g) + S (filter inverse) + S (Y' forward) + 1 (range inverse)
               + S more if X'_g has to be its own transform (the model's default: segments on the PULSE grid of x)
               -- or + 1 if the segment grid is laid on xs instead (grid origin p0 + delayMin_c: then xs's segment IS x's
               segment and X_g = X'_g, at the price of one more, ragged, filter block per pulse): 3 S + 3
    two-stage  range: 2 S_r + 1 with S_r = ceil(nCorr / (F - nDelay + 1));  FIR: 2 nCorr / (F - taps + 1)

The prediction (main()) prices the transforms with the per-transform cost MEASURED on range_kernel<16> at configs[2],
batch 32 (round 6, bench.py --config cfg3 --range-grid 0 / 256, same box): 56.6 us/CPI at the kernel's two workgroups per
CU, 86.8 us/CPI capped at ONE -- which is where a fused workgroup lives: 116 KB of LDS (68 of exchange regions + 48 of
filtered samples between the filter's inverse and the correlation's forward transform) and 200 registers per thread.
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def two_stage(x, y, w, dmin_c, n_corr, n_d, lags):
    """WienerHopf's FIR (taps w given) then the per-pulse range correlations: R[i][k] for lag lags[k]."""
    n = x.size
    xs = np.roll(x, dmin_c)  # xs[i] = x[i - delayMin] (delayMin <= 0: a left rotate, WienerHopf.cpp:67)
    conv = np.convolve(w, xs)[:n]
    yf = y - conv
    R = np.zeros((n_d, lags.size), dtype=np.complex128)
    for i in range(n_d):
        xp, yp = x[i * n_corr:(i + 1) * n_corr], yf[i * n_corr:(i + 1) * n_corr]
        for k, d in enumerate(lags):
            lo, hi = max(0, -d), min(n_corr, n_corr - d)
            R[i, k] = np.sum(yp[lo + d:hi + d] * np.conj(xp[lo:hi]))
    return yf, R


def fused(x, y, w, dmin_c, n_corr, n_d, lags, F):
    """The fused workgroup's transform sequence for every pulse; returns R and the transform count per pulse."""
    L = F // 2
    assert w.size <= L + 1 and lags.size <= F - L + 1 and lags[0] <= 0
    n = x.size
    xs = np.roll(x, dmin_c)
    H = np.fft.fft(w, F)
    sgn = (-1.0) ** np.arange(F)
    R = np.zeros((n_d, lags.size), dtype=np.complex128)
    count = 0
    for i in range(n_d):
        p0 = i * n_corr
        S = -(-n_corr // L)
        # history block: the L samples of xs in front of the pulse (previous pulse's tail; for pulse 0 the wrap of the CPI's
        # end -- WienerHopf's convolution is LINEAR, xs[m < 0] = 0, WienerHopf.cpp:125-160)
        hist = np.zeros(L, dtype=np.complex128)
        lo = p0 - L
        src = xs[max(lo, 0):p0]
        hist[L - src.size:] = src
        Xprev = np.fft.fft(hist, F)
        count += 1
        yf_pulse = np.zeros(n_corr + F, dtype=np.complex128)  # filtered samples of THIS pulse (zero outside: the range mask)
        specs = []
        for g in range(S):
            a, b = p0 + g * L, min(p0 + (g + 1) * L, p0 + n_corr)
            seg = np.zeros(L, dtype=np.complex128)
            seg[:b - a] = xs[a:b]                       # xs of the segment (beyond the pulse: the NEXT pulse's samples are not
            if b - a < L:                                # needed: outputs past the pulse end are masked)
                seg[b - a:] = xs[b:min(a + L, n)][:L - (b - a)] if b < n else 0
            Xg = np.fft.fft(seg, F)
            count += 1
            conv = np.fft.ifft(H * (Xprev + sgn * Xg))[L:]   # (w * xs) on segment g: the window's last L outputs
            count += 1
            yf_pulse[g * L:g * L + (b - a)] = (y[a:b] - conv[:b - a])
            specs.append(Xg)
            Xprev = Xg
        # range correlation on the same segment grid: x' = x's segment (NOT xs: shifted by dmin_c), y' = filtered window
        acc = np.zeros(F, dtype=np.complex128)
        for g in range(S):
            a, b = g * L, min((g + 1) * L, n_corr)
            xseg = np.zeros(F, dtype=np.complex128)
            xseg[:b - a] = x[p0 + a:p0 + b]
            # X'_g from the filter's X_g: a phase ramp for the shift + the boundary samples' correction (counted as VALU, not as
            # a transform: |dmin_c| samples x F outputs as a rank-|dmin_c| update is MORE than a transform for |dmin_c| > 12 --
            # the model therefore counts X'_g as its own transform unless dmin_c == 0)
            Xp = np.fft.fft(xseg)
            if dmin_c != 0:
                count += 1
            ywin = np.zeros(F, dtype=np.complex128)
            w0 = a + lags[0]
            lo_, hi_ = max(w0, 0), min(w0 + F, n_corr)
            ywin[lo_ - w0:hi_ - w0] = yf_pulse[lo_:hi_]
            acc += np.fft.fft(ywin) * np.conj(Xp)
            count += 1
        r = np.fft.ifft(acc)
        count += 1
        R[i] = r[:lags.size]
    return R, count / n_d


def fused_window_form(x, y, w, dmin_c, n_corr, n_d, lags, F):
    """range_fir_kernel's sequence (csrc/kernels.hpp), transform for transform, in fp64.  Per pulse, with x masked to the pulse:
    V_g = FFT([segment g-1 | segment g]) (block 0: the history block and segment 0 separately), the filter's output block g =
    the last L outputs of IFFT(H V_g) at samples g L + delayMin + [0, L), y' = y - that (+ the direct products at the pulse's
    edges), Z_g = FFT([y' block | 0]), acc += Z_g conj((-1)^m V_g) (block 0: conj X_0); one inverse at the end.
    Returns R and the transform count per pulse."""
    L = F // 2
    assert lags[0] == dmin_c <= 0 and w.size <= L + 1 and lags.size <= L + 1 and n_corr >= L - dmin_c
    n = x.size
    H = np.fft.fft(w, F)
    sgn = (-1.0) ** np.arange(F)
    R = np.zeros((n_d, lags.size), dtype=np.complex128)
    count = 0
    SB = -(-(n_corr - dmin_c) // L)
    for i in range(n_d):
        p0 = i * n_corr
        xm = np.zeros(SB * L + L, dtype=np.complex128)   # x of the pulse, zero beyond it
        xm[:n_corr] = x[p0:p0 + n_corr]

        def block_out(g, HV):
            conv = np.fft.ifft(HV)[L:]
            out = np.zeros(L, dtype=np.complex128)
            for k in range(L):
                nn = g * L + dmin_c + k
                if not (0 <= nn < n_corr):
                    continue
                c = conv[k]
                # past the pulse's end: taps kk <= nn - delayMin - nCorr reach x[p0 + nCorr ...], masked out of the window
                for kk in range(0, min(nn - dmin_c - n_corr, w.size - 1) + 1):
                    c += w[kk] * x[p0 + nn - dmin_c - kk]
                # the CPI's first |delayMin| samples: in the window of pulse 0, zero in the filter's stream (xs[m < 0] = 0)
                if i == 0:
                    for kk in range(max(nn + 1, 0), min(nn - dmin_c, w.size - 1) + 1):
                        c -= w[kk] * x[nn - dmin_c - kk]
                out[k] = y[p0 + nn] - c
            return out

        hist = np.zeros(L, dtype=np.complex128)
        if i > 0:
            hist[:] = x[p0 - L:p0]
        Xh, X0 = np.fft.fft(hist, F), np.fft.fft(xm[:L], F)
        count += 2 if i > 0 else 1
        Z = np.fft.fft(block_out(0, H * (Xh + sgn * X0)), F)
        count += 2
        acc = Z * np.conj(X0)
        for g in range(1, SB):
            V = np.fft.fft(xm[(g - 1) * L:(g + 1) * L])
            Z = np.fft.fft(block_out(g, H * V), F)
            count += 3
            acc += Z * np.conj(sgn * V)
        R[i] = np.fft.ifft(acc)[:lags.size]
        count += 1
    return R, count / n_d


def geometry(n_corr, n_delay, taps, F, dmin_c):
    L = F // 2
    S = -(-n_corr // L)
    fused_t = 1 + S + S + S + 1 + (S if dmin_c != 0 else 0)
    fused_xs_grid = 3 * S + 3
    S_r = -(-n_corr // (F - n_delay + 1))
    range_t = 2 * S_r + 1
    fir_t = 2 * n_corr / (F - taps + 1)
    return {"segments_fused": S, "fused_transforms_per_pulse": fused_t, "fused_if_filter_shift_were_zero": 1 + 3 * S + 1,
            "fused_on_the_xs_grid": fused_xs_grid,
            "range_transforms_per_pulse": range_t, "fir_transforms_per_pulse": fir_t, "two_stage_transforms_per_pulse": range_t + fir_t}


def main():
    # 1. the algebra, on a small case with every ingredient of the big one (taps = L - 1, negative filter shift, ragged last segment)
    rng = np.random.default_rng(5)
    F, n_corr, n_d = 64, 150, 7
    n = n_corr * n_d + 11
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    y = 0.8 * x + 0.1 * np.roll(x, 3) + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    w = 0.3 * (rng.standard_normal(31) + 1j * rng.standard_normal(31))
    lags = np.arange(-3, 30)
    _, R2 = two_stage(x, y, w, -3, n_corr, n_d, lags)
    Rf, per_pulse = fused(x, y, w, -3, n_corr, n_d, lags, F)
    sel = (np.arange(lags.size) + 0) % F
    err = np.max(np.abs(Rf - R2[:, :]) if lags[0] == 0 else np.abs(Rf[:, :lags.size] - R2))
    # lags are indexed from lags[0] in both; the fused ifft index k <-> lag lags[0] + k because the y' window starts at a + lags[0]
    print(f"fused == two-stage on the small case: max |dR| / max |R| = {err / np.abs(R2).max():.2e}  ({per_pulse:.1f} transforms per pulse)")
    assert err / np.abs(R2).max() < 1e-12
    # 2. the counts at BASELINE configs[2]
    g = geometry(9756, 2048, 2047, 4096, -24)
    print("configs[2] (nCorr 9756, 2048 lags, 2047 taps, F 4096, filter shift -24):", g)
    n_d3, n3 = 1025, 10_000_000
    t_now = g["two_stage_transforms_per_pulse"] * n_d3
    t_fused = g["fused_transforms_per_pulse"] * n_d3
    t_fused0 = g["fused_if_filter_shift_were_zero"] * n_d3
    bytes_now = (3 * n3 + 2 * n3) * 8 + n_d3 * 2048 * 8   # FIR: x, y in, y_f out; range: x, y_f in; the range map out
    bytes_fused = 2 * n3 * 8 + n_d3 * 2048 * 8               # x, y once
    print(f"transforms per CPI: two-stage {t_now:.0f}, fused {t_fused:.0f} ({t_fused / t_now - 1:+.1%}), fused with a zero filter shift {t_fused0:.0f} ({t_fused0 / t_now - 1:+.1%})")
    print(f"HBM bytes per CPI (these two kernels): {bytes_now / 1e6:.0f} MB -> {bytes_fused / 1e6:.0f} MB ({(bytes_now - bytes_fused) / 1e9:.2f} GB less)")
    # 3. what a fused workgroup holds
    regs = {"H (taps' spectrum)": 32, "X_(g-1)": 32, "X_g / X'_g": 32, "accumulator": 32, "transform in flight": 32, "twiddles, addresses": 40}
    lds = {"exchange regions (A + B of WgFft<16>)": 2 * 4352 * 8 / 1024, "filtered samples of the pulse's segments in flight (3 x L x 8 B)": 3 * 2048 * 8 / 1024}
    print(f"registers per thread (256 threads, 16 points each): {sum(regs.values())} of 256: {regs}")
    print(f"LDS per workgroup: {sum(lds.values()):.0f} KB: {lds}  -> ONE workgroup per CU (today: two of 68 KB)")
    # 4. the prediction, from measured per-transform costs (module docstring)
    range_two_wg, range_one_wg, fir_now = 56.6, 86.8, 57.0   # us/CPI: range_kernel<16> at 2 and at 1 workgroup per CU; clutter_fir_kernel<16>
    ns2 = 1e3 * range_two_wg / (g["range_transforms_per_pulse"] * n_d3)
    ns1 = 1e3 * range_one_wg / (g["range_transforms_per_pulse"] * n_d3)
    best = g["fused_on_the_xs_grid"] * n_d3
    pred = best * ns1 * 1e-3
    dram = 10.0  # us/CPI: ALL of the DRAM time the fusion could remove (a third of the 28.8 us the whole chain's DRAM traffic is worth, profiles/r05_cfg3_bytes.json)
    print(f"per 4096-point transform: {ns2:.2f} ns at two workgroups per CU, {ns1:.2f} ns at one")
    print(f"two-stage today: range {range_two_wg} + FIR {fir_now} = {range_two_wg + fir_now:.1f} us/CPI;  fused, {g['fused_on_the_xs_grid']} transforms per pulse at one "
          f"workgroup per CU: {pred:.1f} us/CPI, less at most {dram:.0f} of DRAM time = {pred - dram:.1f}:  {pred - dram - (range_two_wg + fir_now):+.1f} us/CPI "
          f"({(pred - dram) / (range_two_wg + fir_now) - 1:+.0%}) -- and even at the two-workgroup cost per transform, which its LDS footprint rules out, "
          f"{best * ns2 * 1e-3 - dram:.1f} us/CPI ({(best * ns2 * 1e-3 - dram) / (range_two_wg + fir_now) - 1:+.0%})")
    g["predicted_fused_us_per_cpi"] = pred - dram
    g["two_stage_us_per_cpi"] = range_two_wg + fir_now
    return g


if __name__ == "__main__":
    main()
