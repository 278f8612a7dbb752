#!/usr/bin/env python3
"""Wave-level NumPy model of cfar2d_stream_kernel (blah2_amd/csrc/cfar_kernels.hpp): the same decomposition -- strips of
64 - 2 hC output columns with hC halo lanes either side, segments of R Doppler rows walked top to bottom, the neighbours'
values as PAIR SUMS at fixed lane offsets, the column sums through rings of U slots with compile-time delays, the test
`sq n > alpha tot` with n from the clipped window -- in fp64 array arithmetic, one "wave" = one array of 64 lanes.
What it pins down is the index algebra (offsets of the pair sums for odd and even counts, ring delays, which row a
streamed row completes, clipped counts at all four map edges, the lanes and rows that test nothing); the arithmetic is
NumPy's.  In the CPU test suite against oracle.cfar2d (tests/test_cfar_stream_model.py)."""
import math

import numpy as np


def ring_len(ntf, ngf, v=2):
    return (v + ntf + 2 * ngf + 1 + v - 1) // v * v


def stream_strip(sq, j0, i0, i1, ntd, ngd, ntf, ngf, alpha, tested_col, tested_row, v=2):
    """One wave: output columns j0 + hC .. j0 + 63 - hC (lane l holds column j0 + l), output rows i0 .. i1 - 1 of the
    |z|^2 map `sq` [nD][nC].  Returns the (row, column) pairs it reports."""
    nD, nC = sq.shape
    hC, hR = ntd + ngd, ntf + ngf
    GH = 2 * ngf + 1
    U = ring_len(ntf, ngf, v)
    lanes = np.arange(64)
    cols = j0 + lanes
    inmap = (cols >= 0) & (cols < nC)
    PAD = 16
    assert hC + 1 <= PAD

    def at(arr, k):  # value of lane l + k (a pad of 16 either side, like the LDS piece; pads hold garbage = NaN here)
        ext = np.full(64 + 2 * PAD, np.nan)
        ext[PAD:PAD + 64] = arr
        return ext[PAD + k:PAD + k + 64]

    clampc = lambda x: np.clip(x, 0, nC)
    ncols = lambda a, b: np.maximum(b, 1) - np.maximum(a, 1)  # column 0 never trains
    lane_ok = (lanes >= hC) & (lanes < 64 - hC) & inmap & tested_col[np.clip(cols, 0, nC - 1)]
    n_all = np.where(lane_ok, ncols(clampc(cols - hC), clampc(cols + hC + 1)), 0)
    n_guard = np.where(lane_ok, ncols(clampc(cols - ngd), clampc(cols + ngd + 1)), 0)

    Bh, Ah, TBh, TAh, sqh = (np.zeros((U, 64)) for _ in range(5))
    hits = []
    r_start, r_last = i0 - hR, i1 - 1 + hR
    rb = r_start
    while rb <= r_last:
        for u in range(U):
            r = rb + u
            row_ok = 0 <= r < nD and r <= r_last
            s = np.where(inmap, sq[r, np.clip(cols, 0, nC - 1)], 0.0) if row_ok else np.zeros(64)
            s1 = np.where(cols == 0, 0.0, s)
            s2 = s1 + at(s1, 1)  # pair sums
            # the 2 nGd + 1 guard columns: one cell and nGd pairs; nTd columns either side: pairs, and one cell if nTd is odd
            G = at(s1, -ngd)
            for m in range(ngd):
                G = G + at(s2, -ngd + 1 + 2 * m)
            A = np.zeros(64)
            if ntd > 0:
                L = R = None
                for m in range(ntd // 2):
                    l_, r_ = at(s2, -hC + 2 * m), at(s2, ngd + 1 + 2 * m)
                    L = l_ if L is None else L + l_
                    R = r_ if R is None else R + r_
                if ntd & 1:
                    l_, r_ = at(s1, -ngd - 1), at(s1, hC)
                    L = l_ if L is None else L + l_
                    R = r_ if R is None else R + r_
                A = L + R
            Bh[u], Ah[u], sqh[u] = A + G, A, s
            TB = np.zeros(64)
            if ntf > 0:
                TB = Bh[u].copy()
                for k in range(1, ntf):
                    TB = TB + Bh[(u - k) % U]
            TA = Ah[u].copy()
            for k in range(1, GH):
                TA = TA + Ah[(u - k) % U]
            TBh[u], TAh[u] = TB, TA
            tot = TAh[(u - ntf) % U]
            if ntf > 0:
                tot = (TB + tot) + TBh[(u - ntf - GH) % U]
            cut = sqh[(u - hR) % U]
            i = r - hR
            if i0 <= i < i1 and tested_row[i]:
                rA = min(i + hR + 1, nD) - max(i - hR, 0)
                rG = min(i + ngf + 1, nD) - max(i - ngf, 0)
                nn = rA * n_all - rG * n_guard
                with np.errstate(invalid="ignore"):
                    hit = cut * nn > alpha[nn] * tot  # alpha[0] = NaN: never
                for l in np.nonzero(hit)[0]:
                    hits.append((i, int(cols[l])))
        rb += U
    return hits


def cfar2d_stream(m, delay_axis, doppler_axis, pfa, ng_d, nt_d, ng_f, nt_f, min_delay, min_doppler, rows_per_seg=32, v=2):
    """The whole map the way the kernel's tasks cover it; returns the sorted list of (row, column) detections."""
    m = np.asarray(m, dtype=np.complex128)
    nD, nC = m.shape
    sq = m.real * m.real + m.imag * m.imag
    hC = ng_d + nt_d
    outw = 64 - 2 * hC
    assert outw >= 16
    nmax = (2 * hC + 1) * (2 * (ng_f + nt_f) + 1)
    alpha = np.full(nmax + 1, np.nan)
    for n in range(1, nmax + 1):
        alpha[n] = n * (math.pow(pfa, -1.0 / n) - 1)
    tested_col = np.asarray(delay_axis) >= min_delay
    tested_row = ~(np.abs(np.asarray(doppler_axis)) < min_doppler)
    hits = []
    for i0 in range(0, nD, rows_per_seg):
        for strip in range(-(-nC // outw)):
            hits += stream_strip(sq, strip * outw - hC, i0, min(i0 + rows_per_seg, nD), nt_d, ng_d, nt_f, ng_f, alpha,
                                 tested_col, tested_row, v)
    return sorted(hits)
