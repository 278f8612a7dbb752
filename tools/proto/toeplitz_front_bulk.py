#!/usr/bin/env python3
"""Wave-level emulation (NumPy, fp64) of the multi-workgroup Toeplitz solve of csrc/solve_la.hpp -- the index algebra
of the device kernel, run sequentially; not part of the product, nothing imports it but tests/test_toeplitz_prototype.py.

State per index j (three complex arrays), role decided by the order boundary m (indices <= m are "lower"):
    lower  j <= m : U = F[j]   V = B[j-1]   Z = x[j]          (predictors, the backward one stored one index up)
    upper  j >  m : U = A[j]   V = C[j-1]   Z = -g[j]         (their residuals T F, T B and the right-hand side's)
and ONE order is the same element-wise map on every index followed by a shift,
    W = V - conj(ef) U,   U' = U - ef V,   Z' = Z + dt W,   V'[j] = W[j-1],        ef = U[m+1] / s,  dt = -Z[m+1] / s'
    index m+1 (it turns lower):  U' = -ef,  Z' = dt
so new values at j after k orders depend on old values at [j-k, j] only.  That is what is distributed:

  * a FRONT wave keeps two 64-lane sets of the 32-index sub-windows W_b = [32b+1, 32b+32]: A = [W_b | W_b+1] and
    B = [W_b+1 | W_b+2].  Phase 1 of slot b runs the 32 orders of block b on A and reads (ef, dt) off the leading lane;
    phase 2 applies them to B.  Invalid values creep up one lane per order from lane 0, so the upper halves stay valid:
    A' = [A.hi | B.hi], B' = [B.hi | feed-in W_b+3 valid through block b].
  * BULK waves own fixed slices (S = 64E - 32 indices + a halo of 32 below), apply the published coefficients block by
    block, refresh the halo from the wave below after every block, and supply the front's feed-in.
"""
import numpy as np

K = 32  # orders per block = halo width = sub-window width


def _init_index(r, b, j):
    """(U, V, Z) of index j before order 0."""
    n = r.size
    if j < 0 or j >= n:
        return 0j, 0j, 0j
    x0 = b[0] / r[0].real
    if j == 0:
        return 1.0 + 0j, 0j, x0
    return r[j], r[j - 1], -(b[j] - r[j] * x0)


def _step(U, V, Z, ef, dt):
    """One order on a contiguous run of positions (position 0 shifts in zero)."""
    W = V - np.conj(ef) * U
    Un = U - ef * V
    Zn = Z + dt * W
    Vn = np.concatenate(([0j], W[:-1]))
    return Un, Vn, Zn


class Front:
    def __init__(self, r, b):
        self.r, self.b = r, b
        self.n = r.size
        self.inv_s = 1.0 / r[0].real
        self.A = self._load(1, 64)        # [W_0 | W_1]
        self.B = self._load(1 + K, 64)    # [W_1 | W_2]
        self.blk = 0

    def _load(self, j0, cnt):
        vals = [_init_index(self.r, self.b, j) for j in range(j0, j0 + cnt)]
        return [np.array([v[i] for v in vals]) for i in range(3)]

    def slot(self, feed):
        """Block self.blk: returns the 32 (ef, dt) pairs or None (not positive definite).  `feed` = (U, V, Z) of
        W_{blk+2} valid through block blk-1 ... supplied BEFORE phase 2 (it is B's upper half already, see below)."""
        n, m0 = self.n, K * self.blk
        U, V, Z = self.A
        efs, dts = [], []
        for i in range(K):
            m = m0 + i
            if m > n - 2:                      # orders beyond the last: identity coefficients
                efs.append(0j); dts.append(0j)
                U, V, Z = _step(U, V, Z, 0j, 0j)
                continue
            ef = U[i] * self.inv_s
            D = 1.0 - abs(ef) ** 2
            if not (D > 0.0) or not np.isfinite(D):
                return None
            self.inv_s = self.inv_s / D
            # dt needs Z of the leading index BEFORE this order's update (g after the previous order)
            dt = -Z[i] * self.inv_s
            U, V, Z = _step(U, V, Z, ef, dt)
            efs.append(ef); dts.append(dt)
        self.A = [U, V, Z]
        # phase 2: the block applied to B
        U, V, Z = self.B
        for ef, dt in zip(efs, dts):
            U, V, Z = _step(U, V, Z, ef, dt)
        self.B = [U, V, Z]
        self.blk += 1
        return efs, dts

    def advance(self, feed):
        """A' = [A.hi | B.hi], B' = [B.hi | feed]; feed = W_{blk+2} (blk already advanced) valid through block blk-1."""
        self.A = [np.concatenate((a[K:], bb[K:])) for a, bb in zip(self.A, self.B)]
        self.B = [np.concatenate((bb[K:], f)) for bb, f in zip(self.B, feed)]


class Bulk:
    def __init__(self, r, b, q, E):
        self.q, self.E = q, E
        self.S = 64 * E - K
        self.n = r.size
        self.j0 = q * self.S + 1 - K            # index of position 0
        vals = [_init_index(r, b, self.j0 + p) for p in range(64 * E)]
        self.st = [np.array([v[i] for v in vals]) for i in range(3)]

    def block(self, blk, efs, dts):
        U, V, Z = self.st
        for i, (ef, dt) in enumerate(zip(efs, dts)):
            m = K * blk + i
            U, V, Z = _step(U, V, Z, ef, dt)
            p = m + 1 - self.j0                  # the index that turns lower at this order
            if 0 <= p < U.size and m <= self.n - 2:
                U[p] = -ef
                Z[p] = dt
        self.st = [U, V, Z]

    def top(self):
        return [a[-K:].copy() for a in self.st]

    def refresh_halo(self, halo):
        if self.q == 0:
            return                               # positions below index 0 are exact zeros: nothing creeps in
        for a, h in zip(self.st, halo):
            a[:K] = h

    def window(self, wb):
        """(U, V, Z) of sub-window W_wb if this wave OWNS it, else None."""
        p = K * wb + 1 - self.j0
        if p < K or p + K > 64 * self.E:
            return None
        return [a[p:p + K].copy() for a in self.st]


class FrontPair:
    """The front as TWO waves (round 4): the chain wave keeps only the 32-index triangle W_b; its companion wave
    supplies every next triangle.  Per block b the companion (i) applies c_{b-1} to D = [W_b | W_b+1] valid through b-2
    (the old phase 2, one block later: its feed-in from the bulk has a whole block of slack more), then (ii) follows the
    chain wave's orders of block b on C = [W_b | W_b+1] valid through b-1 and hands C.hi = W_b+1 valid through b over."""

    def __init__(self, r, b):
        self.r, self.b, self.n = r, b, r.size
        self.inv_s = 1.0 / r[0].real
        self.T = self._load(1, K)                      # W_0
        self.C = self._load(1, 2 * K)                  # [W_0 | W_1]
        self.Dhi_prev = self._load(1 + K, K)           # W_1, "valid through -2": initial
        self.blk = 0

    def _load(self, j0, cnt):
        vals = [_init_index(self.r, self.b, j) for j in range(j0, j0 + cnt)]
        return [np.array([v[i] for v in vals]) for i in range(3)]

    def block(self, feed):
        """feed: W_{blk+1} valid through block blk-2 from the bulk (blk >= 2), else None (initial values)."""
        b, n = self.blk, self.n
        # companion, step (i)
        if b >= 1:
            hi = feed if feed is not None else (self._load(1 + K * (b + 1), K) if b < 2 else [np.zeros(K, complex)] * 3)
            D = [np.concatenate((lo, h)) for lo, h in zip(self.Dhi_prev, hi)]
            for ef, dt in zip(*self.prev_coef):
                D = list(_step(*D, ef, dt))
            self.Dhi_prev = [a[K:].copy() for a in D]
            self.C = [np.concatenate((c_lo, d_hi)) for c_lo, d_hi in zip(self.Chi_prev, self.Dhi_prev)]
        # chain wave: the triangle alone
        U, V, Z = self.T
        efs, dts = [], []
        for i in range(K):
            m = K * b + i
            if m > n - 2:
                efs.append(0j); dts.append(0j)
                U, V, Z = _step(U, V, Z, 0j, 0j)
                continue
            ef = U[i] * self.inv_s
            D_ = 1.0 - abs(ef) ** 2
            if not (D_ > 0.0) or not np.isfinite(D_):
                return None
            self.inv_s = self.inv_s / D_
            dt = -Z[i] * self.inv_s
            U, V, Z = _step(U, V, Z, ef, dt)
            efs.append(ef); dts.append(dt)
        # companion, step (ii): the same orders on C
        C = self.C
        for ef, dt in zip(efs, dts):
            C = list(_step(*C, ef, dt))
        self.Chi_prev = [a[K:].copy() for a in C]
        self.T = [a.copy() for a in self.Chi_prev]     # the hand-over
        self.prev_coef = (efs, dts)
        self.blk += 1
        return efs, dts


def solve_front_pair_bulk(r, b, E=3):
    """solve_front_bulk with the two-wave front: same bulk, same feed-in schedule (W_{k+3} after the bulk's block k)."""
    r = np.asarray(r, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    n = r.size
    if not (r[0].real > 0) or not np.isfinite(r[0].real):
        return np.zeros(n, complex), False
    S = 64 * E - K
    nbulk = max(1, -(-(n - 1) // S))
    NB = max(1, -(-(n - 1) // K))
    front = FrontPair(r, b)
    bulk = [Bulk(r, b, q, E) for q in range(nbulk)]
    feeds = {}
    for blk in range(NB):
        c = front.block(feeds.get(blk + 1))
        if c is None:
            return np.zeros(n, complex), False
        efs, dts = c
        for w in bulk:
            w.block(blk, efs, dts)
        tops = [w.top() for w in bulk]
        for q in range(1, nbulk):
            for a in bulk[q].st:
                a[:K] = np.nan
            bulk[q].refresh_halo(tops[q - 1])
        for w in bulk:                                   # W_{blk+3} valid through blk
            f = w.window(blk + 3)
            if f is not None:
                feeds[blk + 3] = f
    x = np.zeros(n, complex)
    for w in bulk:
        for p in range(K - 1 if w.q == 0 else K, 64 * E):
            j = w.j0 + p
            if 0 <= j < n:
                x[j] = w.st[2][p]
    return x, True


def solve_front_bulk(r, b, E=3, check_halo_garbage=True):
    r = np.asarray(r, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    n = r.size
    if not (r[0].real > 0) or not np.isfinite(r[0].real):
        return np.zeros(n, complex), False
    S = 64 * E - K
    nbulk = max(1, -(-(n - 1) // S))
    NB = max(1, -(-(n - 1) // K))
    front = Front(r, b)
    bulk = [Bulk(r, b, q, E) for q in range(nbulk)]
    for blk in range(NB):
        c = front.slot(None)
        if c is None:
            return np.zeros(n, complex), False
        efs, dts = c
        for w in bulk:
            w.block(blk, efs, dts)
        tops = [w.top() for w in bulk]
        for q in range(1, nbulk):
            if check_halo_garbage:               # poison the halo first: the refresh must restore every position used
                for a in bulk[q].st:
                    a[:K] = np.nan
            bulk[q].refresh_halo(tops[q - 1])
        # feed-in for the NEXT slot's phase 2: W_{blk+3} valid through block blk
        feed = None
        for w in bulk:
            f = w.window(blk + 3)
            if f is not None:
                feed = f
        if feed is None:                         # beyond the last owned sub-window: never consumed by a live order
            feed = [np.zeros(K, complex)] * 3
        front.advance(feed)
    x = np.zeros(n, complex)
    for w in bulk:
        for p in range(K - 1 if w.q == 0 else K, 64 * E):
            j = w.j0 + p
            if 0 <= j < n:
                x[j] = w.st[2][p]
    return x, True


def _selftest():
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("la", os.path.join(os.path.dirname(__file__), "toeplitz_lookahead.py"))
    la = importlib.util.module_from_spec(spec); spec.loader.exec_module(la)
    rng = np.random.default_rng(5)
    for n, colour in ((2, 0.0), (3, 0.0), (33, 0.0), (34, 0.5), (97, 0.0), (161, 0.9), (410, 0.9), (700, 0.98), (1025, 0.5)):
        sig = rng.standard_normal(8 * n) + 1j * rng.standard_normal(8 * n)
        for i in range(1, sig.size):
            sig[i] += colour * sig[i - 1]
        full = np.correlate(sig, sig, mode="full")
        r = full[sig.size - 1:sig.size - 1 + n].copy()
        r[0] = r[0].real
        bb = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * abs(r[0])
        w1, ok1 = la.solve_stepwise(r, bb)
        assert ok1
        for E in (1, 2, 3, 6, 12):
            w2, ok2 = solve_front_bulk(r, bb, E)
            assert ok2
            w3, ok3 = solve_front_pair_bulk(r, bb, E)
            assert ok3 and np.linalg.norm(w3 - w2) <= 1e-9 * np.linalg.norm(w2), (n, E)
            d = np.linalg.norm(w2 - w1) / np.linalg.norm(w1)
            T = la._toeplitz(r)
            res = np.linalg.norm(T @ w2 - bb) / np.linalg.norm(bb)
            print(f"n={n:5d} colour={colour:4.2f} E={E:2d}: |w - w_step|/|w| = {d:.1e}  residual {res:.1e}")
            assert d < 1e-9 * max(1.0, np.linalg.cond(T) * 1e-6)
    r = np.array([1.0, 0.9, 1.2, 0.1], complex)
    assert not solve_front_bulk(r, np.ones(4))[1]
    print("ok")


if __name__ == "__main__":
    _selftest()
