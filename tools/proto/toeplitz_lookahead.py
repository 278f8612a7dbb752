#!/usr/bin/env python3
"""Prototype (NumPy, fp64) of a LOOK-AHEAD form of the Toeplitz solve of the clutter filter -- preparation for a
multi-CU kernel; not part of the product, nothing imports it.

The device kernel (csrc/clutter.hip: clutter_solve_kernel) runs the Levinson recursion with its inner products replaced
by Schur-type residual recursions: per order m ONE reflection coefficient ef_m and ONE gain dt_m are formed from the
leading elements, then every index is updated element-wise -- 2047 orders, one workgroup barrier each, 1.6 ms on one CU.
An order's update of the residual pair (A, C) and of the predictor pair (F, B) is the same 2 x 2 map with a shift,

    [U'(z)]   [      1        -ef    ] [U(z)]          U = A,  V[j] = C[j-1]      (residuals of the zero-extended predictors)
    [V'(z)] = [ -conj(ef) z     z    ] [V(z)]    or    U = F,  V = B = conj(rev F) (the predictors themselves)

so k orders are ONE 2 x 2 matrix of polynomials of degree <= k, Theta_k(z), and the two accumulators (the solution x and
the residual g of the right-hand side) move by polynomial combinations Psi of the same pairs.  The k coefficient pairs
(ef, dt) of a block depend only on the k leading elements of A, C and g: a k x k triangle of work for one wave, after
which applying Theta / Psi to the other n elements is embarrassingly parallel -- one exchange between CUs per BLOCK instead
of one barrier per order, at about twice the arithmetic (4(k+1) instead of 2k multiply-adds per element and block).

`solve_stepwise` is the recursion as the kernel runs it; `solve_lookahead` the block form.  Both return (w, ok) with
ok = False exactly when a prediction-error power is not positive (the matrix is not positive definite: the condition
under which the reference's chol() fails, WienerHopf.cpp:111).  Self-test: python tools/proto/toeplitz_lookahead.py
"""
import numpy as np


def solve_stepwise(r, b):
    """T w = b, T[i][j] = r[i-j] Hermitian Toeplitz (r[0] real), by Levinson with Schur residual recursions."""
    r = np.asarray(r, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    n = r.size
    if not (r[0].real > 0):
        return np.zeros(n, complex), False
    # unnormalised quantities: T F = s e_first (on the leading block), B = conj(rev F)
    F = np.zeros(n, complex); F[0] = 1.0
    B = np.zeros(n, complex); B[0] = 1.0
    s = r[0].real
    A = r.copy()                 # A[j] = (T F_ext)[j]; order 0: F = B = [1], (T e_0)[j] = r[j]
    C = r.copy()                 # C[j] = (T B_ext)[j]
    x = np.zeros(n, complex); x[0] = b[0] / s
    g = b - r * x[0]             # g[j] = b[j] - (T x_ext)[j]
    for m in range(n - 1):
        ef = A[m + 1] / s
        D = 1.0 - abs(ef) ** 2
        if not (D > 0.0) or not np.isfinite(D):
            return np.zeros(n, complex), False
        s_new = s * D
        # predictors: F' = F - ef z B, B' = z B - conj(ef) F  (z = one index up)
        Bsh = np.concatenate(([0.0], B[:-1]))
        Fn = F - ef * Bsh
        Bn = Bsh - np.conj(ef) * F
        # residuals of the zero-extended predictors: the same map; C gets the shift
        Csh = np.concatenate(([0.0], C[:-1]))
        An = A - ef * Csh
        Cn = Csh - np.conj(ef) * A
        F, B, A, C, s = Fn, Bn, An, Cn, s_new
        d = g[m + 1]
        dt = d / s
        x = x + dt * B
        g = g - dt * C
    return x, True


def _lookahead_coefficients(A, C, g, s, m, k):
    """The k pairs (ef, dt) of orders m .. m+k-1 from the k leading elements of A, C and g (indices m+1 .. m+k; C one
    index lower): the stepwise recursion on that window alone.  Returns (ef[], dt[], s after the block) or None if a
    prediction-error power is not positive."""
    a = A[m + 1:m + 1 + k].copy()
    c = C[m:m + k].copy()        # c[i] = C[j - 1] for j = m + 1 + i
    gg = g[m + 1:m + 1 + k].copy()
    efs, dts = [], []
    for i in range(k):
        ef = a[i] / s
        D = 1.0 - abs(ef) ** 2
        if not (D > 0.0) or not np.isfinite(D):
            return None
        s = s * D
        a_new = a - ef * c                 # A'[j] = A[j] - ef C[j-1]
        c_same = c - np.conj(ef) * a       # C'[j] = C[j-1] - conj(ef) A[j]
        dt = gg[i] / s
        gg = gg - dt * c_same              # g'[j] = g[j] - dt C'[j]
        a = a_new
        c = np.concatenate(([0.0], c_same[:-1]))  # the next order pairs index j with C'[j-1]
        efs.append(ef)
        dts.append(dt)
    return efs, dts, s


def _apply(poly, x):
    """(poly(z) x)(j) = sum_i poly[i] x[j - i], same length as x."""
    return np.convolve(x, poly)[:x.size]


def solve_lookahead(r, b, k=16):
    r = np.asarray(r, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    n = r.size
    if not (r[0].real > 0):
        return np.zeros(n, complex), False
    F = np.zeros(n, complex); F[0] = 1.0
    B = np.zeros(n, complex); B[0] = 1.0
    s = r[0].real
    A = r.copy()
    C = r.copy()
    x = np.zeros(n, complex); x[0] = b[0] / s
    g = b - r * x[0]
    m = 0
    while m < n - 1:
        kk = min(k, n - 1 - m)
        # V arrays of the two pairs in the shifted convention of the stepwise code: the map uses C[j-1] and B[j-1]
        la = _lookahead_coefficients(A, C, g, s, m, kk)
        if la is None:
            return np.zeros(n, complex), False
        efs, dts, s = la
        # one order: U' = U - ef z V, V' = z V - conj(ef) U   (z = shift up by one index)  => per order the matrix
        # [[1, -ef z], [-conj(ef), z]]; built here in that convention
        T = np.zeros((2, 2, kk + 1), complex)
        T[0, 0, 0] = 1.0; T[1, 1, 0] = 1.0
        P = np.zeros((2, kk + 1), complex)
        for ef, dt in zip(efs, dts):
            zT1 = np.zeros_like(T[1]); zT1[:, 1:] = T[1][:, :-1]
            row0 = T[0] - ef * zT1
            row1 = zT1 - np.conj(ef) * T[0]
            T = np.stack([row0, row1])
            P = P + dt * row1
        An = _apply(T[0, 0], A) + _apply(T[0, 1], C)
        Cn = _apply(T[1, 0], A) + _apply(T[1, 1], C)
        g = g - (_apply(P[0], A) + _apply(P[1], C))
        Fn = _apply(T[0, 0], F) + _apply(T[0, 1], B)
        Bn = _apply(T[1, 0], F) + _apply(T[1, 1], B)
        x = x + (_apply(P[0], F) + _apply(P[1], B))
        A, C, F, B = An, Cn, Fn, Bn
        m += kk
    return x, True


def solve_lookahead_unified(r, b, k=16):
    """The block form on the DEVICE's storage: three arrays of n entries whose meaning changes at the order boundary m --
    index j <= m carries (F[j], B[j], x[j]), index j > m carries (A[j], C[j-1], g[j]) -- so that every index does the work
    of ONE pair.  Per block [m, m + k): the look-ahead triangle reads the k entries above m; then, out of place,
        j <= m + k : (U, V, acc)[j] <- Theta, Psi applied to the LOWER entries (F, B at indices <= m, zero above)
        j >  m + k : (U, V, acc)[j] <- Theta, Psi applied to the UPPER entries (A, zC at indices > m: a halo of k below j)
    which is the split a multi-CU kernel distributes: a CU needs, beyond its own slice, the k entries below it (old
    values) and the block's Theta / Psi."""
    r = np.asarray(r, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    n = r.size
    if not (r[0].real > 0):
        return np.zeros(n, complex), False
    s = r[0].real
    U = r.copy(); V = np.concatenate(([0.0], r[:-1])); acc = b - r * (b[0] / s)   # upper roles: A, C[j-1], g
    U[0] = 1.0; V[0] = 1.0; acc[0] = b[0] / s                                      # lower roles at j = 0: F, B, x
    m = 0
    while m < n - 1:
        kk = min(k, n - 1 - m)
        # look-ahead on the kk entries above m (upper roles there: A[j], C[j-1], g[j])
        a = U[m + 1:m + 1 + kk].copy(); c = V[m + 1:m + 1 + kk].copy(); gg = acc[m + 1:m + 1 + kk].copy()
        efs, dts = [], []
        for i in range(kk):
            ef = a[i] / s
            D = 1.0 - abs(ef) ** 2
            if not (D > 0.0) or not np.isfinite(D):
                return np.zeros(n, complex), False
            s = s * D
            a_new = a - ef * c
            c_same = c - np.conj(ef) * a
            dt = gg[i] / s
            gg = gg - dt * c_same
            a = a_new
            c = np.concatenate(([0.0], c_same[:-1]))
            efs.append(ef); dts.append(dt)
        # lower pair (F, B): one order is [[1, -ef z], [-conj(ef), z]], x' = x + dt B'
        # upper pair in its shifted storage (U, V) = (A[j], C[j-1]): [[1, -ef], [-conj(ef) z, z]], g' = g - dt (V - conj(ef) U)
        # (the same transformation seen through V = z C: Theta_up = diag(1, z) Theta_low diag(1, 1/z))
        T = np.zeros((2, 2, kk + 1), complex); T[0, 0, 0] = 1.0; T[1, 1, 0] = 1.0
        Tu = T.copy()
        Pl = np.zeros((2, kk + 1), complex)
        Pu = np.zeros((2, kk + 1), complex)
        for ef, dt in zip(efs, dts):
            zT1 = np.zeros_like(T[1]); zT1[:, 1:] = T[1][:, :-1]
            T = np.stack([T[0] - ef * zT1, zT1 - np.conj(ef) * T[0]])
            Pl = Pl + dt * T[1]
            inner = Tu[1] - np.conj(ef) * Tu[0]          # C' of this order in terms of the block's input pair
            Pu = Pu + dt * inner
            zin = np.zeros_like(inner); zin[:, 1:] = inner[:, :-1]
            Tu = np.stack([Tu[0] - ef * Tu[1], zin])
        lo = slice(0, m + 1)                  # the lower entries that exist
        Fl = np.zeros(n, complex); Fl[lo] = U[lo]
        Bl = np.zeros(n, complex); Bl[lo] = V[lo]
        xl = np.zeros(n, complex); xl[lo] = acc[lo]
        Au = U.copy(); Au[lo] = 0.0           # the upper entries; what sits below m is never reached by a halo of kk from j > m + kk
        Cu = V.copy(); Cu[lo] = 0.0
        nU = np.empty(n, complex); nV = np.empty(n, complex); nacc = np.empty(n, complex)
        top = m + kk
        # lower formula for j <= top
        nU[:top + 1] = (_apply(T[0, 0], Fl) + _apply(T[0, 1], Bl))[:top + 1]
        nV[:top + 1] = (_apply(T[1, 0], Fl) + _apply(T[1, 1], Bl))[:top + 1]
        nacc[:top + 1] = (xl + _apply(Pl[0], Fl) + _apply(Pl[1], Bl))[:top + 1]
        # upper formula for j > top
        nU[top + 1:] = (_apply(Tu[0, 0], Au) + _apply(Tu[0, 1], Cu))[top + 1:]
        nV[top + 1:] = (_apply(Tu[1, 0], Au) + _apply(Tu[1, 1], Cu))[top + 1:]
        nacc[top + 1:] = (acc - (_apply(Pu[0], Au) + _apply(Pu[1], Cu)))[top + 1:]
        U, V, acc = nU, nV, nacc
        m = top
    return acc, True


def _toeplitz(r):
    n = r.size
    i, j = np.indices((n, n))
    return np.where(i >= j, r[np.abs(i - j)], np.conj(r[np.abs(i - j)]))


def _selftest():
    rng = np.random.default_rng(3)
    worst = 0.0
    for n, colour in ((33, 0.0), (200, 0.0), (411, 0.9), (700, 0.98), (512, 0.9995)):
        sig = rng.standard_normal(8 * n) + 1j * rng.standard_normal(8 * n)
        if colour:  # band-limited reference: ill-conditioned normal equations
            for i in range(1, sig.size):
                sig[i] += colour * sig[i - 1]
        full = np.correlate(sig, sig, mode="full")
        r = full[sig.size - 1:sig.size - 1 + n] / sig.size
        r[0] = r[0].real * (1 + 1e-9)
        b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        T = _toeplitz(r)
        ref = np.linalg.solve(T, b)
        w1, ok1 = solve_stepwise(r, b)
        assert ok1
        e1 = np.linalg.norm(T @ w1 - b) / np.linalg.norm(b)
        for k in (1, 4, 16, 32):
            w2, ok2 = solve_lookahead(r, b, k)
            assert ok2
            w3, ok3 = solve_lookahead_unified(r, b, k)
            assert ok3 and np.linalg.norm(w3 - w2) / np.linalg.norm(w2) < 1e3 * np.finfo(float).eps * max(1.0, np.linalg.cond(T)) , (n, k)
            e2 = np.linalg.norm(T @ w2 - b) / np.linalg.norm(b)
            dw = np.linalg.norm(w2 - w1) / np.linalg.norm(w1)
            worst = max(worst, dw)
            print(f"n={n:4d} colour={colour:4.2f} cond={np.linalg.cond(T):9.2e} k={k:2d}: residual stepwise {e1:.1e} look-ahead {e2:.1e} "
                  f"|w_la - w_step|/|w| {dw:.1e}  |w_step - w_lapack|/|w| {np.linalg.norm(w1 - ref) / np.linalg.norm(ref):.1e}")
    # not positive definite: both refuse
    r = np.array([1.0, 0.9, 1.2, 0.1], complex)
    assert not solve_stepwise(r, np.ones(4))[1] and not solve_lookahead(r, np.ones(4), 2)[1]
    print("worst stepwise / look-ahead difference", worst)


if __name__ == "__main__":
    _selftest()
