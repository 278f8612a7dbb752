"""The look-ahead form of the clutter filter's Toeplitz solve (tools/proto/toeplitz_lookahead.py: preparation for a
multi-CU kernel) against the stepwise recursion the device runs, LAPACK, and the normal equations
of a compiled-reference fixture."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import load_golden
from oracle import blah2_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("toeplitz_lookahead", os.path.join(ROOT, "tools", "proto", "toeplitz_lookahead.py"))
P = importlib.util.module_from_spec(spec)
spec.loader.exec_module(P)


def coloured_normal_equations(n, colour, seed):
    rng = np.random.default_rng(seed)
    sig = rng.standard_normal(8 * n) + 1j * rng.standard_normal(8 * n)
    for i in range(1, sig.size):
        sig[i] += colour * sig[i - 1]
    r = np.correlate(sig, sig, mode="full")[sig.size - 1:sig.size - 1 + n] / sig.size
    r[0] = r[0].real
    return r, rng.standard_normal(n) + 1j * rng.standard_normal(n)


@pytest.mark.parametrize("n,colour", [(97, 0.0), (300, 0.9), (411, 0.99)])
@pytest.mark.parametrize("k", [1, 8, 32])
def test_lookahead_equals_stepwise(n, colour, k):
    r, b = coloured_normal_equations(n, colour, n)
    T = P._toeplitz(r)
    w1, ok1 = P.solve_stepwise(r, b)
    w2, ok2 = P.solve_lookahead(r, b, k)
    assert ok1 and ok2
    ref = np.linalg.solve(T, b)
    scale = np.linalg.cond(T) * 1e-15
    assert np.linalg.norm(w2 - w1) / np.linalg.norm(w1) <= 50 * scale
    assert np.linalg.norm(w2 - ref) / np.linalg.norm(ref) <= 50 * scale
    assert np.linalg.norm(T @ w2 - b) <= 2 * max(np.linalg.norm(T @ w1 - b), 1e-13 * np.linalg.norm(b))
    # the same on the device's storage: three arrays whose meaning changes at the order boundary, every index doing the
    # work of one pair, the upper pair through its shifted transformation
    w3, ok3 = P.solve_lookahead_unified(r, b, k)
    assert ok3 and np.linalg.norm(w3 - w2) / np.linalg.norm(w2) <= 50 * scale


def test_not_positive_definite_is_refused_by_both():
    r = np.array([1.0, 0.9, 1.2, 0.1, 0.0, 0.3], complex)
    for k in (1, 2, 5):
        assert not P.solve_lookahead(r, np.ones(6), k)[1]
        assert not P.solve_lookahead_unified(r, np.ones(6), k)[1]
    assert not P.solve_stepwise(r, np.ones(6))[1]
    assert not P.solve_lookahead(np.array([0.0, 0.1], complex), np.ones(2))[1]


def test_on_the_normal_equations_of_a_reference_fixture():
    g = load_golden("medium")
    dmin, dmax = (int(v) for v in g["clutter_params"])
    ok, _, w_ref, r_ref, b_ref = O.wiener_hopf(g["x"], g["y"], dmin, dmax, return_filter=True)
    assert ok and r_ref.size == dmax - dmin
    w, ok2 = P.solve_lookahead(r_ref, b_ref, 16)
    assert ok2
    assert np.linalg.norm(w - w_ref) / np.linalg.norm(w_ref) <= 1e-9


# ---- the wave-level model of the device's look-ahead kernel (csrc/solve_la.hpp): front wave + bulk slices ---------------
spec2 = importlib.util.spec_from_file_location("toeplitz_front_bulk", os.path.join(ROOT, "tools", "proto", "toeplitz_front_bulk.py"))
FB = importlib.util.module_from_spec(spec2)
spec2.loader.exec_module(FB)


@pytest.mark.parametrize("n,colour", [(1, 0.0), (2, 0.0), (33, 0.0), (34, 0.5), (97, 0.9), (300, 0.9), (411, 0.5)])
@pytest.mark.parametrize("E", [1, 2, 3, 6, 12])
def test_front_bulk_model_equals_stepwise(n, colour, E):
    """Slices of 64 E - 32 indices with a halo of 32, blocks of 32 orders, the front's two 64-lane sets and its feed-in
    schedule: the index algebra of the kernel, with the halo poisoned before every refresh."""
    r, b = coloured_normal_equations(n, colour, 3 * n + E)
    w1, ok1 = P.solve_stepwise(r, b)
    w2, ok2 = FB.solve_front_bulk(r, b, E)
    assert ok1 and ok2
    scale = np.linalg.cond(P._toeplitz(r)) * 1e-15 if n > 1 else 1e-15
    assert np.linalg.norm(w2 - w1) <= 50 * scale * np.linalg.norm(w1)


def test_front_bulk_model_refuses_what_is_not_positive_definite():
    r = np.array([1.0, 0.9, 1.2, 0.1, 0.0, 0.3], complex)
    assert not FB.solve_front_bulk(r, np.ones(6), 2)[1]
    assert not FB.solve_front_bulk(np.array([-1.0, 0.1], complex), np.ones(2), 3)[1]


@pytest.mark.parametrize("n,colour", [(2, 0.0), (34, 0.5), (97, 0.9), (300, 0.9), (411, 0.5)])
@pytest.mark.parametrize("E", [2, 3, 12])
def test_two_wave_front_model_equals_the_one_wave_front(n, colour, E):
    """The front as a chain wave (triangle only) and a companion wave (catch-up one block later + following): the
    hand-over and feed-in schedule of csrc/solve_la.hpp's front_chain / front_companion."""
    r, b = coloured_normal_equations(n, colour, 5 * n + E)
    w1, ok1 = FB.solve_front_bulk(r, b, E)
    w2, ok2 = FB.solve_front_pair_bulk(r, b, E)
    assert ok1 and ok2
    assert np.linalg.norm(w2 - w1) <= 1e-9 * np.linalg.norm(w1)
