"""The clutter filter's Toeplitz solve (WienerHopf.cpp:85-122: toeplitz(r), chol, two triangular solves) on its own,
through blah2hip_clutter_solve: the look-ahead form on several workgroups per CPI (csrc/solve_la.hpp) at every slice
width, beside the one-workgroup stepwise kernel, against LAPACK in fp64.

Tolerance: the taps come back as complex fp32, so |w - w_ref| / |w_ref| <= 4e-7 * max(1, cond * 1e-9) -- fp32 rounding of
the output for well-conditioned systems, the fp64 recursion's cond * eps above that (measured ~1e-7 / cond * 3e-16)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def b2(built_lib):
    import blah2_amd
    assert blah2_amd.device_count() > 0
    return blah2_amd


def normal_equations(n, colour, seed):
    """r = autocorrelation of a (coloured) sequence at the scale the filter sees (sum over ~1e5 samples of |300|^2),
    b = a random right-hand side at that scale."""
    rng = np.random.default_rng(seed)
    m = max(8 * n, 4096)
    sig = 300.0 * (rng.standard_normal(m) + 1j * rng.standard_normal(m))
    for i in range(1, m):
        sig[i] += colour * sig[i - 1]
    f = np.fft.fft(sig, 2 * m)
    r = np.fft.ifft(np.abs(f) ** 2)[:n].copy()
    r[0] = r[0].real
    b = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * abs(r[0]) * 0.3
    return r, b


def toeplitz(r):
    n = r.size
    i, j = np.indices((n, n))
    return np.where(i >= j, r[np.abs(i - j)], np.conj(r[np.abs(i - j)]))


def handle(b2, n, max_batch=1):
    return b2.WienerHopf(-1, n - 1, 8192, max_batch=max_batch)


FORMS = [("stepwise", 0), ("lookahead", 0), ("lookahead", 2), ("lookahead", 3), ("lookahead", 6), ("lookahead", 12)]


@pytest.mark.parametrize("n,colour", [(1, 0.0), (2, 0.0), (3, 0.3), (33, 0.0), (34, 0.5), (65, 0.0), (97, 0.9), (161, 0.9),
                                      (410, 0.9), (411, 0.0), (737, 0.5), (1025, 0.9), (2047, 0.5), (3000, 0.3)])
@pytest.mark.parametrize("form,E", FORMS)
def test_solve_against_lapack(b2, n, colour, form, E):
    r, b = normal_equations(n, colour, 7 * n + 1)
    T = toeplitz(r)
    ref = np.linalg.solve(T, b)
    cond = np.linalg.cond(T) if n <= 1100 else 1e4
    wh = handle(b2, n)
    wh.set_solve_form(form, E)
    ok, w = wh.solve(r, b)
    info = wh.solve_info()
    assert info["fault"] == 0
    assert info["form"] == (1 if form == "stepwise" else 2)
    if E:
        assert info["E"] == E
    assert ok[0]
    err = np.linalg.norm(w[0] - ref) / np.linalg.norm(ref)
    assert err <= 4e-7 * max(1.0, cond * 1e-9), f"n={n} {form} E={E}: {err:.2e} (cond {cond:.1e})"


@pytest.mark.parametrize("form,E", FORMS)
def test_solve_batches_and_relaunches(b2, form, E):
    """Distinct systems per CPI, batch sizes either side of a multiple of 8 (the grid is padded to whole groups of 8
    CPIs), and the SAME handle launched repeatedly: the mailboxes are never cleared, only the launch epoch moves."""
    n = 300
    wh = handle(b2, n, max_batch=19)
    wh.set_solve_form(form, E)
    for rep, B in enumerate((1, 8, 19, 3, 9)):
        rs, bs = zip(*(normal_equations(n, 0.1 * (c % 5), 1000 * rep + c) for c in range(B)))
        ok, w = wh.solve(np.stack(rs), np.stack(bs))
        assert wh.solve_info()["fault"] == 0
        assert ok.all()
        for c in range(B):
            ref = np.linalg.solve(toeplitz(rs[c]), bs[c])
            assert np.linalg.norm(w[c] - ref) / np.linalg.norm(ref) <= 1e-6, (rep, B, c)


@pytest.mark.parametrize("form,E", FORMS)
@pytest.mark.parametrize("n,bad_at", [(4, 2), (100, 40), (100, 99), (700, 0), (700, 333)])
def test_not_positive_definite_is_refused(b2, form, E, n, bad_at):
    """The reference's chol() fails on a matrix that is not positive definite and the CPI is skipped (WienerHopf.cpp:111-115):
    ok = 0 and zero taps, wherever in the recursion the prediction-error power stops being positive -- also in a batch
    whose other systems are fine."""
    r, b = normal_equations(n, 0.5, n)
    if bad_at == 0:
        r[0] = -abs(r[0])
    else:
        r[bad_at] = 1.5 * abs(r[0])  # |r[k]| > r[0]: impossible for an autocorrelation
    assert np.linalg.eigvalsh(toeplitz(r)).min() < 0
    good_r, good_b = normal_equations(n, 0.2, n + 5)
    wh = handle(b2, n, max_batch=3)
    wh.set_solve_form(form, E)
    ok, w = wh.solve(np.stack([good_r, r, good_r]), np.stack([good_b, b, good_b]))
    assert wh.solve_info()["fault"] == 0
    assert ok.tolist() == [True, False, True]
    assert not w[1].any()
    ref = np.linalg.solve(toeplitz(good_r), good_b)
    for c in (0, 2):
        assert np.linalg.norm(w[c] - ref) / np.linalg.norm(ref) <= 1e-6


def test_the_planner_spreads_a_small_batch_and_packs_a_large_one(b2):
    n = 2047
    r, b = normal_equations(n, 0.3, 5)
    ref = np.linalg.solve(toeplitz(r), b)
    wh = handle(b2, n, max_batch=256)
    for B, want_g_at_least in ((1, 4), (32, 4), (256, 1)):
        ok, w = wh.solve(np.repeat(r[None], B, 0), np.repeat(b[None], B, 0))
        info = wh.solve_info()
        assert info["fault"] == 0 and info["form"] == 2
        assert info["G"] >= want_g_at_least and info["G"] * ((B + 7) // 8 * 8) <= 256
        assert ok.all()
        for c in (0, B - 1):
            assert np.linalg.norm(w[c] - ref) / np.linalg.norm(ref) <= 1e-6


# ---- the look-ahead form beside other work: bounded waits that run out cost time, never a result -------------------------

@pytest.mark.parametrize("E", [0, 2, 6])
@pytest.mark.parametrize("n", [410, 2047])
def test_a_wait_that_runs_out_is_solved_again_not_reported_as_not_positive_definite(b2, n, E):
    """A spin limit of ONE poll makes the look-ahead form's waits run out at once (what a chip busy with other kernels does
    to workgroups it dispatches late): the one-workgroup kernel gated behind the launch solves those CPIs again -- same
    taps, ok = 1 -- and a matrix that really is not positive definite still comes back ok = 0 (WienerHopf.cpp:111-115)."""
    B = 5
    rs, bs = map(list, zip(*(normal_equations(n, 0.1 * c, 50 * n + c) for c in range(B))))
    rs[3] = rs[3].copy()
    rs[3][n // 2] = 1.5 * abs(rs[3][0])
    wh = handle(b2, n, max_batch=B)
    wh.set_solve_form("lookahead", E)
    wh.set_solve_spin_limit(1)
    ok, w = wh.solve(np.stack(rs), np.stack(bs))
    info = wh.solve_info()
    assert info["form"] == 2
    if info["G"] > 1:  # several workgroups per CPI, the form that waits across CUs: waits DO run out at one poll
        assert info["fault"] == 1 and info["retries"] >= 1, info
    assert ok.tolist() == [True, True, True, False, True]
    assert not w[3].any()
    for c in (0, 1, 2, 4):
        ref = np.linalg.solve(toeplitz(rs[c]), bs[c])
        assert np.linalg.norm(w[c] - ref) / np.linalg.norm(ref) <= 1e-6, c
    # back at the default limit the same handle solves without a retry
    wh.set_solve_spin_limit(0)
    before = wh.solve_info()["retries"]
    ok, w2 = wh.solve(np.stack(rs), np.stack(bs))
    assert wh.solve_info()["retries"] == before
    assert ok.tolist() == [True, True, True, False, True]
    for c in (0, 1, 2, 4):
        ref = np.linalg.solve(toeplitz(rs[c]), bs[c])
        assert np.linalg.norm(w2[c] - ref) / np.linalg.norm(ref) <= 1e-6, c


def test_two_handles_on_two_streams_and_a_chip_filling_kernel_beside_them(b2):
    """Two filter handles enqueue their look-ahead solves (G > 1: workgroups that wait for each other) on two streams at
    the same time, round after round, while a third stream keeps the chip full with the batched range kernel of
    BASELINE configs[1] (persistent workgroups on every CU).  Every solve agrees with LAPACK, no wait ran out for good
    (retries are allowed -- they are the mechanism -- and counted)."""
    import torch
    n, B = 410, 2
    whs = [handle(b2, n, max_batch=B) for _ in range(2)]
    for wh in whs:
        wh.set_solve_form("lookahead", 0)
    sys_ = [[normal_equations(n, 0.2 * (c + 1), 9000 + 10 * h + c) for c in range(B)] for h in range(2)]
    d_rb = [torch.from_numpy(np.stack([np.stack([r, b]) for r, b in s])).cuda() for s in sys_]
    d_w = [torch.zeros((B, n), dtype=torch.complex64, device="cuda") for _ in range(2)]
    d_ok = [torch.full((B,), -7, dtype=torch.int32, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    # the chip-filling neighbour: 16 CPIs of configs[1] per launch on the one-wave range kernel + the tile Doppler kernel
    geom = (-10, 400, -256, 256, 2_000_000, 2_000_000)
    nb = 16
    amb = b2.Ambiguity(*geom, True, max_batch=nb)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    x = torch.view_as_complex(300 * torch.randn((nb, geom[5], 2), generator=g, device="cuda"))
    y = torch.view_as_complex(300 * torch.randn((nb, geom[5], 2), generator=g, device="cuda"))
    torch.cuda.synchronize()
    for rnd in range(20):
        for _ in range(2):
            amb.process_dev(b2.FMT_C32, x.data_ptr(), y.data_ptr(), nb, geom[5], None, None, streams[2].cuda_stream)
        for h in range(2):
            whs[h].solve_dev(d_rb[h].data_ptr(), B, d_w[h].data_ptr(), d_ok[h].data_ptr(), streams[h].cuda_stream)
    torch.cuda.synchronize()
    infos = [wh.solve_info() for wh in whs]
    print(f"\n[solve beside other work] {infos}")
    for h in range(2):
        assert infos[h]["form"] == 2 and infos[h]["G"] > 1
        assert d_ok[h].cpu().tolist() == [1] * B
        w = d_w[h].cpu().numpy()
        for c, (r, b) in enumerate(sys_[h]):
            ref = np.linalg.solve(toeplitz(r), b)
            assert np.linalg.norm(w[c] - ref) / np.linalg.norm(ref) <= 1e-6, (h, c)
