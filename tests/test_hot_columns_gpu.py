"""GPU: the fp64 transform of the delay columns under the map's tallest peaks (BLAH2HIP_OPT_HOT_COLUMNS, csrc/capi.hip).

The fp32 Doppler transform leaves up to 1.2e-7 of a column's peak in the other rows of that column; under a peak 1000x the
map's mean level -- a strong echo behind the clutter filter -- that is beyond north_star's 1e-4 on a mean-level cell
(tools/gpu_chain_split_diag.py).  The engine transforms such columns again in fp64.  Checked against the oracle
(Ambiguity.cpp:152-169 in fp64): the column's error with and without it, that no other cell moves, which columns are
picked, every Doppler kernel family, batches whose CPIs differ.
"""
import numpy as np
import pytest

from gates import map_cell_gate
from oracle import blah2_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def b2(built_lib):
    import blah2_amd
    assert blah2_amd.device_count() > 0
    return blah2_amd


def echo_cpi(n, fs, seed, delay, doppler, amp=1.0, noise=0.02, direct=0.0):
    """x white; y = direct x + amp x(t - delay) e^{2 pi i doppler t} + noise: one echo far above the floor."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 200.0
    t = np.arange(n) / fs
    xd = np.roll(x, delay)
    xd[:delay] = 0
    y = direct * x + amp * xd * np.exp(2j * np.pi * doppler * t) + noise * 200.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64), y.astype(np.complex64)


def col_err(got, ref, lvl, col, skip_row):
    """Largest error of the column's cells, each relative to itself or to the mean level if it lies below (an echo between two
    Doppler rows fills its column with sidelobes far above the mean level: those cells answer for their own size)."""
    e = np.abs(got[:, col].astype(np.complex128) - ref[:, col]) / np.maximum(np.abs(ref[:, col]), lvl)
    e[skip_row] = 0.0
    return float(e.max())


GEOMETRIES = [
    # (delayMin, delayMax, dopplerMin, dopplerMax, fs, n), echo (delay, doppler)
    ((-10, 400, -256, 256, 2_000_000, 2_000_000), (37, -63.0)),   # nD 513: the headline's geometry
    ((-10, 300, -512, 512, 2_000_000, 1_000_000), (120, 200.0)),  # nD 513 at 0.5 s (2 Hz rows)
    ((-8, 200, -512, 512, 1_000_000, 1_000_000), (50, 101.0)),    # nD 1025
    ((-8, 120, -1024, 1024, 1_000_000, 1_000_000), (9, -700.0)),  # nD 2049
    ((-4, 60, -100, 100, 500_000, 250_000), (20, 33.0)),          # nD 101: short pulses
]


@pytest.mark.parametrize("args,echo", GEOMETRIES)
def test_the_echo_column_is_transformed_in_fp64(b2, args, echo):
    n, fs = args[5], args[4]
    x, y = echo_cpi(n, fs, 11, *echo)
    d = O.ambiguity_dims(*args, True)
    ref = O.ambiguity_process(d, x.astype(np.complex128), y.astype(np.complex128))
    lvl = 10.0 ** (O.map_metrics(ref)[0] / 10.0)
    col = int(np.argmin(np.abs(d.delay - echo[0])))
    row = int(np.argmin(np.abs(d.doppler - echo[1])))
    assert np.abs(ref[row, col]) > 250.0 * lvl                      # the echo stands where the rule looks
    maps, hot = {}, {}
    for mode in ("off", "auto"):
        amb = b2.Ambiguity(*args, True)
        amb.set_hot_columns(mode)
        maps[mode] = amb.process(x, y).data.copy()
        hot[mode] = amb.hot_columns()
        kern = amb.last_doppler_kernel()
        amb.close()
    assert hot["off"] == 0 and hot["auto"] == 1, hot
    e_off, e_on = col_err(maps["off"], ref, lvl, col, row), col_err(maps["auto"], ref, lvl, col, row)
    other = np.ones(ref.shape, dtype=bool)
    other[:, col] = False
    floor = float((np.abs(maps["off"].astype(np.complex128) - ref) / np.maximum(np.abs(ref), lvl))[other].max())
    print(f"\n[hot] nD {ref.shape[0]} Doppler kernel {kern}: echo {np.abs(ref[row, col]) / lvl:.0f}x the mean level; its column's "
          f"largest error / mean level {e_off:.2e} -> {e_on:.2e} (other columns {floor:.2e})")
    assert np.array_equal(maps["off"][other], maps["auto"][other])  # no other cell moves
    assert e_on <= 2.0 * floor + 2e-6 and e_on < e_off
    g = map_cell_gate(maps["auto"], ref)
    assert g["ok"] and g["cell_rel_above_mean"] <= 3e-5, g
    # the peak cell itself: fp64 of the fp32 range map
    assert abs(maps["auto"][row, col] - ref[row, col]) <= 2e-6 * abs(ref[row, col])


def test_noise_has_no_hot_column_and_the_map_keeps_its_bits(b2):
    args = (-10, 400, -256, 256, 2_000_000, 2_000_000)
    rng = np.random.default_rng(3)
    x = ((rng.standard_normal(args[5]) + 1j * rng.standard_normal(args[5])) * 100).astype(np.complex64)
    y = ((rng.standard_normal(args[5]) + 1j * rng.standard_normal(args[5])) * 100).astype(np.complex64)
    out = {}
    for mode in ("off", "auto", "always"):
        amb = b2.Ambiguity(*args, True)
        amb.set_hot_columns(mode)
        out[mode] = amb.process(x, y).data.copy()
        assert amb.hot_columns() == 0
        amb.close()
    assert np.array_equal(out["off"], out["auto"]) and np.array_equal(out["off"], out["always"])


def test_the_direct_path_column_is_left_to_the_doppler_kernel(b2):
    """y = 0.8 x + noise: the lag-0 column holds a peak 1000x the mean level AT ZERO DOPPLER, which the Doppler kernels take out
    exactly before they transform (the first pulse's value) -- nothing to transform again."""
    args = (-10, 400, -256, 256, 2_000_000, 2_000_000)
    x, y = echo_cpi(args[5], args[4], 9, 37, -63.0, amp=0.02, noise=0.1, direct=0.8)
    d = O.ambiguity_dims(*args, True)
    ref = O.ambiguity_process(d, x.astype(np.complex128), y.astype(np.complex128))
    lvl = 10.0 ** (O.map_metrics(ref)[0] / 10.0)
    c0 = int(np.argmin(np.abs(d.delay)))
    assert np.abs(ref[:, c0]).max() > 800.0 * lvl
    amb = b2.Ambiguity(*args, True)
    m = amb.process(x, y).data.copy()
    assert amb.hot_columns() == 0
    amb.close()
    assert map_cell_gate(m, ref)["ok"]


def test_short_cpis_are_left_alone_in_auto_mode(b2):
    """Under 35 000 samples no peak can stand 250x above the mean level: auto mode does not launch the kernel; "always" does."""
    args = (-4, 40, -50, 50, 100_000, 20_000)
    x, y = echo_cpi(args[5], args[4], 5, 10, 7.0, noise=1e-3)
    d = O.ambiguity_dims(*args, True)
    ref = O.ambiguity_process(d, x.astype(np.complex128), y.astype(np.complex128))
    amb = b2.Ambiguity(*args, True)
    m_auto = amb.process(x, y).data.copy()
    assert amb.hot_columns() == 0
    amb.set_hot_columns("always")
    m_always = amb.process(x, y).data.copy()
    lvl = 10.0 ** (O.map_metrics(ref)[0] / 10.0)
    n_hot = amb.hot_columns()
    amb.close()
    for m in (m_auto, m_always):
        assert np.max(np.abs(m.astype(np.complex128) - ref)) <= 1e-5 * np.abs(ref).max()
    col = int(np.argmin(np.abs(d.delay - 10)))
    if np.abs(ref[:, col]).max() > 250.0 * lvl:
        assert n_hot >= 1


def test_each_cpi_of_a_batch_has_its_own_columns(b2):
    import torch
    args = (-10, 400, -256, 256, 2_000_000, 2_000_000)
    B = 5
    d = O.ambiguity_dims(*args, True)
    echoes = [(37, -63.0), None, (200, 10.0), (37, 100.0), None]
    data = []
    for c, e in enumerate(echoes):
        if e is None:
            rng = np.random.default_rng(40 + c)
            mk = lambda: ((rng.standard_normal(args[5]) + 1j * rng.standard_normal(args[5])) * 100).astype(np.complex64)
            data.append((mk(), mk()))
        else:
            data.append(echo_cpi(args[5], args[4], 40 + c, *e))
    xs = torch.from_numpy(np.stack([v[0] for v in data])).cuda()
    ys = torch.from_numpy(np.stack([v[1] for v in data])).cuda()
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for mode in ("off", "auto"):
        amb = b2.Ambiguity(*args, True, max_batch=B)
        amb.set_hot_columns(mode)
        out = torch.zeros((B, d.n_doppler_bins, d.n_delay_bins), dtype=torch.complex64, device="cuda")
        met = torch.zeros((B, 2), dtype=torch.float64, device="cuda")
        amb.process_dev(b2.FMT_C32, xs.data_ptr(), ys.data_ptr(), B, args[5], out.data_ptr(), met.data_ptr(), st)
        torch.cuda.synchronize()
        res[mode] = (out.cpu().numpy(), met.cpu().numpy())
        assert amb.hot_columns() == (1 if mode == "auto" else 0)       # of CPI 0
        amb.close()
    assert np.array_equal(res["off"][1], res["auto"][1])               # Map::set_metrics: taken before the rewrite
    for c, e in enumerate(echoes):
        a, b = res["off"][0][c], res["auto"][0][c]
        changed = np.flatnonzero(np.any(a != b, axis=0))
        if e is None:
            assert changed.size == 0, (c, changed)
            continue
        col = int(np.argmin(np.abs(d.delay - e[0])))
        assert list(changed) == [col], (c, changed, col)
    c = 2
    ref = O.ambiguity_process(d, data[c][0].astype(np.complex128), data[c][1].astype(np.complex128))
    g = map_cell_gate(res["auto"][0][c], ref)
    assert g["ok"] and g["cell_rel_above_mean"] <= 3e-5, g
    nz, mx = O.map_metrics(ref)
    assert abs(res["auto"][1][c][0] - nz) < 1e-3 and abs(res["auto"][1][c][1] - mx) < 1e-3


def test_the_strongest_sixteen_of_many(b2):
    """More candidates than the kernel rewrites: 16 of them are taken, by strength as four pulses of the range map (less the
    first pulse's value) show it -- which depends on where in its cycle an echo's Doppler phase is sampled --, the rest keep
    their fp32 values; the echo at 0 Hz is no candidate (its column is constant over the pulses)."""
    args = (-10, 400, -256, 256, 2_000_000, 2_000_000)
    n, fs = args[5], args[4]
    rng = np.random.default_rng(8)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 200.0
    t = np.arange(n) / fs
    y = 0.01 * 200.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    lags = list(range(20, 20 + 8 * 24, 8))                             # 24 echoes
    for i, lag in enumerate(lags):
        xd = np.roll(x, lag)
        xd[:lag] = 0
        y = y + (1.0 + 0.05 * i) * xd * np.exp(2j * np.pi * (5.0 * i - 60.0) * t)
    x, y = x.astype(np.complex64), y.astype(np.complex64)
    d = O.ambiguity_dims(*args, True)
    out = {}
    for mode in ("off", "auto"):
        amb = b2.Ambiguity(*args, True)
        amb.set_hot_columns(mode)
        out[mode] = amb.process(x, y).data.copy()
        hot = amb.hot_columns()
        amb.close()
    assert hot == 16
    changed = np.flatnonzero(np.any(out["off"] != out["auto"], axis=0))
    cols = [int(np.argmin(np.abs(d.delay - lag))) for lag in lags]
    assert changed.size == 16 and set(changed) <= set(cols) and cols[12] not in changed, (changed, cols)
