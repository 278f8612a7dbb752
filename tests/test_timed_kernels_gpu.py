"""GPU parity for the kernels bench.py actually times.

The benchmark runs the device-resident chain on BATCHES of CPIs; the launch-size
rule of the engine then selects kernels that a single-CPI call never reaches
(doppler_tile_kernel<8> at nD = 513, the range kernel's grid-stride path with
cpi > 0, the multi-wave tile kernel at nD = 1025 / 2049).  Every test here
asserts WHICH Doppler kernel ran (blah2hip_amb_get_info) and compares every CPI
of the batch with the fp64 oracle (Ambiguity.cpp:92-172, Map.cpp:187-206) at the
gates of tests/test_ambiguity_gpu.py:
    max|dM| / max|M| <= 1e-5, cell-wise <= 1e-4 above the mean level, metrics within 1e-3 dB.
"""
import numpy as np
import pytest

from oracle import blah2_oracle as O

pytestmark = pytest.mark.gpu

PEAK_TOL, CELL_TOL, DB_TOL = 1e-5, 1e-4, 1e-3


@pytest.fixture(scope="module")
def b2(built_lib):
    import blah2_amd
    assert blah2_amd.device_count() > 0
    return blah2_amd


def assert_cpi(got, met, ref, tag, cell_tol=CELL_TOL):
    got = got.astype(np.complex128)
    peak = np.max(np.abs(ref))
    err = np.abs(got - ref)
    assert err.max() / peak <= PEAK_TOL, f"{tag}: peak-relative error {err.max() / peak:.3e}"
    strong = np.abs(ref) > np.mean(np.abs(ref))
    rel = np.max(err[strong] / np.abs(ref[strong]))
    assert rel <= cell_tol, f"{tag}: element-wise relative error {rel:.3e}"
    noise, mx = O.map_metrics(ref)
    assert abs(met[0] - noise) <= DB_TOL and abs(met[1] - mx) <= DB_TOL, f"{tag}: metrics {met} vs {(noise, mx)}"


def run_batch(b2, geom, B, kernel, seeds, fmt="c32", expect=None, cell_tol=CELL_TOL, targets=((37, -63.0, 0.05),),
              range_kernel=0, doppler_grid=0, range_grid=0, fft_len=0, db_gate=False, synth=None):
    """B distinct CPIs through blah2hip_amb_process_dev in ONE call; every CPI against the oracle."""
    import torch
    dmin, dmax, fmin, fmax, fs, n = geom
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=B)
    if kernel != "auto":
        amb.set_doppler_kernel(kernel)
    if fft_len:
        amb.set_fft_len(fft_len)
    if range_kernel:
        amb.set_range_kernel(range_kernel)
    if doppler_grid:
        amb.set_doppler_grid(doppler_grid)
    if range_grid:
        amb.set_range_grid(range_grid)
    synth = synth or O.synth_iq
    xs, ys = zip(*(synth(n, seed=s, fs=fs, targets=targets, quantise=(fmt != "f16")) for s in seeds))
    nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
    out = torch.zeros((B, nD, nC), dtype=torch.complex64, device="cuda")
    met = torch.zeros((B, 2), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    if fmt == "c32":
        x = torch.from_numpy(np.stack(xs).astype(np.complex64)).cuda()
        y = torch.from_numpy(np.stack(ys).astype(np.complex64)).cuda()
        amb.process_dev(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
    elif fmt == "f16":
        # fp16 IQ storage (BASELINE configs[4]): the oracle is fed the ALREADY-QUANTISED values,
        # so that storage error is not counted as kernel error
        xh = np.stack([np.stack([x_.real, x_.imag], axis=-1) for x_ in xs]).astype(np.float16)
        yh = np.stack([np.stack([y_.real, y_.imag], axis=-1) for y_ in ys]).astype(np.float16)
        xs = [h[:, 0].astype(np.float64) + 1j * h[:, 1].astype(np.float64) for h in xh]
        ys = [h[:, 0].astype(np.float64) + 1j * h[:, 1].astype(np.float64) for h in yh]
        x, y = torch.from_numpy(xh).cuda(), torch.from_numpy(yh).cuda()
        amb.process_dev(b2.FMT_F16, x.data_ptr(), y.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
    else:
        iq = np.stack([np.stack([x_.real, x_.imag, y_.real, y_.imag], axis=-1) for x_, y_ in zip(xs, ys)]).astype(np.int16)
        d = torch.from_numpy(iq).cuda()
        amb.process_dev(b2.FMT_I16, d.data_ptr(), 0, B, n, out.data_ptr(), met.data_ptr(), st)
    torch.cuda.synchronize()
    ran = amb.last_doppler_kernel()
    assert ran == (expect or kernel), f"Doppler kernel that ran: {ran}"
    o, m = out.cpu().numpy(), met.cpu().numpy()
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
    assert (d.n_doppler_bins, d.n_delay_bins) == (nD, nC)
    for c in range(B):
        ref = O.ambiguity_process(d, xs[c], ys[c])
        assert_cpi(o[c], m[c], ref, f"cpi {c} of {B} [{ran}]", cell_tol)
        if db_gate:  # the JSON-map gate of tests/gates.py on this CPI too
            from gates import db_map_gate
            g = db_map_gate(o[c], m[c][0], ref)
            print(f"\n[dB map, cpi {c} of {B}, {fmt}, {ran}] {g}")
            assert g["ok"], g
    return amb


CFG2 = (-10, 400, -256, 256, 2_000_000, 2_000_000)


@pytest.mark.parametrize("B", [3, 8])
def test_cfg2_batched_takes_the_tile_kernel(b2, B):
    """BASELINE configs[1] in batches: B*52 half-tiles >= numCU/2 -> doppler_tile_kernel<8> at
    nD = 513 (all nine row groups, the k = 8 tail row, the row rotation by 257), and the range
    kernel's grid-stride loop reaches pulses of cpi > 0."""
    amb = run_batch(b2, CFG2, B, "auto", seeds=range(40, 40 + B), expect="tile8")
    from blah2_amd import _lib
    # a handle whose largest launch has a pulse for each of the 12 wave slots per CU of the 1024-point one-wave kernel
    # plans F = 1024 for it (same butterfly count as F = 2048 here, measured 1-2 % faster); smaller ones F = 2048: the
    # one-wave kernel from 8 pulses per CU on, else the workgroup kernel
    w1k = B * 513 >= 12 * amb.info(_lib.INFO_NUM_CU)
    assert (amb.get_n_doppler_bins(), amb.get_n_delay_bins(), amb.dims.fft_len) == (513, 411, 1024 if w1k else 2048)
    full = B * 513 >= 8 * amb.info(_lib.INFO_NUM_CU)
    assert amb.info(_lib.INFO_LAST_RANGE_KERNEL) == (_lib.RANGE_WAVE1K if w1k else _lib.RANGE_WAVE if full else _lib.RANGE_E16)


@pytest.mark.parametrize("fmt", ["c32", "i16"])
@pytest.mark.parametrize("which", ["wave", "e16"])
def test_cfg2_both_range_kernels_forced(b2, fmt, which):
    """The one-wave range kernel (32 points per lane, one LDS exchange, no barriers; the default at
    F = 2048) and the workgroup kernel it replaced there (16 points per thread), each forced, on batches
    of BASELINE configs[1]: grid-stride over the pulses of several CPIs, three segments per pulse."""
    from blah2_amd import _lib
    k = _lib.RANGE_WAVE if which == "wave" else _lib.RANGE_E16
    amb = run_batch(b2, CFG2, 3, "auto", seeds=(70, 71, 72), fmt=fmt, expect="tile8", range_kernel=k)
    assert amb.info(_lib.INFO_LAST_RANGE_KERNEL) == k


def test_one_wave_range_kernel_ragged_geometry(b2):
    """Ragged pulse length, a lag window that starts at a positive lag and a last segment of a few
    samples, single CPI and a batch."""
    from blah2_amd import _lib
    geom = (1, 299, -100, 100, 1_000_000, 777_001)
    for B in (1, 2):
        amb = run_batch(b2, geom, B, "auto", seeds=range(80, 80 + B), range_kernel=_lib.RANGE_WAVE, fft_len=2048,
                        expect="sub4", targets=((40, 30.0, 0.05),))
        assert amb.dims.fft_len == 2048 and amb.info(_lib.INFO_LAST_RANGE_KERNEL) == _lib.RANGE_WAVE


def test_cfg2_batched_int16_wire_format(b2):
    run_batch(b2, CFG2, 3, "auto", seeds=(50, 51, 52), fmt="i16", expect="tile8")


@pytest.mark.parametrize("kernel", ["tile8", "tile8k", "tile16", "sub4", "column", "direct"])
def test_cfg2_every_doppler_kernel(b2, kernel):
    """The same two CPIs through each Doppler kernel that covers nD = 513, forced."""
    run_batch(b2, CFG2, 2, kernel, seeds=(60, 61))


# Doppler lengths either side of each tile kernel's row grouping (64 rows per register at one wave
# per column), ragged delay counts (last tile partly empty: 411 = 51*8+3, 300 = 18*16+12), B = 2.
@pytest.mark.parametrize("fmax,n,nD", [(32, 130_000, 65), (255, 1_022_000, 511), (256, 1_026_000, 513)])
@pytest.mark.parametrize("kernel", ["tile8", "tile8k", "tile16"])
def test_tile_kernels_row_groups_and_ragged_tiles(b2, fmax, n, nD, kernel):
    geom = (-7, 292, -fmax, fmax, n, n)  # fs = n: 1 s CPI, 1 Hz Doppler resolution, nCorr = 2000
    amb = run_batch(b2, geom, 2, kernel, seeds=(70 + nD, 71 + nD), targets=((37, -13.0, 0.05),))
    assert amb.get_n_doppler_bins() == nD and amb.get_n_delay_bins() == 300


@pytest.mark.parametrize("fmax,n,nD", [(257, 1_030_000, 515), (400, 1_602_000, 801), (512, 2_050_000, 1025)])
@pytest.mark.parametrize("kernel", ["tilew", "tilem"])
def test_one_wave_and_two_wave_tile_kernels(b2, fmax, n, nD, kernel):
    """513 < nD <= 1025 -> doppler_tilew_kernel (one-wave 2048-point columns, 8 per workgroup; the
    automatic choice) and doppler_tilem_kernel<8> (two-wave columns), each forced; ragged last tile."""
    geom = (-7, 292, -fmax, fmax, n, n)
    amb = run_batch(b2, geom, 2, kernel, seeds=(80 + nD, 81 + nD), targets=((37, -13.0, 0.05),))
    assert amb.get_n_doppler_bins() == nD
    from blah2_amd import _lib
    assert amb.info(_lib.INFO_DOPPLER_FFT_LEN) == 2048


@pytest.mark.parametrize("fmax,n,nD", [(513, 2_054_000, 1027), (1024, 4_098_000, 2049)])
def test_four_wave_tile_kernel(b2, fmax, n, nD):
    """1025 < nD <= 2049 -> doppler_tilem_kernel<16> (four-wave columns, 4 per workgroup)."""
    geom = (-7, 292, -fmax, fmax, n, n)
    amb = run_batch(b2, geom, 2, "tilem", seeds=(90 + nD, 91 + nD), targets=((37, -13.0, 0.05),), cell_tol=1e-4)
    assert amb.get_n_doppler_bins() == nD
    from blah2_amd import _lib
    assert amb.info(_lib.INFO_DOPPLER_FFT_LEN) == 4096


def test_auto_rule_small_launch_takes_the_four_column_kernel(b2):
    """A lone CPI at nD <= 513: doppler_sub1k_kernel (four columns per workgroup, Map::set_metrics finished by the last
    workgroup to arrive: no metrics launch); the handle is then reused, so the arrival counter must be back at zero."""
    amb = run_batch(b2, CFG2, 1, "auto", seeds=(99,), expect="sub4")
    with pytest.raises(b2.Blah2HipError):  # nD = 513 is outside the multi-wave tile kernel's plan (M = 1024)
        amb.set_doppler_kernel("tilem")


@pytest.mark.parametrize("geom,fft_len,kernel", [(CFG2, 2048, "e16"), ((-24, 2023, -64, 64, 1_260_000, 1_260_000), 4096, "e16"),
                                                  ((-10, 100, -100, 100, 1_000_000, 100_000), 1024, "e8")])
def test_range_kernel_of_every_transform_length_batched(b2, geom, fft_len, kernel):
    """F = 1024 -> the 8-point-per-thread kernel whose last transform stage runs across lanes (DPP),
    F = 2048 / 4096 -> the 16-point workgroup kernel (two CPIs are below the launch size at which F = 2048
    switches to the one-wave kernel); two distinct CPIs per launch."""
    from blah2_amd import _lib
    # (F = 1024 in launches this small is the pulse-per-workgroup kernel's since round 4: the 8-point kernel is forced)
    amb = run_batch(b2, geom, 2, "auto", seeds=(5, 6), expect=_expected_doppler(b2, geom),
                    targets=((37, -13.0, 0.05),), cell_tol=1e-4, range_kernel=_lib.RANGE_E8 if kernel == "e8" else 0)
    assert amb.dims.fft_len == fft_len
    assert amb.info(_lib.INFO_LAST_RANGE_KERNEL) == {"e8": _lib.RANGE_E8, "e16": _lib.RANGE_E16, "wave": _lib.RANGE_WAVE}[kernel]


def _expected_doppler(b2, geom):
    # what the launch-size rule picks for two CPIs of this geometry (asserted inside run_batch)
    dmin, dmax, fmin, fmax, fs, n = geom
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
    if d.n_doppler_bins <= 513:  # 256 CUs: whole tiles from one per CU on, half tiles from one per two CUs
        if 2 * -(-d.n_delay_bins // 16) >= 256:
            return "tile16"
        return "tile8" if 2 * -(-d.n_delay_bins // 8) >= 128 else "sub4"
    if d.n_doppler_bins <= 1025:
        return "tilew" if 2 * -(-d.n_delay_bins // 8) >= 128 else "column"
    return "tilew2" if 2 * -(-d.n_delay_bins // 4) >= 128 else "column"


@pytest.mark.parametrize("geom,fft_len", [(CFG2, 2048), (CFG2, 1024), ((-24, 2023, -64, 64, 1_260_000, 1_260_000), 4096),
                                          ((-10, 100, -100, 100, 1_000_000, 100_000), 1024), ((-10, 400, -300, 200, 2_000_000, 1_000_000), 0)])
def test_mixed_format_int16_reference_fp32_surveillance(b2, geom, fft_len):
    """BLAH2HIP_FMT_I16X_C32Y -- x from the .rspduo words, y from an fp32 plane (the ambiguity stage behind the int16
    clutter filter): every range kernel (one per transform length; F = 2048 both forms) and the rotate kernel
    (asymmetric Doppler limits) against the fp32-plane path on the same values: identical maps."""
    import torch
    from blah2_amd import _lib
    dmin, dmax, fmin, fmax, fs, n = geom
    B = 9 if geom == CFG2 else 2  # nine CPIs of cfg 2 reach the launch size of the one-wave kernels (F = 2048 and F = 1024)
    xs, ys = zip(*(O.synth_iq(n, seed=700 + c, fs=fs, targets=((37, -13.0, 0.05),)) for c in range(B)))
    iq = np.stack([np.stack([x.real, x.imag, y.real, y.imag], axis=-1) for x, y in zip(xs, ys)]).astype(np.int16)
    d_iq = torch.from_numpy(iq).cuda()
    dx = torch.from_numpy(np.stack(xs).astype(np.complex64)).cuda()
    dy = torch.from_numpy(np.stack(ys).astype(np.complex64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=B)
    if fft_len:
        amb.set_fft_len(fft_len)
    nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
    maps = []
    for fmt, px in ((b2.FMT_C32, dx), (b2.FMT_I16X_C32Y, d_iq)):
        out = torch.zeros((B, nD, nC), dtype=torch.complex64, device="cuda")
        amb.process_dev(fmt, px.data_ptr(), dy.data_ptr(), B, n, out.data_ptr(), None, st)
        torch.cuda.synchronize()
        maps.append(out.cpu().numpy())
        if fft_len:
            assert amb.dims.fft_len == fft_len
    if fft_len == 2048:
        assert amb.info(_lib.INFO_LAST_RANGE_KERNEL) == _lib.RANGE_WAVE
    if geom == CFG2 and fft_len == 1024:
        assert amb.info(_lib.INFO_LAST_RANGE_KERNEL) == _lib.RANGE_WAVE1K
    assert np.array_equal(maps[0], maps[1])
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
    ref = O.ambiguity_process(d, xs[B - 1], ys[B - 1])
    assert np.max(np.abs(maps[1][B - 1] - ref)) / np.max(np.abs(ref)) <= PEAK_TOL


@pytest.mark.parametrize("geom,fmt,xhalf", [((-24, 2023, -64, 64, 1_260_000, 1_260_000), "c32", True),
                                            ((-24, 2023, -64, 64, 1_260_000, 1_260_000), "i16", True),
                                            ((-10, 89, -20, 20, 123_000, 123_000), "c32", False),
                                            ((1, 299, -100, 100, 1_000_000, 777_001), "c32", False)])
def test_4096_point_range_kernel_shapes(b2, geom, fmt, xhalf):
    """range_kernel<16> (F = 4096, the workgroup transform), three CPIs per launch: the half-zero reference segments of the
    configs[2] shape (9 of 16 x loads, pruned first step), full segments, a ragged pulse length with a lag window that
    starts at a positive lag, and the int16 wire format.  (Round 3 ran these shapes on the two-wave rangew2_kernel too;
    it measured 5 % slower than this one, spilled, and was removed in round 4.)"""
    from blah2_amd import _lib
    amb = run_batch(b2, geom, 3, "auto", seeds=(120, 121, 122), fmt=fmt, fft_len=4096, range_kernel=_lib.RANGE_E16,
                    expect=_expected_doppler3(geom), targets=((37, -13.0, 0.05),), cell_tol=1e-4)
    assert amb.dims.fft_len == 4096 and amb.info(_lib.INFO_LAST_RANGE_KERNEL) == _lib.RANGE_E16
    assert (amb.dims.seg_len <= 2048) == xhalf


def _expected_doppler3(geom):
    dmin, dmax, fmin, fmax, fs, n = geom
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
    if d.n_doppler_bins <= 513:
        if 3 * -(-d.n_delay_bins // 16) >= 256:
            return "tile16"
        return "tile8" if 3 * -(-d.n_delay_bins // 8) >= 128 else "sub4"
    return "tilew" if 3 * -(-d.n_delay_bins // 8) >= 128 else "column"


@pytest.mark.parametrize("geom,out7", [((-7, 292, -50, 50, 171_700, 171_700), True), ((-7, 492, -50, 50, 155_540, 155_540), False)])
def test_one_wave_range_kernel_long_segments(b2, geom, out7):
    """The unpruned instantiations of rangew_kernel (segLen > 24*64: every load issued): with few lags the inverse still
    computes only the seven wanted outputs per lane (the configs[4] shape: segLen 1627, 411 lags), with more all 32."""
    from blah2_amd import _lib
    amb = run_batch(b2, geom, 2, "auto", seeds=(140, 141), range_kernel=_lib.RANGE_WAVE, fft_len=2048,
                    expect="sub4", targets=((37, -13.0, 0.05),))
    assert amb.dims.fft_len == 2048 and amb.dims.seg_len > 1536
    assert (amb.get_n_delay_bins() <= 448) == out7
    assert amb.info(_lib.INFO_LAST_RANGE_KERNEL) == _lib.RANGE_WAVE


@pytest.mark.parametrize("geom,fmt,seg576,shortx,out7,grid", [
    (CFG2, "c32", True, True, True, 0),                                        # BASELINE configs[1]: 7 segments of 576, carried y' registers
    (CFG2, "i16", True, True, True, 3),                                        # the same from .rspduo words, 36 waves: 43 pulses per wave
    ((-7, 492, -50, 50, 155_540, 155_540), "c32", False, True, False, 0),      # 500 lags: all 16 outputs of the inverse, no carry
    ((-5, 94, -40, 40, 240_000, 240_000), "c32", False, False, True, 2),       # long segments (every x load issued), few lags
    ((1, 299, -100, 100, 1_000_000, 777_001), "i16", False, False, True, 1),   # ragged pulse length, window starts at a positive lag
    ((-10, 400, -2, 2, 20_000, 20_000), "c32", True, True, True, 0)])          # 5 pulses per CPI: most waves leave without a pulse
def test_one_wave_1024_range_kernel(b2, geom, fmt, seg576, shortx, out7, grid):
    """rangew1k_kernel (one wave per pulse on the one-wave 1024-point transform, the next segment's loads in flight
    during the transforms, three CPIs per launch), forced: every instantiation, the carried-over y' registers of the
    576-sample segmentation, pulse boundaries inside a wave's walk (grid cap), waves without work."""
    from blah2_amd import _lib
    amb = run_batch(b2, geom, 3, "auto", seeds=(150, 151, 152), fmt=fmt, fft_len=1024, range_kernel=_lib.RANGE_WAVE1K,
                    expect=_expected_doppler3(geom), targets=((37, -13.0, 0.05),), cell_tol=1e-4, range_grid=grid)
    assert amb.dims.fft_len == 1024 and amb.info(_lib.INFO_LAST_RANGE_KERNEL) == _lib.RANGE_WAVE1K
    assert (amb.dims.seg_len == 576) == seg576 and (amb.dims.seg_len <= 576) == shortx and (amb.get_n_delay_bins() <= 448) == out7
    if grid:
        assert amb.info(_lib.INFO_RANGE_GRID) == grid


def test_f1024_kernels_agree(b2):
    """The two F = 1024 range kernels (8 points per thread in a workgroup, 16 per lane in one wave) on the same batch."""
    import torch
    from blah2_amd import _lib
    geom = (-10, 400, -32, 32, 250_000, 250_000)
    n = geom[5]
    xs, ys = zip(*(O.synth_iq(n, seed=s, fs=geom[4], targets=((100, 11.0, 0.1),)) for s in (7, 8)))
    x = torch.from_numpy(np.stack(xs).astype(np.complex64)).cuda()
    y = torch.from_numpy(np.stack(ys).astype(np.complex64)).cuda()
    maps = []
    for k in (_lib.RANGE_E8, _lib.RANGE_WAVE1K):
        amb = b2.Ambiguity(*geom, True, max_batch=2)
        amb.set_fft_len(1024)
        amb.set_range_kernel(k)
        out = torch.zeros((2, amb.get_n_doppler_bins(), amb.get_n_delay_bins()), dtype=torch.complex64, device="cuda")
        amb.process_dev(b2.FMT_C32, x.data_ptr(), y.data_ptr(), 2, n, out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert amb.info(_lib.INFO_LAST_RANGE_KERNEL) == k
        maps.append(out.cpu().numpy())
    assert np.abs(maps[0] - maps[1]).max() / np.abs(maps[0]).max() <= 5e-7


def test_tile16_on_both_transforms(b2):
    """The 16-column Doppler tile kernel on the one-wave 1024-point transform (the default: chirp, rotation targets and row
    indices carried in registers across the tiles) against the workgroup-transform version it replaced: same maps to
    rounding, both against the oracle, at nD = 513 and at a short, even Doppler length (rows that do not exist)."""
    for geom, B in ((CFG2, 10), ((-10, 400, -50, 50, 300_000, 300_000), 12)):
        a = run_batch(b2, geom, B, "tile16", seeds=range(500, 500 + B))
        b = run_batch(b2, geom, B, "tile16wg", seeds=range(500, 500 + B))
        assert a.last_doppler_kernel() == "tile16" and b.last_doppler_kernel() == "tile16wg"


def test_four_column_kernel_finishes_the_metrics_itself_launch_after_launch(b2):
    """doppler_sub1k_kernel forced on ONE handle for launches of 1, 3 and 2 CPIs: Map::set_metrics comes out of the last
    workgroup of each CPI (an arrival counter per CPI that must be back at zero for the next launch), the map from 103
    four-column workgroups per CPI; every CPI of every launch against the oracle.  Ragged: 411 = 102 * 4 + 3."""
    import torch
    dmin, dmax, fmin, fmax, fs, n = (-10, 400, -100, 100, 1_000_000, 402_000)
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=3)
    amb.set_doppler_kernel("sub4")
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
    nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
    st = torch.cuda.current_stream().cuda_stream
    for rep, B in enumerate((1, 3, 2, 1)):
        xs, ys = zip(*(O.synth_iq(n, seed=300 + 10 * rep + c, fs=fs, targets=((37, -13.0, 0.05),)) for c in range(B)))
        x = torch.from_numpy(np.stack(xs).astype(np.complex64)).cuda()
        y = torch.from_numpy(np.stack(ys).astype(np.complex64)).cuda()
        out = torch.zeros((B, nD, nC), dtype=torch.complex64, device="cuda")
        met = torch.full((B, 2), float("nan"), dtype=torch.float64, device="cuda")
        amb.process_dev(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, n, out.data_ptr(), met.data_ptr(), st)
        torch.cuda.synchronize()
        assert amb.last_doppler_kernel() == "sub4"
        o, m = out.cpu().numpy(), met.cpu().numpy()
        for c in range(B):
            assert_cpi(o[c], m[c], O.ambiguity_process(d, xs[c], ys[c]), f"launch {rep} cpi {c} of {B}")


@pytest.mark.parametrize("geom,fmt,B", [(CFG2, "c32", 1), (CFG2, "i16", 2),
                                        ((-10, 89, -100, 100, 1_000_000, 603_000), "c32", 2),    # 4 segments of 750: every x load, 7 outputs
                                        ((-7, 492, -50, 50, 155_540, 155_540), "c32", 1),         # 500 lags: all 16 outputs
                                        ((1, 299, -100, 100, 1_000_000, 777_001), "i16", 3),      # ragged pulse, first lag positive
                                        ((-10, 400, -50, 50, 800_000, 100_000), "c32", 1)])       # 13 segments: more than three per wave
def test_pulse_per_workgroup_range_kernel(b2, geom, fmt, B):
    """rangeps_kernel (F = 1024, small launches): a workgroup of FOUR waves per pulse, wave q walking segments q, q + 4, ...
    with its partial sum in registers; the four sums meet in LDS and the last wave inverts -- forced, on the cfg 2 shape
    (seven segments of 557, pruned x', seven outputs), with fewer and longer segments, with more than 448 lags, on a ragged
    geometry, on the int16 wire format, and with thirteen segments (four passes of the walk: the automatic choice stops at
    eight, use_ps_range in csrc/capi.hip); every CPI against the oracle."""
    from blah2_amd import _lib
    amb = run_batch(b2, geom, B, "auto", seeds=range(400, 400 + B), fmt=fmt, fft_len=1024, range_kernel=_lib.RANGE_PS,
                    expect=None if B > 3 else _expected_small(geom, B), targets=((37, -13.0, 0.05),))
    assert amb.dims.fft_len == 1024 and amb.info(_lib.INFO_LAST_RANGE_KERNEL) == _lib.RANGE_PS
    assert amb.dims.n_seg == 13 if geom[5] == 100_000 else amb.dims.n_seg <= 7


def test_automatic_choice_leaves_long_walks_to_the_eight_point_kernel(b2):
    """Thirteen segments per pulse on a lone CPI: use_ps_range's automatic branch stops at two segments per wave (the shape
    its timings and the planner's cost factor cover), so the launch runs range8_kernel."""
    from blah2_amd import _lib
    geom = (-10, 400, -50, 50, 800_000, 100_000)
    amb = run_batch(b2, geom, 1, "auto", seeds=(431,), fft_len=1024, expect=_expected_small(geom, 1), targets=((37, -13.0, 0.05),))
    assert amb.dims.n_seg == 13 and amb.info(_lib.INFO_LAST_RANGE_KERNEL) == _lib.RANGE_E8


def _expected_small(geom, B):
    dmin, dmax, fmin, fmax, fs, n = geom
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
    if d.n_doppler_bins <= 513:
        if B * -(-d.n_delay_bins // 16) >= 256:
            return "tile16"
        return "tile8" if B * -(-d.n_delay_bins // 8) >= 128 else "sub4"
    return "column"


def test_a_lone_cpi_takes_the_small_launch_kernels(b2):
    """The planner's choice for a handle of one CPI at configs[1]: F = 1024 on the pulse-per-workgroup range kernel and the
    four-column Doppler kernel (two launches for the whole chain)."""
    from blah2_amd import _lib
    amb = run_batch(b2, CFG2, 1, "auto", seeds=(77,), expect="sub4")
    assert amb.dims.fft_len == 1024 and amb.info(_lib.INFO_LAST_RANGE_KERNEL) == _lib.RANGE_PS


# ---- fp16 IQ storage (BASELINE configs[4]): every range-kernel instantiation of InF16 the planner can reach ----------------

CFG5 = (-10, 400, -512, 512, 20_000_000, 40_000_000)


def synth_iq_device(n, seed, fs, targets, quantise=False, ref_amp=300.0, noise_amp=30.0, direct=0.8):
    """oracle.synth_iq's signal model drawn on the device (40 M samples take 35 s per CPI with NumPy's generator):
    test INPUT only -- the oracle then works on exactly the values the kernels read."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    x = ref_amp * torch.view_as_complex(torch.randn((n, 2), generator=g, device="cuda", dtype=torch.float64))
    y = direct * x
    t = torch.arange(n, device="cuda", dtype=torch.float64) / fs
    for d, f, a in targets:
        xd = torch.roll(x, d)
        xd[:d] = 0
        y = y + a * xd * torch.exp(2j * torch.pi * f * t)
    y = y + noise_amp * torch.view_as_complex(torch.randn((n, 2), generator=g, device="cuda", dtype=torch.float64))
    if quantise:
        x, y = (torch.view_as_complex(torch.clamp(torch.round(torch.view_as_real(v)), -32768, 32767)) for v in (x, y))
    return x.cpu().numpy(), y.cpu().numpy()


def test_cfg5_fp16_timed_combination(b2):
    """BASELINE configs[4] as `bench.py --config cfg5 --fmt f16` times it: rangew_kernel<InF16,false,true> (F = 2048,
    segments of 1627 samples: every load issued, seven outputs per lane) into doppler_tilew2_kernel (nD = 2049), two
    CPIs of 40 M fp16 samples per launch, each against the fp64 oracle on the quantised values
    (Ambiguity.cpp:106-169)."""
    from blah2_amd import _lib
    amb = run_batch(b2, CFG5, 2, "auto", seeds=(900, 901), fmt="f16", expect="tilew2",
                    targets=((37, -63.0, 0.05), (300, 250.25, 0.05)), db_gate=True, synth=synth_iq_device)
    assert (amb.get_n_doppler_bins(), amb.get_n_delay_bins(), amb.dims.fft_len) == (2049, 411, 2048)
    assert amb.dims.seg_len > 24 * 64
    assert amb.info(_lib.INFO_LAST_RANGE_KERNEL) == _lib.RANGE_WAVE


@pytest.mark.parametrize("geom,fft_len,kernel,B", [
    (CFG2, 1024, "wave1k", 3),                                          # rangew1k_kernel<InF16,true,true,true>: carried y' registers
    ((-7, 492, -50, 50, 155_540, 155_540), 1024, "wave1k", 3),          # <InF16,true,false>: 500 lags, no carry
    ((-5, 94, -40, 40, 240_000, 240_000), 1024, "wave1k", 3),           # <InF16,false,true>: long segments
    (CFG2, 2048, "wave", 3),                                            # rangew_kernel<InF16,true,true>: pruned windows
    ((-7, 492, -50, 50, 155_540, 155_540), 2048, "wave", 2),            # rangew_kernel<InF16,false,false>
    ((-10, 100, -100, 100, 1_000_000, 100_000), 1024, "e8", 2),         # range8_kernel<2,InF16>
    (CFG2, 2048, "e16", 2),                                             # range_kernel<8,InF16>
    ((-24, 2023, -64, 64, 1_260_000, 1_260_000), 4096, "e16", 2),       # range_kernel<16,InF16>: half-zero x segments
    ((-10, 89, -20, 20, 123_000, 123_000), 4096, "e16", 2),             # range_kernel<16,InF16>: full segments
    (CFG2, 1024, "ps", 1),                                              # rangeps_kernel<InF16,true,true>: a lone CPI
    ((1, 299, -100, 100, 1_000_000, 777_001), 1024, "ps", 2)])          # rangeps_kernel<InF16,...>: ragged pulse
def test_fp16_every_range_kernel_forced(b2, geom, fft_len, kernel, B):
    """Each InF16 range-kernel instantiation forced at a small geometry, every CPI against the oracle on the
    quantised values; asserts which range kernel ran."""
    from blah2_amd import _lib
    k = {"wave1k": _lib.RANGE_WAVE1K, "wave": _lib.RANGE_WAVE, "e8": _lib.RANGE_E8, "e16": _lib.RANGE_E16,
         "ps": _lib.RANGE_PS}[kernel]
    amb = run_batch(b2, geom, B, "direct", seeds=range(910, 910 + B), fmt="f16", fft_len=fft_len, range_kernel=k,
                    targets=((37, -13.0, 0.05),), cell_tol=1e-4)
    assert amb.dims.fft_len == fft_len and amb.info(_lib.INFO_LAST_RANGE_KERNEL) == k
