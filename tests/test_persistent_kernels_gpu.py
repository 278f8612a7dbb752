"""The STEADY STATE of the persistent, software-pipelined Doppler tile kernels against the oracle.

doppler_tile_kernel<8|16> and doppler_tilew_kernel (and their successors for longer Doppler axes) loop
`for (it = blockIdx.x; it < tiles; it += gridDim.x)` and request tile `it + gridDim.x` while tile `it`
is being transformed (kernels.hpp).  With the natural grid (one or two workgroups per CU) a test-sized
launch has fewer tiles than workgroups and every workgroup runs ONE iteration: the prefetch into live
registers, the reuse of the column regions after the end-of-body barrier and the reuse of wsum / wmax
never execute.  Here BLAH2HIP_OPT_DOPPLER_GRID caps the grid at a few workgroups, so each one walks
>= 3 tiles, across CPI boundaries (grid and tiles-per-CPI are coprime) and through ragged last tiles;
the natural-size tests at the end do the same at the sizes bench.py times.  Every CPI of every launch is
compared with the fp64 oracle (Ambiguity.cpp:152-169, Map.cpp:187-206) at the gates of
tests/test_timed_kernels_gpu.py; `tiles > grid` is asserted through blah2hip_amb_get_info.
"""
import pytest

from test_timed_kernels_gpu import CFG2, b2, run_batch  # noqa: F401  (b2 is a fixture)

pytestmark = pytest.mark.gpu


def _assert_steady(amb, min_iters=3):
    from blah2_amd import _lib
    grid, tiles = amb.info(_lib.INFO_DOPPLER_GRID), amb.info(_lib.INFO_DOPPLER_TILES)
    assert tiles >= min_iters * grid, f"{tiles} tiles on {grid} workgroups: not {min_iters} iterations each"
    return grid, tiles


# nD = 513 (all nine row groups), 300 delay bins: 38 half tiles / 19 whole tiles per CPI, the last one ragged
G513 = (-7, 292, -256, 256, 1_026_000, 1_026_000)
# nD = 65: a single row group, the other eight compile-time rows are masked
G65 = (-7, 292, -32, 32, 130_000, 130_000)


@pytest.mark.parametrize("kernel,grid", [("tile8", 5), ("tile8", 7), ("tile8k", 5), ("tile8k", 7), ("tile16", 4), ("tile16", 5)])
@pytest.mark.parametrize("geom", [G513, G65], ids=["nD513", "nD65"])
def test_tile_kernels_forced_small_grid(b2, kernel, grid, geom):
    amb = run_batch(b2, geom, 3, kernel, seeds=(11, 12, 13), targets=((37, -13.0, 0.05),), doppler_grid=grid)
    g, tiles = _assert_steady(amb)
    assert g == grid and tiles == 3 * (38 if kernel in ("tile8", "tile8k") else 19)
    assert tiles % grid != 0  # a last round in which only some workgroups still have a tile


@pytest.mark.parametrize("fmax,n,nD", [(257, 1_030_000, 515), (400, 1_602_000, 801), (512, 2_050_000, 1025)])
@pytest.mark.parametrize("grid", [4, 7])
def test_one_wave_tile_kernel_forced_small_grid(b2, fmax, n, nD, grid):
    """doppler_tilew_kernel: spectrum request before, next-tile request after the spectrum product."""
    geom = (-7, 292, -fmax, fmax, n, n)
    amb = run_batch(b2, geom, 3, "tilew", seeds=(21 + nD, 22 + nD, 23 + nD), targets=((37, -13.0, 0.05),), doppler_grid=grid)
    assert amb.get_n_doppler_bins() == nD
    g, tiles = _assert_steady(amb)
    assert g == grid and tiles == 3 * 38


def test_even_delay_count_takes_the_wide_stores(b2):
    """nDelay even -> the 16-byte row-piece stores of doppler_tilew_kernel (odd counts take the 8-byte path)."""
    geom = (-7, 296, -400, 400, 1_602_000, 1_602_000)  # 304 delay bins = 38 full half tiles
    amb = run_batch(b2, geom, 2, "tilew", seeds=(31, 32), targets=((37, -13.0, 0.05),), doppler_grid=5)
    assert amb.get_n_delay_bins() == 304
    _assert_steady(amb)


def test_cfg2_x16_natural_grid_runs_second_iterations(b2):
    """BASELINE configs[1] x 16 CPIs, automatic plan: 16 x 26 = 416 whole tiles on one workgroup per CU
    -> doppler_tile_kernel<16>, 160 workgroups run a second iteration (the shape bench.py times, 13
    iterations per workgroup there)."""
    from blah2_amd import _lib
    amb = run_batch(b2, CFG2, 16, "auto", seeds=range(300, 316), expect="tile16")
    grid, tiles = amb.info(_lib.INFO_DOPPLER_GRID), amb.info(_lib.INFO_DOPPLER_TILES)
    assert tiles == 416 and grid == min(416, amb.info(_lib.INFO_NUM_CU)) and tiles > grid


def test_cfg2_tile8_natural_grid_runs_second_iterations(b2):
    """The half-tile kernel, two workgroups per CU: 12 x 52 = 624 half tiles on 512 workgroups."""
    from blah2_amd import _lib
    amb = run_batch(b2, CFG2, 12, "tile8", seeds=range(330, 342))
    grid, tiles = amb.info(_lib.INFO_DOPPLER_GRID), amb.info(_lib.INFO_DOPPLER_TILES)
    assert tiles == 624 and tiles > grid


def test_cfg2_tile8k_natural_grid_runs_second_iterations(b2):
    """doppler_tile1k_kernel<8> (round 5: the 16-column kernel on half tiles, two workgroups per CU, the even kernel
    spectrum as a half table): 12 x 52 = 624 half tiles on 512 workgroups, every CPI against the oracle."""
    from blah2_amd import _lib
    amb = run_batch(b2, CFG2, 12, "tile8k", seeds=range(360, 372))
    grid, tiles = amb.info(_lib.INFO_DOPPLER_GRID), amb.info(_lib.INFO_DOPPLER_TILES)
    assert tiles == 624 and tiles > grid


def test_cfg3_geometry_x2_natural_grid(b2):
    """BASELINE configs[2] geometry (1025 x 2048, F = 4096) x 2 CPIs, automatic plan: 2 x 256 half tiles
    on one workgroup per CU -> the one-wave tile kernel's second iteration, every CPI against the oracle."""
    from blah2_amd import _lib
    cfg3 = (-24, 2023, -512, 512, 10_000_000, 10_000_000)
    amb = run_batch(b2, cfg3, 2, "auto", seeds=(350, 351), expect="tilew",
                    targets=((37, -63.0, 0.05), (1500, 300.0, 0.05)), cell_tol=1e-4)
    grid, tiles = amb.info(_lib.INFO_DOPPLER_GRID), amb.info(_lib.INFO_DOPPLER_TILES)
    assert tiles == 512 and tiles > grid


@pytest.mark.parametrize("fmax,n,nD", [(513, 2_054_000, 1027), (800, 3_202_000, 1601), (1024, 4_098_000, 2049)])
@pytest.mark.parametrize("ndelay", [300, 304])
def test_two_wave_tile_kernel(b2, fmax, n, nD, ndelay):
    """doppler_tilew2_kernel (1025 < nD <= 2049: a pair of waves per column on the two-wave 4096-point transform, four
    columns per persistent workgroup, XCD-aware walk over the quarter tiles), forced, on 32 workgroups: three CPIs x 19
    sixteen-column groups x 4 quarters = 228 quarter tiles, seven iterations per workgroup, ragged last group (300 = 18
    x 16 + 12: its fourth quarter does not exist) and both store widths (odd / even delay counts)."""
    from blah2_amd import _lib
    geom = (-7, ndelay - 8, -fmax, fmax, n, n)
    amb = run_batch(b2, geom, 3, "tilew2", seeds=(95 + nD, 96 + nD, 97 + nD), targets=((37, -13.0, 0.05),), cell_tol=1e-4,
                    doppler_grid=32)
    assert amb.get_n_doppler_bins() == nD and amb.get_n_delay_bins() == ndelay
    assert amb.info(_lib.INFO_DOPPLER_FFT_LEN) == 4096
    grid, tiles = _assert_steady(amb)
    assert grid == 32 and tiles == 3 * 19 * 4


@pytest.mark.parametrize("fmax,n,nD", [(513, 2_054_000, 1027), (800, 3_202_000, 1601), (1024, 4_098_000, 2049)])
@pytest.mark.parametrize("ndelay,grid", [(300, 5), (304, 7)])
def test_one_wave_4096_tile_kernel(b2, fmax, n, nD, ndelay, grid):
    """doppler_tilew4_kernel (round 5; 1025 < nD <= 2049: ONE wave per column, the 4096-point transform as four one-wave
    1024-point transforms of the samples 4 n + r and a radix-4 step across them in registers, eight columns per persistent
    workgroup), forced, on 5 / 7 workgroups: three CPIs x 38 half tiles, 16-23 iterations per workgroup across CPI
    boundaries, a ragged last half tile (300 = 37 x 8 + 4) and both store widths (odd / even delay counts); every CPI
    against the oracle."""
    from blah2_amd import _lib
    geom = (-7, ndelay - 8, -fmax, fmax, n, n)
    amb = run_batch(b2, geom, 3, "tilew4", seeds=(195 + nD, 196 + nD, 197 + nD), targets=((37, -13.0, 0.05),), cell_tol=1e-4,
                    doppler_grid=grid)
    assert amb.get_n_doppler_bins() == nD and amb.get_n_delay_bins() == ndelay
    assert amb.info(_lib.INFO_DOPPLER_FFT_LEN) == 4096
    g, tiles = _assert_steady(amb)
    assert g == grid and tiles == 3 * 38
