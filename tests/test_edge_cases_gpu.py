"""GPU: edge geometries of the ambiguity engine against the fp64 oracle: tiny
pulses, one-sided lag windows at the limits the reference allows
(Ambiguity.cpp:132-146 needs delayMin <= 1 and delayMax >= -1), ragged tails,
lag windows that force every transform length, and the documented limits."""
import numpy as np
import pytest

from oracle import blah2_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def b2(built_lib):
    import blah2_amd
    assert blah2_amd.device_count() > 0
    return blah2_amd


def run(b2, args, seed=0, quantise=True):
    dmin, dmax, fmin, fmax, fs, n, rh = args
    x, y = O.synth_iq(n, seed=seed, fs=fs, targets=((max(dmin, 0) + 1, 0.3 * fmax, 0.1),), quantise=quantise)
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, rh)
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, rh)
    assert (amb.get_n_doppler_bins(), amb.get_n_delay_bins(), amb.get_n_corr(), amb.get_nfft()) == \
        (d.n_doppler_bins, d.n_delay_bins, d.n_corr, d.nfft)
    m = amb.process(x, y)
    ref = O.ambiguity_process(d, x, y)
    err = np.abs(m.data.astype(np.complex128) - ref)
    assert err.max() / np.abs(ref).max() <= 1e-5, err.max() / np.abs(ref).max()
    noise, mx = O.map_metrics(ref)
    assert abs(m.noisePower - noise) <= 1e-3 and abs(m.maxPower - mx) <= 1e-3
    return amb


@pytest.mark.parametrize("args", [
    (-1, 1, -2, 2, 1000, 1000, False),          # smallest lag window the reference allows, 5 pulses of 200
    (1, 60, -5, 5, 10_000, 10_000, True),       # delayMin = +1: positive lags only
    (-60, -1, -5, 5, 10_000, 10_000, True),     # delayMax = -1: negative lags only
    (0, 0 + 300, -3, 3, 20_000, 19_997, True),  # ragged: n not a multiple of anything
    (-10, 400, -1, 1, 2_000_000, 2_000_000, True),    # 3 pulses of 666 666 samples: nCorr > uint16 -> wraps like the reference
    (-5, 1500, -20, 20, 1_000_000, 500_000, True),   # nDelay = 1506 -> F = 2048 or 4096
    (-100, 3000, -10, 10, 1_000_000, 1_000_000, True),  # nDelay = 3101 -> only F = 4096 fits
    (-10, 100, -7, 30, 100_000, 100_000, True),      # asymmetric Doppler window -> rotate kernel
])
def test_geometries(b2, args):
    dmin, dmax, fmin, fmax, fs, n, rh = args
    d = O.ambiguity_dims(*args)
    if d.n_corr == 0:
        with pytest.raises(b2.Blah2HipError):
            b2.Ambiguity(*args)
        return
    run(b2, args, seed=abs(dmin) + dmax)


def test_doppler_transform_lengths(b2):
    # nDoppler = 2*floor(fMax*n/fs)+1: exercise 3, 65, 513 (M = 1024 boundary), 515 (first M = 2048), 1025
    for fmax, n, fs in [(1, 100_000, 100_000), (32, 200_000, 200_000), (256, 400_000, 400_000),
                        (257, 400_000, 400_000), (512, 800_000, 800_000)]:
        amb = run(b2, (-4, 40, -fmax, fmax, fs, n, True), seed=fmax)
        assert amb.get_n_doppler_bins() == 2 * fmax + 1


def test_unsupported_geometries_fail_loudly(b2):
    # outside the range where the reference's own lag gather is in bounds
    with pytest.raises(b2.Blah2HipError):
        b2.Ambiguity(5, 100, -10, 10, 1_000_000, 1_000_000, True)
    with pytest.raises(b2.Blah2HipError):
        b2.Ambiguity(-10, 100, 10, -10, 1_000_000, 1_000_000, True)
    # |delay| >= nfft: Ambiguity.cpp:132-146 would index outside its nfft-point buffer (2049 pulses of 488 samples, nfft = 1000)
    with pytest.raises(b2.Blah2HipError) as e:
        b2.Ambiguity(-10, 1000, -2048, 2048, 2_000_000, 1_000_000, True)
    assert e.value.code == -3


def test_more_delay_bins_than_one_transform_holds(b2):
    """4301 delay bins (the reference accepts up to 65535, Ambiguity.h:80-89): the lag window runs as three chunks of at
    most 2048 lags on the 4096-point transform, each into its own columns of the range map."""
    amb = run(b2, (-100, 4200, -10, 10, 1_000_000, 1_000_000, True), seed=3)
    assert amb.get_n_delay_bins() == 4301 and amb.dims.fft_len == 4096
    # the same through the batched device chain (two CPIs), every Doppler kernel choice left to the planner
    import torch
    n, fs = 400_000, 400_000
    args = (-30, 4400, -4, 4, fs, n)
    amb = b2.Ambiguity(*args, True, max_batch=2)
    xs, ys = zip(*(O.synth_iq(n, seed=s_, fs=fs, targets=((4000, 2.0, 0.1), (17, -3.0, 0.05))) for s_ in (1, 2)))
    dx = torch.from_numpy(np.stack(xs).astype(np.complex64)).cuda()
    dy = torch.from_numpy(np.stack(ys).astype(np.complex64)).cuda()
    out = torch.zeros((2, amb.get_n_doppler_bins(), amb.get_n_delay_bins()), dtype=torch.complex64, device="cuda")
    amb.process_dev(b2.FMT_C32, dx.data_ptr(), dy.data_ptr(), 2, n, out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    d = O.ambiguity_dims(*args, True)
    for c in range(2):
        ref = O.ambiguity_process(d, xs[c], ys[c])
        assert np.abs(out[c].cpu().numpy() - ref).max() / np.abs(ref).max() <= 1e-5
        i, j = np.unravel_index(np.argmax(np.abs(ref[:, 3000:])), ref[:, 3000:].shape)
        assert d.delay[3000 + j] == 4000  # the far target sits in the third chunk


def test_lag_chunks_with_an_asymmetric_doppler_window(b2):
    """Chunked lag windows behind the rotation of the reference channel about the Doppler centre (Ambiguity.cpp:95-102):
    4301 delay bins in three chunks, Doppler limits -30 .. +10 Hz; and the aliased lags of a short-pulse geometry with
    an off-centre window."""
    run(b2, (-100, 4200, -30, 10, 400_000, 400_000, True), seed=5)
    args = (-600, 10, -2048, 1024, 2_000_000, 1_000_000, True)
    x, y = O.synth_iq(args[5], seed=22, fs=args[4], targets=((20, 600.0, 0.1), (-37, -200.0, 0.1)))
    amb = b2.Ambiguity(*args)
    m = amb.process(x, y)
    ref = O.ambiguity_process(O.ambiguity_dims(*args), x, y)
    assert np.abs(m.data.astype(np.complex128) - ref).max() / np.abs(ref).max() <= 1e-5


@pytest.mark.parametrize("dmin,dmax", [(-10, 600), (-600, 10), (-700, 700), (-999, 999)])
def test_lags_the_reference_aliases(b2, dmin, dmax):
    """Short pulses: 2049 pulses of 488 samples, nfft = 1000.  Delays beyond nfft - nCorr = 512 read, in the reference's
    nfft-point CIRCULAR correlation (Ambiguity.cpp:132-146), the opposite-sign lag d -+ nfft; delays 488..512 are zero.  The
    engine maps every delay to its one linear lag and runs the runs of consecutive lags as chunks (here up to three, with
    column offsets that are not multiples of 16)."""
    args = (dmin, dmax, -2048, 2048, 2_000_000, 1_000_000, True)
    x, y = O.synth_iq(args[5], seed=21, fs=args[4], targets=((20, 600.0, 0.1), (-37, -200.0, 0.1)))
    amb = b2.Ambiguity(*args)
    assert (amb.get_n_corr(), amb.get_nfft()) == (488, 1000)
    m = amb.process(x, y)
    ref = O.ambiguity_process(O.ambiguity_dims(*args), x, y)
    peak = np.abs(ref).max()
    assert np.abs(m.data.astype(np.complex128) - ref).max() / peak <= 1e-5
    far = np.abs(amb.delay) > 512
    assert far.any() and np.abs(ref[:, far]).max() / peak > 1e-3  # the aliased columns are not empty


def test_lags_longer_than_a_pulse_are_zero_like_the_reference(b2):
    # same geometry with delayMax = 500 <= nfft - nCorr: lags 488..500 exceed the pulse length and are
    # zero (up to rounding) in the reference (zero padding to nfft) and here
    # (the reference's values there are fp64 rounding noise ~1e-13 of the peak, so Map::set_metrics' mean
    # of the dB values is not comparable for such a geometry: the map is what is checked)
    args = (-10, 500, -2048, 2048, 2_000_000, 1_000_000, True)
    x, y = O.synth_iq(args[5], seed=11, fs=args[4], targets=((20, 600.0, 0.1),))
    amb = b2.Ambiguity(*args)
    assert (amb.get_n_doppler_bins(), amb.get_n_corr(), amb.get_nfft()) == (2049, 488, 1000)
    m = amb.process(x, y)
    ref = O.ambiguity_process(O.ambiguity_dims(*args), x, y)
    peak = np.abs(ref).max()
    assert np.abs(m.data.astype(np.complex128) - ref).max() / peak <= 1e-5
    beyond = amb.delay >= 488
    assert beyond.sum() == 13 and np.abs(ref[:, beyond]).max() / peak < 1e-9 and np.abs(m.data[:, beyond]).max() / peak < 1e-5


def test_zero_cells_poison_the_mean_like_the_reference(b2):
    # an all-zero surveillance channel gives |z| = 0 cells: 10*log10(0) = -inf and
    # Map::set_metrics' mean becomes -inf (Map.cpp:187-206 has no guard)
    n, fs = 20_000, 200_000
    amb = b2.Ambiguity(-3, 20, -50, 50, fs, n, True)
    x = np.ones(n, dtype=np.complex128)
    m = amb.process(x, np.zeros(n, dtype=np.complex128))
    assert np.all(m.data == 0)
    assert m.noisePower == -np.inf


def test_two_wave_doppler_tile_kernel(b2):
    # 513 < nD <= 1025 with enough delay tiles to take the multi-wave tile kernel from a single
    # CPI (csrc/kernels.hpp doppler_tilem_kernel): nD = 601 / 1025, ragged last tile (nDelay % 8 != 0)
    for fmax, n, fs, dmax in [(300, 1_000_000, 1_000_000, 1100), (512, 2_000_000, 2_000_000, 1029)]:
        amb = run(b2, (-5, dmax, -fmax, fmax, fs, n, True), seed=fmax)
        assert amb.get_n_doppler_bins() == 2 * fmax + 1


def test_explicit_power_of_two_doppler_bins(b2):
    """Extension of SURVEY.md 8g: exactly 512 / 1024 / 2048 Doppler bins (the reference's
    constructor only yields odd counts); oracle = the NumPy restatement with the same count."""
    for nD, n, fs, dmax in [(512, 2_000_000, 2_000_000, 400), (1024, 2_000_000, 2_000_000, 100), (2048, 4_000_000, 2_000_000, 60),
                            (64, 100_000, 100_000, 30), (2, 50_000, 100_000, 10)]:
        args = (-10, dmax, -(nD // 2), nD // 2, fs, n, True)
        amb = b2.Ambiguity(*args, n_doppler_bins=nD)
        d = O.ambiguity_dims(*args, n_doppler_bins=nD)
        assert (amb.get_n_doppler_bins(), amb.get_n_corr(), amb.get_nfft()) == (nD, d.n_corr, d.nfft) == (nD, n // nD, d.nfft)
        assert np.allclose(amb.doppler, d.doppler, rtol=0, atol=1e-9) and amb.doppler.size == nD
        f_t = float(d.doppler[nD // 4])  # a target exactly on a bin
        x, y = O.synth_iq(n, seed=nD, fs=fs, targets=((7, f_t, 0.1),))
        m = amb.process(x, y)
        ref = O.ambiguity_process(d, x, y)
        err = np.abs(m.data.astype(np.complex128) - ref)
        assert err.max() / np.abs(ref).max() <= 1e-5
        noise, mx = O.map_metrics(ref)
        assert abs(m.noisePower - noise) <= 1e-3 and abs(m.maxPower - mx) <= 1e-3
    with pytest.raises(b2.Blah2HipError):
        b2.Ambiguity(-10, 100, -50, 50, 100_000, 1000, True, n_doppler_bins=70_000)
