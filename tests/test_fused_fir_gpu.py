"""GPU: the clutter filter's FIR fused into the range correlation (range_fir_kernel, csrc/kernels.hpp;
WienerHopf.cpp:124-160 feeding Ambiguity.cpp:106-149 without the filtered channel crossing HBM).

The fused chain -- WienerHopf.estimate_dev_fmt (correlations, reduction, solve) + Ambiguity.set_fir + process_dev on the
UNFILTERED channels -- against the two-stage chain (same taps: bit-identical filter estimate) and against the oracle's
chain in fp64, over the geometries that exercise its edges: pulses that end inside a block, on a block boundary and within
|delayMin| of one (a sixth, nearly empty block), the CPI's first pulse (the filter's stream is zero on its first |delayMin|
samples), the last pulse's look-ahead, both sample formats, a batch, a failed solve (all-zero reference: taps zero, the
surveillance channel passes through), and the refusals.
"""
import numpy as np
import pytest

from gates import map_cell_gate, notch_mask
from oracle import blah2_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def b2(built_lib):
    import blah2_amd
    assert blah2_amd.device_count() > 0
    return blah2_amd


def synth(n, fs, seed, direct=0.8, echo=(37, -3.0, 0.05), noise=30.0, integer=True):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 300.0
    t = np.arange(n) / fs
    xd = np.roll(x, echo[0])
    xd[:echo[0]] = 0
    y = direct * x + echo[2] * xd * np.exp(2j * np.pi * echo[1] * t) + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    if integer:
        x = np.round(x.real) + 1j * np.round(x.imag)
        y = np.round(y.real) + 1j * np.round(y.imag)
    return x.astype(np.complex64), y.astype(np.complex64)


def run_chains(b2, args, xs, ys, fmt="c32", stride=None):
    """(two-stage maps, fused maps, ok flags of both) for a batch [B][n] of complex64 channels."""
    import torch
    dmin, dmax, fmin, fmax, fs, n = args
    B = xs.shape[0]
    stride = stride or n
    dev = torch.device("cuda", 0)
    xb = np.zeros((B, stride), dtype=np.complex64)
    yb = np.zeros((B, stride), dtype=np.complex64)
    xb[:, :n], yb[:, :n] = xs, ys
    amb = b2.Ambiguity(dmin, dmax, fmin, fmax, fs, n, True, max_batch=B)
    amb.set_fft_len(4096)
    wh = b2.WienerHopf(dmin, dmax, n, max_batch=B)
    nD, nC = amb.get_n_doppler_bins(), amb.get_n_delay_bins()
    st = torch.cuda.current_stream().cuda_stream
    out = [torch.zeros((B, nD, nC), dtype=torch.complex64, device=dev) for _ in range(2)]
    met = torch.zeros((B, 2), dtype=torch.float64, device=dev)
    ok = [torch.full((B,), -1, dtype=torch.int32, device=dev) for _ in range(2)]
    if fmt == "c32":
        x, y = torch.from_numpy(xb).to(dev), torch.from_numpy(yb).to(dev)
        yf = torch.empty_like(y)
        wh.process_dev_fmt(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, stride, yf.data_ptr(), stride, ok[0].data_ptr(), st)
        amb.process_dev(b2.FMT_C32, x.data_ptr(), yf.data_ptr(), B, stride, out[0].data_ptr(), met.data_ptr(), st)
        assert amb.fir_fusable(wh, b2.FMT_C32) is None
        amb.set_fir(wh)
        wh.estimate_dev_fmt(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, stride, ok[1].data_ptr(), st)
        amb.process_dev(b2.FMT_C32, x.data_ptr(), y.data_ptr(), B, stride, out[1].data_ptr(), met.data_ptr(), st)
    else:
        iq = np.stack([xb.real, xb.imag, yb.real, yb.imag], axis=-1).astype(np.int16)
        d_iq = torch.from_numpy(iq).to(dev)
        yf = torch.empty((B, stride), dtype=torch.complex64, device=dev)
        wh.process_dev_fmt(b2.FMT_I16, d_iq.data_ptr(), None, B, stride, yf.data_ptr(), stride, ok[0].data_ptr(), st)
        amb.process_dev(b2.FMT_I16X_C32Y, d_iq.data_ptr(), yf.data_ptr(), B, stride, out[0].data_ptr(), met.data_ptr(), st)
        assert amb.fir_fusable(wh, b2.FMT_I16) is None
        amb.set_fir(wh)
        wh.estimate_dev_fmt(b2.FMT_I16, d_iq.data_ptr(), None, B, stride, ok[1].data_ptr(), st)
        amb.process_dev(b2.FMT_I16, d_iq.data_ptr(), None, B, stride, out[1].data_ptr(), met.data_ptr(), st)
    torch.cuda.synchronize()
    assert amb.info(b2._lib.INFO_LAST_RANGE_KERNEL) == b2._lib.RANGE_FIR
    res = out[0].cpu().numpy(), out[1].cpu().numpy(), ok[0].cpu().numpy(), ok[1].cpu().numpy()
    amb.close()
    wh.close()
    return res


def check(b2, args, seed, fmt="c32", **kw):
    dmin, dmax, fmin, fmax, fs, n = args
    x, y = synth(n, fs, seed, **kw)
    two, fus, ok2, okf = run_chains(b2, args, x[None], y[None], fmt)
    assert ok2[0] == 1 and okf[0] == 1
    xh, yh = x.astype(np.complex128), y.astype(np.complex128)
    okr, yfr = O.wiener_hopf(xh, yh, dmin, dmax)[:2]
    assert okr
    d = O.ambiguity_dims(dmin, dmax, fmin, fmax, fs, n, True)
    ref = O.ambiguity_process(d, xh, yfr)
    noise = O.map_metrics(ref)[0]
    nm = notch_mask(ref.shape, d.doppler, d.delay, dmin, dmax)
    lvl = 10.0 ** (noise / 10.0)
    g2, gf = map_cell_gate(two[0], ref, noise, notch=nm), map_cell_gate(fus[0], ref, noise, notch=nm)
    diff = float(np.abs(fus[0].astype(np.complex128) - two[0]).max() / lvl)
    print(f"\n[fused {fmt}] {ref.shape[0]} pulses of {d.n_corr} samples, {dmax - dmin} taps: "
          f"cell-rel outside the notch two-stage {g2['cell_rel_above_mean_outside_notch']:.2e} fused {gf['cell_rel_above_mean_outside_notch']:.2e}; "
          f"fused - two-stage {diff:.2e} of the mean level")
    assert g2["ok"] and gf["ok"], (g2, gf)
    assert gf["cell_rel_above_mean_outside_notch"] <= max(2.0 * g2["cell_rel_above_mean_outside_notch"], 2e-5)
    # the notch cells: the same taps, so the same cancellation residue to within the transforms' rounding
    assert diff <= 2e-3, diff
    return two[0], fus[0]


# (delayMin, delayMax, dopplerMin, dopplerMax, fs, n) with fs = n (a 1 s CPI): nD = dopplerMax - dopplerMin + 1 pulses of
# nCorr = n // nD samples, and n - nD nCorr >= |delayMin| samples behind the last pulse for the filter's look-ahead
GEOMETRIES = {
    "ragged last block (nCorr 9523)": (-8, 1200, -10, 10, 199_993, 199_993),
    "pulse ends on a block boundary (nCorr 4096: a third block of |delayMin| samples)": (-16, 900, -12, 12, 102_420, 102_420),
    "pulse ends 5 samples past a boundary (nCorr 6149)": (-24, 2023, -15, 15, 190_647, 190_647),
    "pulse ends 5 samples short of a boundary (nCorr 6139)": (-24, 2023, -15, 15, 190_337, 190_337),
    "the shortest pulse the kernel takes (nCorr 2056 = L - delayMin)": (-8, 300, -20, 20, 84_308, 84_308),
    "the longest filter and lag window (2048 taps, 2049 delay bins)": (0, 2048, -5, 5, 110_003, 110_003),
}


@pytest.mark.parametrize("name", list(GEOMETRIES))
def test_fused_chain_matches_the_two_stage_chain_and_the_oracle(b2, name):
    args = GEOMETRIES[name]
    check(b2, args, seed=21)


def test_int16_words(b2):
    check(b2, GEOMETRIES["ragged last block (nCorr 9523)"], seed=22, fmt="i16")
    check(b2, GEOMETRIES["pulse ends 5 samples past a boundary (nCorr 6149)"], seed=23, fmt="i16")


def test_a_batch_with_a_stride_and_a_failed_solve(b2):
    """CPI 1's reference channel is all zero: WienerHopf.cpp:107-112 gives up (ok = 0), the taps are zero and the surveillance
    channel passes through -- in the fused kernel exactly as in the two-stage chain (the map of y against a zero x: zeros)."""
    args = GEOMETRIES["ragged last block (nCorr 9523)"]
    n, fs = args[5], args[4]
    data = [synth(n, fs, 30 + c) for c in range(3)]
    xs = np.stack([v[0] for v in data])
    ys = np.stack([v[1] for v in data])
    xs[1] = 0
    two, fus, ok2, okf = run_chains(b2, args, xs, ys, stride=n + 4096)
    assert list(ok2) == [1, 0, 1] and list(okf) == [1, 0, 1]
    assert not fus[1].any() and not two[1].any()
    for c in (0, 2):
        scale = np.abs(two[c]).max()
        assert np.abs(fus[c].astype(np.complex128) - two[c]).max() <= 2e-6 * scale


def test_refusals(b2):
    # (set_fft_len, args, filter window, reason)
    a = b2.Ambiguity(-8, 1200, -10, 10, 200_000, 200_000, True)
    w_ok = b2.WienerHopf(-8, 1200, 200_000)
    a.set_fft_len(2048)
    assert "4096" in a.fir_fusable(w_ok, b2.FMT_C32)
    a.set_fft_len(4096)
    assert a.fir_fusable(w_ok, b2.FMT_C32) is None and a.fir_fusable(w_ok, b2.FMT_I16) is None
    assert "fp32 planes or int16" in a.fir_fusable(w_ok, b2.FMT_F16)
    w_shift = b2.WienerHopf(-4, 1200, 200_000)
    assert "first lag" in a.fir_fusable(w_shift, b2.FMT_C32)
    # a process call with an unfusable filter fails loudly instead of filtering wrongly
    import torch
    x = torch.zeros(200_000, dtype=torch.complex64, device="cuda")
    a.set_fir(w_shift)
    with pytest.raises(b2.Blah2HipError):
        a.process_dev(b2.FMT_C32, x.data_ptr(), x.data_ptr(), 1, 200_000)
    a.set_fir(None)
    a.process_dev(b2.FMT_C32, x.data_ptr(), x.data_ptr(), 1, 200_000)
    torch.cuda.synchronize()
    for h in (a, w_ok, w_shift):
        h.close()
    b = b2.Ambiguity(-8, 300, -100, 100, 200_000, 200_000, True)  # 201 pulses of 995 samples
    b.set_fft_len(4096)
    w = b2.WienerHopf(-8, 300, 200_000)
    assert "shorter" in b.fir_fusable(w, b2.FMT_C32)
    b.close()
    w.close()
    c = b2.Ambiguity(1, 1200, -10, 10, 200_000, 200_000, True)
    c.set_fft_len(4096)
    w = b2.WienerHopf(1, 1200, 200_000)
    assert "<= 0" in c.fir_fusable(w, b2.FMT_C32)
    c.close()
    w.close()


def test_fused_chain_is_deterministic_and_blind_to_the_batch_position(b2):
    """The same CPI alone, twice: bit-identical maps (fixed-order reductions in the filter's estimate, no atomics in the fused
    kernel, the hot-column rewrite included: the echo below is one).  The same CPI as CPI 0 and CPI 2 of a batch of three:
    bit-identical to each other; against the lone run only to rounding -- the estimate cuts a CPI into a number of partial
    sums that depends on how many CPIs share the launch (csrc/clutter.hip launch_clutter), so the fp32 partials group
    differently."""
    args = GEOMETRIES["pulse ends 5 samples past a boundary (nCorr 6149)"]
    n, fs = args[5], args[4]
    x, y = synth(n, fs, 51, echo=(37, -3.0, 0.2), noise=3.0)
    xo, yo = synth(n, fs, 52)
    _, a1, _, _ = run_chains(b2, args, x[None], y[None])
    _, a2, _, _ = run_chains(b2, args, x[None], y[None])
    assert np.array_equal(a1, a2)
    _, b3, _, ok3 = run_chains(b2, args, np.stack([x, xo, x]), np.stack([y, yo, y]))
    assert list(ok3) == [1, 1, 1]
    assert np.array_equal(b3[0], b3[2]) and not np.array_equal(b3[1], b3[0])
    assert np.max(np.abs(b3[0].astype(np.complex128) - a1[0])) <= 1e-5 * np.max(np.abs(a1[0]))
