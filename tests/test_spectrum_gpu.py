"""GPU parity for SpectrumAnalyser through the C ABI:
the folded nS-point evaluation in csrc/spectrum.hip against the compiled
reference's output (tests/golden) and the NumPy restatement."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_names, load_golden
from oracle import blah2_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-9  # fp64 sums in a different order; |delta| / max|spectrum|


@pytest.fixture(scope="module")
def b2(built_lib):
    import blah2_amd
    assert blah2_amd.device_count() > 0
    return blah2_amd


def rel(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


@pytest.mark.parametrize("name", golden_names())
def test_capture_fixtures(b2, name):
    g = load_golden(name)
    n = int(g["params"][1])
    sa = b2.SpectrumAnalyser(n, 2000)
    assert (sa.decimation, sa.nSpectrum, sa.nfft) == O.spectrum_dims(n, 2000)
    spec, freq = sa.process(g["x"])
    assert freq.size == int(g["spectrum_n_frequency"]) == 0  # the reference's axis loop never runs
    assert spec.shape == g["spectrum"].shape
    assert rel(spec, g["spectrum"]) <= TOL
    # complex64 entry point: int16-valued samples are exact in fp32
    spec32, _ = sa.process(g["x"].astype(np.complex64))
    assert np.array_equal(spec32, spec)


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(os.path.join(GOLDEN, "spectrum"))))
def test_geometry_fixtures(b2, name):
    z = np.load(os.path.join(GOLDEN, "spectrum", name + ".npz"))
    n, bw = int(z["params"][0]), float(z["params"][1])
    x = z["iq"][:, 0].astype(np.float64) + 1j * z["iq"][:, 1].astype(np.float64)
    sa = b2.SpectrumAnalyser(n, bw)
    assert (sa.decimation, sa.nSpectrum, sa.nfft) == O.spectrum_dims(n, bw)
    spec, freq = sa.process(x)
    assert freq.size == int(z["n_frequency"])
    assert rel(spec, z["spectrum"]) <= TOL


def test_baseline_cpi_and_device_batch(b2):
    import torch
    n, fs = 2_000_000, 2_000_000
    sa = b2.SpectrumAnalyser(n, 2000, max_batch=3)
    assert (sa.decimation, sa.nSpectrum, sa.nfft) == (1000, 2000, 2_000_000)
    xs = [O.synth_iq(n, fs=fs, seed=40 + i)[0] for i in range(3)]
    refs = [O.spectrum_process(x, n, 2000)[0] for x in xs]
    got, _ = sa.process(xs[0])
    assert rel(got, refs[0]) <= TOL
    # a tone exactly on a kept bin: X[(k*D + N/2 + 1) mod N] for k = 1234
    k = 1234
    b = (k * 1000 + n // 2 + 1) % n
    tone = np.round(1000 * np.exp(2j * np.pi * b * np.arange(n) / n))
    spec, _ = sa.process(tone)
    assert np.argmax(np.abs(spec)) == k
    assert rel(spec, O.spectrum_process(tone, n, 2000)[0]) <= TOL
    # device entry points: complex64 planes and the interleaved int16 capture layout, batched
    dev = torch.device("cuda", 0)
    xb = torch.from_numpy(np.stack(xs).astype(np.complex64)).to(dev)
    out = torch.zeros((3, sa.nSpectrum), dtype=torch.complex128, device=dev)
    sa.process_dev(b2.FMT_C32, xb.data_ptr(), 3, n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i in range(3):
        assert rel(out[i].cpu().numpy(), refs[i]) <= TOL
    iq = torch.zeros((3, n, 4), dtype=torch.int16, device=dev)
    iq[:, :, 0] = xb.real.to(torch.int16)
    iq[:, :, 1] = xb.imag.to(torch.int16)
    iq[:, :, 2] = 77  # tuner 2 must be ignored
    out.zero_()
    sa.process_dev(b2.FMT_I16, iq.data_ptr(), 3, n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i in range(3):
        assert rel(out[i].cpu().numpy(), refs[i]) <= TOL


@pytest.mark.parametrize("n,bw", [(2_000_000, 10_000), (300_000, 7000), (70_000, 20_000)])
def test_wide_spectra(b2, n, bw):
    """nSpectrum above 4096 (the root tables leave LDS for L2): same sums, same tolerance."""
    rng = np.random.default_rng(n + bw)
    x = np.round(rng.normal(0, 300, n)) + 1j * np.round(rng.normal(0, 300, n))
    t = 500 * np.exp(2j * np.pi * 0.0613 * np.arange(n))
    x += np.round(t.real) + 1j * np.round(t.imag)  # int16-valued like the wire samples: exact in the fp32 upload
    sa = b2.SpectrumAnalyser(n, bw)
    assert (sa.decimation, sa.nSpectrum, sa.nfft) == O.spectrum_dims(n, bw) and sa.nSpectrum > 4096
    spec, _ = sa.process(x)
    assert rel(spec, O.spectrum_process(x, n, bw)[0]) <= TOL


def test_limits(b2):
    with pytest.raises(b2.Blah2HipError):
        b2.SpectrumAnalyser(1000, 2000)       # decimation 0: the reference divides by zero
    # nSpectrum 100000 (round 6: refused above 65536 until the chirp-z path; SpectrumAnalyser.cpp:9-30 has no such bound):
    # a CPI of the headline's size with twenty samples per bin, against the restatement
    n, bw = 2_000_000, 100_000
    x, _ = O.synth_iq(n, fs=2_000_000, seed=29)
    sa = b2.SpectrumAnalyser(n, bw)
    assert (sa.decimation, sa.nSpectrum, sa.nfft) == (20, 100_000, 2_000_000)
    spec, _ = sa.process(x)
    assert rel(spec, O.spectrum_process(x, n, bw)[0]) <= TOL
    sa = b2.SpectrumAnalyser(20_000, 2000)
    with pytest.raises(b2.Blah2HipError):
        sa.process(np.zeros(19_999, dtype=np.complex128))
