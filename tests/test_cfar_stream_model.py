"""CPU: the wave-level NumPy model of cfar2d_stream_kernel (tools/proto/cfar_stream_model.py: strips with halo lanes, row
segments, pair sums at fixed lane offsets, ring delays, clipped counts) against the oracle's 2-D CA-CFAR on random maps --
every window shape the kernel is instantiated for (C2S_SHAPES in csrc/cfar_kernels.hpp) and a few it is not, maps that are
ragged in strips and segments, thresholds low enough that hundreds of cells fire, cells excluded by minDelay / minDoppler."""
import re

import numpy as np
import pytest

from oracle import blah2_oracle as O
from tools.proto import cfar_stream_model as M


def shapes_in_header():
    src = open("blah2_amd/csrc/cfar_kernels.hpp").read()
    body = src[src.index("#define C2S_SHAPES(X)"):]
    body = body[:body.index("\n\n")] if "\n\n" in body else body
    return sorted({tuple(int(v) for v in g) for g in re.findall(r"X\((\d+), (\d+), (\d+), (\d+)\)", body)})


def test_the_header_lists_the_shapes_this_test_covers():
    assert (6, 2, 3, 1) in shapes_in_header() and len(shapes_in_header()) >= 8


@pytest.mark.parametrize("shape", shapes_in_header() + [(7, 0, 2, 2), (2, 5, 0, 1), (1, 1, 1, 1)])
@pytest.mark.parametrize("nD,nC,rows_per_seg", [(45, 111, 8), (33, 50, 32), (70, 129, 16)])
def test_stream_model_equals_the_oracle(shape, nD, nC, rows_per_seg):
    ntd, ngd, ntf, ngf = shape
    rng = np.random.default_rng(nD * 1000 + nC + sum(shape))
    m = (rng.standard_normal((nD, nC)) + 1j * rng.standard_normal((nD, nC))) * np.exp(rng.standard_normal((nD, nC)))
    m[:, 0] *= 50.0  # delay column 0: a cell under test that never trains
    delay = np.arange(nC) - 3
    doppler = np.linspace(-20.0, 20.0, nD)
    pfa, min_delay, min_doppler = 0.05, -1, 1.5
    got = M.cfar2d_stream(m, delay, doppler, pfa, ngd, ntd, ngf, ntf, min_delay, min_doppler, rows_per_seg)
    dl, dp, _, margin = O.cfar2d(m, delay, doppler, 0.0, pfa, ngd, ntd, ngf, ntf, min_delay, min_doppler, return_margin=True)
    row = {f: i for i, f in enumerate(doppler)}
    ref = sorted((row[f], int(d - delay[0])) for d, f in zip(dl, dp))
    assert len(ref) > 20
    for i, j in set(ref) ^ set(got):  # summation order only
        assert abs(margin[i, j] - 1) < 1e-9, (i, j, margin[i, j])
    assert len(set(got)) == len(got)  # no cell reported by two strips or two segments


def test_ring_length_holds_a_block_and_the_rows_behind_it():
    for ntf in range(0, 7):
        for ngf in range(0, 3):
            for v in (1, 2, 4):
                U = M.ring_len(ntf, ngf, v)
                assert U % v == 0 and U >= v + ntf + 2 * ngf + 1 and U < v + ntf + 2 * ngf + 1 + v
