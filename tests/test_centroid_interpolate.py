"""CPU suite: Centroid and Interpolate are host arithmetic behind the C ABI, so
they are checked here (no GPU) against the compiled reference's lists in
tests/golden, feeding them the reference map rounded to fp32 (what the GPU map
holds)."""
import numpy as np
import pytest

from conftest import golden_names, load_golden


def objs(g):
    import blah2_amd
    m = blah2_amd.Map(None, g["map"].astype(np.complex64), g["delay"].astype(np.int32), g["doppler"],
                      float(g["metrics"][0]), float(g["metrics"][1]))
    det = blah2_amd.Detection(*g["cfar"])
    return blah2_amd, m, det


@pytest.mark.parametrize("name", golden_names())
def test_centroid_then_interpolate_golden(built_lib, name):
    g = load_golden(name)
    b2, m, det = objs(g)
    nc, res = int(g["det_params"][5]), float(g["det_params"][6])
    c = b2.Centroid(nc, nc, res).process(det)
    assert np.array_equal(c.get_delay(), g["centroid"][0])
    assert np.array_equal(c.get_doppler(), g["centroid"][1])
    assert np.array_equal(c.get_snr(), g["centroid"][2])
    i = b2.Interpolate(True, True).process(c, m)
    assert i.get_nDetections() == g["interp"].shape[1]
    # interpolation of fp32-rounded cells: sub-bin offsets agree to 1e-4 bins / 1e-4 dB
    assert np.allclose(i.get_delay(), g["interp"][0], rtol=0, atol=1e-4)
    assert np.allclose(i.get_doppler(), g["interp"][1], rtol=0, atol=1e-3)
    assert np.allclose(i.get_snr(), g["interp"][2], rtol=0, atol=1e-4)


def test_centroid_suppresses_weaker_neighbours(built_lib):
    import blah2_amd
    det = blah2_amd.Detection([20, 22, 40, 3], [10.0, 12.0, 10.0, 0.0], [5.0, 9.0, 4.0, 1.0])
    out = blah2_amd.Centroid(6, 6, 2.0).process(det)
    # (20,10) is inside (22,12)'s box and weaker -> dropped; (3,0): 3-6 wraps in uint16, so it is never suppressed
    assert out.get_delay().tolist() == [22.0, 40.0, 3.0]


def test_interpolate_drops_edges_and_non_peaks(built_lib):
    import blah2_amd
    nD, nC = 5, 7
    z = np.ones((nD, nC), dtype=np.complex64)
    z[2, 3] = 100.0  # a clean peak
    m = blah2_amd.Map(None, z, np.arange(-1, nC - 1, dtype=np.int32), np.linspace(-2, 2, nD), 0.0, 0.0)
    det = blah2_amd.Detection([2.0, -1.0, 4.0], [0.0, 0.0, 1.0], [20.0, 0.0, 0.0])  # peak, delay edge, flat cell
    out = blah2_amd.Interpolate(True, True).process(det, m)
    # the peak is symmetric -> no offset; the edge detection is dropped (:46-49); a flat cell gives 0/0
    assert out.get_delay()[0] == 2.0 and out.get_doppler()[0] == 0.0
    assert out.get_nDetections() >= 1
