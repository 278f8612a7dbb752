"""Runs the C++ tests of the drop-in host classes on the GPU: blah2_amd/host/test/test_ambiguity.cpp (modelled on the
reference's TestAmbiguity.cpp) and test_golden.cpp -- the class surface a blah2 maintainer links (IqData::push_back,
SpectrumAnalyser / WienerHopf / Ambiguity::process(IqData*, IqData*), Map::set_metrics, CfarDetector1D, Centroid,
Interpolate) against the compiled reference's values of tests/golden/*.npz, on the per-CPI path, the eager pinned-shadow
path over three CPIs of one FIFO, with a wrapped ring front and with a reader of y between filter and map."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_classes(built_lib):
    exe = os.path.join(ROOT, "blah2_amd", "host", "test", "test_ambiguity")
    assert os.path.exists(exe), "host test program was not built"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout


def write_flat(g, path):
    """One fixture as the flat little-endian record test_golden.cpp reads (every array: int64 count, then the items)."""
    def arr(f, a, dtype):
        a = np.ascontiguousarray(np.asarray(a).astype(dtype, copy=False)).ravel()
        n_items = a.size if a.dtype != np.complex128 else a.size
        f.write(struct.pack("<q", n_items))
        f.write(a.tobytes())

    def dets(f, d):
        for k in range(3):
            arr(f, d[k], np.float64)

    with open(path, "wb") as f:
        f.write(b"B2GOLD01")
        f.write(struct.pack("<7q", *(int(v) for v in g["params"])))
        f.write(struct.pack("<4q", *(int(v) for v in g["dims"])))
        f.write(struct.pack("<2d", float(g["cpi"]), float(g["doppler_middle"])))
        arr(f, g["iq"], np.int16)
        arr(f, g["delay"], np.float64)
        arr(f, g["doppler"], np.float64)
        arr(f, g["map"], np.complex128)
        f.write(struct.pack("<2d", *(float(v) for v in g["metrics"])))
        f.write(struct.pack("<7d", *(float(v) for v in g["det_params"])))
        dets(f, g["cfar"]); dets(f, g["centroid"]); dets(f, g["interp"])
        f.write(struct.pack("<3q", int(g["clutter_params"][0]), int(g["clutter_params"][1]), int(bool(g["clutter_ok"]))))
        arr(f, g["clutter_y"], np.complex128)
        arr(f, g["chain_map"], np.complex128)
        f.write(struct.pack("<2d", *(float(v) for v in g["chain_metrics"])))
        dets(f, g["chain_cfar"])
        arr(f, g["spectrum"], np.complex128)
        f.write(struct.pack("<q", int(g["spectrum_n_frequency"])))


@pytest.mark.parametrize("name", golden_names())
def test_cpp_classes_against_the_compiled_reference(built_lib, tmp_path, name):
    exe = os.path.join(ROOT, "blah2_amd", "host", "test", "test_golden")
    assert os.path.exists(exe), "host golden test program was not built"
    path = str(tmp_path / (name + ".bin"))
    write_flat(load_golden(name), path)
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    assert "OK" in out.stdout


def test_cpp_golden_check_can_fail(built_lib, tmp_path):
    """The same program on a fixture whose reference map is off by 2e-4: it must report the cells and exit non-zero."""
    exe = os.path.join(ROOT, "blah2_amd", "host", "test", "test_golden")
    g = load_golden("small_sym")
    g["map"] = g["map"] * (1.0 + 2e-4)
    path = str(tmp_path / "small_sym_off.bin")
    write_flat(g, path)
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=900)
    assert out.returncode == 1 and "CHECK FAILED" in out.stdout and "relmax" in out.stdout
