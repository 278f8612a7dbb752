"""CPU suite: compiles the device FFT / range-correlation code for the host and
emulates a workgroup thread by thread (tests/host/emulate_fft.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emu") / "emulate_fft")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe,
                           os.path.join(ROOT, "tests", "host", "emulate_fft.cpp")])
    return exe


def test_workgroup_fft_forward_inverse(emu):
    # both transform families: 16 points/thread (16x16xR3) and 8 points/thread (8x8x8xR4)
    out = subprocess.run([emu, "fft"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("E8 R4=") == 3 and out.stdout.count("R3=") == 3
    # the one-wave 2048-point transform (32 points per lane, lane exchange emulated on lane pairs) and the
    # pruned 16- / 32-point kernels of the zero-padded segments
    assert "WAVE F=2048" in out.stdout and "pruned_vs_full_abs_err=0.000e+00" in out.stdout


# (R3, nCorr, nDoppler, delayMin, delayMax, nSeg, segLen)
@pytest.mark.parametrize("case", [
    (8, 1818, 3, -3, 20, 1, 1818),      # one segment
    (4, 1818, 3, -3, 20, 3, 606),       # three segments, F = 1024
    (16, 3898, 2, -10, 400, 2, 1949),   # BASELINE cfg 2 pulse, F = 4096
    (8, 3898, 2, -10, 400, 3, 1300),    # BASELINE cfg 2 pulse, F = 2048 (the planned shape)
    (4, 700, 3, 0, 40, 2, 350),         # delayMin = 0
    (8, 1001, 2, -24, 100, 1, 1001),    # ragged tail, larger negative lag
    (4, 37, 4, -1, 1, 1, 37),           # tiny pulse, minimum lag span the reference allows
])
def test_segmented_range_correlation(emu, case):
    out = subprocess.run([emu, "range", *[str(v) for v in case], "5"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
