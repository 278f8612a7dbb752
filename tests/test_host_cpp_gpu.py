"""Runs the C++ test of the drop-in host classes (blah2_amd/host/test/
test_ambiguity.cpp, modelled on the reference's TestAmbiguity.cpp) on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_classes(built_lib):
    exe = os.path.join(ROOT, "blah2_amd", "host", "test", "test_ambiguity")
    assert os.path.exists(exe), "host test program was not built"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout
