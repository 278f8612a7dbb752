"""IqData's fp32 shadow (the eager upload of the drop-in classes, blah2_amd/host/data/IqData.h) against a model of the
device ring: tests/host/test_iqdata_shadow.cpp, GPU-free."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shadow_follows_the_fifo(tmp_path):
    exe = str(tmp_path / "test_iqdata_shadow")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "blah2_amd", "host"), "-o", exe,
                    os.path.join(ROOT, "tests", "host", "test_iqdata_shadow.cpp"),
                    os.path.join(ROOT, "blah2_amd", "host", "data", "IqData.cpp")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
