"""GPU: the kernel variants that are not the default for a geometry stay correct.
The library reads its BLAH2HIP_* switches once per process, so each variant runs
the golden-fixture parity tests in a fresh interpreter."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = {
    "clamped loads instead of raw buffer loads": ({"BLAH2HIP_RANGE_BUF": "0"}, ["tests/test_ambiguity_gpu.py", "tests/test_edge_cases_gpu.py"]),
    "8-point-per-thread range kernel for every transform length": ({"BLAH2HIP_RANGE_E8": "1"}, ["tests/test_ambiguity_gpu.py", "tests/test_edge_cases_gpu.py"]),
    "16-point-per-thread range kernel for F = 1024": ({"BLAH2HIP_RANGE_E8": "0"}, ["tests/test_ambiguity_gpu.py"]),
    "sequential (non-interleaved) x/y transforms": ({"BLAH2HIP_RANGE_ILV": "0"}, ["tests/test_ambiguity_gpu.py"]),
    "16-column Doppler tiles": ({"BLAH2HIP_DOPPLER_TILE": "16"}, ["tests/test_ambiguity_gpu.py"]),
    "per-column Doppler kernel only": ({"BLAH2HIP_DOPPLER_TILE": "0", "BLAH2HIP_DOPPLER_TILEM": "0"}, ["tests/test_ambiguity_gpu.py", "tests/test_edge_cases_gpu.py"]),
    "16-wave Toeplitz solve": ({"BLAH2HIP_SOLVE_WAVES": "16"}, ["tests/test_clutter_gpu.py"]),
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_variant(name, built_lib):
    env_extra, files = VARIANTS[name]
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider", *files],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"{name}: {env_extra}\n{r.stdout[-3000:]}\n{r.stderr[-1000:]}"
