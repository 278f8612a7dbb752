"""CPU suite: the C-ABI library builds, loads, and exports every symbol that
include/blah2hip.h declares.  No compute calls (there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "blah2hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(blah2hip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(built_lib):
    import ctypes
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert isinstance(getattr(built_lib, n), ctypes._CFuncPtr), n


def test_binding_covers_header():
    from blah2_amd import _lib
    assert sorted(_lib.SYMBOLS) == declared_symbols()


def test_host_only_helpers(built_lib):
    import blah2_amd
    assert b"gfx950" in built_lib.blah2hip_version()
    # HammingNumber.cpp:38-48 is integer host arithmetic; TestHammingNumber.cpp:15-17
    assert [blah2_amd.next_hamming(v) for v in (104, 3322, 19043)] == [108, 3375, 19200]


def test_fails_loudly_without_gpu(built_lib):
    import blah2_amd
    if blah2_amd.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(blah2_amd.Blah2HipError) as e:
        blah2_amd.Ambiguity(-10, 300, -300, 300, 2_000_000, 1_000_000)
    assert e.value.code == -5  # BLAH2HIP_ERR_NO_DEVICE: no CPU fallback exists


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "blah2_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(base, f), errors="replace").read()
                assert "from oracle" not in txt and "import oracle" not in txt, f


def test_python_constants_mirror_the_header():
    """Every BLAH2HIP_* option / kernel / format code that blah2_amd/_lib.py names has the header's value (the codes
    are what crosses the ABI; a renumbering on one side only would select the wrong kernel silently)."""
    from blah2_amd import _lib
    src = open(os.path.join(ROOT, "include", "blah2hip.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^#define\s+BLAH2HIP_([A-Z0-9_]+)\s+(-?(?:0x[0-9a-fA-F]+|\d+))\b", src, re.M)}
    assert len(defs) >= 40
    checked = 0
    for name, value in vars(_lib).items():
        if name.isupper() and isinstance(value, int) and name in defs:
            assert defs[name] == value, (name, defs[name], value)
            checked += 1
    assert checked >= 30
    for must in ("RANGE_WAVE1K", "DOP_TILE16WG", "OPT_FFT_LEN", "FMT_I16X_C32Y", "CLUTTER_OPT_CORR"):
        assert must in defs and getattr(_lib, must) == defs[must], must
