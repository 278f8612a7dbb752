"""CPU: the JSON products of the host classes (blah2_amd/host, through include/blah2host.h).

rapidjson (lib/vcpkg.json: 1.1.0) is not installed, so the byte-level pin is by HAND-DERIVED
vectors: each expected string below was worked out from rapidjson 1.1.0's published number
formatting -- ``Writer::WriteDouble`` -> ``dtoa(value, buffer, maxDecimalPlaces)`` -> Grisu2
shortest digits -> ``Prettify(buffer, length, k, maxDecimalPlaces)`` -- whose own comments name the
cases ("When maxDecimalPlaces = 2, 1.2345 -> 1.23, 1.102 -> 1.1", "0.123 -> 0.12, 0.102 -> 0.1",
"Truncate to zero", "1e30", "1234e30 -> 1.234e33").  With digits d (length L) and exponent k,
kk = L + k:
  0 <= k, kk <= 21        digits, zeros up to kk, then ".0"                    12 -> "12.0"
  0 < kk <= 21            point after kk digits; if k + 2 < 0 TRUNCATE to 2 decimals, then strip
                          trailing zeros but keep one decimal                  2.995 -> "2.99"
  -6 < kk <= 0            "0." + zeros + digits, truncated the same way        0.019 -> "0.01", 0.001 -> "0.0"
  kk < -2                 "0.0"                                                1e-7 -> "0.0"
  otherwise               exponent form                                        1e21 -> "1e21"
dtoa() itself prints zero as "0.0" ("-0.0" when the sign bit is set) and a leading '-' for negatives.
"""
import json
import os
import shutil
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden
from oracle import blah2_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRETTIFY_VECTORS = [
    (1.2345, "1.23"), (1.102, "1.1"),            # the two cases rapidjson's own comment names
    (0.123, "0.12"), (0.102, "0.1"),             # the -6 < kk <= 0 branch's comment
    (0.001, "0.0"), (0.009, "0.0"), (0.019, "0.01"), (0.5, "0.5"), (0.25, "0.25"),
    (1e-7, "0.0"), (-1e-7, "-0.0"),              # "Truncate to zero" keeps the sign dtoa already wrote
    (2.995, "2.99"), (-2.995, "-2.99"), (9.999, "9.99"), (0.995, "0.99"),  # truncation, never rounding
    (12.0, "12.0"), (100.5, "100.5"), (-37.456, "-37.45"), (3.0, "3.0"), (1500.0, "1500.0"),
    (10.0, "10.0"), (10.004, "10.0"), (10.01, "10.01"), (10.10, "10.1"),
    (0.0, "0.0"), (-0.0, "-0.0"),
    (1e20, "100000000000000000000.0"), (1e21, "1e21"), (1.5e22, "1.5e22"),
    (1.234e33, "1.234e33"), (1.5e300, "1.5e300"),  # "1234e30 -> 1.234e33"
    (-63.0, "-63.0"), (76.918, "76.91"), (30.2816, "30.28"),  # the reference test's metrics (TestAmbiguity.cpp:176-177)
    (0.14989622900000001, "0.14"),               # one delay bin in km at 2 MS/s: c/fs/1000
]


@pytest.fixture(scope="module")
def H(built_lib):
    from blah2_amd import _hostlib
    _hostlib.load()
    return _hostlib


def test_header_symbols_are_exported_and_bound(H):
    import re
    src = open(os.path.join(ROOT, "include", "blah2host.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(blah2host_[a-z0-9_]+)\s*\(", src)))
    assert names == sorted(H.SYMBOLS) and len(names) == 3
    for n in names:
        assert getattr(H.load(), n)


@pytest.mark.parametrize("value,text", PRETTIFY_VECTORS)
def test_double_formatting_hand_derived_vectors(H, value, text):
    assert H.format_double(value, 2) == text


def test_double_formatting_matches_truncation_rule_on_random_values(H):
    """Property: for 1e-2 <= |v| < 1e15 the printed value is v truncated (toward zero) to two decimals."""
    rng = np.random.default_rng(5)
    for v in np.concatenate([rng.uniform(-200, 200, 2000), rng.uniform(-1, 1, 500), 10 ** rng.uniform(-2, 12, 500)]):
        s = H.format_double(float(v), 2)
        from decimal import ROUND_DOWN, Decimal
        want = Decimal(repr(float(v))).quantize(Decimal("0.01"), rounding=ROUND_DOWN)  # repr = shortest round-trip digits
        assert Decimal(s) == want, (v, s)
        frac = s.split(".")[1]
        assert 1 <= len(frac) <= 2 and (len(frac) == 1 or frac[-1] != "0")


def test_map_json_document(H):
    """Map::to_json + delay_bin_to_km on the compiled reference's `medium` map: field order of
    Map.cpp:148-155, two-decimal truncated cells equal to the oracle's Map::to_json values."""
    g = load_golden("medium")
    fs = int(g["params"][0])
    m, noise, mx = g["map"], float(g["metrics"][0]), float(g["metrics"][1])
    doc = H.map_json(m.astype(np.complex64), g["delay"], g["doppler"], noise, mx, 1702595171000, fs)
    assert doc.startswith('{"timestamp":1702595171000,"nRows":%d,"nCols":%d,"noisePower":' % m.shape)
    assert doc.endswith("]]}") and "\n" not in doc and " " not in doc
    d = json.loads(doc)
    assert list(d) == ["timestamp", "nRows", "nCols", "noisePower", "maxPower", "delay", "doppler", "data"]
    assert (d["nRows"], d["nCols"]) == m.shape == (len(d["data"]), len(d["data"][0]))
    assert d["doppler"] == [float(H.format_double(v)) for v in g["doppler"]]
    km = g["delay"].astype(np.float64) * (O.C_LIGHT / fs) / 1000  # Map.cpp:175
    assert d["delay"] == [float(H.format_double(v)) for v in km]
    want = O.map_db(m.astype(np.complex64).astype(np.complex128), noise)
    got = np.array(d["data"])
    # truncation toward zero at two decimals: |printed| <= |value| < |printed| + 0.01
    assert np.all(np.abs(got) <= np.abs(want) + 1e-9) and np.all(np.abs(want) - np.abs(got) < 0.01 + 1e-9)
    assert np.all((np.sign(got) == np.sign(want)) | (got == 0))
    # without fs the delay axis stays in bins (Map::to_json alone)
    assert json.loads(H.map_json(m.astype(np.complex64), g["delay"], g["doppler"], noise, mx, 1, 0))["delay"] == g["delay"].tolist()


def test_detection_json_document(H):
    g = load_golden("medium")
    fs = int(g["params"][0])
    dl, dp, sn = g["cfar"]
    doc = H.detection_json(dl, dp, sn, 42, fs)
    d = json.loads(doc)
    assert list(d) == ["timestamp", "delay", "doppler", "snr"] and d["timestamp"] == 42  # Detection.cpp:47-85
    assert d["doppler"] == [float(H.format_double(v)) for v in dp]
    assert d["snr"] == [float(H.format_double(v)) for v in sn]
    assert d["delay"] == [float(H.format_double(v * (O.C_LIGHT / fs) / 1000)) for v in dl]
    assert H.detection_json([], [], [], 7, fs) == '{"timestamp":7,"delay":[],"doppler":[],"snr":[]}'


@pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")
def test_node_api_listener_accepts_our_frames(H):
    """The documents, sent like Socket::sendData does (1024-byte writes, no terminator), are accepted by
    the Node API's TCP listener rule (api/server.js:123-136) and parse in Node."""
    from blah2_amd import replay as R
    g = load_golden("medium")
    fs = int(g["params"][0])
    docs = [H.map_json(g["map"].astype(np.complex64), g["delay"], g["doppler"], float(g["metrics"][0]),
                       float(g["metrics"][1]), 1000 * k, fs) for k in range(3)]
    docs.append(H.detection_json(*g["cfar"], 3000, fs))
    p = subprocess.Popen(["node", os.path.join(ROOT, "tests", "host", "frame_consumer.js"), str(len(docs))],
                         stdout=subprocess.PIPE, text=True)
    try:
        port = int(p.stdout.readline().split()[1])
        with socket.create_connection(("127.0.0.1", port)) as s:
            for doc in docs:
                assert len(doc) > R.MTU or "snr" in doc
                R.send_frame(s, doc)
                line = json.loads(p.stdout.readline())  # the listener completed a frame: wait for it before the next
                assert line["bytes"] == len(doc)
                if "nRows" in line:
                    assert (line["rows"], line["cols"]) == g["map"].shape == (line["nRows"], line["nCols"])
                    assert line["keys"] == ["timestamp", "nRows", "nCols", "noisePower", "maxPower", "delay", "doppler", "data"]
                    assert abs(line["dataMax"] - float(g["metrics"][1])) < 0.011  # max cell = maxPower, truncated
                else:
                    assert line["nDetections"] == len(g["cfar"][0])
        assert p.wait(timeout=20) == 0
    finally:
        if p.poll() is None:
            p.kill()
