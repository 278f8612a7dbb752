"""ctypes binding of include/blah2host.h: the host-side product classes (Map, Detection and the
JSON number formatting) of ``blah2_amd/host`` behind a C ABI, so that the replay driver and the tests
serialise through the same C++ code that drops into blah2.cpp."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libblah2host.so")

_vp, _u32, _dbl, _sz = C.c_void_p, C.c_uint32, C.c_double, C.c_size_t
SYMBOLS = {
    "blah2host_map_json": (C.c_int, [_vp, _u32, _u32, _vp, _vp, _dbl, _dbl, C.c_uint64, _u32, C.c_char_p, _sz, C.POINTER(_sz)]),
    "blah2host_detection_json": (C.c_int, [_vp, _vp, _vp, _u32, C.c_uint64, _u32, C.c_char_p, _sz, C.POINTER(_sz)]),
    "blah2host_format_double": (C.c_int, [_dbl, C.c_int, C.c_char_p, _sz, C.POINTER(_sz)]),
}

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m blah2_amd.build`")
    from . import _lib as hip  # libblah2host.so links libblah2hip.so: same HIP-runtime load order rule
    hip.load()
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _call(fn, *args, guess=1 << 16):
    cap = guess
    for _ in range(2):
        buf = C.create_string_buffer(cap)
        n = C.c_size_t(0)
        rc = fn(*args, buf, cap, C.byref(n))
        if rc == 0:
            return buf.raw[:n.value].decode("ascii")
        if rc != -6:
            raise ValueError(f"blah2host call failed with {rc}")
        cap = n.value + 1
    raise RuntimeError("blah2host: capacity negotiation failed")


def _p(a):
    return C.c_void_p(a.ctypes.data)


def map_json(data, delay, doppler, noise_power, max_power, timestamp, fs=0):
    """Map::to_json (+ delay_bin_to_km when fs > 0), blah2.cpp:304-305."""
    m = np.ascontiguousarray(data, dtype=np.complex64)
    dl = np.ascontiguousarray(delay, dtype=np.int32)
    dp = np.ascontiguousarray(doppler, dtype=np.float64)
    assert m.shape == (dp.size, dl.size)
    return _call(load().blah2host_map_json, _p(m), m.shape[0], m.shape[1], _p(dl), _p(dp), float(noise_power),
                 float(max_power), int(timestamp), int(fs), guess=m.size * 8 + 4096)


def detection_json(delay, doppler, snr, timestamp, fs=0):
    """Detection::to_json (+ delay_bin_to_km when fs > 0), blah2.cpp:315-316."""
    d, f, s = (np.ascontiguousarray(v, dtype=np.float64) for v in (delay, doppler, snr))
    return _call(load().blah2host_detection_json, _p(d), _p(f), _p(s), d.size, int(timestamp), int(fs), guess=d.size * 48 + 256)


def format_double(v, max_decimals=2):
    return _call(load().blah2host_format_double, float(v), int(max_decimals), guess=64)
