"""TEST INFRASTRUCTURE ONLY -- ctypes loader for ``oracle/_ref/libblah2ref.so``.

That library is the reference's OWN sources (Ambiguity.cpp, HammingNumber.cpp,
WienerHopf.cpp, CfarDetector1D.cpp, Centroid.cpp, Interpolate.cpp), compiled
where they lie under /root/reference/src by ``oracle/Makefile`` against the
FFTW / Armadillo shims in ``oracle/shim``.  It is the strongest oracle this
image allows (FFTW, Armadillo and rapidjson are not installed).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libblah2ref.so")

_dp = C.POINTER(C.c_double)


def available() -> bool:
    return os.path.exists(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.ref_next_hamming.restype = C.c_uint32
        L.ref_next_hamming.argtypes = [C.c_uint32]
        L.ref_amb_create.restype = C.c_void_p
        L.ref_amb_create.argtypes = [C.c_int32] * 4 + [C.c_uint32, C.c_uint32, C.c_int]
        L.ref_amb_destroy.argtypes = [C.c_void_p]
        L.ref_amb_dims.argtypes = [C.c_void_p, _dp]
        L.ref_amb_process.restype = C.c_double
        L.ref_amb_process.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32, _dp, _dp, _dp, _dp,
                                      C.POINTER(C.c_uint32)]
        L.ref_detect.restype = C.c_int64
        L.ref_detect.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double,
                                 C.c_int, C.c_double, C.c_int, _dp, _dp, _dp, C.c_int64]
        L.ref_wiener_process.restype = C.c_int
        L.ref_wiener_process.argtypes = [C.c_int32, C.c_int32, C.c_uint32, _dp, _dp, _dp, _dp]
        if hasattr(L, "ref_spectrum_process"):
            L.ref_spectrum_process.restype = C.c_int
            L.ref_spectrum_process.argtypes = [C.c_uint32, C.c_double, _dp, _dp, C.c_uint32, C.POINTER(C.c_uint32)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def next_hamming(v: int) -> int:
    return int(lib().ref_next_hamming(v))


class RefAmbiguity:
    """The reference ``Ambiguity`` + ``Map::set_metrics`` + detector chain."""

    def __init__(self, delay_min, delay_max, doppler_min, doppler_max, fs, n, round_hamming=False):
        self._h = lib().ref_amb_create(delay_min, delay_max, doppler_min, doppler_max, fs, n,
                                       1 if round_hamming else 0)
        self.n = n
        d = np.zeros(7)
        lib().ref_amb_dims(self._h, _p(d))
        self.n_doppler_bins, self.n_delay_bins, self.n_corr, self.nfft = (int(v) for v in d[:4])
        self.cpi, self.doppler_middle = float(d[4]), float(d[5])
        self.last_seconds = None

    def close(self):
        if self._h:
            lib().ref_amb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process(self, x, y):
        """-> (map complex128 [nD, nDelay], delay axis, doppler axis, noisePower, maxPower, leftover)."""
        x = np.ascontiguousarray(x, dtype=np.complex128)
        y = np.ascontiguousarray(y, dtype=np.complex128)
        assert x.shape[0] == self.n and y.shape[0] == self.n
        m = np.zeros((self.n_doppler_bins, self.n_delay_bins), dtype=np.complex128)
        delay = np.zeros(self.n_delay_bins)
        dop = np.zeros(self.n_doppler_bins)
        met = np.zeros(2)
        left = (C.c_uint32 * 2)()
        self.last_seconds = lib().ref_amb_process(
            self._h, _p(x.view(np.float64)), _p(y.view(np.float64)), self.n,
            _p(m.view(np.float64)), _p(delay), _p(dop), _p(met), left)
        return m, delay.astype(np.int64), dop, float(met[0]), float(met[1]), (left[0], left[1])

    def detect(self, pfa, n_guard, n_train, min_delay, min_doppler, n_centroid=0, centroid_res=0.0,
               stage=0, cap=1 << 22):
        d = np.zeros(cap)
        f = np.zeros(cap)
        s = np.zeros(cap)
        n = lib().ref_detect(self._h, pfa, n_guard, n_train, min_delay, min_doppler, n_centroid,
                             centroid_res, stage, _p(d), _p(f), _p(s), cap)
        if n < 0:
            raise RuntimeError("ref_detect called before process")
        n = min(n, cap)
        return d[:n].copy(), f[:n].copy(), s[:n].copy()


def wiener_hopf(x, y, delay_min, delay_max):
    """The reference ``WienerHopf::process`` -> (ok, y_filtered, seconds)."""
    x = np.ascontiguousarray(x, dtype=np.complex128)
    y = np.ascontiguousarray(y, dtype=np.complex128)
    n = x.shape[0]
    out = np.zeros(n, dtype=np.complex128)
    secs = C.c_double(0)
    ok = lib().ref_wiener_process(delay_min, delay_max, n, _p(x.view(np.float64)),
                                  _p(y.view(np.float64)), _p(out.view(np.float64)),
                                  C.cast(C.pointer(secs), _dp))
    return bool(ok), (out if ok else y.copy()), secs.value


def spectrum(x, n, bandwidth, cap=1 << 16):
    """The reference ``SpectrumAnalyser(n, bandwidth).process`` -> (spectrum, n_frequency)."""
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.zeros(cap, dtype=np.complex128)
    nf = C.c_uint32(0)
    m = lib().ref_spectrum_process(n, float(bandwidth), _p(x.view(np.float64)), _p(out.view(np.float64)), cap, C.byref(nf))
    return out[:m].copy(), nf.value
