"""Python mirror of the reference's hot-path classes, on top of the C ABI.

Same names, constructor arguments, getters and call order as
/root/reference/src/process/ambiguity/Ambiguity.h:34-58,
src/data/Map.h:19-112, src/data/Detection.h:13-70 and
src/process/detection/CfarDetector1D.h:46-55, so the parity tests read like
test/unit/process/ambiguity/TestAmbiguity.cpp.  All numerics run in the HIP
library; this file only marshals buffers.  (The C++ mirror that drops into
blah2.cpp lives in blah2_amd/host/.)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import AmbDims, Blah2HipError, check


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


class Map:
    """src/data/Map.h: rows = Doppler, cols = delay.  ``data`` is complex64."""

    def __init__(self, owner, data, delay, doppler, noise_power, max_power, cpi_index=0):
        self._owner = owner
        self._cpi_index = cpi_index
        self.data = data
        self.delay = delay
        self.doppler = doppler
        self.noisePower = noise_power
        self.maxPower = max_power
        # which device copy this map mirrors: the engine's process-call counter and a fingerprint of the cells
        # (like Map::fingerprint of the C++ class): the detectors run on the engine's copy only while both still match
        self._gen = getattr(owner, "_gen", None)
        self._fp = self._fingerprint()

    def _fingerprint(self):
        # position-dependent (CRC-32 of the bytes + Adler-32, with the shape): a swap of two cells, a roll or equal edits
        # at two places change it, like the C++ Map::fingerprint
        import zlib
        raw = np.ascontiguousarray(self.data).view(np.uint8).ravel()
        return (zlib.crc32(raw), zlib.adler32(raw), self.data.shape)

    def device_copy_is_current(self) -> bool:
        """True while the engine's device copy is still this map: no later process call on the engine and no change of
        ``data`` by the caller."""
        return self._owner is not None and self._gen == getattr(self._owner, "_gen", None) and self._fp == self._fingerprint()

    def get_nRows(self):
        return self.data.shape[0]

    def get_nCols(self):
        return self.data.shape[1]

    def set_metrics(self):
        """Map::set_metrics (Map.cpp:187-206).  The reduction is fused into the
        Doppler kernel, so the values are already there; kept so that callers
        can follow blah2.cpp:278-279 verbatim."""
        return None


class Detection:
    """src/data/Detection.h:13-70."""

    def __init__(self, delay, doppler, snr):
        self.delay = np.asarray(delay, dtype=np.float64)
        self.doppler = np.asarray(doppler, dtype=np.float64)
        self.snr = np.asarray(snr, dtype=np.float64)

    def get_delay(self):
        return self.delay.copy()

    def get_doppler(self):
        return self.doppler.copy()

    def get_snr(self):
        return self.snr.copy()

    def get_nDetections(self):
        return int(self.delay.size)


class Ambiguity:
    """src/process/ambiguity/Ambiguity.h:34-58."""

    def __init__(self, delayMin, delayMax, dopplerMin, dopplerMax, fs, n, roundHamming=False,
                 device=0, max_batch=1, n_doppler_bins=0):
        """``n_doppler_bins`` (extension): an explicit number of Doppler bins, e.g. exactly 512;
        0 = the reference's rule (always odd)."""
        L = _lib.load()
        h = C.c_void_p()
        check(L.blah2hip_amb_create_ex(delayMin, delayMax, dopplerMin, dopplerMax, fs, n,
                                       1 if roundHamming else 0, n_doppler_bins, device, max_batch, C.byref(h)))
        self._h = h
        self._L = L
        self.device = device
        self.dims = AmbDims()
        check(L.blah2hip_amb_get_dims(h, C.byref(self.dims)))
        self.delay = np.zeros(self.dims.n_delay_bins, dtype=np.int32)
        self.doppler = np.zeros(self.dims.n_doppler_bins, dtype=np.float64)
        check(L.blah2hip_amb_get_axes(h, _ptr(self.delay), _ptr(self.doppler)))
        self._n_samples = n
        self._gen = 0  # process calls so far: a Map remembers which one produced it

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.blah2hip_amb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- getters (Ambiguity.cpp:174-199) --------------------------------------
    def get_doppler_middle(self):
        return self.dims.doppler_middle

    def get_n_delay_bins(self):
        return self.dims.n_delay_bins

    def get_n_doppler_bins(self):
        return self.dims.n_doppler_bins

    def get_n_corr(self):
        return self.dims.n_corr

    def get_cpi(self):
        return self.dims.cpi

    def get_nfft(self):
        return self.dims.nfft

    def get_n_samples(self):
        return self._n_samples

    # -- process --------------------------------------------------------------
    def _result(self, out, met, cpi_index=0):
        return Map(self, out, self.delay.copy(), self.doppler.copy(), float(met[0]), float(met[1]), cpi_index)

    def process(self, x, y):
        """Ambiguity::process (+ Map::set_metrics) on host arrays of complex
        samples (x = reference, y = surveillance).  Like the reference it
        consumes n_corr*n_doppler_bins samples and raises on underflow
        (IqData::pop_front, IqData.cpp:57-59)."""
        x = np.ascontiguousarray(x)
        y = np.ascontiguousarray(y)
        nD, nC = self.dims.n_doppler_bins, self.dims.n_delay_bins
        out = np.empty((nD, nC), dtype=np.complex64)
        met = np.zeros(2, dtype=np.float64)
        if x.dtype == np.complex64 and y.dtype == np.complex64:
            fn = self._L.blah2hip_amb_process_c32
        else:
            x = x.astype(np.complex128, copy=False)
            y = y.astype(np.complex128, copy=False)
            fn = self._L.blah2hip_amb_process_c64
        n = min(x.shape[0], y.shape[0])
        rc = fn(self._h, _ptr(x), _ptr(y), n, _ptr(out), _ptr(met))
        if rc == _lib.ERR_UNDERFLOW:
            raise RuntimeError("Attempting to pop from an empty deque")
        check(rc)
        self._n_samples = self.dims.n_used  # Ambiguity.cpp:105
        self._gen += 1
        return self._result(out, met)

    def process_i16(self, iq):
        """Same, on the .rspduo wire layout: int16 array [n, 4] = I1 Q1 I2 Q2."""
        iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1, 4)
        nD, nC = self.dims.n_doppler_bins, self.dims.n_delay_bins
        out = np.empty((nD, nC), dtype=np.complex64)
        met = np.zeros(2, dtype=np.float64)
        rc = self._L.blah2hip_amb_process_i16(self._h, _ptr(iq), iq.shape[0], _ptr(out), _ptr(met))
        if rc == _lib.ERR_UNDERFLOW:
            raise RuntimeError("Attempting to pop from an empty deque")
        check(rc)
        self._n_samples = self.dims.n_used
        self._gen += 1
        return self._result(out, met)

    def process_dev(self, fmt, d_x, d_y, n_cpi, cpi_stride, d_map=None, d_metrics=None, stream=0):
        """Enqueue the device-resident chain on ``stream`` (raw pointers/ints)."""
        fir = getattr(self, "_fir", None)
        if fir is not None:
            if fir._h is None:
                raise Blah2HipError(_lib.ERR_INVALID, "set_fir: the WienerHopf handle has been closed")
            if n_cpi > fir.max_batch:
                raise Blah2HipError(_lib.ERR_INVALID, f"set_fir: {n_cpi} CPIs, the filter handle holds taps for {fir.max_batch}")
        check(self._L.blah2hip_amb_process_dev(self._h, fmt, d_x, d_y, n_cpi, cpi_stride, d_map,
                                               d_metrics, stream))
        self._gen += 1

    def read_last(self, cpi=0):
        nD, nC = self.dims.n_doppler_bins, self.dims.n_delay_bins
        out = np.empty((nD, nC), dtype=np.complex64)
        met = np.zeros(2, dtype=np.float64)
        check(self._L.blah2hip_amb_read_last(self._h, cpi, _ptr(out), _ptr(met)))
        return self._result(out, met, cpi)

    def db_dev(self, d_map, d_metrics, n_cpi, d_db, stream=0):
        """Enqueue the fp32 dB map Map::to_json prints (10*log10|M| - noisePower) for n_cpi maps."""
        check(self._L.blah2hip_amb_db_dev(self._h, d_map, d_metrics, n_cpi, d_db, stream))

    # -- execution plan (engine-specific, not part of the reference surface) ----
    def set_doppler_kernel(self, which):
        """Force one of the Doppler kernels (``_lib.DOP_*`` or its name); 'auto' picks by launch size."""
        if isinstance(which, str):
            which = {v: k for k, v in _lib.DOPPLER_KERNEL_NAMES.items()}[which]
        check(self._L.blah2hip_amb_set_option(self._h, _lib.OPT_DOPPLER_KERNEL, int(which)))

    def set_range_grid(self, n):
        check(self._L.blah2hip_amb_set_option(self._h, _lib.OPT_RANGE_GRID, int(n)))

    def set_doppler_grid(self, n):
        """Workgroup cap of the persistent Doppler tile kernels (0 = their residency)."""
        check(self._L.blah2hip_amb_set_option(self._h, _lib.OPT_DOPPLER_GRID, int(n)))

    def set_cfar2d_kernel(self, which):
        """'auto' / 'tile' (one pass over the map) / 'sat' (summed-area table) for :class:`CfarDetector2D`."""
        if isinstance(which, str):
            which = {"auto": _lib.CFAR2D_AUTO, "tile": _lib.CFAR2D_TILE, "sat": _lib.CFAR2D_SAT, "stream": _lib.CFAR2D_STREAM}[which]
        check(self._L.blah2hip_amb_set_option(self._h, _lib.OPT_CFAR2D_KERNEL, int(which)))

    def set_fir(self, wiener_hopf):
        """Run the clutter filter's FIR fused into the range kernel with the taps of ``wiener_hopf`` (a WienerHopf whose
        ``estimate_dev_fmt`` precedes each ``process_dev`` on the same stream); None: back to the plain range kernels."""
        if wiener_hopf is None:
            check(self._L.blah2hip_amb_set_fir(self._h, None, 0, 0))
        else:
            p, nb, dm = wiener_hopf.taps_dev()
            check(self._L.blah2hip_amb_set_fir(self._h, p, nb, dm))
        # the ambiguity handle reads the filter handle's device array on every call: keep it alive, and remember how many
        # CPIs' taps it holds
        self._fir = wiener_hopf

    def fir_fusable(self, wiener_hopf, fmt):
        """None if ``set_fir(wiener_hopf)`` is covered for samples in format ``fmt`` (FMT_C32 / FMT_I16), else the reason."""
        _, nb, dm = wiener_hopf.taps_dev()
        rc = self._L.blah2hip_amb_fir_fusable(self._h, int(fmt), nb, dm)
        if rc == _lib.OK:
            return None
        if rc == _lib.ERR_UNSUPPORTED:
            return self._L.blah2hip_last_error().decode()
        check(rc)

    def set_hot_columns(self, mode):
        """BLAH2HIP_OPT_HOT_COLUMNS: "off", "auto" (default) or "always" -- the fp64 Doppler transform of the delay columns
        under the map's tallest peaks (include/blah2hip.h)."""
        mode = {"off": 0, "auto": 1, "always": 2}.get(mode, mode)
        check(self._L.blah2hip_amb_set_option(self._h, _lib.OPT_HOT_COLUMNS, int(mode)))

    def hot_columns(self):
        """Columns of the last call's first CPI that were transformed again in fp64 (waits for the device)."""
        return self.info(_lib.INFO_HOT_COLUMNS)

    def set_leak_compensation(self, mode):
        """BLAH2HIP_OPT_LEAK_COMPENSATION: "off", "auto" (default) or "always" (include/blah2hip.h)."""
        mode = {"off": _lib.LEAK_OFF, "auto": _lib.LEAK_AUTO, "always": _lib.LEAK_ALWAYS}.get(mode, mode)
        check(self._L.blah2hip_amb_set_option(self._h, _lib.OPT_LEAK_COMPENSATION, int(mode)))

    def leak_info(self):
        """(cells of the zero-Doppler row the last call corrected, largest |g| of the kernel pair it ran)."""
        return self.info(_lib.INFO_LEAK_LAGS), self.info(_lib.INFO_LEAK_MAX_E12) * 1e-12

    def set_fft_len(self, F):
        """Force the range transform length (1024 / 2048 / 4096; 0 = planner); re-plans the segmentation."""
        check(self._L.blah2hip_amb_set_option(self._h, _lib.OPT_FFT_LEN, int(F)))
        check(self._L.blah2hip_amb_get_dims(self._h, C.byref(self.dims)))

    def set_range_kernel(self, which):
        """0 = by transform length, ``_lib.RANGE_WAVE`` = the one-wave kernel (F = 2048 only)."""
        check(self._L.blah2hip_amb_set_option(self._h, _lib.OPT_RANGE_KERNEL, int(which)))

    def info(self, key):
        v = C.c_int64(0)
        check(self._L.blah2hip_amb_get_info(self._h, key, C.byref(v)))
        return v.value

    def last_doppler_kernel(self):
        return _lib.DOPPLER_KERNEL_NAMES[self.info(_lib.INFO_LAST_DOPPLER_KERNEL)]

    # -- per-kernel timing -----------------------------------------------------
    def set_timing(self, enable=True):
        check(self._L.blah2hip_amb_set_timing(self._h, 1 if enable else 0))

    def get_timing(self):
        ms = np.zeros(_lib.K_COUNT, dtype=np.float64)
        cnt = np.zeros(_lib.K_COUNT, dtype=np.uint32)
        check(self._L.blah2hip_amb_get_timing(self._h, _ptr(ms), _ptr(cnt)))
        return {name: (float(ms[k]), int(cnt[k])) for k, name in _lib.KERNEL_NAMES.items()}


class CfarDetector1D:
    """src/process/detection/CfarDetector1D.h:46-55."""

    def __init__(self, pfa, nGuard, nTrain, minDelay, minDoppler):
        for name, v in (("nGuard", nGuard), ("nTrain", nTrain), ("minDelay", minDelay)):
            if not -128 <= int(v) <= 127:  # int8_t in the reference
                raise ValueError(f"{name} outside int8 range")
        self.pfa, self.nGuard, self.nTrain = float(pfa), int(nGuard), int(nTrain)
        self.minDelay, self.minDoppler = int(minDelay), float(minDoppler)

    def process(self, x: Map) -> Detection:
        """CfarDetector1D::process on ``x.data`` (CfarDetector1D.cpp:23-100).  While ``x`` is still the map the engine holds
        on the device (no later process call, cells untouched) it runs there without an upload; any other Map -- built or
        modified by the caller, or outlived by a newer CPI -- is uploaded and runs through the same kernel
        (blah2hip_cfar1d_map), like the C++ class does."""
        amb = x._owner
        cells = x.data.size
        cap = cells
        d = np.zeros(cap)
        f = np.zeros(cap)
        s = np.zeros(cap)
        n = C.c_uint32(0)
        if x.device_copy_is_current():
            check(amb._L.blah2hip_cfar1d_process(amb._h, x._cpi_index, self.pfa, self.nGuard, self.nTrain,
                                                 self.minDelay, self.minDoppler, _ptr(d), _ptr(f), _ptr(s),
                                                 cap, C.byref(n)))
        else:
            L = _lib.load()
            m = np.ascontiguousarray(x.data, dtype=np.complex64)
            dax = np.ascontiguousarray(x.delay, dtype=np.int32)
            fax = np.ascontiguousarray(x.doppler, dtype=np.float64)
            dev = amb.device if amb is not None else 0
            check(L.blah2hip_cfar1d_map(_ptr(m), m.shape[0], m.shape[1], _ptr(dax), _ptr(fax), float(x.noisePower), self.pfa,
                                        self.nGuard, self.nTrain, self.minDelay, self.minDoppler, dev, _ptr(d), _ptr(f),
                                        _ptr(s), cap, C.byref(n)))
        k = n.value
        return Detection(d[:k].copy(), f[:k].copy(), s[:k].copy())

    def process_dev(self, amb, n_cpi, d_hits, cap, d_count, d_map=None, d_metrics=None, stream=0):
        """Enqueue the detector for n_cpi device-resident maps (None = the engine's internal buffers of the
        last process_dev): hit records into d_hits [n_cpi][cap], counts into d_count [n_cpi]."""
        check(amb._L.blah2hip_cfar1d_dev(amb._h, d_map, d_metrics, n_cpi, self.pfa, self.nGuard, self.nTrain,
                                         self.minDelay, self.minDoppler, d_hits, cap, d_count, stream))


def hits_to_detection(amb, hits, count, cap):
    """One CPI's device hit records (``blah2hip_hit_t``: int32 row, int32 col, double snr; arbitrary
    order) -> :class:`Detection` in the reference's row-major emission order
    (CfarDetector1D.cpp:36-92: delay = col + delay[0], doppler = doppler[row])."""
    if count > cap:
        raise Blah2HipError(_lib.ERR_CAPACITY, f"{count} detections, capacity {cap}")
    h = np.asarray(hits[:count])
    order = np.lexsort((h["col"], h["row"]))
    h = h[order]
    return Detection(h["col"].astype(np.float64) + float(amb.delay[0]), amb.doppler[h["row"]], h["snr"].astype(np.float64))


HIT_DTYPE = np.dtype([("row", np.int32), ("col", np.int32), ("snr", np.float64)])


class CfarDetector2D:
    """2-D cell-averaging CFAR (BASELINE.json configs[2]).  Not a reference class:
    the reference only has the 1-D detector; this is its extension as defined in
    SURVEY.md section 8g, and with nGuardDoppler = nTrainDoppler = 0 it returns
    exactly what :class:`CfarDetector1D` returns."""

    def __init__(self, pfa, nGuardDelay, nTrainDelay, nGuardDoppler, nTrainDoppler, minDelay, minDoppler):
        self.pfa = float(pfa)
        self.p = [int(nGuardDelay), int(nTrainDelay), int(nGuardDoppler), int(nTrainDoppler)]
        self.minDelay, self.minDoppler = int(minDelay), float(minDoppler)

    def process(self, x: Map) -> Detection:
        amb = x._owner
        if not x.device_copy_is_current():
            raise ValueError("CfarDetector2D.process: this Map is no longer the engine's device copy (a later process call, or "
                             "its cells were changed); the 2-D detector has no host-map entry point")
        cap = x.data.size
        d, f, s = np.zeros(cap), np.zeros(cap), np.zeros(cap)
        n = C.c_uint32(0)
        check(amb._L.blah2hip_cfar2d_process(amb._h, x._cpi_index, self.pfa, *self.p, self.minDelay,
                                             self.minDoppler, _ptr(d), _ptr(f), _ptr(s), cap, C.byref(n)))
        k = n.value
        return Detection(d[:k].copy(), f[:k].copy(), s[:k].copy())

    def process_dev(self, amb, n_cpi, d_hits, cap, d_count, d_map=None, d_metrics=None, stream=0):
        check(amb._L.blah2hip_cfar2d_dev(amb._h, d_map, d_metrics, n_cpi, self.pfa, *self.p, self.minDelay,
                                         self.minDoppler, d_hits, cap, d_count, stream))


class Centroid:
    """src/process/detection/Centroid.h: non-maximum suppression of the CFAR list."""

    def __init__(self, nDelay, nDoppler, resolutionDoppler):
        self.nDelay, self.nDoppler, self.resolutionDoppler = int(nDelay), int(nDoppler), float(resolutionDoppler)

    def process(self, x: Detection) -> Detection:
        L = _lib.load()
        n = x.get_nDetections()
        d, f, s = (np.ascontiguousarray(v, dtype=np.float64) for v in (x.delay, x.doppler, x.snr))
        od, of, os_ = np.zeros(n), np.zeros(n), np.zeros(n)
        k = C.c_uint32(0)
        check(L.blah2hip_centroid(_ptr(d), _ptr(f), _ptr(s), n, self.nDelay, self.nDoppler, self.resolutionDoppler,
                                  _ptr(od), _ptr(of), _ptr(os_), C.byref(k)))
        return Detection(od[:k.value], of[:k.value], os_[:k.value])


class Interpolate:
    """src/process/detection/Interpolate.h: quadratic peak interpolation."""

    def __init__(self, doDelay, doDoppler):
        self.doDelay, self.doDoppler = bool(doDelay), bool(doDoppler)

    def process(self, x: Detection, y: Map) -> Detection:
        L = _lib.load()
        n = x.get_nDetections()
        d, f, s = (np.ascontiguousarray(v, dtype=np.float64) for v in (x.delay, x.doppler, x.snr))
        m = np.ascontiguousarray(y.data, dtype=np.complex64)
        dax = np.ascontiguousarray(y.delay, dtype=np.int32)
        fax = np.ascontiguousarray(y.doppler, dtype=np.float64)
        od, of, os_ = np.zeros(n), np.zeros(n), np.zeros(n)
        k = C.c_uint32(0)
        check(L.blah2hip_interpolate(_ptr(d), _ptr(f), _ptr(s), n, _ptr(m), m.shape[0], m.shape[1], _ptr(dax),
                                     _ptr(fax), float(y.noisePower), int(self.doDelay), int(self.doDoppler),
                                     _ptr(od), _ptr(of), _ptr(os_), C.byref(k)))
        return Detection(od[:k.value], of[:k.value], os_[:k.value])


class WienerHopf:
    """src/process/clutter/WienerHopf.h:68-78: least-squares clutter canceller.

    ``process(x, y)`` returns ``(ok, y_filtered)``; the reference returns the
    bool and replaces the contents of the y FIFO (WienerHopf.cpp:156-160).  When
    the normal equations are not positive definite ``ok`` is False and y is
    returned unchanged (blah2.cpp:270-273 then skips the CPI)."""

    def __init__(self, delayMin, delayMax, nSamples, device=0, max_batch=1):
        L = _lib.load()
        h = C.c_void_p()
        check(L.blah2hip_clutter_create(delayMin, delayMax, nSamples, device, max_batch, C.byref(h)))
        self._h, self._L = h, L
        self.nSamples = nSamples
        self.max_batch = max(1, int(max_batch))
        nb, fl, sl = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(L.blah2hip_clutter_get_dims(h, C.byref(nb), C.byref(fl), C.byref(sl)))
        self.nBins, self.fft_len, self.seg_len = nb.value, fl.value, sl.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.blah2hip_clutter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process(self, x, y):
        x = np.ascontiguousarray(x)
        y = np.ascontiguousarray(y)
        if x.shape[0] != self.nSamples or y.shape[0] != self.nSamples:
            # the reference sizes its plans for nSamples (WienerHopf.cpp:7-56) and reads exactly that many
            raise ValueError(f"WienerHopf.process needs {self.nSamples} samples per channel, got {x.shape[0]} and {y.shape[0]}")
        ok = C.c_int(0)
        if x.dtype == np.complex64 and y.dtype == np.complex64:
            out = np.empty(self.nSamples, dtype=np.complex64)
            check(self._L.blah2hip_clutter_process_c32(self._h, _ptr(x), _ptr(y), x.shape[0], _ptr(out), C.byref(ok)))
        else:
            x = x.astype(np.complex128, copy=False)
            y = y.astype(np.complex128, copy=False)
            out = np.empty(self.nSamples, dtype=np.complex128)
            check(self._L.blah2hip_clutter_process_c64(self._h, _ptr(x), _ptr(y), x.shape[0], _ptr(out), C.byref(ok)))
        return (True, out) if ok.value else (False, y.copy())

    def process_dev(self, d_x, d_y, n_cpi, cpi_stride, d_y_out, d_ok=None, stream=0):
        """Enqueue on ``stream``: device complex64 planes, output may alias d_y."""
        check(self._L.blah2hip_clutter_process_dev(self._h, d_x, d_y, n_cpi, cpi_stride, d_y_out, d_ok, stream))

    def process_dev_fmt(self, fmt, d_x, d_y, n_cpi, cpi_stride, d_y_out, out_stride, d_ok=None, stream=0):
        """Enqueue on ``stream`` with the input in format ``fmt`` (FMT_C32 planes, or FMT_I16: d_x = the interleaved
        .rspduo buffer); the filtered channel is written as a complex64 plane with ``out_stride`` samples per CPI."""
        check(self._L.blah2hip_clutter_process_dev_fmt(self._h, fmt, d_x, d_y, n_cpi, cpi_stride, d_y_out, out_stride,
                                                       d_ok, stream))

    def estimate_dev_fmt(self, fmt, d_x, d_y, n_cpi, cpi_stride, d_ok=None, stream=0):
        """The filter's correlations, reduction and Toeplitz solve only: the taps stay in the handle (``taps_dev``) for an
        Ambiguity handle that runs the FIR fused into its range kernel (``Ambiguity.set_fir``)."""
        check(self._L.blah2hip_clutter_estimate_dev_fmt(self._h, fmt, d_x, d_y, n_cpi, cpi_stride, d_ok, stream))

    def taps_dev(self):
        """(device pointer of the taps [max_batch][nBins] complex64, nBins, the filter's first lag)."""
        p, nb, dm = C.c_void_p(), C.c_uint32(), C.c_int32()
        check(self._L.blah2hip_clutter_taps_dev(self._h, C.byref(p), C.byref(nb), C.byref(dm)))
        return p.value, nb.value, dm.value

    def _refresh_dims(self):
        nb, fl, sl = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(self._L.blah2hip_clutter_get_dims(self._h, C.byref(nb), C.byref(fl), C.byref(sl)))
        self.nBins, self.fft_len, self.seg_len = nb.value, fl.value, sl.value

    def set_fft_len(self, F):
        """Force the transform length of the correlation / FIR kernels (0 = planner)."""
        check(self._L.blah2hip_clutter_set_option(self._h, _lib.CLUTTER_OPT_FFT_LEN, int(F)))
        self._refresh_dims()

    def set_fir_carry(self, on):
        """Blocks of F/2 samples with the window overlap carried in registers (filters with nBins - 1 just under F/2); re-plans."""
        check(self._L.blah2hip_clutter_set_option(self._h, _lib.CLUTTER_OPT_FIR_CARRY, 1 if on else 0))

    def set_corr_form(self, which):
        """'auto' / 'half' / 'window' (``_lib.CLUTTER_CORR_*``)."""
        if isinstance(which, str):
            which = {"auto": _lib.CLUTTER_CORR_AUTO, "half": _lib.CLUTTER_CORR_HALF, "window": _lib.CLUTTER_CORR_WINDOW}[which]
        check(self._L.blah2hip_clutter_set_option(self._h, _lib.CLUTTER_OPT_CORR, int(which)))
        self._refresh_dims()

    def set_solve_indices_per_thread(self, k):
        check(self._L.blah2hip_clutter_set_option(self._h, _lib.CLUTTER_OPT_SOLVE_K, int(k)))

    def set_solve_form(self, which, indices_per_lane=0):
        """'auto' / 'stepwise' (one workgroup per CPI, a barrier per order) / 'lookahead' (blocks of 32 orders on several
        workgroups per CPI); ``indices_per_lane`` in {0, 2, 3, 6, 12} fixes the look-ahead form's slice width."""
        if isinstance(which, str):
            which = {"auto": _lib.CLUTTER_SOLVE_AUTO, "stepwise": _lib.CLUTTER_SOLVE_STEPWISE,
                     "lookahead": _lib.CLUTTER_SOLVE_LOOKAHEAD}[which]
        check(self._L.blah2hip_clutter_set_option(self._h, _lib.CLUTTER_OPT_SOLVE_FORM, int(which)))
        check(self._L.blah2hip_clutter_set_option(self._h, _lib.CLUTTER_OPT_SOLVE_E, int(indices_per_lane)))

    def set_solve_spin_limit(self, polls):
        """Polls after which a wait of the look-ahead solve gives up (0 = default, about a second).  Tests: a CPI whose
        solve gave up is solved again by the one-workgroup kernel behind it, so the result does not change."""
        check(self._L.blah2hip_clutter_set_option(self._h, _lib.CLUTTER_OPT_SOLVE_SPIN_LIMIT, int(polls)))

    def solve(self, r, b):
        """The filter's Toeplitz solve alone: toeplitz(r) w = b for [n_cpi][nBins] (or [nBins]) complex r, b.
        Returns (ok[n_cpi] bool, w[n_cpi][nBins] complex64)."""
        r = np.atleast_2d(np.asarray(r, dtype=np.complex128))
        b = np.atleast_2d(np.asarray(b, dtype=np.complex128))
        if r.shape != b.shape or r.shape[1] != self.nBins:
            raise ValueError(f"solve needs [n_cpi][{self.nBins}] arrays")
        rb = np.ascontiguousarray(np.stack([r, b], axis=1))
        w = np.empty((r.shape[0], self.nBins), dtype=np.complex64)
        ok = np.zeros(r.shape[0], dtype=np.int32)
        check(self._L.blah2hip_clutter_solve(self._h, _ptr(rb), r.shape[0], _ptr(w), _ptr(ok)))
        return ok.astype(bool), w

    def solve_dev(self, d_rb, n_cpi, d_w, d_ok, stream=0):
        """Enqueue the solve on device arrays: d_rb [n_cpi][2][nBins] complex128, d_w [n_cpi][nBins] complex64, d_ok int32."""
        check(self._L.blah2hip_clutter_solve_dev(self._h, d_rb, n_cpi, d_w, d_ok, stream))

    def solve_info(self):
        """What the last call's Toeplitz solve ran: {'form', 'E' (indices per lane), 'G' (workgroups per CPI), 'fault' (sticky:
        a bounded wait of the look-ahead form ran out at some point), 'retries' (CPIs the gated one-workgroup kernel has solved)}."""
        out = {}
        for name, what in (("form", _lib.CLUTTER_INFO_SOLVE_FORM), ("E", _lib.CLUTTER_INFO_SOLVE_E),
                           ("G", _lib.CLUTTER_INFO_SOLVE_G), ("fault", _lib.CLUTTER_INFO_SOLVE_FAULT),
                           ("retries", _lib.CLUTTER_INFO_SOLVE_RETRIES)):
            v = C.c_int64(0)
            check(self._L.blah2hip_clutter_get_info(self._h, what, C.byref(v)))
            out[name] = int(v.value)
        return out

    def read_last(self, cpi=0):
        """(ok, w, r, b) of CPI ``cpi`` of the last call: the nBins filter taps (complex64) and the fp64
        correlations r, b of the normal equations A w = b, A[i][j] = r[i-j] (diagnostics)."""
        w = np.empty(self.nBins, dtype=np.complex64)
        rb = np.empty(2 * self.nBins, dtype=np.complex128)
        ok = C.c_int(0)
        check(self._L.blah2hip_clutter_read_last(self._h, cpi, _ptr(w), _ptr(rb), C.byref(ok)))
        return bool(ok.value), w, rb[:self.nBins].copy(), rb[self.nBins:].copy()

    def set_timing(self, enable=True):
        check(self._L.blah2hip_clutter_set_timing(self._h, 1 if enable else 0))

    def get_timing(self):
        ms = np.zeros(_lib.CK_COUNT, dtype=np.float64)
        cnt = np.zeros(_lib.CK_COUNT, dtype=np.uint32)
        check(self._L.blah2hip_clutter_get_timing(self._h, _ptr(ms), _ptr(cnt)))
        return {name: (float(ms[k]), int(cnt[k])) for k, name in _lib.CLUTTER_KERNEL_NAMES.items()}


class SpectrumAnalyser:
    """src/process/spectrum/SpectrumAnalyser.h:53-62: decimated spectrum of the
    reference channel.  ``process(x)`` returns ``(spectrum, frequency)``: the
    nSpectrum complex values the reference hands to ``IqData::update_spectrum``
    (SpectrumAnalyser.cpp:43-55) and the frequency axis it hands to
    ``update_frequency``, which is always empty (the axis loop runs on a uint32
    that starts at (2^32 - nSpectrum)/2, :64)."""

    def __init__(self, n, bandwidth, device=0, max_batch=1):
        L = _lib.load()
        h = C.c_void_p()
        check(L.blah2hip_spectrum_create(n, float(bandwidth), device, max_batch, C.byref(h)))
        self._h, self._L = h, L
        d, ns, nfft = C.c_uint32(), C.c_uint32(), C.c_uint64()
        check(L.blah2hip_spectrum_get_dims(h, C.byref(d), C.byref(ns), C.byref(nfft)))
        self.decimation, self.nSpectrum, self.nfft = d.value, ns.value, nfft.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.blah2hip_spectrum_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process(self, x):
        x = np.ascontiguousarray(x)
        out = np.empty(self.nSpectrum, dtype=np.complex128)
        if x.dtype == np.complex64:
            check(self._L.blah2hip_spectrum_process_c32(self._h, _ptr(x), x.shape[0], _ptr(out)))
        else:
            x = x.astype(np.complex128, copy=False)
            check(self._L.blah2hip_spectrum_process_c64(self._h, _ptr(x), x.shape[0], _ptr(out)))
        return out, np.empty(0, dtype=np.float64)

    def process_dev(self, fmt, d_x, n_cpi, cpi_stride, d_out, stream=0):
        """Enqueue on ``stream``; d_out is [n_cpi][nSpectrum] complex128 in HBM."""
        check(self._L.blah2hip_spectrum_process_dev(self._h, fmt, d_x, n_cpi, cpi_stride, d_out, stream))


def next_hamming(v: int) -> int:
    """src/process/meta/HammingNumber.cpp:38-48."""
    return int(_lib.load().blah2hip_next_hamming(v))
