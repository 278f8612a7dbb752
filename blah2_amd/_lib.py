"""ctypes binding of include/blah2hip.h.

The HIP library is the product; there is no CPU fallback.  Importing this
module fails loudly when ``libblah2hip.so`` has not been built, and every call
raises :class:`Blah2HipError` on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libblah2hip.so")

OK = 0
ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_UNDERFLOW, ERR_NO_DEVICE, ERR_CAPACITY = -1, -2, -3, -4, -5, -6
FMT_C32, FMT_I16, FMT_F16, FMT_I16X_C32Y = 0, 1, 2, 3
K_RANGE, K_DOPPLER, K_METRICS, K_CFAR, K_SAT_ROWS, K_SAT_COLS, K_ROTATE, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 8
KERNEL_NAMES = {K_RANGE: "range", K_DOPPLER: "doppler", K_METRICS: "metrics", K_CFAR: "cfar",
                K_SAT_ROWS: "sat_rows", K_SAT_COLS: "sat_cols", K_ROTATE: "rotate"}
CK_CORR, CK_REDUCE, CK_SOLVE, CK_FIR, CK_COUNT = 0, 1, 2, 3, 4
CLUTTER_KERNEL_NAMES = {CK_CORR: "clutter_corr", CK_REDUCE: "clutter_reduce", CK_SOLVE: "clutter_solve",
                        CK_FIR: "clutter_fir"}
OPT_DOPPLER_KERNEL, OPT_RANGE_GRID, OPT_RANGE_KERNEL, OPT_DOPPLER_GRID, OPT_FFT_LEN, OPT_CFAR2D_KERNEL = 1, 2, 3, 4, 5, 6
OPT_LEAK_COMPENSATION = 7
OPT_HOT_COLUMNS = 8
LEAK_OFF, LEAK_AUTO, LEAK_ALWAYS = 0, 1, 2
CFAR2D_AUTO, CFAR2D_TILE, CFAR2D_SAT, CFAR2D_STREAM = 0, 1, 2, 3
CLUTTER_OPT_SOLVE_K, CLUTTER_OPT_FFT_LEN, CLUTTER_OPT_CORR, CLUTTER_OPT_SOLVE_FORM, CLUTTER_OPT_SOLVE_E, CLUTTER_OPT_FIR_CARRY = 1, 2, 3, 4, 5, 6
CLUTTER_OPT_SOLVE_SPIN_LIMIT = 7
CLUTTER_SOLVE_AUTO, CLUTTER_SOLVE_STEPWISE, CLUTTER_SOLVE_LOOKAHEAD = 0, 1, 2
CLUTTER_INFO_SOLVE_FORM, CLUTTER_INFO_SOLVE_E, CLUTTER_INFO_SOLVE_G, CLUTTER_INFO_SOLVE_FAULT, CLUTTER_INFO_SOLVE_RETRIES = 1, 2, 3, 4, 5
CLUTTER_CORR_AUTO, CLUTTER_CORR_HALF, CLUTTER_CORR_WINDOW = 0, 1, 2
DOP_AUTO, DOP_TILE8, DOP_TILE16, DOP_TILEM, DOP_COLUMN, DOP_DIRECT, DOP_TILEW, DOP_TILEW2, DOP_TILE16WG, DOP_SUB4, DOP_TILE8K, DOP_TILEW4 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
DOPPLER_KERNEL_NAMES = {DOP_AUTO: "auto", DOP_TILE8: "tile8", DOP_TILE16: "tile16", DOP_TILEM: "tilem",
                        DOP_COLUMN: "column", DOP_DIRECT: "direct", DOP_TILEW: "tilew", DOP_TILEW2: "tilew2", DOP_TILE16WG: "tile16wg", DOP_SUB4: "sub4", DOP_TILE8K: "tile8k", DOP_TILEW4: "tilew4"}
RANGE_E16, RANGE_E8, RANGE_WAVE, RANGE_WAVE1K, RANGE_PS, RANGE_FIR = 1, 2, 3, 5, 6, 7
INFO_LAST_DOPPLER_KERNEL, INFO_LAST_RANGE_KERNEL, INFO_DOPPLER_FFT_LEN, INFO_RANGE_GRID, INFO_NUM_CU = 1, 2, 3, 4, 5
INFO_DOPPLER_GRID, INFO_DOPPLER_TILES = 6, 7
INFO_LEAK_LAGS, INFO_LEAK_MAX_E12 = 8, 9
INFO_HOT_COLUMNS = 10


class Blah2HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"blah2hip error {code}: {msg}")
        self.code = code


class AmbDims(C.Structure):
    _fields_ = [("n_doppler_bins", C.c_uint32), ("n_delay_bins", C.c_uint32), ("n_corr", C.c_uint32),
                ("nfft", C.c_uint32), ("n_samples", C.c_uint32), ("n_used", C.c_uint32),
                ("cpi", C.c_double), ("doppler_middle", C.c_double), ("fft_len", C.c_uint32),
                ("n_seg", C.c_uint32), ("seg_len", C.c_uint32), ("max_batch", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("row", C.c_int32), ("col", C.c_int32), ("snr", C.c_double)]


# every symbol include/blah2hip.h declares: name -> (restype, argtypes)
_vp, _u32, _i32, _dbl = C.c_void_p, C.c_uint32, C.c_int32, C.c_double
SYMBOLS = {
    "blah2hip_last_error": (C.c_char_p, []),
    "blah2hip_version": (C.c_char_p, []),
    "blah2hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "blah2hip_next_hamming": (_u32, [_u32]),
    "blah2hip_amb_create": (C.c_int, [_i32, _i32, _i32, _i32, _u32, _u32, C.c_int, C.c_int, _u32, C.POINTER(_vp)]),
    "blah2hip_amb_create_ex": (C.c_int, [_i32, _i32, _i32, _i32, _u32, _u32, C.c_int, _u32, C.c_int, _u32, C.POINTER(_vp)]),
    "blah2hip_amb_destroy": (C.c_int, [_vp]),
    "blah2hip_amb_get_dims": (C.c_int, [_vp, C.POINTER(AmbDims)]),
    "blah2hip_amb_get_axes": (C.c_int, [_vp, _vp, _vp]),
    "blah2hip_amb_set_option": (C.c_int, [_vp, C.c_int, C.c_int64]),
    "blah2hip_amb_get_info": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64)]),
    "blah2hip_amb_process_c64": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp]),
    "blah2hip_amb_process_c32": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp]),
    "blah2hip_amb_process_i16": (C.c_int, [_vp, _vp, _u32, _vp, _vp]),
    "blah2hip_amb_process_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _u32, C.c_uint64, _vp, _vp, _vp]),
    "blah2hip_amb_read_last": (C.c_int, [_vp, _u32, _vp, _vp]),
    "blah2hip_amb_db_dev": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp]),
    "blah2hip_cfar1d_dev": (C.c_int, [_vp, _vp, _vp, _u32, _dbl, _i32, _i32, _i32, _dbl, _vp, _u32, _vp, _vp]),
    "blah2hip_cfar1d_prepare": (C.c_int, [_vp, _dbl, _i32]),
    "blah2hip_cfar2d_prepare": (C.c_int, [_vp, _dbl, _i32, _i32, _i32, _i32]),
    "blah2hip_cfar1d_process": (C.c_int, [_vp, _u32, _dbl, _i32, _i32, _i32, _dbl, _vp, _vp, _vp, _u32, C.POINTER(_u32)]),
    "blah2hip_cfar1d_map": (C.c_int, [_vp, _u32, _u32, _vp, _vp, _dbl, _dbl, _i32, _i32, _i32, _dbl, C.c_int, _vp, _vp, _vp, _u32,
                                      C.POINTER(_u32)]),
    "blah2hip_cfar2d_dev": (C.c_int, [_vp, _vp, _vp, _u32, _dbl, _i32, _i32, _i32, _i32, _i32, _dbl, _vp, _u32, _vp, _vp]),
    "blah2hip_cfar2d_process": (C.c_int, [_vp, _u32, _dbl, _i32, _i32, _i32, _i32, _i32, _dbl, _vp, _vp, _vp, _u32,
                                          C.POINTER(_u32)]),
    "blah2hip_centroid": (C.c_int, [_vp, _vp, _vp, _u32, C.c_uint16, C.c_uint16, _dbl, _vp, _vp, _vp, C.POINTER(_u32)]),
    "blah2hip_interpolate": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _u32, _u32, _vp, _vp, _dbl, C.c_int, C.c_int,
                                       _vp, _vp, _vp, C.POINTER(_u32)]),
    "blah2hip_clutter_create": (C.c_int, [_i32, _i32, _u32, C.c_int, _u32, C.POINTER(_vp)]),
    "blah2hip_clutter_destroy": (C.c_int, [_vp]),
    "blah2hip_clutter_process_c64": (C.c_int, [_vp, _vp, _vp, _u32, _vp, C.POINTER(C.c_int)]),
    "blah2hip_clutter_process_c32": (C.c_int, [_vp, _vp, _vp, _u32, _vp, C.POINTER(C.c_int)]),
    "blah2hip_clutter_process_dev": (C.c_int, [_vp, _vp, _vp, _u32, C.c_uint64, _vp, _vp, _vp]),
    "blah2hip_clutter_process_dev_fmt": (C.c_int, [_vp, C.c_int, _vp, _vp, _u32, C.c_uint64, _vp, C.c_uint64, _vp, _vp]),
    "blah2hip_clutter_set_option": (C.c_int, [_vp, C.c_int, C.c_int64]),
    "blah2hip_clutter_solve": (C.c_int, [_vp, _vp, _u32, _vp, _vp]),
    "blah2hip_clutter_solve_dev": (C.c_int, [_vp, _vp, _u32, _vp, _vp, _vp]),
    "blah2hip_clutter_get_info": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64)]),
    "blah2hip_clutter_get_dims": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "blah2hip_clutter_read_last": (C.c_int, [_vp, _u32, _vp, _vp, C.POINTER(C.c_int)]),
    "blah2hip_clutter_set_timing": (C.c_int, [_vp, C.c_int]),
    "blah2hip_clutter_get_timing": (C.c_int, [_vp, _vp, _vp]),
    "blah2hip_spectrum_create": (C.c_int, [_u32, C.c_double, C.c_int, _u32, C.POINTER(_vp)]),
    "blah2hip_spectrum_destroy": (C.c_int, [_vp]),
    "blah2hip_spectrum_get_dims": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(C.c_uint64)]),
    "blah2hip_spectrum_process_c64": (C.c_int, [_vp, _vp, _u32, _vp]),
    "blah2hip_spectrum_process_c32": (C.c_int, [_vp, _vp, _u32, _vp]),
    "blah2hip_spectrum_process_dev": (C.c_int, [_vp, C.c_int, _vp, _u32, C.c_uint64, _vp, _vp]),
    "blah2hip_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "blah2hip_ctx_destroy": (C.c_int, [_vp]),
    "blah2hip_ctx_stream": (_vp, [_vp]),
    "blah2hip_ctx_sync": (C.c_int, [_vp]),
    "blah2hip_ctx_malloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "blah2hip_ctx_free": (C.c_int, [_vp, _vp]),
    "blah2hip_ctx_malloc_host": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "blah2hip_ctx_free_host": (C.c_int, [_vp, _vp]),
    "blah2hip_ctx_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "blah2hip_ctx_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "blah2hip_ctx_d2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "blah2hip_stream_read_dev": (C.c_int, [_vp, C.c_size_t, _vp, _vp]),
    "blah2hip_clutter_estimate_dev_fmt": (C.c_int, [_vp, C.c_int, _vp, _vp, _u32, C.c_uint64, _vp, _vp]),
    "blah2hip_clutter_taps_dev": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_u32), C.POINTER(_i32)]),
    "blah2hip_amb_set_fir": (C.c_int, [_vp, _vp, _u32, _i32]),
    "blah2hip_amb_fir_fusable": (C.c_int, [_vp, C.c_int, _u32, _i32]),
    "blah2hip_amb_result_ptrs": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "blah2hip_amb_set_timing": (C.c_int, [_vp, C.c_int]),
    "blah2hip_amb_get_timing": (C.c_int, [_vp, _vp, _vp]),
}

_lib = None


def load():
    """Loads libblah2hip.so (built by ``python -m blah2_amd.build``) or raises."""
    global _lib
    if _lib is not None:
        return _lib
    # BLAH2HIP_LIBRARY: another BUILD of the same HIP library (tools/ A/B runs of compile-time variants)
    path = os.environ.get("BLAH2HIP_LIBRARY", LIB_PATH)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -m blah2_amd.build` "
            "(there is no CPU fallback for the HIP path)")
    # The PyTorch-ROCm wheel bundles its own libamdhip64/libhsa-runtime64.  Two HIP
    # runtimes in one process cannot both own the device, so when torch is installed
    # it is imported FIRST: libblah2hip.so's libamdhip64.so.7 dependency then binds
    # to the runtime torch already loaded (torch is only used by the Python harness
    # for device buffers/streams; C++ callers link /opt/rocm's runtime directly).
    if os.environ.get("BLAH2HIP_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError when the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise Blah2HipError(rc, load().blah2hip_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = C.c_int(0)
    rc = load().blah2hip_device_count(C.byref(n))
    return n.value if rc == OK else 0
