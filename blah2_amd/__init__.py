"""blah2_amd -- MI355X (gfx950) cross-ambiguity engine for the blah2 passive radar.

Product code: ``csrc/`` (hand-written HIP kernels + the C ABI of
``include/blah2hip.h``), ``host/`` (C++ classes with the reference's own
surface) and the thin Python mirror in :mod:`blah2_amd.process`.  There is no
CPU implementation in this package: without ``libblah2hip.so`` and a GPU the
calls fail loudly.
"""
from ._lib import Blah2HipError, FMT_C32, FMT_F16, FMT_I16, FMT_I16X_C32Y, device_count, load  # noqa: F401
from .process import HIT_DTYPE, hits_to_detection  # noqa: F401
from .process import Ambiguity, Centroid, CfarDetector1D, CfarDetector2D, Detection, Interpolate, Map, SpectrumAnalyser, WienerHopf, next_hamming  # noqa: F401

__all__ = ["Ambiguity", "Centroid", "Interpolate", "CfarDetector1D", "CfarDetector2D", "Detection", "Map", "WienerHopf", "SpectrumAnalyser", "next_hamming", "Blah2HipError",
           "FMT_C32", "FMT_I16", "FMT_F16", "FMT_I16X_C32Y", "device_count", "load"]
