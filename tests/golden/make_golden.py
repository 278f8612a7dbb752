#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the reference's OWN sources.

Run in the build container (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py [fixture names ...]

Each fixture holds a seeded synthetic capture in the .rspduo wire layout
(int16 I1 Q1 I2 Q2, /root/reference/src/capture/rspduo/RspDuo.cpp:512-526) and
what the compiled reference (oracle/_ref/libblah2ref.so) produced for it:
the derived sizes, both axes, the complex128 map of Ambiguity::process, the
Map::set_metrics pair, the CfarDetector1D / Centroid / Interpolate detection
lists and the WienerHopf-filtered surveillance channel.  The GPU box has no
/root/reference, so these files are what `-m gpu` parity tests compare with.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import blah2_oracle as O  # noqa: E402
from oracle import ref_lib as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (fs, n, delayMin, delayMax, dopplerMin, dopplerMax, roundHamming, seed, targets, clutter(dMin,dMax))
CASES = {
    "small_sym": (200_000, 20_000, -3, 20, -50, 50, True, 11, ((7, -40.0, 0.05),), (-3, 20)),
    "small_nohamming": (200_000, 20_000, -3, 20, -50, 50, False, 12, ((5, 20.0, 0.05),), (-3, 20)),
    "small_asym": (200_000, 20_000, -2, 30, -20, 60, True, 13, ((11, 40.0, 0.06),), (-2, 30)),
    "medium": (1_000_000, 100_000, -10, 100, -100, 100, True, 14,
               ((37, -63.0, 0.05), (80, 30.0, 0.03)), (-10, 100)),
    "delay_pos_only": (200_000, 30_000, 0, 40, -30, 30, True, 15, ((20, 10.0, 0.05),), (0, 40)),
    # round 3, the widened envelope.  199-sample pulses, nfft = 2*199 - 1 = 397 (no Hamming rounding, so that no delay falls
    # into the all-zero band between nCorr and nfft - nCorr, whose cells are FFT rounding noise in the reference and make
    # Map::set_metrics' mean of dB values incomparable): delays from 199 on read the opposite-sign lag d - 397 out of the
    # reference's nfft-point circular correlation (Ambiguity.cpp:132-146)
    "aliased_lags": (200_000, 40_000, -30, 350, -500, 500, False, 16, ((20, 100.0, 0.05), (-15, -200.0, 0.05)), (-3, 20)),
    # clutter filter with a POSITIVE delayMin: the first delayMin reference samples come from the uint32 wrap of
    # WienerHopf.cpp:61-70 (index i - delayMin below zero), the piece the engine's windows treat element by element
    "clutter_pos_delay": (200_000, 30_000, 0, 40, -30, 30, True, 18, ((20, 10.0, 0.05),), (3, 40)),
    # nCorr = 200000 / 3 = 66666 does not fit the reference's uint16 (Ambiguity.h:80-89): it processes 1130 samples per pulse
    "ncorr_wraps_uint16": (200_000, 200_000, -3, 20, -1, 1, True, 19, ((7, 0.4, 0.05),), (-3, 20)),
    # 4201 delay bins (more than one on-chip transform holds: the engine runs the window as lag chunks)
    "many_delay_bins": (60_000, 60_000, -20, 4180, -2, 2, True, 17, ((4000, 1.0, 0.05), (17, -1.0, 0.05)), (-3, 20)),
    # round 6, deep cancellation: receiver noise at 1 LSB (sigma = 1 per component before the int16 rounding), the direct
    # path 58 dB above it (0.8 x 1000), one target 40 dB under the direct path.  After WienerHopf::process the channel is
    # 800 x smaller than what the filter subtracted: the case where fp32 `y - w*x` is nearest the 1e-4 gate
    "deep_cancel": (1_000_000, 100_000, -10, 100, -100, 100, True, 26, ((37, -63.0, 0.008),), (-10, 100)),
    # round 6: 9011 delay bins -- more than blah2hip_cfar1d_map's fp64 row used to fit (8192); delays beyond nCorr = 6000 read the
    # reference's aliased lags (Ambiguity.cpp:132-146), no Hamming rounding as in `aliased_lags`
    "wide_delay": (30_000, 30_000, -10, 9000, -2, 2, False, 28, ((5000, 1.0, 0.05), (23, -1.0, 0.05)), (-3, 20)),
    # round 6: a clutter filter of 4610 taps -- more than one on-chip transform holds (4081): the engine runs it as three
    # chunks of 2048 taps (csrc/clutter.hip, "LONG filters"); the reference's 4610 x 4610 Cholesky takes 45 s here
    "long_filter": (60_000, 60_000, -10, 100, -2, 2, True, 29, ((300, 1.0, 0.05), (50, -1.0, 0.05)), (-10, 4600)),
}
# name -> synth_iq keyword overrides (amplitudes); everything else uses the generator's defaults
SYNTH_KW = {"deep_cancel": dict(ref_amp=1000.0, noise_amp=1.0, direct=0.8)}
# name -> (n, bandwidth, seed)
SPECTRUM_CASES = {
    "ragged_odd_decimation": (30_011, 2000.0, 21),     # D = 15, nS = 2000, nfft = 30000 < n
    "odd_nfft": (44_000, 2001.0, 22),                  # D = 21, nS = 2095, nfft = 43995 (odd)
    "even_decimation_odd_bins": (45_122, 2000.0, 23), # D = 22, nS = 2051: bins congruent to D/2+1 mod D
    "small": (5_000, 400.0, 24),                       # D = 12, nS = 416
    "wide": (70_000, 20_000.0, 25),                    # D = 3, nS = 23333 (round 3: more bins than the on-chip tables hold)
    "beyond_direct_dft": (140_001, 70_000.0, 27),      # D = 2, nS = 70000 > 65536 (round 6: the chirp-z path; nfft = 140000)
}
DET0 = dict(pfa=1e-5, n_guard=2, n_train=6, min_delay=5, min_doppler=15.0, n_centroid=6)
# name -> detector overrides (recorded in the fixture's det_params): `wide_delay` has five Doppler rows within +-2 Hz
DET_KW = {"wide_delay": dict(pfa=1e-3, min_doppler=0.5)}


def main():
    if not R.available():
        sys.exit("oracle/_ref/libblah2ref.so missing: run `make -C oracle ref` first")
    only = set(sys.argv[1:])  # optional: fixture names to (re)generate
    for name, (fs, n, dmin, dmax, fmin, fmax, rh, seed, targets, clut) in CASES.items():
        if only and name not in only:
            continue
        DET = dict(DET0, **DET_KW.get(name, {}))
        x, y = O.synth_iq(n, seed=seed, fs=fs, targets=targets, **SYNTH_KW.get(name, {}))
        iq = np.empty((n, 4), dtype=np.int16)
        iq[:, 0], iq[:, 1], iq[:, 2], iq[:, 3] = x.real, x.imag, y.real, y.imag
        amb = R.RefAmbiguity(dmin, dmax, fmin, fmax, fs, n, rh)
        m, delay, doppler, noise, peak, left = amb.process(x, y)
        tcpi = n / fs
        det0 = amb.detect(DET["pfa"], DET["n_guard"], DET["n_train"], DET["min_delay"],
                          DET["min_doppler"], stage=0)
        det1 = amb.detect(DET["pfa"], DET["n_guard"], DET["n_train"], DET["min_delay"],
                          DET["min_doppler"], DET["n_centroid"], 1.0 / tcpi, stage=1)
        det2 = amb.detect(DET["pfa"], DET["n_guard"], DET["n_train"], DET["min_delay"],
                          DET["min_doppler"], DET["n_centroid"], 1.0 / tcpi, stage=2)
        ok, yf, _ = R.wiener_hopf(x, y, clut[0], clut[1])
        # the full chain as blah2.cpp:268-287 runs it: clutter filter, then ambiguity
        amb2 = R.RefAmbiguity(dmin, dmax, fmin, fmax, fs, n, rh)
        m2, _, _, noise2, peak2, _ = amb2.process(x, yf)
        det_chain = amb2.detect(DET["pfa"], DET["n_guard"], DET["n_train"], DET["min_delay"],
                                DET["min_doppler"], stage=0)
        # SpectrumAnalyser on the reference channel with the bandwidth blah2.cpp:198 hard-codes
        spec, n_freq = R.spectrum(x, n, 2000.0)
        out = os.path.join(HERE, name + ".npz")
        np.savez_compressed(
            out, iq=iq, spectrum=spec, spectrum_n_frequency=np.int64(n_freq),
            params=np.array([fs, n, dmin, dmax, fmin, fmax, int(rh)], dtype=np.int64),
            dims=np.array([amb.n_doppler_bins, amb.n_delay_bins, amb.n_corr, amb.nfft], dtype=np.int64),
            cpi=np.float64(amb.cpi), doppler_middle=np.float64(amb.doppler_middle),
            leftover=np.array(left, dtype=np.int64),
            delay=delay, doppler=doppler, map=m, metrics=np.array([noise, peak]),
            det_params=np.array([DET["pfa"], DET["n_guard"], DET["n_train"], DET["min_delay"],
                                 DET["min_doppler"], DET["n_centroid"], 1.0 / tcpi]),
            cfar=np.stack(det0), centroid=np.stack(det1), interp=np.stack(det2),
            clutter_params=np.array(clut, dtype=np.int64), clutter_ok=np.bool_(ok), clutter_y=yf,
            chain_map=m2, chain_metrics=np.array([noise2, peak2]), chain_cfar=np.stack(det_chain))
        print(f"{name}: nD={amb.n_doppler_bins} nDelay={amb.n_delay_bins} nCorr={amb.n_corr} "
              f"nfft={amb.nfft} cfar={det0[0].size} centroid={det1[0].size} interp={det2[0].size} "
              f"clutter_ok={ok} chain_cfar={det_chain[0].size} -> {os.path.getsize(out)/1024:.0f} KiB")
    # SpectrumAnalyser geometries the capture fixtures do not reach: nfft < n, odd
    # decimation, odd nfft, nSpectrum != 2000
    os.makedirs(os.path.join(HERE, "spectrum"), exist_ok=True)
    for name, (n, bw, seed) in SPECTRUM_CASES.items():
        if only and name not in only:
            continue
        x, _ = O.synth_iq(n, seed=seed, fs=2_000_000)
        spec, n_freq = R.spectrum(x, n, bw, cap=1 << 18)
        iq = np.empty((n, 2), dtype=np.int16)
        iq[:, 0], iq[:, 1] = x.real, x.imag
        out = os.path.join(HERE, "spectrum", name + ".npz")
        np.savez_compressed(out, iq=iq, params=np.array([n, bw], dtype=np.float64), spectrum=spec,
                            n_frequency=np.int64(n_freq))
        print(f"spectrum/{name}: n={n} bw={bw} dims={O.spectrum_dims(n, bw)} nFrequency={n_freq} "
              f"-> {os.path.getsize(out)/1024:.0f} KiB")


if __name__ == "__main__":
    main()
