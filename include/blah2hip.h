/* blah2hip.h -- C ABI of the MI355X (gfx950) cross-ambiguity engine for blah2.
 *
 * The reference (30hours/blah2) has no plugin/FFI layer: the boundary of its
 * hot path is the C++ class surface that src/blah2.cpp constructs and calls
 * once per CPI on its processing thread (blah2.cpp:154-177, 263-289):
 *
 *   Ambiguity(delayMin,delayMax,dopplerMin,dopplerMax,fs,n,roundHamming)
 *       src/process/ambiguity/Ambiguity.h:34   ctor   Ambiguity.cpp:11-82
 *   Map<complex<double>>* Ambiguity::process(IqData* x, IqData* y)
 *       src/process/ambiguity/Ambiguity.h:44          Ambiguity.cpp:92-172
 *   void Map::set_metrics()                            src/data/Map.cpp:187-206
 *   CfarDetector1D(pfa,nGuard,nTrain,minDelay,minDoppler)::process(Map*)
 *       src/process/detection/CfarDetector1D.h:46,55   CfarDetector1D.cpp:23-100
 *   WienerHopf(delayMin,delayMax,nSamples)::process(IqData* x, IqData* y) -> bool
 *       src/process/clutter/WienerHopf.h:68,78         WienerHopf.cpp:58-163
 *
 * This header is what a binding for that path binds: opaque handles, plain
 * pointers and sizes, int status returns, no exceptions, no C++ or torch types.
 * blah2_amd/host/ holds source-compatible C++ classes (same names, arguments
 * and error behaviour as the reference's) implemented on top of it, and
 * INTEGRATION.md shows how they replace the FFTW path in blah2.cpp.
 *
 * Conventions
 *   - every function returns BLAH2HIP_OK (0) or a negative error code;
 *     blah2hip_last_error() returns a thread-local description of the last one.
 *   - "host" entry points take caller-owned host buffers, run on the handle's
 *     own stream and return when the results are in the output buffers.
 *   - "dev" entry points take device pointers that are already resident in HBM
 *     and a hipStream_t (as void*); they only enqueue work and never
 *     synchronise.  This is the chain the benchmark times.  Everything a dev
 *     call needs is allocated at *_create; the two exceptions are named where
 *     they occur (blah2hip_cfar1d_prepare / blah2hip_cfar2d_prepare: a threshold
 *     table per parameter tuple and the 2-D detector's summed-area table, built
 *     by the first call that needs them or ahead of time by *_prepare).
 *   - a handle is not re-entrant: one thread at a time, like the reference's
 *     objects (all three process() calls run on blah2.cpp's single t2 thread).
 *   - maps are complex fp32, interleaved (re,im), row-major [doppler][delay],
 *     the layout of Map<T>::data (src/data/Map.h:27).
 */
#ifndef BLAH2HIP_H
#define BLAH2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLAH2HIP_OK 0
#define BLAH2HIP_ERR_INVALID (-1)     /* bad argument */
#define BLAH2HIP_ERR_HIP (-2)         /* a HIP runtime call failed */
#define BLAH2HIP_ERR_UNSUPPORTED (-3) /* configuration outside what the kernels cover */
#define BLAH2HIP_ERR_UNDERFLOW (-4)   /* fewer samples than one CPI (IqData::pop_front throws, IqData.cpp:57-59) */
#define BLAH2HIP_ERR_NO_DEVICE (-5)   /* no gfx950 device visible */
#define BLAH2HIP_ERR_CAPACITY (-6)    /* output capacity too small */

/* input sample formats of the dev entry points */
#define BLAH2HIP_FMT_C32 0 /* two planes of complex fp32: x = reference, y = surveillance */
#define BLAH2HIP_FMT_I16 1 /* one buffer, int16 I1 Q1 I2 Q2 per sample (.rspduo, RspDuo.cpp:512-526) */
#define BLAH2HIP_FMT_F16 2 /* two planes of (re,im) IEEE half pairs; widened to fp32 on load, fp32 accumulate */
#define BLAH2HIP_FMT_I16X_C32Y 3 /* reference channel from the .rspduo buffer (tuner 1 of I1 Q1 I2 Q2), surveillance channel from a
                                  * complex fp32 plane: the ambiguity stage behind blah2hip_clutter_process_dev_fmt(FMT_I16), which
                                  * leaves x untouched and writes the filtered y as fp32 (WienerHopf.cpp:156-160) */

typedef struct blah2hip_amb_s *blah2hip_amb_t;
typedef struct blah2hip_clutter_s *blah2hip_clutter_t;
typedef struct blah2hip_spectrum_s *blah2hip_spectrum_t;

/* Derived sizes: the first block reproduces the reference constructor
 * (Ambiguity.cpp:22-65) and is what its getters return; the second block is
 * this engine's execution plan for the range stage. */
typedef struct blah2hip_amb_dims {
  uint32_t n_doppler_bins; /* Ambiguity::get_n_doppler_bins */
  uint32_t n_delay_bins;   /* Ambiguity::get_n_delay_bins   */
  uint32_t n_corr;         /* Ambiguity::get_n_corr         */
  uint32_t nfft;           /* Ambiguity::get_nfft (value the reference would plan; reported, not used) */
  uint32_t n_samples;      /* constructor argument n        */
  uint32_t n_used;         /* n_corr * n_doppler_bins = samples process() consumes */
  double cpi;              /* Ambiguity::get_cpi            */
  double doppler_middle;   /* Ambiguity::get_doppler_middle */
  uint32_t fft_len;        /* F: on-chip transform length of the range kernel */
  uint32_t n_seg;          /* segments per pulse */
  uint32_t seg_len;        /* reference samples per segment */
  uint32_t max_batch;      /* CPIs one dev call may carry */
} blah2hip_amb_dims_t;

/* One CFAR hit as the device writes it; the host API converts to the
 * reference's Detection triplets. */
typedef struct blah2hip_hit {
  int32_t row; /* Doppler bin index */
  int32_t col; /* delay bin index   */
  double snr;  /* 10*log10|z| - noisePower (CfarDetector1D.cpp:48) */
} blah2hip_hit_t;

const char *blah2hip_last_error(void);
const char *blah2hip_version(void);
int blah2hip_device_count(int *count);
/* smallest 5-smooth integer strictly greater than v (HammingNumber.cpp:38-48) */
uint32_t blah2hip_next_hamming(uint32_t v);

/* ---- Ambiguity (Ambiguity.h:34-58) ------------------------------------- */
int blah2hip_amb_create(int32_t delay_min, int32_t delay_max, int32_t doppler_min,
                        int32_t doppler_max, uint32_t fs, uint32_t n, int round_hamming,
                        int device, uint32_t max_batch, blah2hip_amb_t *out);
/* Extension (not in the reference, whose constructor only yields odd counts): the same
 * engine with an EXPLICIT number of Doppler bins / pulses, n_doppler_bins (0 = the
 * reference's rule).  nCorr = n / n_doppler_bins, cpi, both axes and the row shift
 * (j + nD/2 + 1) % nD of Ambiguity.cpp:165 follow from it unchanged; for an even count
 * the Nyquist bin is the last row.  This is how "512 / 1024 / 2048 Doppler bins" of
 * BASELINE.json can be had literally. */
int blah2hip_amb_create_ex(int32_t delay_min, int32_t delay_max, int32_t doppler_min, int32_t doppler_max,
                           uint32_t fs, uint32_t n, int round_hamming, uint32_t n_doppler_bins,
                           int device, uint32_t max_batch, blah2hip_amb_t *out);
/* Geometry (the reference's own limit is the uint16 narrowing of Ambiguity.h:80-89, reproduced):
 *   - any number of delay bins: a window of more than 4081 lags runs as chunks of 2048 on the 4096-point range
 *     transform, each chunk re-reading the pulses (correct, slower per bin);
 *   - delays beyond nfft - n_corr read, in the reference's nfft-point CIRCULAR correlation (Ambiguity.cpp:132-146), the
 *     opposite-sign lag d -+ nfft: reproduced (every delay maps to one linear lag; runs of consecutive lags are chunks);
 *     |delay| >= nfft, where the reference indexes outside its buffer, is BLAH2HIP_ERR_UNSUPPORTED;
 *   - Doppler lengths above 2049 run on the direct-DFT kernel (correct, slow). */
int blah2hip_amb_destroy(blah2hip_amb_t h);
int blah2hip_amb_get_dims(blah2hip_amb_t h, blah2hip_amb_dims_t *dims);
/* Map::delay (bins, length n_delay_bins) and Map::doppler (Hz, length n_doppler_bins) */
int blah2hip_amb_get_axes(blah2hip_amb_t h, int32_t *delay, double *doppler);

/* Execution plan, per handle.  Options take effect on the next process call. */
#define BLAH2HIP_OPT_DOPPLER_KERNEL 1 /* BLAH2HIP_DOP_*; AUTO picks by launch size */
#define BLAH2HIP_OPT_RANGE_GRID 2     /* workgroups of the range kernel; 0 = the launched kernel's residency */
#define BLAH2HIP_OPT_RANGE_KERNEL 3   /* 0 = by transform length and launch size (4096: E16; 2048: WAVE once a launch has a pulse per wave
                                       * slot of the chip -- 8 x CUs -- else E16, e.g. a single CPI; 1024: WAVE1K from 12 x CUs pulses,
                                       * else E8); BLAH2HIP_RANGE_E16 / _WAVE (F = 2048) / _WAVE2 (F = 4096, measured 5 % slower than
                                       * E16) / _WAVE1K / _E8 (F = 1024) force one */
#define BLAH2HIP_OPT_DOPPLER_GRID 4   /* workgroup cap of the PERSISTENT Doppler tile kernels (TILE8 / TILE16 / TILEW / TILEW2); 0 = their
                                       * residency (one or two workgroups per CU).  A small cap makes every workgroup walk many tiles --
                                       * the steady state of the software-pipelined loops -- on a small fixture (tests) */
#define BLAH2HIP_OPT_FFT_LEN 5        /* range transform length F in {1024, 2048, 4096}; 0 = the planner's choice (cost model).  Re-plans the
                                       * segmentation and re-uploads the root table (blocking); BLAH2HIP_ERR_UNSUPPORTED when the lag window
                                       * does not fit F.  Replaces the BLAH2HIP_FFT_LEN environment variable of earlier versions */
#define BLAH2HIP_OPT_CFAR2D_KERNEL 6  /* 2-D detector: BLAH2HIP_CFAR2D_AUTO (the one-pass stream kernel for the window shapes it is instantiated
                                       * for -- C2S_SHAPES in csrc/cfar_kernels.hpp, among them config.yml's 2 / 6 along delay with 1 / 3 along
                                       * Doppler --, the one-pass tile kernel for other windows with nGf + nTf <= 24 and nGd + nTd <= 40, else the
                                       * summed-area table), _STREAM / _TILE (BLAH2HIP_ERR_UNSUPPORTED at the call for other windows) or _SAT */
#define BLAH2HIP_OPT_LEAK_COMPENSATION 7 /* Fixed-pattern leak of the fp32 transform chain (csrc/capi.hip, "leak compensation"): the butterfly
                                       * and twiddle constants are each off by up to 3e-8 of themselves, the same way in every transform, and
                                       * what that adds up to over a CPI is a FIXED fraction g[d] (<= 2e-8) of the lag-0 column appearing at a
                                       * few dozen lags d -- in the zero-Doppler row, where the direct-path peak stands sqrt(N) above the floor,
                                       * up to 1e-4 of a floor cell at 4e7 samples per CPI.  g is measured once per (range kernel, Doppler
                                       * kernel) on a synthetic white CPI against its exact fp64 direct sum and subtracted from that row.
                                       * 1 (default): where max|g| sqrt(N) >= 3e-5 (not at 2 MS/s x 1 s); 2: wherever a pattern was measured;
                                       * 0: never.  No reference counterpart (the reference computes in fp64). */
#define BLAH2HIP_OPT_HOT_COLUMNS 8    /* fp64 Doppler transform of the delay columns that hold a peak more than 250x (24 dB) above the map's
                                       * mean level (csrc/capi.hip, "hot columns"): the fp32 transform leaves up to 1.2e-7 of a column's peak
                                       * in that column's other rows, which behind the clutter filter (the floor 8x down, a target 1400x above
                                       * it at 10 MS/s) is 1.7e-4 of a mean-level cell.  At most 16 columns a CPI, found from four pulses of
                                       * the range map.  1 (default): CPIs of >= 35 000 samples (a shorter one cannot hold such a peak);
                                       * 2: every call; 0: never.  No reference counterpart (the reference computes in fp64). */
#define BLAH2HIP_CFAR2D_AUTO 0
#define BLAH2HIP_CFAR2D_TILE 1
#define BLAH2HIP_CFAR2D_SAT 2
#define BLAH2HIP_CFAR2D_STREAM 3
#define BLAH2HIP_DOP_AUTO 0
#define BLAH2HIP_DOP_TILE8 1   /* nD <= 513: 8-column tiles, one wave per column */
#define BLAH2HIP_DOP_TILE16 2  /* nD <= 513: 16-column tiles, one wave per column on the one-wave 1024-point transform */
#define BLAH2HIP_DOP_TILEM 3   /* 513 < nD <= 2049: multi-wave columns, 8 or 4 per workgroup */
#define BLAH2HIP_DOP_COLUMN 4  /* nD <= 2049: one column per workgroup (small launches) */
#define BLAH2HIP_DOP_DIRECT 5  /* any nD: direct DFT */
#define BLAH2HIP_DOP_TILEW 6   /* 513 < nD <= 1025: one-wave 2048-point columns, 8 per workgroup */
#define BLAH2HIP_DOP_TILEW2 7  /* 1025 < nD <= 2049: two-wave 4096-point columns, 4 per workgroup */
#define BLAH2HIP_DOP_TILE16WG 8 /* TILE16 on the workgroup transform of rounds 1-2 (twiddles in registers; kept for comparison) */
#define BLAH2HIP_DOP_TILEW4 11 /* 1025 < nD <= 2049: ONE wave per column, the 4096-point transform as four one-wave 1024-point transforms + a radix-4 step in registers, 8 columns per workgroup (round 5) */
#define BLAH2HIP_DOP_TILE8K 10 /* nD <= 513: TILE16's kernel on 8-column half tiles, two workgroups per CU (round 5) */
#define BLAH2HIP_DOP_SUB4 9    /* nD <= 513, small launches (a lone CPI): 4-column workgroups, one wave per SIMD; Map::set_metrics finished by the last workgroup */
#define BLAH2HIP_RANGE_E16 1   /* 16 points per thread, one workgroup per pulse (F = 4096; F = 2048 on request) */
#define BLAH2HIP_RANGE_E8 2    /* 8 points per thread, last stage across lanes (F = 1024) */
#define BLAH2HIP_RANGE_WAVE 3  /* one wave per pulse, 32 points per lane, no barriers (F = 2048) */
/* 4: was BLAH2HIP_RANGE_WAVE2 (a pair of waves per pulse at F = 4096; measured 5 % slower than _E16 and removed in round 4) */
#define BLAH2HIP_RANGE_WAVE1K 5 /* one wave per pulse, 16 points per lane, four waves per SIMD (F = 1024) */
#define BLAH2HIP_RANGE_FIR 7      /* INFO_LAST_RANGE_KERNEL only: range_fir_kernel (blah2hip_amb_set_fir) */
#define BLAH2HIP_RANGE_PS 6     /* small launches (a lone CPI) at F = 1024: one workgroup of four waves per pulse, its segments dealt round-robin to the waves */
/* BLAH2HIP_ERR_UNSUPPORTED when the kernel does not cover the handle's Doppler length */
int blah2hip_amb_set_option(blah2hip_amb_t h, int option, int64_t value);
#define BLAH2HIP_INFO_LAST_DOPPLER_KERNEL 1 /* BLAH2HIP_DOP_* the last process call launched (0 = none yet) */
#define BLAH2HIP_INFO_LAST_RANGE_KERNEL 2   /* BLAH2HIP_RANGE_* */
#define BLAH2HIP_INFO_DOPPLER_FFT_LEN 3     /* chirp-z transform length M (0 = direct DFT only) */
#define BLAH2HIP_INFO_RANGE_GRID 4
#define BLAH2HIP_INFO_NUM_CU 5
#define BLAH2HIP_INFO_DOPPLER_GRID 6        /* workgroups of the last Doppler launch */
#define BLAH2HIP_INFO_LEAK_LAGS 8            /* cells of the zero-Doppler row the last process call corrected (0 = compensation not applied) */
#define BLAH2HIP_INFO_LEAK_MAX_E12 9         /* 1e12 x the largest |g| measured for the kernel pair the last call ran (0 = not measurable: no
                                              * zero-Doppler row / lag-0 column, rotated reference channel, chunked lag window) */
#define BLAH2HIP_INFO_HOT_COLUMNS 10          /* columns of the last call's first CPI rewritten in fp64 (BLAH2HIP_OPT_HOT_COLUMNS); waits for that call's stream */
#define BLAH2HIP_INFO_DOPPLER_TILES 7       /* tiles (units of work the persistent workgroups walk) of the last Doppler launch; 0 for the
                                             * non-persistent kernels */
int blah2hip_amb_get_info(blah2hip_amb_t h, int key, int64_t *value);

/* Ambiguity::process + Map::set_metrics on host buffers (blah2.cpp:278-279).
 * x = reference, y = surveillance, n complex samples each (n >= n_used, only
 * the first n_used are consumed, like the reference's pops).  map_out may be
 * NULL; metrics[0] = noisePower, metrics[1] = maxPower. */
int blah2hip_amb_process_c64(blah2hip_amb_t h, const double *x, const double *y, uint32_t n,
                             float *map_out, double *metrics);
int blah2hip_amb_process_c32(blah2hip_amb_t h, const float *x, const float *y, uint32_t n,
                             float *map_out, double *metrics);
int blah2hip_amb_process_i16(blah2hip_amb_t h, const int16_t *iq, uint32_t n, float *map_out,
                             double *metrics);

/* Device-resident chain: range kernel -> Doppler kernel (+ fused metrics).
 * d_x/d_y: device pointers (fmt C32: two planes; fmt I16: d_x = interleaved
 * buffer, d_y ignored).  CPI c starts at sample c*cpi_stride.  d_map:
 * [n_cpi][n_doppler][n_delay] complex fp32; d_metrics: [n_cpi][2] doubles.
 * Either output may be NULL (then the handle's internal buffer is used and
 * can be read with blah2hip_amb_read_last). */
int blah2hip_amb_process_dev(blah2hip_amb_t h, int fmt, const void *d_x, const void *d_y,
                             uint32_t n_cpi, uint64_t cpi_stride, void *d_map, double *d_metrics,
                             void *stream);
/* copies CPI `cpi` of the handle's internal map/metrics to the host (synchronises) */
int blah2hip_amb_read_last(blah2hip_amb_t h, uint32_t cpi, float *map_out, double *metrics);
/* The values Map::to_json prints (Map.cpp:148-155): d_db[cpi][doppler][delay] =
 * 10*log10|M| - noisePower as fp32, from d_map/d_metrics as written by
 * blah2hip_amb_process_dev (NULL = the handle's internal buffers).  Enqueues only. */
int blah2hip_amb_db_dev(blah2hip_amb_t h, const void *d_map, const double *d_metrics, uint32_t n_cpi,
                        float *d_db, void *stream);

/* ---- CfarDetector1D (CfarDetector1D.h:46-55) ---------------------------- */
/* dev: d_map/d_metrics as written by blah2hip_amb_process_dev (NULL = the
 * handle's internal buffers).  d_hits: [n_cpi][cap]; d_count: [n_cpi], zeroed
 * by the call.  Hits are appended in arbitrary order; count may exceed cap
 * (then only cap were stored). */
int blah2hip_cfar1d_dev(blah2hip_amb_t h, const void *d_map, const double *d_metrics,
                        uint32_t n_cpi, double pfa, int32_t n_guard, int32_t n_train,
                        int32_t min_delay, double min_doppler, blah2hip_hit_t *d_hits,
                        uint32_t cap, uint32_t *d_count, void *stream);
/* Builds the threshold table alpha[n] = n (pfa^(-1/n) - 1) for this (pfa, n_train) ahead of
 * time (one blocking upload).  blah2hip_cfar1d_dev does it on the first call with a new
 * tuple and only enqueues afterwards.  A table built by *_prepare stays resident for the handle's lifetime (a graph
 * captured after it may replay at any time); the tables the dev calls build on their own are a cache of eight, least
 * recently used first out. */
int blah2hip_cfar1d_prepare(blah2hip_amb_t h, double pfa, int32_t n_train);
/* host: runs the detector on CPI `cpi` of the handle's internal map and
 * returns Detection's three vectors (delay bins, Doppler Hz, snr) in the
 * reference's emission order (row-major, CfarDetector1D.cpp:36-92). */
int blah2hip_cfar1d_process(blah2hip_amb_t h, uint32_t cpi, double pfa, int32_t n_guard,
                            int32_t n_train, int32_t min_delay, double min_doppler, double *delay,
                            double *doppler, double *snr, uint32_t cap, uint32_t *count);

/* The same detector on a map held by the HOST (any Map<complex<double>>, e.g. one the caller
 * modified or built itself): complex fp32 cells [n_doppler][n_delay], Map::delay (bins),
 * Map::doppler (Hz) and Map::noisePower.  Uploads, runs the GPU kernel on `device`, frees. */
int blah2hip_cfar1d_map(const float *map, uint32_t n_doppler, uint32_t n_delay, const int32_t *delay_axis,
                        const double *doppler_axis, double noise_power, double pfa, int32_t n_guard,
                        int32_t n_train, int32_t min_delay, double min_doppler, int device, double *delay,
                        double *doppler, double *snr, uint32_t cap, uint32_t *count);

/* 2-D cell-averaging CFAR (BASELINE.json configs[2]; NOT in the reference, which
 * only has the 1-D detector).  Extension defined in SURVEY.md section 8g: training
 * rectangle (2(ngd+ntd)+1) x (2(ngf+ntf)+1) minus the guard box, in-bounds cells
 * only, delay column 0 never trains, |z|^2 statistic, alpha = N(pfa^(-1/N)-1);
 * with n_guard_doppler = n_train_doppler = 0 it is exactly blah2hip_cfar1d_*.
 * Same output conventions as the 1-D entry points. */
int blah2hip_cfar2d_dev(blah2hip_amb_t h, const void *d_map, const double *d_metrics, uint32_t n_cpi,
                        double pfa, int32_t n_guard_delay, int32_t n_train_delay,
                        int32_t n_guard_doppler, int32_t n_train_doppler, int32_t min_delay,
                        double min_doppler, blah2hip_hit_t *d_hits, uint32_t cap, uint32_t *d_count,
                        void *stream);
/* Builds the threshold table for this parameter tuple and, for windows beyond the one-pass tile kernel's
 * halo (see BLAH2HIP_OPT_CFAR2D_KERNEL), allocates the summed-area table ([max_batch][nD+1][nDelay+1]
 * doubles); blah2hip_cfar2d_dev calls it implicitly. */
int blah2hip_cfar2d_prepare(blah2hip_amb_t h, double pfa, int32_t n_guard_delay, int32_t n_train_delay,
                            int32_t n_guard_doppler, int32_t n_train_doppler);
int blah2hip_cfar2d_process(blah2hip_amb_t h, uint32_t cpi, double pfa, int32_t n_guard_delay,
                            int32_t n_train_delay, int32_t n_guard_doppler, int32_t n_train_doppler,
                            int32_t min_delay, double min_doppler, double *delay, double *doppler,
                            double *snr, uint32_t cap, uint32_t *count);

/* ---- Centroid / Interpolate (host-side, tens of detections) -------------- */
/* Centroid::process (src/process/detection/Centroid.cpp:19-73): keeps a
 * detection unless a stronger one lies strictly inside its
 * (+-n_delay bins, +-n_doppler*resolution Hz) box.  The box's delay limits are
 * uint16_t in the reference (they wrap for delay - n_delay < 0); reproduced.
 * In/out arrays may not alias.  *count_out <= count. */
int blah2hip_centroid(const double *delay, const double *doppler, const double *snr, uint32_t count,
                      uint16_t n_delay, uint16_t n_doppler, double resolution_doppler,
                      double *delay_out, double *doppler_out, double *snr_out, uint32_t *count_out);
/* Interpolate::process (src/process/detection/Interpolate.cpp:20-91): 3-point
 * quadratic peak interpolation in delay and/or Doppler on the map cells
 * (complex fp32, [n_doppler][n_delay]); drops detections on the map edge or
 * whose neighbours are stronger.  The Doppler branch stores its SNR estimate
 * into the delay variable like the reference does (:80). */
int blah2hip_interpolate(const double *delay, const double *doppler, const double *snr, uint32_t count,
                         const float *map, uint32_t n_doppler, uint32_t n_delay, const int32_t *delay_axis,
                         const double *doppler_axis, double noise_power, int do_delay, int do_doppler,
                         double *delay_out, double *doppler_out, double *snr_out, uint32_t *count_out);

/* ---- WienerHopf clutter filter (WienerHopf.h:68-78) ---------------------
 * nBins = delay_max - delay_min taps (WienerHopf.cpp:12).  Up to 4081 taps run on one on-chip transform (fp32 planes or the int16
 * words).  More ("long" filters, round 6): the same kernels chunk by chunk of 2048 lags / taps on rotated and shifted copies of the
 * channels, the Toeplitz solve by one workgroup on vectors in global memory -- fp32 planes only (BLAH2HIP_FMT_I16: ERR_UNSUPPORTED),
 * no blah2hip_clutter_set_option; built for coverage, not speed (2 nChunks - 1 correlation passes, nChunks FIR passes, a few
 * microseconds per order of the solve).  nBins > n_samples: ERR_UNSUPPORTED (the reference reads its n_samples correlation lags out
 * of bounds there, WienerHopf.cpp:76-108). */
int blah2hip_clutter_create(int32_t delay_min, int32_t delay_max, uint32_t n_samples, int device,
                            uint32_t max_batch, blah2hip_clutter_t *out);
int blah2hip_clutter_destroy(blah2hip_clutter_t h);
/* host: y_out = y - w*x (WienerHopf.cpp:156-160); *ok = 0 when the normal
 * equations are not positive definite (the reference returns false and the CPI
 * is skipped, blah2.cpp:270-273), in which case y_out is not written. */
int blah2hip_clutter_process_c64(blah2hip_clutter_t h, const double *x, const double *y, uint32_t n,
                                 double *y_out, int *ok);
int blah2hip_clutter_process_c32(blah2hip_clutter_t h, const float *x, const float *y, uint32_t n,
                                 float *y_out, int *ok);
/* dev: complex fp32 planes resident in HBM; d_y_out may alias d_y.
 * d_ok: [n_cpi] int32 flags.  Enqueues only. */
int blah2hip_clutter_process_dev(blah2hip_clutter_t h, const void *d_x, const void *d_y,
                                 uint32_t n_cpi, uint64_t cpi_stride, void *d_y_out, int32_t *d_ok,
                                 void *stream);
/* The same with the INPUT in format fmt: BLAH2HIP_FMT_C32 (d_x, d_y planes) or BLAH2HIP_FMT_I16 (d_x = the interleaved
 * .rspduo buffer I1 Q1 I2 Q2, d_y ignored: the correlation and FIR kernels read the int16 words directly,
 * RspDuo.cpp:512-526).  The filtered surveillance channel is always written as a complex fp32 plane, CPI c at
 * d_y_out + c * out_stride samples (for FMT_C32 it may alias d_y with out_stride = cpi_stride).  Enqueues only. */
int blah2hip_clutter_process_dev_fmt(blah2hip_clutter_t h, int fmt, const void *d_x, const void *d_y, uint32_t n_cpi,
                                     uint64_t cpi_stride, void *d_y_out, uint64_t out_stride, int32_t *d_ok, void *stream);
/* Execution plan of the filter.  SOLVE_K: indices of the Toeplitz recursion per thread (0 = by
 * size: 1 up to 1024 taps, 2 up to 2048, 4 above; the workgroup has ceil(nBins / K) threads rounded up to a wave). */
#define BLAH2HIP_CLUTTER_OPT_SOLVE_K 1
/* Planner overrides (re-plan and re-allocate the handle's work buffers, blocking; they replace the BLAH2HIP_CLUTTER_FFT_LEN /
 * BLAH2HIP_CLUTTER_CORR environment variables of earlier versions).  FFT_LEN: transform length in {1024, 2048, 4096}, 0 = planner.
 * CORR: BLAH2HIP_CLUTTER_CORR_AUTO / _HALF (two transforms per F/2 samples; needs nBins - 1 <= F/2) / _WINDOW (three per F - nBins + 1). */
#define BLAH2HIP_CLUTTER_OPT_FFT_LEN 2
#define BLAH2HIP_CLUTTER_OPT_CORR 3
#define BLAH2HIP_CLUTTER_CORR_AUTO 0
#define BLAH2HIP_CLUTTER_CORR_HALF 1
#define BLAH2HIP_CLUTTER_CORR_WINDOW 2
/* The Toeplitz solve of the normal equations (WienerHopf.cpp:85-122).  SOLVE_FORM: _LOOKAHEAD = blocks of 32 orders on
 * several workgroups per CPI (as many CUs as the launch leaves free; the default), _STEPWISE = the one-workgroup kernel,
 * one barrier per order (SOLVE_K applies to it).  SOLVE_E: indices per lane of the look-ahead form's slices
 * (2, 3, 6 or 12: 96 ... 736 indices per wave; 0 = the smallest whose workgroups each get a CU at the launch's batch). */
#define BLAH2HIP_CLUTTER_OPT_SOLVE_FORM 4
#define BLAH2HIP_CLUTTER_OPT_SOLVE_E 5
/* FIR_CARRY (planner, re-plans like FFT_LEN): 1 (default) = a filter whose history nBins - 1 is just under F/2 (cfg 3: 2047 taps on
 * F = 4096) runs on blocks of exactly F/2 samples and the FIR kernel carries the window overlap in registers; 0 = blocks of
 * F - nBins + 1 samples, every window read whole (tests, A/B timing). */
#define BLAH2HIP_CLUTTER_OPT_FIR_CARRY 6
/* SOLVE_SPIN_LIMIT (tests): polls after which a wait of the look-ahead solve gives up (0 = default, 2^20, about a second).
 * A CPI whose solve gave up is solved again by the one-workgroup kernel enqueued behind it, so a low limit changes the
 * time, not the result. */
#define BLAH2HIP_CLUTTER_OPT_SOLVE_SPIN_LIMIT 7
#define BLAH2HIP_CLUTTER_SOLVE_AUTO 0
#define BLAH2HIP_CLUTTER_SOLVE_STEPWISE 1
#define BLAH2HIP_CLUTTER_SOLVE_LOOKAHEAD 2
int blah2hip_clutter_set_option(blah2hip_clutter_t h, int option, int64_t value);
/* What the last process call launched: the solve's form, indices per lane and workgroups per CPI.  The look-ahead
 * form's workgroups wait for each other with BOUNDED waits; a CPI whose wait ran out (workgroups dispatched late on a chip
 * busy with other work) is solved again by the one-workgroup kernel that every look-ahead launch has behind it, gated per
 * CPI, so ok = 0 always and only means "not positive definite" (WienerHopf.cpp:111-115).  SOLVE_FAULT reads
 * (synchronising) the sticky word set when any wait ran out since the handle was created, SOLVE_RETRIES how many CPIs the
 * gated kernel has solved. */
#define BLAH2HIP_CLUTTER_INFO_SOLVE_FORM 1
#define BLAH2HIP_CLUTTER_INFO_SOLVE_E 2
#define BLAH2HIP_CLUTTER_INFO_SOLVE_G 3
#define BLAH2HIP_CLUTTER_INFO_SOLVE_FAULT 4
#define BLAH2HIP_CLUTTER_INFO_SOLVE_RETRIES 5
int blah2hip_clutter_get_info(blah2hip_clutter_t h, int what, int64_t *value);
/* The filter's Toeplitz solve on its own: n_cpi systems toeplitz(r) w = b given as rb = [n_cpi][2][nBins] complex fp64 (r then b,
 * interleaved re, im; the layout blah2hip_clutter_read_last returns), taps to w ([n_cpi][nBins] complex fp32), ok[c] = 0
 * where the matrix is not positive definite (WienerHopf.cpp:111-115).  Host arrays; synchronises.  For tests and timing of the
 * solve kernels apart from the correlations (set_timing / get_timing: BLAH2HIP_CK_SOLVE). */
int blah2hip_clutter_solve(blah2hip_clutter_t h, const double *rb, uint32_t n_cpi, float *w, int32_t *ok);
/* The same on device-resident arrays; enqueues only. */
int blah2hip_clutter_solve_dev(blah2hip_clutter_t h, const double *d_rb, uint32_t n_cpi, float *d_w, int32_t *d_ok, void *stream);
/* Derived sizes: nBins = delayMax - delayMin taps (WienerHopf.cpp:12), on-chip transform
 * length and samples per overlap-save block. */
int blah2hip_clutter_get_dims(blah2hip_clutter_t h, uint32_t *n_bins, uint32_t *fft_len, uint32_t *seg_len);
/* Copies CPI `cpi` of the last process call's filter to the host (synchronises): the nBins
 * taps w (complex fp32, interleaved), the fp64 correlations the normal equations were built
 * from, r then b (2*nBins complex fp64, interleaved), and the ok flag.  Any output may be NULL.
 * For diagnostics (residual of A w = b) and tests. */
int blah2hip_clutter_read_last(blah2hip_clutter_t h, uint32_t cpi, float *w, double *rb, int *ok);

/* ---- SpectrumAnalyser (SpectrumAnalyser.h:53-62, called blah2.cpp:264) ----
 * decimation = uint32(n/bandwidth), nSpectrum = n/decimation, nfft = nSpectrum*decimation
 * (SpectrumAnalyser.cpp:16-18); spectrum[k] = FFT_nfft(x)[(k*decimation + nfft/2 + 1) mod nfft]
 * (:33-54).  Output: nSpectrum complex fp64 values (re, im interleaved) per CPI.
 * Fails with ERR_INVALID when n < bandwidth (the reference divides by zero) and
 * ERR_UNSUPPORTED when nSpectrum > 4096. */
int blah2hip_spectrum_create(uint32_t n_samples, double bandwidth, int device, uint32_t max_batch,
                             blah2hip_spectrum_t *out);
int blah2hip_spectrum_destroy(blah2hip_spectrum_t h);
int blah2hip_spectrum_get_dims(blah2hip_spectrum_t h, uint32_t *decimation, uint32_t *n_spectrum, uint64_t *nfft);
/* host: the first nfft of the n samples of the reference channel are used (:33-37) */
int blah2hip_spectrum_process_c64(blah2hip_spectrum_t h, const double *x, uint32_t n, double *spectrum_out);
int blah2hip_spectrum_process_c32(blah2hip_spectrum_t h, const float *x, uint32_t n, double *spectrum_out);
/* dev: d_x in format fmt (C32: complex fp32 plane; I16: the interleaved .rspduo
 * buffer, tuner 1 is used; F16: half pairs), d_out [n_cpi][nSpectrum] complex fp64.
 * Enqueues only. */
int blah2hip_spectrum_process_dev(blah2hip_spectrum_t h, int fmt, const void *d_x, uint32_t n_cpi,
                                  uint64_t cpi_stride, double *d_out, void *stream);

/* ---- the clutter filter's FIR fused into the range correlation (round 6; WienerHopf.cpp:124-160 feeding Ambiguity.cpp:106-149)
 * The filtered surveillance channel never crosses HBM: blah2hip_clutter_estimate_dev_fmt runs the filter's correlations,
 * reduction and Toeplitz solve only (the taps stay in the handle, ok flags as in _process_dev); blah2hip_amb_set_fir hands them
 * to an ambiguity handle, whose next blah2hip_amb_process_dev calls (with the UNFILTERED x, y) filter on the fly inside the range
 * kernel (csrc/kernels.hpp range_fir_kernel).  The result is the two-stage result (same linear operations; fp32 rounding in a
 * different order).  Supported where one 4096-point transform covers it -- the handle's transform length is 4096
 * (BLAH2HIP_OPT_FFT_LEN), n_bins <= 2049, at most 2049 delay bins in one chunk, the filter's first lag equal to the map's and
 * <= 0, symmetric Doppler limits, pulses of at least 2048 - delayMin samples, fp32 or int16 samples -- else
 * BLAH2HIP_ERR_UNSUPPORTED at the process call.  d_w = NULL switches back to the plain range kernels.  The ambiguity handle keeps
 * the POINTER: the filter handle must outlive its use, and a process call may not ask for more CPIs than the filter handle's max_batch. */
int blah2hip_clutter_estimate_dev_fmt(blah2hip_clutter_t h, int fmt, const void *d_x, const void *d_y, uint32_t n_cpi,
                                      uint64_t cpi_stride, int32_t *d_ok, void *stream);
/* the handle's taps on the device ([max_batch][n_bins] complex fp32; zero where ok = 0), their count and the filter's first lag */
int blah2hip_clutter_taps_dev(blah2hip_clutter_t h, const float **d_w, uint32_t *n_bins, int32_t *delay_min);
int blah2hip_amb_set_fir(blah2hip_amb_t h, const float *d_w, uint32_t n_bins, int32_t clutter_delay_min);
/* BLAH2HIP_OK if the fused kernel covers this handle with such a filter and sample format, else BLAH2HIP_ERR_UNSUPPORTED (the reason
 * in blah2hip_last_error): lets a caller choose between the fused and the two-stage chain before it enqueues anything */
int blah2hip_amb_fir_fusable(blah2hip_amb_t h, int fmt, uint32_t n_bins, int32_t clutter_delay_min);

/* ---- device context for host bindings ------------------------------------
 * What a host-language binding needs of the HIP runtime to keep a CPI resident on the device across the calls
 * of blah2.cpp:264-287 (Spectrum -> WienerHopf -> Ambiguity -> CFAR on the same x, y) without linking it: one
 * stream, pinned staging memory, device buffers and asynchronous copies.  blah2_amd/host/util/DeviceContext
 * builds the per-process sample cache of the drop-in classes on it. */
typedef struct blah2hip_ctx_s *blah2hip_ctx_t;
int blah2hip_ctx_create(int device, blah2hip_ctx_t *out);
int blah2hip_ctx_destroy(blah2hip_ctx_t c);
void *blah2hip_ctx_stream(blah2hip_ctx_t c);                  /* the hipStream_t the dev entry points take */
int blah2hip_ctx_sync(blah2hip_ctx_t c);                      /* waits for everything enqueued on the stream */
int blah2hip_ctx_malloc(blah2hip_ctx_t c, size_t bytes, void **dptr);
int blah2hip_ctx_free(blah2hip_ctx_t c, void *dptr);
int blah2hip_ctx_malloc_host(blah2hip_ctx_t c, size_t bytes, void **hptr); /* pinned */
int blah2hip_ctx_free_host(blah2hip_ctx_t c, void *hptr);
int blah2hip_ctx_h2d(blah2hip_ctx_t c, void *dptr, const void *hptr, size_t bytes); /* enqueues on the stream */
int blah2hip_ctx_d2h(blah2hip_ctx_t c, void *hptr, const void *dptr, size_t bytes); /* enqueues on the stream */
int blah2hip_ctx_d2d(blah2hip_ctx_t c, void *dst, const void *src, size_t bytes);  /* enqueues on the stream */
/* Measurement helper (no reference counterpart): enqueues a kernel that READS `bytes` of device memory at d_src
 * (16-byte aligned) and does nothing else -- the streaming-read rate of the memory system, which bench.py reports
 * beside the copy rate as the ceiling for the range kernel's loads.  d_sink: 4 device bytes, never written, or NULL. */
int blah2hip_stream_read_dev(const void *d_src, size_t bytes, void *d_sink, void *stream);
/* device pointers of a handle's internal results of the last blah2hip_amb_process_dev with NULL outputs:
 * map [max_batch][n_doppler][n_delay] complex fp32 and metrics [max_batch][2] doubles */
int blah2hip_amb_result_ptrs(blah2hip_amb_t h, const void **d_map, const double **d_metrics);

/* ---- per-kernel timing (HIP events on the launch stream) ---------------- */
#define BLAH2HIP_K_RANGE 0
#define BLAH2HIP_K_DOPPLER 1
#define BLAH2HIP_K_METRICS 2
#define BLAH2HIP_K_CFAR 3     /* cfar1d_kernel, cfar2d_tile_kernel or cfar2d_kernel */
#define BLAH2HIP_K_SAT_ROWS 4 /* 2-D CFAR through the summed-area table (large windows): row prefix sums */
#define BLAH2HIP_K_SAT_COLS 5 /* ... column prefix sums */
#define BLAH2HIP_K_ROTATE 6   /* Doppler-centre shift (asymmetric limits only) */
#define BLAH2HIP_K_COUNT 8
/* enable != 0: every dev call brackets each kernel with hipEvents */
int blah2hip_amb_set_timing(blah2hip_amb_t h, int enable);
/* synchronises the recorded events; ms_total[k] = summed duration of kernel k
 * over launches[k] launches since the last reset; then resets */
int blah2hip_amb_get_timing(blah2hip_amb_t h, double *ms_total, uint32_t *launches);
/* the same for the clutter filter's kernels */
#define BLAH2HIP_CK_CORR 0
#define BLAH2HIP_CK_REDUCE 1
#define BLAH2HIP_CK_SOLVE 2
#define BLAH2HIP_CK_FIR 3
#define BLAH2HIP_CK_COUNT 4
int blah2hip_clutter_set_timing(blah2hip_clutter_t h, int enable);
int blah2hip_clutter_get_timing(blah2hip_clutter_t h, double *ms_total, uint32_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* BLAH2HIP_H */
