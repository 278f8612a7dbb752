// Per-thread pieces of the range (Hot loop A) kernel that are independent of
// the launch machinery, shared by the HIP kernel (ambiguity_kernels.hip) and
// the host emulation test (tests/host/emulate_fft.cpp).
//
// What it computes -- /root/reference/src/process/ambiguity/Ambiguity.cpp:106-149:
// for pulse i and lag d in [delayMin, delayMax]
//     R[i][d] = sum_{n=0}^{nCorr-1} y[i*nCorr + n + d] * conj(x[i*nCorr + n])
// with y taken as zero outside the pulse (the reference zero-pads each pulse to
// nfft >= 2*nCorr-1, :114-118, so neighbouring pulses never contribute).
//
// How: the reference does one nfft-point circular correlation per pulse and
// keeps nDelayBins of the nfft lags.  Only those lags are wanted, so the pulse is
// cut into nSeg segments of segLen samples; for segment s
//     x'[m] = x[s*segLen + m]              m in [0, segLen)      , 0 elsewhere
//     y'[m] = y[s*segLen + delayMin + m]   (0 outside the pulse)
// and with F >= segLen + nDelayBins - 1 the F-point circular correlation
//     c_s[k] = sum_n y'[(n+k) mod F] conj(x'[n]),  k in [0, nDelayBins)
// never wraps, so R[i][delayMin + k] = sum_s c_s[k].  The sum over segments is
// taken in the frequency domain (registers), followed by ONE inverse transform
// per pulse.  The result is the same linear correlation the reference computes,
// i.e. mathematically identical, not an approximation.
#pragma once

#include "fft_wg.hpp"

#include <stdint.h>

namespace blah2 {

B2_HD int min_i(int a, int b) { return a < b ? a : b; }

struct RangePlan {
  int32_t nCorr;     // samples per pulse (Ambiguity.cpp:39)
  int32_t nDoppler;  // pulses per CPI   (Ambiguity.cpp:37)
  int32_t nDelay;    // lags kept        (Ambiguity.cpp:22)
  int32_t delayMin;  // first lag
  int32_t nSeg;      // segments per pulse
  int32_t segLen;    // samples of x per segment
  float scale;       // 1/F (the reference divides by nfft before its backward FFT, :126)
  // The lag window of one launch is a CHUNK of the map's delay axis: lags delayMin .. delayMin + nDelay - 1 land in
  // map columns colOff .. colOff + nDelay - 1 of a range map that is nTilesOut sixteen-column tiles wide.  One chunk with
  // colOff = 0 is the usual case; more than 4081 delay bins, or a window that reaches the lags the reference's nfft-point
  // circular correlation aliases (Ambiguity.cpp:132-146), run as several (capi.hip: lag_chunks).
  int32_t colOff;
  int32_t nTilesOut;
};

// complex fp32 planes: x = reference channel, y = surveillance channel
struct InC32 {
  const cf *x;
  const cf *y;
  B2_HD cf lx(int64_t i) const { return x[i]; }
  B2_HD cf ly(int64_t i) const { return y[i]; }
  B2_HD InC32 at(int64_t i) const { return InC32{x + i, y + i}; } // the same channels, sample i first
};

// .rspduo wire layout: int16 I1 Q1 I2 Q2 per sample pair
// (/root/reference/src/capture/rspduo/RspDuo.cpp:512-526); tuner 1 = reference.
struct InI16 {
  const int16_t *iq;
  B2_HD cf lx(int64_t i) const { return cmake((float)iq[4 * i], (float)iq[4 * i + 1]); }
  B2_HD cf ly(int64_t i) const { return cmake((float)iq[4 * i + 2], (float)iq[4 * i + 3]); }
  B2_HD InI16 at(int64_t i) const { return InI16{iq + 4 * i}; }
};

// Reference channel straight from the .rspduo words, surveillance channel from a complex fp32 plane: what the
// ambiguity stage reads behind the clutter filter of a replay (the filter leaves x untouched and writes the
// filtered y as fp32, WienerHopf.cpp:156-160), without a conversion pass over x.
struct InI16C32 {
  const int16_t *iq;
  const cf *y;
  B2_HD cf lx(int64_t i) const { return cmake((float)iq[4 * i], (float)iq[4 * i + 1]); }
  B2_HD cf ly(int64_t i) const { return y[i]; }
};

// load_seg_* / mask_seg_* below DEFINE the segment windows (clamped index, then a
// select).  The GPU kernels fetch the same windows with raw buffer loads whose
// range check does the zero padding (bufload.hpp); these portable forms are what
// tests/host/emulate_fft.cpp runs on the CPU to check the index algebra.
#if defined(__clang__)
// fp16 IQ storage with fp32 accumulate (BASELINE.json configs[4]): two planes of
// (re, im) half pairs, 4 bytes per sample; every value is widened to fp32 on load
// and all arithmetic stays fp32.
struct InF16 {
  const _Float16 *x;
  const _Float16 *y;
  B2_HD cf lx(int64_t i) const { return cmake((float)x[2 * i], (float)x[2 * i + 1]); }
  B2_HD cf ly(int64_t i) const { return cmake((float)y[2 * i], (float)y[2 * i + 1]); }
};
#endif

template <int R3, class In>
B2_HD void load_seg_x(const In &in, const RangePlan &p, int64_t pulseBase, int s, int t, cf *v)
{
  constexpr int T = 16 * R3;
  const int s0 = s * p.segLen + t;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int idx = s0 + T * k;
    v[k] = in.lx(pulseBase + (idx < p.nCorr ? idx : p.nCorr - 1));
  }
}

template <int R3>
B2_HD void mask_seg_x(const RangePlan &p, int s, int t, cf *v)
{
  constexpr int T = 16 * R3;
  const int s0 = s * p.segLen;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int m = t + T * k;
    const bool ok = (m < p.segLen) && (s0 + m < p.nCorr);
    v[k] = ok ? v[k] : cmake(0.f, 0.f);
  }
}

template <int R3, class In>
B2_HD void load_seg_y(const In &in, const RangePlan &p, int64_t pulseBase, int s, int t, cf *v)
{
  constexpr int T = 16 * R3;
  const int s0 = s * p.segLen + p.delayMin + t;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int idx = s0 + T * k;
    const int cl = idx < 0 ? 0 : (idx < p.nCorr ? idx : p.nCorr - 1);
    v[k] = in.ly(pulseBase + cl);
  }
}

template <int R3>
B2_HD void mask_seg_y(const RangePlan &p, int s, int t, cf *v)
{
  constexpr int T = 16 * R3;
  const int s0 = s * p.segLen + p.delayMin + t;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int idx = s0 + T * k;
    v[k] = (idx >= 0 && idx < p.nCorr) ? v[k] : cmake(0.f, 0.f);
  }
}

B2_HD int64_t rmap_index(int nDoppler, int nTiles, int cpi, int pulse, int lag);

// ---- lag store for a transform with E points per thread and T threads (fft_wg8.hpp: E = 8)
template <int T, int E>
B2_HD void store_lags_g(cf *out, const RangePlan &p, int cpi, int pulse, int t, const cf *v)
{
#pragma unroll
  for (int c = 0; c < E; c++) {
    const int j = t + T * c;
    if (j < p.nDelay)
      out[rmap_index(p.nDoppler, p.nTilesOut, cpi, pulse, j + p.colOff)] = cmake(v[c].x * p.scale, v[c].y * p.scale);
  }
}

// Range map layout in HBM ("tiled"): [cpi][colTile][pulse][16] complex fp32,
// colTile = lag/16.  The range kernel writes 128-byte runs (16 lags of one
// pulse), and the Doppler kernel, which needs whole columns, finds the 16
// columns of a tile for all pulses in one contiguous nDoppler*128-byte block.
B2_HD int64_t rmap_index(int nDoppler, int nTiles, int cpi, int pulse, int lag)
{
  return (((int64_t)cpi * nTiles + (lag >> 4)) * nDoppler + pulse) * 16 + (lag & 15);
}

template <int R3>
B2_HD void store_lags(cf *out, const RangePlan &p, int cpi, int pulse, int t, const cf *v)
{
  constexpr int T = 16 * R3;
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const int j = t + T * c;
    if (j < p.nDelay)
      out[rmap_index(p.nDoppler, p.nTilesOut, cpi, pulse, j + p.colOff)] = cmake(v[c].x * p.scale, v[c].y * p.scale);
  }
}

} // namespace blah2
