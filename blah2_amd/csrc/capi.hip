// C-ABI implementation of include/blah2hip.h: handle management, execution
// planning, table generation and kernel launches.  gfx950 only; links against
// libamdhip64 and nothing else.
#include "kernels.hpp"
#include "cfar_kernels.hpp"
#include "timing.hpp"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

using namespace blah2;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
  g_err = msg;
  return code;
}

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(BLAH2HIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

extern "C" hipError_t blah2hip_ensure_lds_(const void *kern, int bytes);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel): the
// attribute belongs to the function object of the CURRENT device, and a process
// may hold handles on several devices.
#define LDSCFG(kern, bytes)                                                                   \
  do {                                                                                       \
    hipError_t e_ = blah2hip_ensure_lds_((const void *)(kern), (int)(bytes));                 \
    if (e_ != hipSuccess)                                                                    \
      return fail(BLAH2HIP_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e_)); \
  } while (0)

// exp(-2*pi*i*k/n) from the exactly reduced angle, fp64 -> fp32
cf root_of_unity(int64_t k, int64_t n)
{
  k %= n;
  if (k < 0) k += n;
  const double a = -2.0 * M_PI * (double)k / (double)n;
  // octant folding keeps the argument of sin/cos small; fp64 is ample for an fp32 result
  return cmake((float)std::cos(a), (float)std::sin(a));
}

} // namespace

struct blah2hip_amb_s {
  int device = 0;
  blah2hip_amb_dims_t dims{};
  int32_t delayMin = 0, delayMax = 0, dopplerMin = 0, dopplerMax = 0;
  uint32_t fs = 0;
  int r3 = 8;
  RangePlan plan{};                 // segmentation (of the longest chunk) + the first chunk's lag window
  struct LagChunk { int32_t lag0, count, col0; }; // linear lags lag0 .. lag0 + count - 1 -> map columns col0 ..
  std::vector<LagChunk> chunks;     // the delay axis as runs of consecutive LINEAR lags (one chunk in the usual case)
  int32_t maxChunk = 0;
  std::vector<int32_t> delayAxis;
  std::vector<double> dopplerAxis;
  hipStream_t stream = nullptr;
  int numCU = 256;
  int rangeGridForce = 0;           // BLAH2HIP_OPT_RANGE_GRID (0 = the launched kernel's residency)
  int rangeGridLast = 0;            // BLAH2HIP_INFO_RANGE_GRID: workgroup cap of the last launch
  void *h_pin = nullptr;            // pinned host staging of the c64 entry point
  size_t h_pin_bytes = 0;

  cf *d_tw = nullptr;
  cf *d_dopW = nullptr;
  double2 *d_dopW64 = nullptr;   // hot_columns_kernel: exp(-2 pi i k / nD) in fp64
  uint32_t *d_hotCount = nullptr; // [max_batch] columns the last call's hot_columns_kernel rewrote
  cf *d_R = nullptr;
  cf *d_map = nullptr;
  double *d_partSum = nullptr;
  uint32_t *d_tickets = nullptr; // arrival counters of the fused Map::set_metrics finish (doppler_sub1k_kernel), zero between launches
  float *d_partMax = nullptr;
  double *d_metrics = nullptr;
  double *d_doppler = nullptr;
  void *d_in = nullptr; // staging for the host entry points
  size_t d_in_bytes = 0;
  cf *d_rot = nullptr; // rotated reference / converted planes for asymmetric Doppler limits
  blah2hip_hit_t *d_hits = nullptr;
  uint32_t *d_count = nullptr;
  uint32_t hitCap = 0;
  int dopTilesX = 0, dopTilesY = 0; // direct-DFT fallback grid
  int dopR3 = 0;                    // 0 = direct fallback, else Bluestein on WgFft<dopR3>
  int dopGridX = 0;
  int nParts = 0;                   // metrics partials per CPI
  int nTiles = 0;                   // 16-column tiles of the range map
  // execution plan: per handle, fixed at create or through blah2hip_amb_set_option
  int dopForce = BLAH2HIP_DOP_AUTO; // BLAH2HIP_OPT_DOPPLER_KERNEL
  int lastDoppler = 0;              // BLAH2HIP_INFO_LAST_DOPPLER_KERNEL
  int lastRange = 0;                // BLAH2HIP_INFO_LAST_RANGE_KERNEL
  int rangeKernel = 0;              // BLAH2HIP_OPT_RANGE_KERNEL (0 = by transform length)
  int fftLenForce = 0;              // BLAH2HIP_OPT_FFT_LEN (0 = planner)
  int cfar2dForce = 0;              // BLAH2HIP_OPT_CFAR2D_KERNEL
  int dopGridForce = 0;             // BLAH2HIP_OPT_DOPPLER_GRID (0 = residency of the persistent kernel)
  int dopGridLast = 0, dopTilesLast = 0; // BLAH2HIP_INFO_DOPPLER_GRID / _TILES
  struct AlphaTable { double pfa; size_t n; double *d; bool pinned; }; // pinned: handed out by *_prepare, never evicted
  std::vector<AlphaTable> alphaTables; // CFAR threshold factors, one per (pfa, size) seen
  cf *d_dtw = nullptr;              // exp(-2 pi i k/M)
  cf *d_chirp = nullptr;            // exp(-i pi n^2/nD)
  cf *d_bf = nullptr;               // chirp-kernel spectrum / M in register layout
  cf *d_bfn = nullptr;              // the same in natural order (M = 2048 only: doppler_tilew_kernel)
  double *d_sat = nullptr;          // 2-D CFAR summed-area table [max_batch][nD+1][nDelay+1]

  KernelTimer<BLAH2HIP_K_COUNT> timer;

  // the clutter filter's FIR fused into the range correlation (blah2hip_amb_set_fir): the filter handle's taps
  const cf *firW = nullptr; // [max_batch][firBins]; nullptr: the plain range kernels
  int firBins = 0, firDmin = 0;
  cf *d_H = nullptr;        // [max_batch][16][256]: the taps' spectrum in the transform's register layout (taps_spectrum_kernel)
  int32_t *d_firK0 = nullptr; // [max_batch]: the largest tap's index (left out of d_H, applied in the time domain)

  // Fixed-pattern leak compensation (see leak_calibrate): per (range kernel, Doppler kernel) the lags of the zero-Doppler row
  // into which the fp32 transform chain leaks a fixed fraction g of the lag-0 cell, measured once on a synthetic CPI
  struct LeakCal { int nLags = 0; double maxAbs = 0.0; bool active = false; int32_t *d_lag = nullptr; cf *d_g = nullptr; };
  std::map<int, LeakCal> leak;      // key = range kernel id * 64 + Doppler kernel id
  int leakMode = 1;                 // BLAH2HIP_OPT_LEAK_COMPENSATION: 0 off, 1 auto (applied where it can reach 3e-5 of the mean level), 2 always
  bool inLeakCal = false;
  int hotMode = 1;                  // BLAH2HIP_OPT_HOT_COLUMNS: 0 off, 1 auto (CPIs long enough for a peak HOT_RATIO above the mean level), 2 always
  bool lastHot = false;             // the last process call launched hot_columns_kernel
  hipStream_t lastHotStream = nullptr; // ... on this stream (BLAH2HIP_INFO_HOT_COLUMNS waits for that one only)
  int lastLeakLags = 0;             // BLAH2HIP_INFO_LEAK_LAGS: lags corrected by the last process call (0 = not applied)
  double lastLeakMax = 0.0;         // BLAH2HIP_INFO_LEAK_MAX_E12: the largest |g| of the calibration the last call ran under
};

namespace {

// ---------------------------------------------------------------- planning --
// Ambiguity::Ambiguity, Ambiguity.cpp:11-82 (same fp64 expressions, same
// uint16 narrowing of nDelayBins / nDopplerBins / nCorr, Ambiguity.h:80-89)
void derive_dims(blah2hip_amb_s *h, uint32_t n, bool roundHamming, uint32_t nDopplerExplicit)
{
  auto &d = h->dims;
  d.n_samples = n;
  d.n_delay_bins = (uint16_t)(h->delayMax - h->delayMin + 1);
  d.doppler_middle = (h->dopplerMin + h->dopplerMax) / 2.0;
  std::deque<double> doppler;
  double res = 1.0 / ((double)n / (double)h->fs);
  doppler.push_back(d.doppler_middle);
  int i = 1;
  while (d.doppler_middle + (i * res) <= h->dopplerMax) {
    doppler.push_back(d.doppler_middle + (i * res));
    doppler.push_front(d.doppler_middle - (i * res));
    i++;
  }
  d.n_doppler_bins = (uint16_t)doppler.size();
  // extension: an explicit bin count, e.g. exactly 512; everything below is the
  // reference's arithmetic with that count, including the shift (j + nD/2 + 1) % nD of :165
  if (nDopplerExplicit) d.n_doppler_bins = (uint16_t)nDopplerExplicit;
  d.n_corr = (uint16_t)(n / d.n_doppler_bins);
  d.cpi = ((double)d.n_corr * d.n_doppler_bins) / h->fs;
  res = 1.0 / d.cpi;
  h->delayAxis.resize(d.n_delay_bins);
  for (uint32_t j = 0; j < d.n_delay_bins; j++) h->delayAxis[j] = h->delayMin + (int32_t)j;
  std::deque<double> ax;
  ax.push_front(d.doppler_middle);
  i = 1;
  while (ax.size() < d.n_doppler_bins) {
    ax.push_back(d.doppler_middle + (i * res));
    if (ax.size() < d.n_doppler_bins) ax.push_front(d.doppler_middle - (i * res)); // even counts: the extra bin is the last one
    i++;
  }
  h->dopplerAxis.assign(ax.begin(), ax.end());
  d.nfft = 2 * d.n_corr - 1;
  if (roundHamming) d.nfft = blah2hip_next_hamming(d.nfft);
  d.n_used = d.n_corr * d.n_doppler_bins;
}

// The delay axis as runs of consecutive linear lags.  Ambiguity.cpp:132-146 gathers delay d from index d mod nfft of an
// nfft-point CIRCULAR correlation of nCorr-sample pulses: c[L] = R[L] for L < nCorr, R[L - nfft] for L > nfft - nCorr (the
// opposite-sign lag aliasing in; nfft >= 2 nCorr - 1, so never both), 0 in between.  This engine computes linear
// correlations of a lag window, so every delay maps to ONE linear lag l(d); l(d) is consecutive in d except where the
// circular index crosses nfft - nCorr, and windows inside |d| <= nfft - nCorr (every practical one) are a single run
// with l(d) = d.  Runs longer than `cap` lags are cut into chunks (multiples of 16 lags, so that the chunks of a run
// keep whole 128-byte runs of the range map): that is how more than 4081 delay bins run on the 4096-point transform.
void lag_chunks(blah2hip_amb_s *h, int cap)
{
  const int64_t nfft = h->dims.nfft, nCorr = h->dims.n_corr;
  h->chunks.clear();
  auto lin = [&](int64_t d) {
    int64_t L = d % nfft;
    if (L < 0) L += nfft;
    return (L > nfft - nCorr) ? L - nfft : L;
  };
  const int32_t nDelay = (int32_t)h->dims.n_delay_bins;
  int32_t col = 0;
  while (col < nDelay) {
    const int64_t l0 = lin((int64_t)h->delayMin + col);
    int32_t len = 1;
    while (col + len < nDelay && lin((int64_t)h->delayMin + col + len) == l0 + len) len++;
    for (int32_t o = 0; o < len; o += cap) h->chunks.push_back({(int32_t)(l0 + o), std::min<int32_t>(cap, len - o), col + o});
    col += len;
  }
  h->maxChunk = 0;
  for (const auto &c : h->chunks) h->maxChunk = std::max(h->maxChunk, c.count);
}

// pick F = 256*R3 and the segmentation minimising (2*nSeg+1) * F*log2(F)
bool choose_plan(blah2hip_amb_s *h)
{
  const int nCorr = h->dims.n_corr, nDelay = h->maxChunk; // the longest chunk decides; shorter ones reuse its segmentation
  const int forced = h->fftLenForce;
  double best = 1e300;
  bool found = false;
  for (int r3 : {4, 8, 16}) {
    const int F = 256 * r3;
    if (forced && F != forced) continue;
    const int lmax = F - nDelay + 1;
    if (lmax < 16) continue;
    const int nSeg = (nCorr + lmax - 1) / lmax;
    int segLen = (nCorr + nSeg - 1) / nSeg;
    // F = 1024: segments of exactly 9*64 samples when that costs no extra segment -- consecutive y' windows then overlap by
    // whole registers of the one-wave kernel, which carries them over instead of reading them again (kernels.hpp: REUSE)
    if (r3 == 4 && lmax >= 9 * 64 && (nCorr + 9 * 64 - 1) / (9 * 64) == nSeg) segLen = 9 * 64;
    // measured on MI355X at equal butterfly count: the F = 1024 workgroup kernel (8 points per thread) is ~3 % slower than
    // the F = 2048 kernels (round 2, forced lengths at three geometries); the F = 1024 ONE-WAVE kernel, which runs once a
    // launch has a pulse for each of its wave slots, is 1-2 % faster than the F = 2048 one-wave kernel (round 3, cfg 2 x 128
    // on four boxes, fp32 input; equal for int16)
    const bool w1k = (int64_t)h->dims.max_batch * h->dims.n_doppler_bins >= (int64_t)4 * RANGEW1K_WAVES_PER_SIMD * h->numCU;
    // handles whose largest launch stays below that (a lone CPI) run F = 1024 on the pulse-per-workgroup kernel when a
    // pulse has at most two segments per wave of it: cfg 2, one CPI per launch: 19.0 vs 21.7 us on the F = 2048 workgroup
    // kernel (round 4, the eight-wave form; the four-wave form of round 5 runs any segment count when forced)
    constexpr int RANGEPS_SEG = 2 * RANGEPS_WAVES;
    const bool ps = r3 == 4 && nSeg <= RANGEPS_SEG && (int64_t)h->dims.max_batch * h->dims.n_doppler_bins <= (int64_t)4 * h->numCU;
    const double cost = (2.0 * nSeg + 1.0) * F * std::log2((double)F) * (r3 == 4 ? (w1k ? 0.985 : (ps ? 0.6 : 1.03)) : 1.0);
    if (cost < best) {
      best = cost;
      found = true;
      h->r3 = r3;
      h->plan.nSeg = nSeg;
      h->plan.segLen = segLen;
      h->plan.scale = 1.0f / (float)F;
    }
  }
  if (!found) return false;
  h->plan.nCorr = nCorr;
  h->plan.nDoppler = h->dims.n_doppler_bins;
  h->plan.nDelay = h->chunks[0].count;
  h->plan.delayMin = h->chunks[0].lag0;
  h->plan.colOff = h->chunks[0].col0;
  h->plan.nTilesOut = (int32_t)((h->dims.n_delay_bins + 15) / 16);
  h->dims.fft_len = 256 * h->r3;
  h->dims.n_seg = h->plan.nSeg;
  h->dims.seg_len = h->plan.segLen;
  return true;
}


// fp64 radix-2 FFT on the host, used once per handle for the chirp-kernel spectrum
void host_fft(std::vector<std::complex<double>> &a)
{
  const size_t n = a.size();
  for (size_t i = 1, j = 0; i < n; i++) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < len / 2; k++) {
        const double ang = -2.0 * M_PI * (double)k / (double)len;
        const std::complex<double> w(std::cos(ang), std::sin(ang));
        const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
        a[i + k] = u + v;
        a[i + k + len / 2] = u - v;
      }
  }
}

// Bluestein tables for the Doppler DFT of length nD on an M = 256*r3 point FFT
void doppler_tables(int nD, int r3, std::vector<cf> &tw, std::vector<cf> &chirp, std::vector<cf> &bf, std::vector<cf> &bfn)
{
  const int M = 256 * r3, T = 16 * r3;
  tw.resize(M);
  for (int k = 0; k < M; k++) tw[k] = root_of_unity(k, M);
  std::vector<std::complex<double>> c(nD);
  chirp.resize(nD);
  for (int64_t n = 0; n < nD; n++) {
    const int64_t q = (n * n) % (2 * (int64_t)nD); // exp(-i pi n^2/nD) = exp(-2 pi i q/(2 nD))
    const double ang = -M_PI * (double)q / (double)nD;
    c[n] = {std::cos(ang), std::sin(ang)};
    chirp[n] = cmake((float)c[n].real(), (float)c[n].imag());
  }
  std::vector<std::complex<double>> b(M, {0.0, 0.0});
  b[0] = std::conj(c[0]);
  for (int m = 1; m < nD; m++) b[m] = b[M - m] = std::conj(c[m]);
  host_fft(b);
  bf.resize(M);
  for (int t = 0; t < T; t++)
    for (int j = 0; j < 16 / r3; j++)
      for (int sidx = 0; sidx < r3; sidx++) {
        const int p = t + T * j, q = p / 16, r = p % 16;
        const int m = q + 16 * r + 256 * sidx;
        const int e = j * r3 + sidx;
        const std::complex<double> v = b[m] / (double)M;
        bf[(size_t)e * T + t] = cmake((float)v.real(), (float)v.imag());
      }
  bfn.resize(M);
  for (int m = 0; m < M; m++) bfn[m] = cmake((float)(b[m].real() / M), (float)(b[m].imag() / M));
}

template <int R3> int launch_doppler_t(blah2hip_amb_s *h, const DopplerArgs &a, uint32_t n_cpi, hipStream_t st)
{
  using W = WgFft<R3>;
  const size_t lds = (size_t)(R3 == 4 ? W::A_ELEMS : W::A_ELEMS + W::B_ELEMS) * sizeof(cf);
  auto kern = doppler_fft_kernel<R3>;
  LDSCFG(kern, lds);
  hipLaunchKernelGGL(kern, dim3(h->dopGridX, n_cpi), dim3(W::T), lds, st, a);
  HIPCHK(hipGetLastError());
  return BLAH2HIP_OK;
}

// F = 2048 runs on the one-wave kernel (measured, round 2, cfg 2 x 128 on one box, steady clocks:
// 1.329 vs 1.385 ms per step of the whole chain) once a launch has a pulse for every wave slot of the
// chip; smaller launches -- a single CPI of the real-time path: 513 pulses -- keep the workgroup
// kernel, where two waves share a pulse (one CPI: range 22.7 vs 34.7 us).
bool use_wave_range(const blah2hip_amb_s *h, int nPulses)
{
  if (h->r3 != 8 || h->rangeKernel == BLAH2HIP_RANGE_E16) return false;
  if (h->rangeKernel == BLAH2HIP_RANGE_WAVE) return true;
  return nPulses >= 4 * RANGEW_WAVES_PER_SIMD * h->numCU;
}

// Workgroups of a range kernel that fit a CU (LDS and registers): the grid of a launch is capped
// there and the kernel walks the pulses with a grid stride.
int range_grid_cap(blah2hip_amb_s *h, size_t lds, int wavesPerWg, int wavesPerCU)
{
  const int perCU = std::max(1, std::min((int)((160 * 1024) / lds), wavesPerCU / std::max(1, wavesPerWg)));
  h->rangeGridLast = h->rangeGridForce ? h->rangeGridForce : perCU * h->numCU;
  return h->rangeGridLast;
}

template <int R3, class In> int launch_range_t(blah2hip_amb_s *h, const RangeArgs &a, In in, hipStream_t st)
{
  using W = WgFft<R3>;
  const size_t lds = (size_t)(W::A_ELEMS + W::B_ELEMS) * sizeof(cf);
  auto kern = range_kernel<R3, In>;
  LDSCFG(kern, lds);
  const int grid = std::min<int>(a.nPulses, range_grid_cap(h, lds, R3 / 4, 8)); // 196-230 VGPRs: 2 waves per SIMD
  hipLaunchKernelGGL(kern, dim3(grid), dim3(W::T), lds, st, a, in);
  HIPCHK(hipGetLastError());
  h->lastRange = BLAH2HIP_RANGE_E16;
  return BLAH2HIP_OK;
}

template <int R4, class In> int launch_range8_t(blah2hip_amb_s *h, const RangeArgs &a, In in, hipStream_t st)
{
  using W = WgFft8<R4>;
  const size_t lds = (size_t)2 * W::BUF_ELEMS * sizeof(cf);
  auto kern = range8_kernel<R4, In>;
  LDSCFG(kern, lds);
  const int grid = std::min<int>(a.nPulses, range_grid_cap(h, lds, R4, 4 * RANGE8_WAVES_PER_SIMD));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(W::T), lds, st, a, in);
  HIPCHK(hipGetLastError());
  h->lastRange = BLAH2HIP_RANGE_E8;
  return BLAH2HIP_OK;
}

template <class In> int launch_rangew_t(blah2hip_amb_s *h, const RangeArgs &a, In in, hipStream_t st)
{
  const size_t lds = (size_t)(WaveFft::TW_ELEMS + RANGEW_WAVES * WaveFft::X_ELEMS) * sizeof(cf);
  const bool shortw = a.plan.segLen <= 24 * 64 && a.plan.segLen + a.plan.nDelay - 1 <= 28 * 64 && a.plan.nDelay <= 7 * 64;
  const bool out7 = a.plan.nDelay <= 7 * 64;
  auto kern = shortw ? rangew_kernel<In, true, true> : (out7 ? rangew_kernel<In, false, true> : rangew_kernel<In, false, false>);
  LDSCFG(kern, lds);
  const int grid = std::min<int>((a.nPulses + RANGEW_WAVES - 1) / RANGEW_WAVES, range_grid_cap(h, lds, RANGEW_WAVES, 4 * RANGEW_WAVES_PER_SIMD));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * RANGEW_WAVES), lds, st, a, in);
  HIPCHK(hipGetLastError());
  h->lastRange = BLAH2HIP_RANGE_WAVE;
  return BLAH2HIP_OK;
}

// F = 1024: 8 points per thread (one-wave transforms: 10.5 vs 14.3 us/CPI at cfg 2 with 16 points per
// thread); F = 2048 / 4096: 16 points per thread (10.0 vs 10.3 us/CPI; equal at 4096).  Measured, round 1.
// F = 1024: 8 points per thread with stage 4 across lanes (3 waves per SIMD); F = 2048 / 4096: 16 points
// per thread.  Measured in round 2 (range kernel, us per launch): F = 1024 (tests/golden `medium`
// geometry x 1024 CPIs) 362 with the lane form, 446 with the LDS form of stage 4 it replaced;
// F = 2048 (cfg 2 x 128): 16-point 1192, 8-point lane form 1363-1372 (4 or 3 waves per SIMD); F = 4096
// (cfg 3 x 8): 568 vs 707.  The 8-point transform executes 26 % more VALU instructions per point
// (radix 8-8-8-4 twiddles + the lane butterflies) and the kernel follows that count, not its occupancy.
// F = 4096 on the two-wave kernel

// F = 1024 on the one-wave kernel with 16 points per lane (four waves per SIMD)
bool use_wave1k_range(const blah2hip_amb_s *h, int nPulses)
{
  if (h->r3 != 4 || h->rangeKernel == BLAH2HIP_RANGE_E8) return false;
  if (h->rangeKernel == BLAH2HIP_RANGE_WAVE1K) return true;
  return nPulses >= 4 * RANGEW1K_WAVES_PER_SIMD * h->numCU;
}

#ifdef B2_RANGEW1K_GLDS
// the LDS-DMA experiment build: fp32 planes with an even first lag run rangew1k_glds_kernel instead (kernels.hpp)
inline bool launch_rangew1k_glds(blah2hip_amb_s *h, const RangeArgs &a, InC32 in, hipStream_t st, int *rc)
{
  if (a.plan.delayMin & 1) return false;
  const size_t lds = (size_t)(Wave1kFft::TW_ELEMS + RANGEG_WAVES * Wave1kFft::X_ELEMS) * sizeof(cf);
  const bool shortx = a.plan.segLen <= 9 * 64, out7 = a.plan.nDelay <= 7 * 64;
  auto kern = shortx ? (out7 ? rangew1k_glds_kernel<true, true> : rangew1k_glds_kernel<true, false>)
                     : (out7 ? rangew1k_glds_kernel<false, true> : rangew1k_glds_kernel<false, false>);
  *rc = BLAH2HIP_OK;
  if (blah2hip_ensure_lds_((const void *)kern, (int)lds) != hipSuccess) { *rc = fail(BLAH2HIP_ERR_HIP, "LDS size (glds experiment)"); return true; }
  const int grid = std::min<int>((a.nPulses + RANGEG_WAVES - 1) / RANGEG_WAVES, range_grid_cap(h, lds, RANGEG_WAVES, 16));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * RANGEG_WAVES), lds, st, a, in);
  if (hipGetLastError() != hipSuccess) *rc = fail(BLAH2HIP_ERR_HIP, "launch (glds experiment)");
  h->lastRange = BLAH2HIP_RANGE_WAVE1K;
  return true;
}
template <class In> inline bool launch_rangew1k_glds(blah2hip_amb_s *, const RangeArgs &, In, hipStream_t, int *) { return false; }
#endif

template <class In> int launch_rangew1k_t(blah2hip_amb_s *h, const RangeArgs &a, In in, hipStream_t st)
{
#ifdef B2_RANGEW1K_GLDS
  { int rc_ = 0; if (launch_rangew1k_glds(h, a, in, st, &rc_)) return rc_; }
#endif
  const size_t lds = (size_t)(Wave1kFft::TW_ELEMS + RANGEW1K_WAVES * Wave1kFft::X_ELEMS) * sizeof(cf);
  const bool shortx = a.plan.segLen <= 9 * 64;
  const bool out7 = a.plan.nDelay <= 7 * 64;
  const bool reuse = a.plan.segLen == 9 * 64 && a.plan.nDelay <= 7 * 64 + 1; // whole-register overlap of consecutive y' windows
  auto kern = reuse ? (out7 ? rangew1k_kernel<In, true, true, true> : rangew1k_kernel<In, true, false, true>)
              : shortx ? (out7 ? rangew1k_kernel<In, true, true> : rangew1k_kernel<In, true, false>)
                       : (out7 ? rangew1k_kernel<In, false, true> : rangew1k_kernel<In, false, false>);
  LDSCFG(kern, lds);
  const int grid = std::min<int>((a.nPulses + RANGEW1K_WAVES - 1) / RANGEW1K_WAVES, range_grid_cap(h, lds, RANGEW1K_WAVES, 4 * RANGEW1K_WAVES_PER_SIMD));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * RANGEW1K_WAVES), lds, st, a, in);
  HIPCHK(hipGetLastError());
  h->lastRange = BLAH2HIP_RANGE_WAVE1K;
  return BLAH2HIP_OK;
}

// F = 1024, a launch that does not fill the one-wave kernel's slots (a lone CPI: 513 pulses): a workgroup of four waves
// per pulse, its segments dealt round-robin to the waves (rangeps_kernel)
bool use_ps_range(const blah2hip_amb_s *h, int nPulses)
{
  if (h->r3 != 4) return false;
  if (h->rangeKernel == BLAH2HIP_RANGE_PS) return true;
  // automatic: below two rounds of its resident workgroups, and at most two segments per wave -- the shape the planner's
  // cost factor and every timing of this kernel assume (a forced kernel walks any segment count; tested up to 13)
  return h->rangeKernel == 0 && nPulses <= 4 * h->numCU && h->plan.nSeg <= 2 * RANGEPS_WAVES;
}

template <class In> int launch_rangeps_t(blah2hip_amb_s *h, const RangeArgs &a, In in, hipStream_t st)
{
  const size_t lds = (size_t)(Wave1kFft::TW_ELEMS + RANGEPS_WAVES * Wave1kFft::X_ELEMS) * sizeof(cf);
  const bool shortx = a.plan.segLen <= 9 * 64;
  const bool out7 = a.plan.nDelay <= 7 * 64;
  auto kern = shortx ? (out7 ? rangeps_kernel<In, true, true> : rangeps_kernel<In, true, false>)
                     : (out7 ? rangeps_kernel<In, false, true> : rangeps_kernel<In, false, false>);
  LDSCFG(kern, lds);
  const int grid = std::min<int>(a.nPulses, range_grid_cap(h, lds, RANGEPS_WAVES, 12)); // 168 VGPRs: three waves per SIMD
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * RANGEPS_WAVES), lds, st, a, in);
  HIPCHK(hipGetLastError());
  h->lastRange = BLAH2HIP_RANGE_PS;
  return BLAH2HIP_OK;
}

template <class In> int launch_range(blah2hip_amb_s *h, const RangeArgs &a, In in, hipStream_t st)
{
  if (use_wave_range(h, a.nPulses)) return launch_rangew_t(h, a, in, st);
  if (use_wave1k_range(h, a.nPulses)) return launch_rangew1k_t(h, a, in, st);
  if (use_ps_range(h, a.nPulses)) return launch_rangeps_t(h, a, in, st);
  switch (h->r3) {
  case 4: return launch_range8_t<2>(h, a, in, st);
  case 8: return launch_range_t<8>(h, a, in, st);
  default: return launch_range_t<16>(h, a, in, st);
  }
}

// exp(-2 pi i k / F) for the handle's current plan (create, and again when BLAH2HIP_OPT_FFT_LEN re-plans)
int upload_range_table(blah2hip_amb_s *h)
{
  const int F = (int)h->dims.fft_len;
  std::vector<cf> tw(F);
  for (int k = 0; k < F; k++) tw[k] = root_of_unity(k, F);
  if (h->d_tw) HIPCHK(hipFree(h->d_tw));
  h->d_tw = nullptr;
  HIPCHK(hipMalloc(&h->d_tw, F * sizeof(cf)));
  HIPCHK(hipMemcpy(h->d_tw, tw.data(), F * sizeof(cf), hipMemcpyHostToDevice));
  return BLAH2HIP_OK;
}

int tic(blah2hip_amb_s *h, int k, hipStream_t st)
{
  HIPCHK(h->timer.tic(k, st));
  return BLAH2HIP_OK;
}

int toc(blah2hip_amb_s *h, int k, hipStream_t st)
{
  HIPCHK(h->timer.toc(k, st));
  return BLAH2HIP_OK;
}

int ensure_staging(blah2hip_amb_s *h, size_t bytes)
{
  if (h->d_in_bytes >= bytes) return BLAH2HIP_OK;
  if (h->d_in) HIPCHK(hipFree(h->d_in));
  h->d_in = nullptr;
  h->d_in_bytes = 0;
  HIPCHK(hipMalloc(&h->d_in, bytes));
  h->d_in_bytes = bytes;
  return BLAH2HIP_OK;
}

int host_tail(blah2hip_amb_s *h, float *map_out, double *metrics)
{
  const size_t cells = (size_t)h->dims.n_doppler_bins * h->dims.n_delay_bins;
  if (map_out) HIPCHK(hipMemcpyAsync(map_out, h->d_map, cells * sizeof(cf), hipMemcpyDeviceToHost, h->stream));
  if (metrics) HIPCHK(hipMemcpyAsync(metrics, h->d_metrics, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return BLAH2HIP_OK;
}

// Which Doppler kernel a launch of n_cpi CPIs runs.  The tile kernels (coalesced tile
// reads, a 512/1024-thread workgroup per 8/16/4 columns) win once a launch carries enough
// tiles to fill the chip; small launches (single CPI) keep the per-column kernel.
bool doppler_kernel_applicable(const blah2hip_amb_s *h, int which)
{
  const int nD = (int)h->dims.n_doppler_bins;
  switch (which) {
  case BLAH2HIP_DOP_TILE8:
  case BLAH2HIP_DOP_TILE8K:
  case BLAH2HIP_DOP_TILE16WG:
  case BLAH2HIP_DOP_SUB4:
  case BLAH2HIP_DOP_TILE16: return h->dopR3 == 4;
  case BLAH2HIP_DOP_TILEM: return (h->dopR3 == 8 && nD <= DopM<8>::MAX_ND) || (h->dopR3 == 16 && nD <= DopM<16>::MAX_ND);
  case BLAH2HIP_DOP_TILEW: return h->dopR3 == 8 && nD <= DOPW_MAX_ND;
  case BLAH2HIP_DOP_TILEW2: return h->dopR3 == 16 && nD <= DOPW2_MAX_ND;
  case BLAH2HIP_DOP_TILEW4: return h->dopR3 == 16 && nD <= DOPW4_MAX_ND;
  case BLAH2HIP_DOP_COLUMN: return h->dopR3 != 0;
  case BLAH2HIP_DOP_DIRECT: return true;
  default: return false;
  }
}

int pick_doppler(const blah2hip_amb_s *h, uint32_t n_cpi)
{
  if (h->dopForce != BLAH2HIP_DOP_AUTO) return h->dopForce;
  if (!h->dopR3) return BLAH2HIP_DOP_DIRECT;
  const int nDelay = (int)h->dims.n_delay_bins;
  const int ncol = h->dopR3 == 4 ? 8 : (h->dopR3 == 8 ? DopM<8>::NCOL : DopM<16>::NCOL);
  const int tiles = (nDelay + ncol - 1) / ncol;
  const bool fills = (int)n_cpi * tiles >= h->numCU / 2;
  // nD <= 513: whole 16-column tiles (128-byte row pieces, one persistent workgroup per CU) once a launch has
  // a tile per CU (cfg 2 x 128: 1.49 vs 1.59 us/CPI); 8-column half tiles, two workgroups per CU, below
  if (h->dopR3 == 4 && (int)n_cpi * ((nDelay + 15) / 16) >= h->numCU) return BLAH2HIP_DOP_TILE16;
  if (fills && h->dopR3 == 4) return BLAH2HIP_DOP_TILE8;
  if (h->dopR3 == 4) return BLAH2HIP_DOP_SUB4; // a lone CPI: every CU gets a 4-column piece
  if (fills && doppler_kernel_applicable(h, BLAH2HIP_DOP_TILEW)) return BLAH2HIP_DOP_TILEW;
  if (fills && doppler_kernel_applicable(h, BLAH2HIP_DOP_TILEW2)) return BLAH2HIP_DOP_TILEW2;
  if (fills && doppler_kernel_applicable(h, BLAH2HIP_DOP_TILEM)) return BLAH2HIP_DOP_TILEM;
  return BLAH2HIP_DOP_COLUMN;
}

// CFAR threshold factors alpha[n] = n*(pfa^(-1/n) - 1), n = 1..maxN, evaluated with the
// same libm pow the reference calls (CfarDetector1D.cpp:76).  A small per-handle cache keyed by
// (pfa, maxN), least recently used first out (a caller that adapts pfa per CPI does not grow device
// memory): a hit only hands out the pointer; a miss allocates and uploads (blocking), which is
// refused while `st` is being captured into a graph -- *_prepare is the call for that.  A table *_prepare has handed out
// lives as long as the handle (`pin`): a graph captured afterwards has its address baked in, and no synchronisation
// at eviction time protects a later replay.  Only unpinned tables are evicted.
constexpr size_t ALPHA_TABLES_MAX = 8;
int alpha_table(blah2hip_amb_s *h, double pfa, size_t maxN, const double **out, hipStream_t st = nullptr, bool pin = false)
{
  auto &T = h->alphaTables;
  for (size_t i = 0; i < T.size(); i++)
    if (T[i].pfa == pfa && T[i].n >= maxN) {
      blah2hip_amb_s::AlphaTable t = T[i];
      t.pinned = t.pinned || pin;
      T.erase(T.begin() + (long)i);
      T.push_back(t); // most recently used last
      *out = t.d;
      return BLAH2HIP_OK;
    }
  if (st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
      return fail(BLAH2HIP_ERR_INVALID, "threshold table of this (pfa, window) is not resident: call blah2hip_cfar1d_prepare / "
                                        "blah2hip_cfar2d_prepare before capturing the stream");
  }
  std::vector<double> alpha(maxN + 1);
  alpha[0] = std::nan("");
  for (size_t n = 1; n <= maxN; n++) alpha[n] = (double)n * (pow(pfa, -1.0 / (double)n) - 1);
  size_t unpinned = 0;
  for (const auto &e : T) unpinned += e.pinned ? 0 : 1;
  if (unpinned >= ALPHA_TABLES_MAX) {
    HIPCHK(hipDeviceSynchronize()); // the oldest unpinned table may still be read by enqueued work
    for (size_t i = 0; i < T.size(); i++)
      if (!T[i].pinned) {
        (void)hipFree(T[i].d);
        T.erase(T.begin() + (long)i);
        break;
      }
  }
  blah2hip_amb_s::AlphaTable t{pfa, maxN, nullptr, pin};
  HIPCHK(hipMalloc(&t.d, (maxN + 1) * sizeof(double)));
  hipError_t e = hipMemcpy(t.d, alpha.data(), (maxN + 1) * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(t.d); return fail(BLAH2HIP_ERR_HIP, std::string("alpha table upload: ") + hipGetErrorString(e)); }
  T.push_back(t);
  *out = t.d;
  return BLAH2HIP_OK;
}

// the one-pass tile kernel covers windows whose halo fits its LDS budget; larger ones take the summed-area table
bool cfar2d_use_tile(const blah2hip_amb_s *h, int ngd, int ntd, int ngf, int ntf)
{
  if (h->cfar2dForce == BLAH2HIP_CFAR2D_SAT) return false;
  return ngf + ntf <= C2T_MAX_HR && ngd + ntd <= C2T_MAX_HC;
}

// the stream kernel exists for the window shapes of C2S_SHAPES (cfar_kernels.hpp)
bool cfar2d_stream_shape(int ngd, int ntd, int ngf, int ntf)
{
#define B2_X(TD, GD, TF, GF) if (ntd == TD && ngd == GD && ntf == TF && ngf == GF) return true;
  C2S_SHAPES(B2_X)
#undef B2_X
  return false;
}
// The rows the detector skips (|doppler| < minDoppler, CfarDetector1D.cpp:40) as ONE interval [lo, hi) of the handle's Doppler axis
// (monotonic: Ambiguity.cpp:60-66), which the stream kernel tests with scalar compares; false if they are not one interval.
bool cfar2d_dead_rows(const blah2hip_amb_s *h, double min_doppler, int *lo, int *hi)
{
  const int nD = (int)h->dims.n_doppler_bins;
  int first = nD, last = -1, count = 0;
  for (int i = 0; i < nD; i++)
    if (std::fabs(h->dopplerAxis[i]) < min_doppler) { first = std::min(first, i); last = i; count++; }
  if (count == 0) { *lo = *hi = 0; return true; }
  *lo = first; *hi = last + 1;
  return count == last + 1 - first;
}
bool cfar2d_use_stream(const blah2hip_amb_s *h, int ngd, int ntd, int ngf, int ntf, double min_doppler)
{
  if (h->cfar2dForce == BLAH2HIP_CFAR2D_SAT || h->cfar2dForce == BLAH2HIP_CFAR2D_TILE) return false;
  int lo, hi;
  return cfar2d_stream_shape(ngd, ntd, ngf, ntf) && cfar2d_dead_rows(h, min_doppler, &lo, &hi);
}

// rows per segment: the fewest rounds of (segment + its 2 hR halo rows) over the wave slots of the chip
void cfar2d_stream_launch(const blah2hip_amb_s *h, const Cfar2dArgs &a, uint32_t n_cpi, hipStream_t st)
{
  Cfar2dStreamArgs ta;
  ta.d = a;
  ta.nCpi = (int32_t)n_cpi;
  const int hC = a.ngD + a.ntD, hR = a.ngF + a.ntF;
  ta.strips = (a.nDelay + (64 - 2 * hC) - 1) / (64 - 2 * hC);
  const int64_t slots = (int64_t)h->numCU * 4 * 5; // about five waves per SIMD at ~ 100 VGPRs
  int64_t best = INT64_MAX;
  for (int R : {8, 16, 32, 64, 128, 256, 512, 1024, (int)a.nD}) { // a lone CPI: short segments, every SIMD a wave
    if (R > a.nD) continue;
    const int64_t segs = (a.nD + R - 1) / R, tasks = segs * ta.strips * n_cpi;
    const int64_t cost = ((tasks + slots - 1) / slots) * (std::min(R, (int)a.nD) + 2 * hR + 16); // 16: start-up of a wave, in rows
    if (cost < best) { best = cost; ta.rowsPerSeg = std::min(R, (int)a.nD); ta.segs = (int32_t)segs; }
  }
  ta.nTasks = ta.nCpi * ta.segs * ta.strips;
  cfar2d_dead_rows(h, a.minDoppler, &ta.deadLo, &ta.deadHi);
  const int grid = (((ta.nTasks + 3) / 4) + 7) & ~7;
#define B2_X(TD, GD, TF, GF) \
  if (a.ntD == TD && a.ngD == GD && a.ntF == TF && a.ngF == GF) { \
    hipLaunchKernelGGL((cfar2d_stream_kernel<TD, GD, TF, GF>), dim3(grid), dim3(256), 0, st, ta); \
    return; \
  }
  C2S_SHAPES(B2_X)
#undef B2_X
}

// ------------------------------------------------------------ leak compensation --
// What it is.  The range transforms multiply by fp32 constants (the 8-/16-/32-point butterflies' roots, the stage twiddles):
// each is off by up to 3e-8 of itself, and it is off THE SAME WAY in every transform.  The data-dependent roundings of a
// transform average out over the 1e6-1e7 samples of a CPI; the constants' errors do not: they add up coherently, and what
// they add up to is a fixed linear response -- a fraction g[d] (measured: 1e-9 rms, 0.8e-8 ... 1.5e-8 at a few dozen lags
// that are multiples of 4 or 8 away from the source) of every pulse's lag-0 correlation appears at lag d.  Summed over the
// pulses that is g[d] x M[k][lag 0] in Doppler row k: invisible wherever column lag 0 is at the floor, and
// g[d] x (direct-path peak) in the zero-Doppler row, where the peak stands sqrt(N) = 1.4e3 ... 6.3e3 above the mean
// level -- 1.2e-5 of a mean-level cell at configs[1], 4e-5 at configs[2], 1e-4 at configs[4] (tools/gpu_cell_err.py:
// two of eight CPIs over north_star's 1e-4 there, every one of them at 2e-5 once the leak is taken out).
// How it is measured.  The chain is run once on a synthetic CPI of white integer-valued samples, x = y, whose zero-Doppler
// row has an EXACT value that a small fp64 kernel sums directly (integers below 2^53); g[d] = (engine - exact)[d] / exact[0].
// The engine's data-dependent rounding on that CPI is 6e-7 of the floor per cell, i.e. 1e-10 of the peak: two orders under
// the pattern.  (A CPI of sparse unit impulses, whose exact row is zero without any reference, was tried first and measures
// a DIFFERENT pattern -- lags F/40, F/20, F/8 ... -- the roundings of products with exact roots of unity are not random;
// subtracting it made the maps worse.)
// How it is applied.  After the Doppler kernel, one 64-thread workgroup per CPI subtracts g[d] x M[k0][lag 0] from the
// listed cells of row k0 (the Map::set_metrics partials were taken before: the mean of 2e5 dB values moves by < 1e-6 dB).
// Auto mode launches it only where max|g| sqrt(N) can reach 3e-5 (not at configs[1]: the headline's launch train is what
// it was).  Calibrated per (range kernel, Doppler kernel) at the first launch that runs the pair.
// calibration input: white, integer-valued (exact in every sample format), x = y
__global__ void leak_noise_kernel(cf *x, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull; // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    x[i] = cmake((float)((int)(z & 1023) - 512), (float)((int)((z >> 20) & 1023) - 512));
  }
}

// ... and its exact zero-Doppler row (Ambiguity.cpp:106-169 at Doppler zero: the plain sum over the pulses of
// sum_n x[n + d] conj(x[n]), samples of the same pulse only), by direct summation in fp64 -- every term is an integer
// below 2^20 and every sum below 2^53: EXACT.  grid (lag blocks of 256, pulse slices); Z += through fp64 atomics.
__global__ __launch_bounds__(256) void leak_ref_kernel(const cf *__restrict__ x, int nCorr, int nPulses, int delayMin, int nDelay,
                                                        double *Z)
{
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int d = delayMin + j;
  double ar = 0.0, ai = 0.0;
  for (int p = blockIdx.y; p < nPulses; p += gridDim.y) {
    const cf *xp = x + (size_t)p * nCorr;
    const int lo = max(0, -d), hi = min(nCorr, nCorr - d);
    if (j < nDelay)
      for (int n = lo; n < hi; n++) {
        const cf a = xp[n], b = xp[n + d]; // b conj(a)
        ar += (double)b.x * (double)a.x + (double)b.y * (double)a.y;
        ai += (double)b.y * (double)a.x - (double)b.x * (double)a.y;
      }
  }
  if (j < nDelay) {
    atomicAdd(&Z[2 * j], ar);
    atomicAdd(&Z[2 * j + 1], ai);
  }
}

__global__ __launch_bounds__(64) void leak_fix_kernel(cf *map, size_t cells, size_t row0, int col0, int n,
                                                       const int32_t *__restrict__ col, const cf *__restrict__ g)
{
  cf *row = map + (size_t)blockIdx.x * cells + row0;
  const cf m0 = row[col0];
  for (int j = threadIdx.x; j < n; j += 64) {
    const cf gj = g[j], v = row[col[j]];
    row[col[j]] = cmake(v.x - (gj.x * m0.x - gj.y * m0.y), v.y - (gj.x * m0.y + gj.y * m0.x));
  }
}

// ------------------------------------------------------------------ hot columns
// A delay column that holds a tone -- the direct path at lag 0, a strong target, the cancelled map's tallest echo -- comes
// out of the fp32 Doppler transform with an error of ~4e-8 ... 1.2e-7 of ITS PEAK in every other row of that column (the
// last radix step alone rounds two half-sums of the peak's size; measured: tools/gpu_chain_split_diag.py, configs[2] behind
// the filter: 1.65e-4 of a mean-level cell in the column of a peak 1430x the mean level, the map's largest error by 5x and
// the only one beyond north_star's 1e-4).  The reference computes in fp64 and has no such floor.  So the few columns whose
// peak can stand more than HOT_RATIO above the map's mean level are transformed again in fp64, straight from the fp32
// range map (direct DFT, nD^2 complex fp64 MACs a column -- 1e6 at nD = 1025, spread over nD/32 workgroups), and written
// over the fp32 result.  Which columns: those whose mean power over four pulses of the range map (less the first pulse's
// value: the Doppler kernels remove the column's zero-Doppler content exactly before they transform it), taken as a coherent tone
// (x nD), would reach HOT_RATIO x the mean level of the Map::set_metrics partials -- noise columns sit sqrt(nD)/0.75 below
// that test, so a launch with nothing hot costs one read of 4 x nDelay cells per workgroup.  At most HOT_MAX columns a CPI
// (the strongest; ties to the lower lag), every workgroup of the CPI deriving the same list.  The Map::set_metrics partials
// were taken before: rewriting 1e-7 of a peak moves noisePower by < 1e-8 dB.
constexpr int HOT_ROWS = 32, HOT_MAX = 16, HOT_CAND = 256, HOT_SAMPLES = 4, HOT_ND_MAX = 4096;
constexpr double HOT_RATIO = 250.0; // 1.2e-7 x 250 = 3e-5 of a mean-level cell: a third of the 1e-4 gate, as LEAK_REACH

struct HotArgs {
  const cf *R;             // tiled range map (rmap_index)
  cf *map;                 // [nCpi][nD][nDelay]
  const double2 *W;        // exp(-2 pi i k / nD), fp64
  const double *partSum;   // [nCpi][nParts] (nParts > 0) ...
  const double *metrics;   // ... or the finished [nCpi][2] (nParts == 0: doppler_sub1k_kernel)
  uint32_t *count;         // [nCpi]: columns rewritten
  int32_t nParts, partStride, nD, nDelay, nTiles;
  int32_t groups;          // 32-row groups a workgroup walks: 1 on small launches (latency), 4 in batches (the candidate scan,
                           // which every workgroup repeats, is most of a launch that finds nothing)
  float ratioDb;           // 10 log10(HOT_RATIO)
};

__global__ __launch_bounds__(256) void hot_columns_kernel(HotArgs a)
{
  extern __shared__ double2 hs[]; // W[nD], column[nD]
  __shared__ double sred[4];
  __shared__ int candLag[HOT_CAND];
  __shared__ float candDb[HOT_CAND];
  __shared__ int waveCnt[4], hot[HOT_MAX], nHot;
  __shared__ double2 red[8][HOT_ROWS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cpi = blockIdx.y, nD = a.nD, nDelay = a.nDelay;
  // (the roots are requested first: they travel while the candidates are looked for, 8 KB a workgroup that finds none)
  double2 *W = hs, *col = hs + nD;
  for (int i = t; i < nD; i += 256) W[i] = a.W[i];
  // mean level (dB) of the map, from the partials in index order (every workgroup of the CPI: the same bits)
  double levelDb;
  if (a.nParts > 0) {
    double s = 0.0;
    for (int i = t; i < a.nParts; i += 256) s += a.partSum[(size_t)cpi * a.partStride + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) sred[wave] = s;
    __syncthreads();
    levelDb = ((sred[0] + sred[1]) + (sred[2] + sred[3])) / ((double)nD * (double)nDelay);
  } else {
    levelDb = a.metrics[2 * cpi];
  }
  const float thr = (float)levelDb + a.ratioDb - 10.f * log10f((float)nD);
  // candidates in lag order: wave w scans its quarter of the lags.  (Eight groups of 64 lags with all their loads in flight
  // together changed nothing: a launch that finds nothing runs 4.5 us on a lone CPI, what every small dependent kernel of the
  // chain runs -- metrics_kernel, clutter_reduce_kernel, the gated solve.)
  const int per = ((nDelay + 3) / 4 + 63) & ~63;
  int mine = 0;
  for (int j0 = wave * per; j0 < min(nDelay, (wave + 1) * per); j0 += 64) {
    const int j = j0 + lane;
    float db = -1e30f;
    if (j < nDelay) {
      // (less the first pulse's value, as the Doppler kernels transform the column: what stands at zero Doppler -- the direct
      // path's column -- is exact there already and is no reason to transform it again)
      const cf r0 = a.R[rmap_index(nD, a.nTiles, cpi, 0, j)];
      float pw = 0.f;
#pragma unroll
      for (int s = 0; s < HOT_SAMPLES; s++) {
        const cf r = a.R[rmap_index(nD, a.nTiles, cpi, (int)(((int64_t)(2 * s + 1) * nD) / (2 * HOT_SAMPLES)), j)];
        pw += (r.x - r0.x) * (r.x - r0.x) + (r.y - r0.y) * (r.y - r0.y);
      }
      db = 5.f * log10f(pw * (1.f / HOT_SAMPLES)); // 10 log10 of the amplitude
    }
    const uint64_t m = __ballot(db > thr);
    if (db > thr) {
      const int slot = mine + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < HOT_CAND / 4) { candLag[wave * (HOT_CAND / 4) + slot] = j; candDb[wave * (HOT_CAND / 4) + slot] = db; }
    }
    mine += __popcll(m);
  }
  if (lane == 0) waveCnt[wave] = min(mine, HOT_CAND / 4);
  __syncthreads();
  if (t == 0) { // the HOT_MAX strongest, ties to the lower lag
    int n = 0;
    for (; n < HOT_MAX; n++) {
      int best = -1;
      float bestDb = -1e30f; // (a picked candidate is marked with this value)
      for (int w = 0; w < 4; w++)
        for (int i = 0; i < waveCnt[w]; i++) {
          const int c = w * (HOT_CAND / 4) + i;
          if (candDb[c] > bestDb) { best = c; bestDb = candDb[c]; }
        }
      if (best < 0) break;
      candDb[best] = -1e30f;
      hot[n] = candLag[best];
    }
    nHot = n;
    if (blockIdx.x == 0) a.count[cpi] = (uint32_t)n;
  }
  __syncthreads();
  const int nh = nHot;
  if (nh == 0) return;
  const int r = t & (HOT_ROWS - 1), part = t / HOT_ROWS; // 8 parts of the pulse axis
  const int chunk = (nD + 7) / 8;
  const int i0 = part * chunk, i1 = min(nD, i0 + chunk);
  for (int h = 0; h < nh; h++) {
    const int j = hot[h];
    __syncthreads();
    for (int i = t; i < nD; i += 256) {
      const cf v = a.R[rmap_index(nD, a.nTiles, cpi, i, j)];
      col[i] = make_double2((double)v.x, (double)v.y);
    }
    __syncthreads();
    for (int gi = 0; gi < a.groups; gi++) {
      const int o = (blockIdx.x * a.groups + gi) * HOT_ROWS + r;
      const int src = ((o < nD ? o : nD - 1) + nD / 2 + 1) % nD; // Ambiguity.cpp:165
      // exp(-2 pi i src i / nD) along the thread's part of the pulse axis by recurrence from two table entries (<= nD / 8 steps
      // in fp64: 6e-14 at nD = 4096; a table read per pulse was the loop's latency: random 16-byte LDS reads.  Four interleaved
      // recurrences of step ws^4 were slower: 9.2 against 8.1 us on a lone CPI, 52 against 40 per 32 CPIs of configs[2])
      const double2 w0 = W[(int)(((int64_t)src * i0) % nD)], ws = W[src];
      double wr = w0.x, wi = w0.y;
      double ar = 0.0, ai = 0.0;
#pragma unroll 8
      for (int i = i0; i < i1; i++) { // (unrolled: the LDS reads of eight pulses travel together; the recurrence is the chain)
        const double2 v0 = col[i];
        ar = fma(v0.x, wr, ar); ar = fma(-v0.y, wi, ar);
        ai = fma(v0.x, wi, ai); ai = fma(v0.y, wr, ai);
        const double nr = fma(wr, ws.x, -wi * ws.y), ni = fma(wr, ws.y, wi * ws.x);
        wr = nr; wi = ni;
      }
      if (gi) __syncthreads(); // the previous group's sums have been read
      red[part][r] = make_double2(ar, ai);
      __syncthreads();
      if (t < HOT_ROWS && o < nD) {
        double2 s = red[0][t];
#pragma unroll
        for (int p = 1; p < 8; p++) { s.x += red[p][t].x; s.y += red[p][t].y; }
        a.map[((size_t)cpi * nD + o) * nDelay + j] = cmake((float)s.x, (float)s.y);
      }
    }
  }
}

int predict_range(const blah2hip_amb_s *h, int nPulses)
{
  if (use_wave_range(h, nPulses)) return BLAH2HIP_RANGE_WAVE;
  if (use_wave1k_range(h, nPulses)) return BLAH2HIP_RANGE_WAVE1K;
  if (use_ps_range(h, nPulses)) return BLAH2HIP_RANGE_PS;
  return h->r3 == 4 ? BLAH2HIP_RANGE_E8 : BLAH2HIP_RANGE_E16;
}

void leak_clear(blah2hip_amb_s *h)
{
  for (auto &kv : h->leak) {
    if (kv.second.d_lag) (void)hipFree(kv.second.d_lag);
    if (kv.second.d_g) (void)hipFree(kv.second.d_g);
  }
  h->leak.clear();
}

// index of the zero-Doppler row / the lag-0 column, or -1
int leak_row0(const blah2hip_amb_s *h)
{
  for (size_t k = 0; k < h->dopplerAxis.size(); k++)
    if (std::fabs(h->dopplerAxis[k]) < 1e-9) return (int)k;
  return -1;
}
int leak_col0(const blah2hip_amb_s *h)
{
  for (size_t j = 0; j < h->delayAxis.size(); j++)
    if (h->delayAxis[j] == 0) return (int)j;
  return -1;
}

constexpr double LEAK_KEEP = 2.0e-9;   // lags whose |g| is above the positions' own rounding (1.4e-9)
constexpr double LEAK_REACH = 3.0e-5;  // auto mode: apply where max|g| sqrt(N) can move a mean-level cell by this much (a third of the 1e-4 gate)
constexpr int LEAK_MAX_LAGS = 512;

int leak_calibrate(blah2hip_amb_s *h, int rangeId, int dopId, hipStream_t st, blah2hip_amb_s::LeakCal *out)
{
  blah2hip_amb_s::LeakCal cal;
  const int k0 = leak_row0(h), c0 = leak_col0(h);
  const int nDelay = (int)h->dims.n_delay_bins, nD = (int)h->dims.n_doppler_bins, nCorr = (int)h->dims.n_corr;
  const int reach = std::max(std::abs(h->delayMin), std::abs(h->delayMax));
  // not calibrated (the entry stays inactive): no zero-Doppler row or lag-0 column, a rotated reference channel, a lag window
  // run as chunks (aliased lags), a window longer than a pulse
  // ... nor, in auto mode, a CPI so short that no pattern of this chain (max|g| <= 3e-8 on every kernel measured) could reach
  // LEAK_REACH: the small handles of a test suite do not pay for a measurement they would never use
  const bool tooShort = h->leakMode == 1 && 4.0e-8 * std::sqrt((double)h->dims.n_used) < LEAK_REACH;
  if (k0 < 0 || c0 < 0 || h->dopplerMin + h->dopplerMax != 0 || h->chunks.size() != 1 || reach >= nCorr || tooShort) {
    *out = cal;
    return BLAH2HIP_OK;
  }
  const size_t n = (size_t)h->dims.n_samples, cells = (size_t)nD * nDelay;
  cf *d_x = nullptr, *d_m = nullptr;
  double *d_met = nullptr, *d_Z = nullptr;
  std::vector<cf> row(nDelay);
  std::vector<double> Z(2 * (size_t)nDelay);
  const int savedRange = h->rangeKernel, savedDop = h->dopForce;
  const bool savedTiming = h->timer.enabled;
  int rc = BLAH2HIP_OK;
  hipError_t e = hipMalloc(&d_x, n * sizeof(cf));
  if (e == hipSuccess) e = hipMalloc(&d_m, cells * sizeof(cf));
  if (e == hipSuccess) e = hipMalloc(&d_met, 2 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&d_Z, 2 * (size_t)nDelay * sizeof(double));
  if (e == hipSuccess) e = hipMemsetAsync(d_Z, 0, 2 * (size_t)nDelay * sizeof(double), st);
  if (e == hipSuccess) {
    leak_noise_kernel<<<dim3(h->numCU * 8), dim3(256), 0, st>>>(d_x, n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) {
    h->rangeKernel = rangeId; h->dopForce = dopId; h->timer.enabled = false; h->inLeakCal = true;
    rc = blah2hip_amb_process_dev(h, BLAH2HIP_FMT_C32, d_x, d_x, 1, 0, d_m, d_met, (void *)st);
    h->rangeKernel = savedRange; h->dopForce = savedDop; h->timer.enabled = savedTiming; h->inLeakCal = false;
    if (rc == BLAH2HIP_OK) {
      const int lagBlocks = (nDelay + 255) / 256;
      const int slices = std::max(1, std::min(nD, (4 * h->numCU + lagBlocks - 1) / lagBlocks));
      leak_ref_kernel<<<dim3(lagBlocks, slices), dim3(256), 0, st>>>(d_x, nCorr, nD, h->delayMin, nDelay, d_Z);
      e = hipGetLastError();
    }
    if (rc == BLAH2HIP_OK && e == hipSuccess) e = hipMemcpyAsync(row.data(), d_m + (size_t)k0 * nDelay, (size_t)nDelay * sizeof(cf), hipMemcpyDeviceToHost, st);
    if (rc == BLAH2HIP_OK && e == hipSuccess) e = hipMemcpyAsync(Z.data(), d_Z, 2 * (size_t)nDelay * sizeof(double), hipMemcpyDeviceToHost, st);
    if (rc == BLAH2HIP_OK && e == hipSuccess) e = hipStreamSynchronize(st);
  }
  if (d_Z) (void)hipFree(d_Z);
  if (d_x) (void)hipFree(d_x);
  if (d_m) (void)hipFree(d_m);
  if (d_met) (void)hipFree(d_met);
  if (rc != BLAH2HIP_OK) return rc;
  if (e != hipSuccess) return fail(BLAH2HIP_ERR_HIP, std::string("leak calibration: ") + hipGetErrorString(e));
  const std::complex<double> peak(Z[2 * c0], Z[2 * c0 + 1]);
  if (!(std::abs(peak) > 0.0)) { *out = cal; return BLAH2HIP_OK; }
  std::vector<std::pair<double, int>> order;
  std::vector<std::complex<double>> g(nDelay);
  for (int j = 0; j < nDelay; j++) {
    g[j] = j == c0 ? std::complex<double>(0.0, 0.0)
                   : (std::complex<double>(row[j].x, row[j].y) - std::complex<double>(Z[2 * j], Z[2 * j + 1])) / peak;
    if (std::abs(g[j]) >= LEAK_KEEP) order.push_back({-std::abs(g[j]), j});
    cal.maxAbs = std::max(cal.maxAbs, std::abs(g[j]));
  }
  std::sort(order.begin(), order.end());
  if ((int)order.size() > LEAK_MAX_LAGS) order.resize(LEAK_MAX_LAGS);
  cal.nLags = (int)order.size();
  if (cal.nLags) {
    std::vector<int32_t> lag(cal.nLags);
    std::vector<cf> gv(cal.nLags);
    for (int i = 0; i < cal.nLags; i++) { lag[i] = order[i].second; gv[i] = cmake((float)g[order[i].second].real(), (float)g[order[i].second].imag()); }
    HIPCHK(hipMalloc(&cal.d_lag, cal.nLags * sizeof(int32_t)));
    HIPCHK(hipMalloc(&cal.d_g, cal.nLags * sizeof(cf)));
    HIPCHK(hipMemcpy(cal.d_lag, lag.data(), cal.nLags * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(cal.d_g, gv.data(), cal.nLags * sizeof(cf), hipMemcpyHostToDevice));
  }
  cal.active = cal.nLags > 0 && cal.maxAbs * std::sqrt((double)h->dims.n_used) >= LEAK_REACH;
  *out = cal;
  return BLAH2HIP_OK;
}

// nullptr if range_fir_kernel covers the handle's geometry with a filter of nBins taps from lag firDmin, else the reason
const char *fir_unfusable(const blah2hip_amb_s *h, int fmt, int nBins, int firDmin)
{
  const int L = 2048;
  const int nDelay = (int)h->dims.n_delay_bins;
  if (h->dims.fft_len != 4096) return "fused FIR: the handle's transform length must be 4096 (BLAH2HIP_OPT_FFT_LEN)";
  if (fmt != BLAH2HIP_FMT_C32 && fmt != BLAH2HIP_FMT_I16) return "fused FIR: fp32 planes or int16 words";
  if (h->chunks.size() != 1 || h->dopplerMin + h->dopplerMax != 0) return "fused FIR: one lag chunk, symmetric Doppler limits";
  if (nBins < 1 || nBins > L + 1 || nDelay > L + 1) return "fused FIR: at most 2049 taps and 2049 delay bins";
  if (firDmin != h->delayMin || h->delayMin > 0) return "fused FIR: the filter's first lag must equal the map's and be <= 0";
  if (nBins < -firDmin) return "fused FIR: the filter's window must reach lag 0";
  if ((int)h->dims.n_corr < L - h->delayMin) return "fused FIR: pulses shorter than 2048 - delayMin samples";
  if ((uint64_t)h->dims.n_used + (uint64_t)(-h->delayMin) > h->dims.n_samples)
    return "fused FIR: the filter's look-ahead past the last pulse wraps around the CPI";
  return nullptr;
}

int ensure_sat(blah2hip_amb_s *h)
{
  if (h->d_sat) return BLAH2HIP_OK;
  const size_t satElems = (size_t)(h->dims.n_doppler_bins + 1) * (h->dims.n_delay_bins + 1);
  HIPCHK(hipMalloc(&h->d_sat, satElems * h->dims.max_batch * sizeof(double)));
  HIPCHK(hipMemset(h->d_sat, 0, satElems * h->dims.max_batch * sizeof(double))); // zero borders
  return BLAH2HIP_OK;
}

} // namespace

extern "C" {

const char *blah2hip_last_error(void) { return g_err.c_str(); }
const char *blah2hip_version(void) { return "blah2hip 0.1 (gfx950)"; }

int blah2hip_device_count(int *count)
{
  if (!count) return fail(BLAH2HIP_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(BLAH2HIP_ERR_NO_DEVICE, hipGetErrorString(e)); }
  *count = n;
  return BLAH2HIP_OK;
}

// HammingNumber.cpp:38-48: first 5-smooth number strictly above v.
uint32_t blah2hip_next_hamming(uint32_t v)
{
  uint64_t best = 0;
  for (uint64_t a = 1; a <= 2 * (uint64_t)v + 2; a *= 2)
    for (uint64_t b = a; b <= 2 * (uint64_t)v + 2; b *= 3)
      for (uint64_t c = b; c <= 2 * (uint64_t)v + 2; c *= 5)
        if (c > v && (best == 0 || c < best)) best = c;
  return (uint32_t)best;
}

int blah2hip_amb_create(int32_t delay_min, int32_t delay_max, int32_t doppler_min,
                        int32_t doppler_max, uint32_t fs, uint32_t n, int round_hamming, int device,
                        uint32_t max_batch, blah2hip_amb_t *out)
{
  return blah2hip_amb_create_ex(delay_min, delay_max, doppler_min, doppler_max, fs, n, round_hamming, 0, device, max_batch, out);
}

int blah2hip_amb_create_ex(int32_t delay_min, int32_t delay_max, int32_t doppler_min,
                           int32_t doppler_max, uint32_t fs, uint32_t n, int round_hamming,
                           uint32_t n_doppler_bins, int device, uint32_t max_batch, blah2hip_amb_t *out)
{
  if (!out) return fail(BLAH2HIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (fs == 0 || n == 0) return fail(BLAH2HIP_ERR_INVALID, "fs and n must be positive");
  if (delay_max < delay_min) return fail(BLAH2HIP_ERR_INVALID, "delayMax < delayMin");
  if (doppler_max < doppler_min) return fail(BLAH2HIP_ERR_INVALID, "dopplerMax < dopplerMin");
  // the reference's lag gather (Ambiguity.cpp:132-146) is only in range for these
  if (delay_min > 1 || delay_max < -1)
    return fail(BLAH2HIP_ERR_UNSUPPORTED, "reference requires delayMin <= 1 and delayMax >= -1");
  if (n_doppler_bins > 65535u || n_doppler_bins > n) return fail(BLAH2HIP_ERR_INVALID, "explicit Doppler bin count outside [1, min(65535, n)]");
  if (max_batch == 0) max_batch = 1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(BLAH2HIP_ERR_NO_DEVICE, "no HIP device visible (the HIP path is the only path)");
  if (device < 0 || device >= ndev) return fail(BLAH2HIP_ERR_INVALID, "device index out of range");
  HIPCHK(hipSetDevice(device));

  auto *h = new blah2hip_amb_s;
  // everything that can fail runs inside `build`; a partially built handle is torn down by destroy()
  auto build = [&]() -> int {
  h->device = device;
  h->delayMin = delay_min; h->delayMax = delay_max;
  h->dopplerMin = doppler_min; h->dopplerMax = doppler_max;
  h->fs = fs;
  derive_dims(h, n, round_hamming != 0, n_doppler_bins);
  h->dims.max_batch = max_batch;
  if (h->dims.n_corr == 0) return fail(BLAH2HIP_ERR_INVALID, "nCorr == 0");
  if ((int64_t)std::max(std::llabs((long long)delay_min), std::llabs((long long)delay_max)) >= (int64_t)h->dims.nfft)
    return fail(BLAH2HIP_ERR_UNSUPPORTED, "|delay| >= nfft: the reference's lag gather (Ambiguity.cpp:132-146) reads outside its buffer there");
  // one chunk when the window fits a transform; windows of more than 4081 lags in chunks of 2048 (F = 4096: 2049 new
  // samples per segment), each re-reading the pulse
  lag_chunks(h, 4081);
  if (h->chunks.size() > 1 || h->maxChunk > 4081) lag_chunks(h, 2048);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  h->numCU = prop.multiProcessorCount;
  if (!choose_plan(h)) {
    return fail(BLAH2HIP_ERR_UNSUPPORTED, "no on-chip transform length fits the lag window (forced length too short?)");
  }
  HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));

  const uint32_t nD = h->dims.n_doppler_bins, nDelay = h->dims.n_delay_bins;

  std::vector<cf> dw(nD);
  for (uint32_t k = 0; k < nD; k++) dw[k] = root_of_unity(k, nD);
  const size_t cells = (size_t)nD * nDelay;
  h->nTiles = (int)((nDelay + 15) / 16);
  const size_t rcells = (size_t)h->nTiles * nD * 16; // tiled range map, padded to 16 columns
  // Doppler plan: Bluestein on the smallest on-chip transform with M >= 2*nD-2
  h->dopR3 = 0;
  for (int r3 : {4, 8, 16})
    if (256 * r3 >= 2 * (int)nD - 2) { h->dopR3 = r3; break; }
  h->dopTilesX = (nDelay + 63) / 64;
  h->dopTilesY = (nD + DOP_KPT - 1) / DOP_KPT;
  h->dopGridX = 8 * ((h->nTiles + 7) / 8) * 16; // per-column kernel: one workgroup per column, tiles padded to a multiple of 8
  // metrics partials per CPI: the largest workgroup count of any Doppler kernel this handle may launch
  h->nParts = std::max(h->dopGridX, h->dopTilesX * h->dopTilesY);

  { const int rc_ = upload_range_table(h); if (rc_) return rc_; }
  HIPCHK(hipMalloc(&h->d_dopW, nD * sizeof(cf)));
  HIPCHK(hipMalloc(&h->d_R, rcells * max_batch * sizeof(cf)));
  HIPCHK(hipMalloc(&h->d_map, cells * max_batch * sizeof(cf)));
  HIPCHK(hipMalloc(&h->d_partSum, (size_t)h->nParts * max_batch * sizeof(double)));
  HIPCHK(hipMalloc(&h->d_partMax, (size_t)h->nParts * max_batch * sizeof(float)));
  HIPCHK(hipMalloc(&h->d_tickets, max_batch * sizeof(uint32_t)));
  HIPCHK(hipMemset(h->d_tickets, 0, max_batch * sizeof(uint32_t)));
  HIPCHK(hipMalloc(&h->d_metrics, 2 * max_batch * sizeof(double)));
  HIPCHK(hipMalloc(&h->d_doppler, nD * sizeof(double)));
  HIPCHK(hipMalloc(&h->d_count, max_batch * sizeof(uint32_t)));
  HIPCHK(hipMemset(h->d_R, 0, rcells * max_batch * sizeof(cf))); // padding columns stay finite
  HIPCHK(hipMemcpy(h->d_dopW, dw.data(), nD * sizeof(cf), hipMemcpyHostToDevice));
  if (nD <= (uint32_t)HOT_ND_MAX) {
    std::vector<double2> dw64(nD);
    for (uint32_t k = 0; k < nD; k++) {
      const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)nD;
      dw64[k] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    HIPCHK(hipMalloc(&h->d_dopW64, nD * sizeof(double2)));
    HIPCHK(hipMemcpy(h->d_dopW64, dw64.data(), nD * sizeof(double2), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&h->d_hotCount, max_batch * sizeof(uint32_t)));
    HIPCHK(hipMemset(h->d_hotCount, 0, max_batch * sizeof(uint32_t)));
  }
  HIPCHK(hipMemcpy(h->d_doppler, h->dopplerAxis.data(), nD * sizeof(double), hipMemcpyHostToDevice));
  if (h->dopR3) {
    std::vector<cf> dtw, chirp, bf;
    std::vector<cf> bfn;
    doppler_tables((int)nD, h->dopR3, dtw, chirp, bf, bfn);
    HIPCHK(hipMalloc(&h->d_bfn, bfn.size() * sizeof(cf)));
    HIPCHK(hipMemcpy(h->d_bfn, bfn.data(), bfn.size() * sizeof(cf), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&h->d_dtw, dtw.size() * sizeof(cf)));
    HIPCHK(hipMalloc(&h->d_chirp, chirp.size() * sizeof(cf)));
    HIPCHK(hipMalloc(&h->d_bf, bf.size() * sizeof(cf)));
    HIPCHK(hipMemcpy(h->d_dtw, dtw.data(), dtw.size() * sizeof(cf), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_chirp, chirp.data(), chirp.size() * sizeof(cf), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_bf, bf.data(), bf.size() * sizeof(cf), hipMemcpyHostToDevice));
  }
  return BLAH2HIP_OK;
  };
  const int rc = build();
  if (rc != BLAH2HIP_OK) {
    blah2hip_amb_destroy(h);
    return rc;
  }
  *out = h;
  return BLAH2HIP_OK;
}

int blah2hip_amb_destroy(blah2hip_amb_t h)
{
  if (!h) return BLAH2HIP_OK;
  // teardown: nothing useful can be done with a failure here
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void *p : {(void *)h->d_tw, (void *)h->d_dopW, (void *)h->d_R, (void *)h->d_map,
                  (void *)h->d_partSum, (void *)h->d_partMax, (void *)h->d_tickets, (void *)h->d_metrics,
                  (void *)h->d_doppler, h->d_in, (void *)h->d_rot,
                  (void *)h->d_hits, (void *)h->d_count, (void *)h->d_sat, (void *)h->d_dtw, (void *)h->d_chirp,
                  (void *)h->d_bf, (void *)h->d_bfn, (void *)h->d_H, (void *)h->d_dopW64, (void *)h->d_hotCount, (void *)h->d_firK0})
    if (p) (void)hipFree(p);
  for (auto &t : h->alphaTables)
    if (t.d) (void)hipFree(t.d);
  if (h->h_pin) (void)hipHostFree(h->h_pin);
  leak_clear(h);
  h->timer.destroy();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return BLAH2HIP_OK;
}

int blah2hip_amb_get_dims(blah2hip_amb_t h, blah2hip_amb_dims_t *dims)
{
  if (!h || !dims) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  *dims = h->dims;
  return BLAH2HIP_OK;
}

int blah2hip_amb_get_axes(blah2hip_amb_t h, int32_t *delay, double *doppler)
{
  if (!h) return fail(BLAH2HIP_ERR_INVALID, "NULL handle");
  if (delay) std::memcpy(delay, h->delayAxis.data(), h->delayAxis.size() * sizeof(int32_t));
  if (doppler) std::memcpy(doppler, h->dopplerAxis.data(), h->dopplerAxis.size() * sizeof(double));
  return BLAH2HIP_OK;
}

int blah2hip_amb_set_option(blah2hip_amb_t h, int option, int64_t value)
{
  if (!h) return fail(BLAH2HIP_ERR_INVALID, "NULL handle");
  switch (option) {
  case BLAH2HIP_OPT_DOPPLER_KERNEL:
    if (value != BLAH2HIP_DOP_AUTO && !doppler_kernel_applicable(h, (int)value))
      return fail(BLAH2HIP_ERR_UNSUPPORTED, "this Doppler kernel does not cover the handle's Doppler length");
    h->dopForce = (int)value;
    return BLAH2HIP_OK;
  case BLAH2HIP_OPT_RANGE_GRID:
    if (value < 0 || value > (1 << 20)) return fail(BLAH2HIP_ERR_INVALID, "range grid outside [0, 2^20]");
    h->rangeGridForce = (int)value;
    return BLAH2HIP_OK;
  case BLAH2HIP_OPT_RANGE_KERNEL:
    if (value != 0 && value != BLAH2HIP_RANGE_WAVE && value != BLAH2HIP_RANGE_E16 && value != BLAH2HIP_RANGE_WAVE1K &&
        value != BLAH2HIP_RANGE_E8 && value != BLAH2HIP_RANGE_PS)
      return fail(BLAH2HIP_ERR_INVALID, "range kernel: 0 (by transform length), BLAH2HIP_RANGE_E16, _E8, _WAVE, _WAVE1K or _PS");
    if (value == BLAH2HIP_RANGE_PS && h->r3 != 4)
      return fail(BLAH2HIP_ERR_UNSUPPORTED, "the pulse-per-workgroup range kernel is a 1024-point transform");
    if ((value == BLAH2HIP_RANGE_WAVE1K || value == BLAH2HIP_RANGE_E8) && h->r3 != 4)
      return fail(BLAH2HIP_ERR_UNSUPPORTED, "the 16-points-per-lane one-wave kernel and the 8-points-per-thread kernel are 1024-point transforms");
    if (value == BLAH2HIP_RANGE_WAVE && h->r3 != 8)
      return fail(BLAH2HIP_ERR_UNSUPPORTED, "the one-wave range kernel is a 2048-point transform");
    if (value == BLAH2HIP_RANGE_E16 && h->r3 == 4)
      return fail(BLAH2HIP_ERR_UNSUPPORTED, "F = 1024 runs on the 8-points-per-thread kernel only");
    h->rangeKernel = (int)value;
    return BLAH2HIP_OK;
  case BLAH2HIP_OPT_CFAR2D_KERNEL:
    if (value < BLAH2HIP_CFAR2D_AUTO || value > BLAH2HIP_CFAR2D_STREAM) return fail(BLAH2HIP_ERR_INVALID, "2-D detector: AUTO, TILE, SAT or STREAM");
    h->cfar2dForce = (int)value;
    return BLAH2HIP_OK;
  case BLAH2HIP_OPT_DOPPLER_GRID:
    if (value < 0 || value > (1 << 20)) return fail(BLAH2HIP_ERR_INVALID, "Doppler grid outside [0, 2^20]");
    h->dopGridForce = (int)value;
    return BLAH2HIP_OK;
  case BLAH2HIP_OPT_FFT_LEN: {
    if (value != 0 && value != 1024 && value != 2048 && value != 4096)
      return fail(BLAH2HIP_ERR_INVALID, "range transform length: 0 (planner), 1024, 2048 or 4096");
    const int prevForce = h->fftLenForce, prevR3 = h->r3;
    const RangePlan prevPlan = h->plan;
    h->fftLenForce = (int)value;
    if (!choose_plan(h)) {
      h->fftLenForce = prevForce; h->r3 = prevR3; h->plan = prevPlan;
      return fail(BLAH2HIP_ERR_UNSUPPORTED, "the lag window does not fit this transform length");
    }
    h->rangeKernel = 0; // a forced kernel belongs to a transform length
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize()); // the table may still be in use by enqueued work
    leak_clear(h);                  // the leak pattern belongs to a transform length too
    int rc = upload_range_table(h);
    return rc;
  }
  case BLAH2HIP_OPT_LEAK_COMPENSATION:
    if (value < 0 || value > 2) return fail(BLAH2HIP_ERR_INVALID, "leak compensation: 0 (off), 1 (auto) or 2 (always)");
    if (h->leakMode != (int)value) { // entries skipped as "too short" in auto mode are measured in always mode
      HIPCHK(hipSetDevice(h->device));
      HIPCHK(hipDeviceSynchronize());
      leak_clear(h);
    }
    h->leakMode = (int)value;
    return BLAH2HIP_OK;
  case BLAH2HIP_OPT_HOT_COLUMNS:
    if (value < 0 || value > 2) return fail(BLAH2HIP_ERR_INVALID, "hot columns: 0 (off), 1 (auto) or 2 (always)");
    h->hotMode = (int)value;
    return BLAH2HIP_OK;
  default: return fail(BLAH2HIP_ERR_INVALID, "unknown option");
  }
}

int blah2hip_amb_set_fir(blah2hip_amb_t h, const float *d_w, uint32_t n_bins, int32_t clutter_delay_min)
{
  if (!h) return fail(BLAH2HIP_ERR_INVALID, "NULL handle");
  if (d_w && n_bins == 0) return fail(BLAH2HIP_ERR_INVALID, "no taps");
  h->firW = reinterpret_cast<const cf *>(d_w);
  h->firBins = (int)n_bins;
  h->firDmin = clutter_delay_min;
  return BLAH2HIP_OK;
}

int blah2hip_amb_fir_fusable(blah2hip_amb_t h, int fmt, uint32_t n_bins, int32_t clutter_delay_min)
{
  if (!h) return fail(BLAH2HIP_ERR_INVALID, "NULL handle");
  const char *why = fir_unfusable(h, fmt, (int)n_bins, clutter_delay_min);
  return why ? fail(BLAH2HIP_ERR_UNSUPPORTED, why) : BLAH2HIP_OK;
}

int blah2hip_amb_get_info(blah2hip_amb_t h, int key, int64_t *value)
{
  if (!h || !value) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  switch (key) {
  case BLAH2HIP_INFO_LAST_DOPPLER_KERNEL: *value = h->lastDoppler; return BLAH2HIP_OK;
  case BLAH2HIP_INFO_LAST_RANGE_KERNEL: *value = h->lastRange; return BLAH2HIP_OK;
  case BLAH2HIP_INFO_DOPPLER_FFT_LEN: *value = 256 * h->dopR3; return BLAH2HIP_OK;
  case BLAH2HIP_INFO_RANGE_GRID: *value = h->rangeGridLast; return BLAH2HIP_OK;
  case BLAH2HIP_INFO_NUM_CU: *value = h->numCU; return BLAH2HIP_OK;
  case BLAH2HIP_INFO_DOPPLER_GRID: *value = h->dopGridLast; return BLAH2HIP_OK;
  case BLAH2HIP_INFO_DOPPLER_TILES: *value = h->dopTilesLast; return BLAH2HIP_OK;
  case BLAH2HIP_INFO_LEAK_LAGS: *value = h->lastLeakLags; return BLAH2HIP_OK;
  case BLAH2HIP_INFO_LEAK_MAX_E12: *value = (int64_t)std::llround(h->lastLeakMax * 1e12); return BLAH2HIP_OK;
  case BLAH2HIP_INFO_HOT_COLUMNS: { // of the last call's first CPI; waits for the device
    *value = 0;
    if (!h->lastHot) return BLAH2HIP_OK;
    uint32_t n = 0;
    HIPCHK(hipSetDevice(h->device));
    // the stream of the call, not the device: a poll in a pipeline must not stall other streams and handles
    HIPCHK(hipMemcpyAsync(&n, h->d_hotCount, sizeof(n), hipMemcpyDeviceToHost, h->lastHotStream));
    HIPCHK(hipStreamSynchronize(h->lastHotStream));
    *value = n;
    return BLAH2HIP_OK;
  }
  default: return fail(BLAH2HIP_ERR_INVALID, "unknown info key");
  }
}

int blah2hip_amb_process_dev(blah2hip_amb_t h, int fmt, const void *d_x, const void *d_y,
                             uint32_t n_cpi, uint64_t cpi_stride, void *d_map, double *d_metrics,
                             void *stream)
{
  if (!h) return fail(BLAH2HIP_ERR_INVALID, "NULL handle");
  if (n_cpi == 0 || n_cpi > h->dims.max_batch) return fail(BLAH2HIP_ERR_INVALID, "n_cpi outside [1, max_batch]");
  if (fmt != BLAH2HIP_FMT_C32 && fmt != BLAH2HIP_FMT_I16 && fmt != BLAH2HIP_FMT_F16 && fmt != BLAH2HIP_FMT_I16X_C32Y)
    return fail(BLAH2HIP_ERR_INVALID, "unknown sample format");
  if (!d_x || (fmt != BLAH2HIP_FMT_I16 && !d_y)) return fail(BLAH2HIP_ERR_INVALID, "NULL input pointer");
#ifndef B2_EXPERIMENT_ALIASED_CPIS // tools/gpu_cfg3_bytes.py: a timing experiment's build lets every CPI of a batch sit at the same addresses
  if (n_cpi > 1 && cpi_stride < h->dims.n_used) return fail(BLAH2HIP_ERR_INVALID, "cpi_stride < samples used per CPI");
#endif
  HIPCHK(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  const uint32_t nD = h->dims.n_doppler_bins, nDelay = h->dims.n_delay_bins;
  cf *map = d_map ? (cf *)d_map : h->d_map;
  double *met = d_metrics ? d_metrics : h->d_metrics;

  // FIR fused into the range kernel: what one 4096-point transform covers (include/blah2hip.h)
  const bool fused = h->firW != nullptr && !h->inLeakCal;
  if (fused) {
    const char *why = fir_unfusable(h, fmt, h->firBins, h->firDmin);
    if (why) return fail(BLAH2HIP_ERR_UNSUPPORTED, why);
    if (n_cpi > 1 && cpi_stride < h->dims.n_samples) return fail(BLAH2HIP_ERR_INVALID, "fused FIR: cpi_stride < nSamples");
    if (!h->d_H) HIPCHK(hipMalloc(&h->d_H, (size_t)h->dims.max_batch * 16 * 256 * sizeof(cf)));
    if (!h->d_firK0) HIPCHK(hipMalloc(&h->d_firK0, (size_t)h->dims.max_batch * sizeof(int32_t)));
  }
  // the fixed-pattern leak of the kernel pair this launch will run (calibrated at the pair's first launch)
  const blah2hip_amb_s::LeakCal *leak = nullptr;
  if (h->leakMode != 0 && !h->inLeakCal && !fused) { // (behind the filter the lag-0 column holds no peak to leak)
    const int rid = predict_range(h, (int)(n_cpi * nD)), did = pick_doppler(h, n_cpi);
    auto it = h->leak.find(rid * 64 + did);
    if (it == h->leak.end()) {
      blah2hip_amb_s::LeakCal cal;
      const int rcc = leak_calibrate(h, rid, did, st, &cal);
      if (rcc) return rcc;
      it = h->leak.emplace(rid * 64 + did, cal).first;
    }
    h->lastLeakMax = it->second.maxAbs;
    if (it->second.nLags > 0 && (it->second.active || h->leakMode == 2)) leak = &it->second;
  }
  if (!h->inLeakCal) h->lastLeakLags = leak ? leak->nLags : 0;

  RangeArgs ra;
  ra.plan = h->plan;
  ra.tw = h->d_tw;
  ra.out = h->d_R;
  ra.cpiStride = (int64_t)cpi_stride;
  ra.nPulses = (int32_t)(n_cpi * nD);

  const int32_t m2 = h->dopplerMin + h->dopplerMax;
  int rc;
  if (m2 != 0) {
    // Ambiguity.cpp:95-102: rotate the reference channel about the Doppler centre.
    // Rare (asymmetric limits only): one extra pass writes both channels as
    // complex fp32 planes with stride n_samples, then the range kernel runs on those.
    const size_t plane = (size_t)h->dims.n_samples;
    if (!h->d_rot) HIPCHK(hipMalloc(&h->d_rot, 2 * plane * h->dims.max_batch * sizeof(cf)));
    cf *xo = h->d_rot, *yo = h->d_rot + plane * h->dims.max_batch;
    const uint32_t nrot = h->dims.n_used; // later samples are never read by the range loop
    dim3 grid(std::min<uint32_t>((nrot + 255) / 256, 2048), n_cpi);
    if ((rc = tic(h, BLAH2HIP_K_ROTATE, st))) return rc;
    if (fmt == BLAH2HIP_FMT_C32) {
      InC32 in{(const cf *)d_x, (const cf *)d_y};
      hipLaunchKernelGGL(rotate_kernel<InC32>, grid, dim3(256), 0, st, in, xo, yo,
                         (int64_t)cpi_stride, (int64_t)plane, nrot, m2, h->fs);
    } else if (fmt == BLAH2HIP_FMT_F16) {
      InF16 in{(const _Float16 *)d_x, (const _Float16 *)d_y};
      hipLaunchKernelGGL(rotate_kernel<InF16>, grid, dim3(256), 0, st, in, xo, yo,
                         (int64_t)cpi_stride, (int64_t)plane, nrot, m2, h->fs);
    } else if (fmt == BLAH2HIP_FMT_I16X_C32Y) {
      InI16C32 in{(const int16_t *)d_x, (const cf *)d_y};
      hipLaunchKernelGGL(rotate_kernel<InI16C32>, grid, dim3(256), 0, st, in, xo, yo,
                         (int64_t)cpi_stride, (int64_t)plane, nrot, m2, h->fs);
    } else {
      InI16 in{(const int16_t *)d_x};
      hipLaunchKernelGGL(rotate_kernel<InI16>, grid, dim3(256), 0, st, in, xo, yo,
                         (int64_t)cpi_stride, (int64_t)plane, nrot, m2, h->fs);
    }
    HIPCHK(hipGetLastError());
    if ((rc = toc(h, BLAH2HIP_K_ROTATE, st))) return rc;
    ra.cpiStride = (int64_t)plane;
    InC32 in2{xo, yo};
    if ((rc = tic(h, BLAH2HIP_K_RANGE, st))) return rc;
    for (const auto &ck : h->chunks) {
      ra.plan.delayMin = ck.lag0; ra.plan.nDelay = ck.count; ra.plan.colOff = ck.col0;
      if ((rc = launch_range(h, ra, in2, st))) return rc;
    }
    if ((rc = toc(h, BLAH2HIP_K_RANGE, st))) return rc;
  } else if (fused) {
    if ((rc = tic(h, BLAH2HIP_K_RANGE, st))) return rc;
    using W = WgFft<16>;
    const size_t lds = (size_t)(W::A_ELEMS + W::B_ELEMS) * sizeof(cf);
    LDSCFG(taps_spectrum_kernel, lds);
    hipLaunchKernelGGL(taps_spectrum_kernel, dim3(n_cpi), dim3(256), lds, st, h->firW, h->firBins, h->d_tw, h->d_H, h->d_firK0);
    RangeFirArgs fa;
    fa.plan = h->plan;
    fa.plan.delayMin = h->chunks[0].lag0; fa.plan.nDelay = h->chunks[0].count; fa.plan.colOff = h->chunks[0].col0;
    fa.tw = h->d_tw; fa.out = h->d_R; fa.cpiStride = (int64_t)cpi_stride; fa.nPulses = (int32_t)(n_cpi * nD);
    fa.H = h->d_H; fa.N = h->dims.n_samples; fa.w = h->firW; fa.nBins = h->firBins; fa.k0 = h->d_firK0;
    const size_t ldsf = lds + 240 * sizeof(cf); // + the stage-3 twiddle table
    const int grid = std::min<int>(fa.nPulses, range_grid_cap(h, ldsf, 4, 8));
    if (fmt == BLAH2HIP_FMT_C32) {
      InC32 in{(const cf *)d_x, (const cf *)d_y};
      LDSCFG(range_fir_kernel<InC32>, ldsf);
      hipLaunchKernelGGL(range_fir_kernel<InC32>, dim3(grid), dim3(256), ldsf, st, fa, in);
    } else {
      InI16 in{(const int16_t *)d_x};
      LDSCFG(range_fir_kernel<InI16>, ldsf);
      hipLaunchKernelGGL(range_fir_kernel<InI16>, dim3(grid), dim3(256), ldsf, st, fa, in);
    }
    HIPCHK(hipGetLastError());
    h->lastRange = BLAH2HIP_RANGE_FIR;
    if ((rc = toc(h, BLAH2HIP_K_RANGE, st))) return rc;
  } else {
    if ((rc = tic(h, BLAH2HIP_K_RANGE, st))) return rc;
    for (const auto &ck : h->chunks) { // one launch in the usual case
      ra.plan.delayMin = ck.lag0; ra.plan.nDelay = ck.count; ra.plan.colOff = ck.col0;
      if (fmt == BLAH2HIP_FMT_C32) {
        InC32 in{(const cf *)d_x, (const cf *)d_y};
        rc = launch_range(h, ra, in, st);
      } else if (fmt == BLAH2HIP_FMT_F16) {
        InF16 in{(const _Float16 *)d_x, (const _Float16 *)d_y};
        rc = launch_range(h, ra, in, st);
      } else if (fmt == BLAH2HIP_FMT_I16X_C32Y) {
        InI16C32 in{(const int16_t *)d_x, (const cf *)d_y};
        rc = launch_range(h, ra, in, st);
      } else {
        InI16 in{(const int16_t *)d_x};
        rc = launch_range(h, ra, in, st);
      }
      if (rc) return rc;
    }
    if ((rc = toc(h, BLAH2HIP_K_RANGE, st))) return rc;
  }

  DopplerArgs da;
  da.R = h->d_R;
  da.map = map;
  da.W = h->d_dopW;
  da.tw = h->d_dtw;
  da.chirp = h->d_chirp;
  da.bf = h->d_bf;
  da.bfn = h->d_bfn;
  da.partSum = h->d_partSum;
  da.partMax = h->d_partMax;
  da.nD = (int32_t)nD;
  da.nDelay = (int32_t)nDelay;
  da.nTiles = h->nTiles;
  int nPartsUsed = 0, dopGrid = 0, dopTiles = 0;
  if ((rc = tic(h, BLAH2HIP_K_DOPPLER, st))) return rc;
  const int which = pick_doppler(h, n_cpi);
  switch (which) {
  case BLAH2HIP_DOP_TILE8:
  case BLAH2HIP_DOP_TILE8K:
  case BLAH2HIP_DOP_TILE16WG:
  case BLAH2HIP_DOP_TILE16: {
    const int ncol = (which == BLAH2HIP_DOP_TILE8 || which == BLAH2HIP_DOP_TILE8K) ? 8 : 16;
    const int grid = (int)((nDelay + ncol - 1) / ncol);
    // persistent: the resident workgroups (LDS: one of 16 columns or two of 8 per CU) walk the tiles of the batch
    if (which == BLAH2HIP_DOP_TILE16) {
      const size_t lds = (size_t)DOPT1K_LDS_ELEMS * sizeof(cf);
      const int wgs = (int)std::min<int64_t>((int64_t)grid * n_cpi, h->dopGridForce ? h->dopGridForce : h->numCU);
      dopGrid = wgs; dopTiles = grid * (int)n_cpi;
      LDSCFG(doppler_tile1k_kernel<16>, lds);
      hipLaunchKernelGGL(doppler_tile1k_kernel<16>, dim3(wgs), dim3(1024), lds, st, da, (int)n_cpi);
    } else if (which == BLAH2HIP_DOP_TILE8K) {
      const size_t lds = (size_t)dopt1k_lds_elems<8>() * sizeof(cf);
      const int wgs = (int)std::min<int64_t>((int64_t)grid * n_cpi, h->dopGridForce ? h->dopGridForce : 2 * h->numCU);
      dopGrid = wgs; dopTiles = grid * (int)n_cpi;
      LDSCFG(doppler_tile1k_kernel<8>, lds);
      hipLaunchKernelGGL(doppler_tile1k_kernel<8>, dim3(wgs), dim3(512), lds, st, da, (int)n_cpi);
    } else if (ncol == 16) {
      const size_t lds = (size_t)dopt_lds_elems<16>() * sizeof(cf);
      const int wgs = (int)std::min<int64_t>((int64_t)grid * n_cpi, h->dopGridForce ? h->dopGridForce : h->numCU);
      dopGrid = wgs; dopTiles = grid * (int)n_cpi;
      LDSCFG(doppler_tile_kernel<16>, lds);
      hipLaunchKernelGGL(doppler_tile_kernel<16>, dim3(wgs), dim3(1024), lds, st, da, (int)n_cpi);
    } else {
      const size_t lds = (size_t)dopt_lds_elems<8>() * sizeof(cf);
      const int wgs = (int)std::min<int64_t>((int64_t)grid * n_cpi, h->dopGridForce ? h->dopGridForce : 2 * h->numCU);
      dopGrid = wgs; dopTiles = grid * (int)n_cpi;
      LDSCFG(doppler_tile_kernel<8>, lds);
      hipLaunchKernelGGL(doppler_tile_kernel<8>, dim3(wgs), dim3(512), lds, st, da, (int)n_cpi);
    }
    nPartsUsed = grid;
    break;
  }
  case BLAH2HIP_DOP_SUB4: {
    const int subs = (int)((nDelay + DOPS_NCOL - 1) / DOPS_NCOL);
    const size_t lds = (size_t)DOPS_LDS_ELEMS * sizeof(cf);
    LDSCFG(doppler_sub1k_kernel, lds);
    da.tickets = h->d_tickets;
    da.metrics = met;
    dopGrid = subs * (int)n_cpi; dopTiles = dopGrid;
    hipLaunchKernelGGL(doppler_sub1k_kernel, dim3(subs * n_cpi), dim3(64 * DOPS_NCOL), lds, st, da, (int)n_cpi);
    nPartsUsed = 0; // Map::set_metrics is finished inside the kernel
    break;
  }
  case BLAH2HIP_DOP_TILEW: {
    const int grid = (int)((nDelay + DOPW_NCOL - 1) / DOPW_NCOL);
    const size_t lds = (size_t)DOPW_LDS_ELEMS * sizeof(cf);
    LDSCFG(doppler_tilew_kernel, lds);
    // persistent: one workgroup per CU (LDS) walks the tiles of the whole batch
    const int wgs = (int)std::min<int64_t>((int64_t)grid * n_cpi, h->dopGridForce ? h->dopGridForce : h->numCU);
    dopGrid = wgs; dopTiles = grid * (int)n_cpi;
    hipLaunchKernelGGL(doppler_tilew_kernel, dim3(wgs), dim3(64 * DOPW_NCOL), lds, st, da, (int)n_cpi);
    nPartsUsed = grid;
    break;
  }
  case BLAH2HIP_DOP_TILEW4: {
    const int grid = (int)((nDelay + DOPW4_NCOL - 1) / DOPW4_NCOL);
    const size_t lds = (size_t)DOPW4_LDS_ELEMS * sizeof(cf);
    LDSCFG(doppler_tilew4_kernel, lds);
    // persistent: one workgroup per CU (LDS) walks the half tiles of the whole batch
    const int wgs = (int)std::min<int64_t>((int64_t)grid * n_cpi, h->dopGridForce ? h->dopGridForce : h->numCU);
    dopGrid = wgs; dopTiles = grid * (int)n_cpi;
    hipLaunchKernelGGL(doppler_tilew4_kernel, dim3(wgs), dim3(64 * DOPW4_NCOL), lds, st, da, (int)n_cpi);
    nPartsUsed = grid;
    break;
  }
  case BLAH2HIP_DOP_TILEW2: {
    const int tiles = (int)((nDelay + DOPW2_NCOL - 1) / DOPW2_NCOL);
    const int groups = (int)((nDelay + 15) / 16);
    const size_t lds = (size_t)DOPW2_LDS_ELEMS * sizeof(cf);
    LDSCFG(doppler_tilew2_kernel, lds);
    // persistent: one workgroup per CU (LDS); a multiple of 32 (the four quarters of a 16-column tile on one XCD)
    const int64_t want = (int64_t)groups * n_cpi * 4;
    int wgs = (int)std::min<int64_t>(want, h->dopGridForce ? h->dopGridForce : h->numCU);
    wgs = std::max(32, (wgs + 31) & ~31);
    if (!h->dopGridForce && wgs > h->numCU) wgs = std::max(32, h->numCU & ~31);
    dopGrid = wgs; dopTiles = groups * (int)n_cpi * 4;
    hipLaunchKernelGGL(doppler_tilew2_kernel, dim3(wgs), dim3(128 * DOPW2_NCOL), lds, st, da, (int)n_cpi);
    nPartsUsed = tiles;
    break;
  }
  case BLAH2HIP_DOP_TILEM: {
    if (h->dopR3 == 8) {
      const int grid = (int)((nDelay + DopM<8>::NCOL - 1) / DopM<8>::NCOL);
      const size_t lds = (size_t)DopM<8>::LDS_ELEMS * sizeof(cf);
      LDSCFG(doppler_tilem_kernel<8>, lds);
      hipLaunchKernelGGL(doppler_tilem_kernel<8>, dim3(grid, n_cpi), dim3(1024), lds, st, da);
      nPartsUsed = grid;
    } else {
      const int grid = (int)((nDelay + DopM<16>::NCOL - 1) / DopM<16>::NCOL);
      const size_t lds = (size_t)DopM<16>::LDS_ELEMS * sizeof(cf);
      LDSCFG(doppler_tilem_kernel<16>, lds);
      hipLaunchKernelGGL(doppler_tilem_kernel<16>, dim3(grid, n_cpi), dim3(1024), lds, st, da);
      nPartsUsed = grid;
    }
    break;
  }
  case BLAH2HIP_DOP_COLUMN:
    if (h->dopR3 == 4) rc = launch_doppler_t<4>(h, da, n_cpi, st);
    else if (h->dopR3 == 8) rc = launch_doppler_t<8>(h, da, n_cpi, st);
    else rc = launch_doppler_t<16>(h, da, n_cpi, st);
    if (rc) return rc;
    nPartsUsed = h->dopGridX;
    break;
  default:
    hipLaunchKernelGGL(doppler_dft_kernel, dim3(h->dopTilesX, h->dopTilesY, n_cpi), dim3(64 * DOP_WAVES), 0, st, da);
    nPartsUsed = h->dopTilesX * h->dopTilesY;
    break;
  }
  HIPCHK(hipGetLastError());
  h->lastDoppler = which;
  h->dopGridLast = dopGrid;
  h->dopTilesLast = dopTiles;
  // fp64 columns under the tallest peaks (hot_columns_kernel), inside the Doppler bracket like the leak fix below
  h->lastHot = false;
  if (h->hotMode != 0 && !h->inLeakCal && h->d_dopW64 &&
      (h->hotMode == 2 || 1.34 * std::sqrt((double)h->dims.n_used) >= HOT_RATIO)) { // |peak| <= sqrt(N) sx sy, mean level ~ 0.75 sqrt(N) sx sy
    HotArgs ha;
    ha.R = h->d_R; ha.map = map; ha.W = h->d_dopW64;
    ha.partSum = h->d_partSum; ha.metrics = met; ha.count = h->d_hotCount;
    ha.nParts = nPartsUsed; ha.partStride = nPartsUsed; ha.nD = (int32_t)nD; ha.nDelay = (int32_t)nDelay; ha.nTiles = h->nTiles;
    ha.ratioDb = (float)(10.0 * std::log10(HOT_RATIO));
    ha.groups = n_cpi >= 4 ? 4 : 1;
    const size_t lds = 2 * (size_t)nD * sizeof(double2);
    LDSCFG(hot_columns_kernel, (size_t)2 * HOT_ND_MAX * sizeof(double2));
    hipLaunchKernelGGL(hot_columns_kernel, dim3((nD + HOT_ROWS * ha.groups - 1) / (HOT_ROWS * ha.groups), n_cpi), dim3(256), lds, st, ha);
    HIPCHK(hipGetLastError());
    h->lastHot = true;
    h->lastHotStream = st;
  }
  if (leak) { // inside the Doppler bracket: one 64-thread workgroup per CPI on a few dozen cells of the zero-Doppler row
    leak_fix_kernel<<<dim3(n_cpi), dim3(64), 0, st>>>(map, (size_t)nD * nDelay, (size_t)leak_row0(h) * nDelay, leak_col0(h),
                                                      leak->nLags, leak->d_lag, leak->d_g);
    HIPCHK(hipGetLastError());
  }
  if ((rc = toc(h, BLAH2HIP_K_DOPPLER, st))) return rc;

  if (nPartsUsed) {
    if ((rc = tic(h, BLAH2HIP_K_METRICS, st))) return rc;
    hipLaunchKernelGGL(metrics_kernel, dim3(n_cpi), dim3(256), 0, st, h->d_partSum, h->d_partMax,
                       nPartsUsed, (double)nD * (double)nDelay, met);
    HIPCHK(hipGetLastError());
    if ((rc = toc(h, BLAH2HIP_K_METRICS, st))) return rc;
  }
  return BLAH2HIP_OK;
}

int blah2hip_amb_read_last(blah2hip_amb_t h, uint32_t cpi, float *map_out, double *metrics)
{
  if (!h) return fail(BLAH2HIP_ERR_INVALID, "NULL handle");
  if (cpi >= h->dims.max_batch) return fail(BLAH2HIP_ERR_INVALID, "cpi index out of range");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());
  const size_t cells = (size_t)h->dims.n_doppler_bins * h->dims.n_delay_bins;
  if (map_out) HIPCHK(hipMemcpy(map_out, h->d_map + cells * cpi, cells * sizeof(cf), hipMemcpyDeviceToHost));
  if (metrics) HIPCHK(hipMemcpy(metrics, h->d_metrics + 2 * cpi, 2 * sizeof(double), hipMemcpyDeviceToHost));
  return BLAH2HIP_OK;
}

int blah2hip_amb_db_dev(blah2hip_amb_t h, const void *d_map, const double *d_metrics, uint32_t n_cpi, float *d_db,
                        void *stream)
{
  if (!h || !d_db) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n_cpi == 0 || n_cpi > h->dims.max_batch) return fail(BLAH2HIP_ERR_INVALID, "n_cpi outside [1, max_batch]");
  HIPCHK(hipSetDevice(h->device));
  const uint32_t cells = h->dims.n_doppler_bins * h->dims.n_delay_bins;
  const cf *map = d_map ? (const cf *)d_map : h->d_map;
  const double *met = d_metrics ? d_metrics : h->d_metrics;
  const int gx = (int)std::min<uint32_t>((cells + 255) / 256, 4u * (uint32_t)h->numCU);
  hipLaunchKernelGGL(db_map_kernel, dim3(gx, n_cpi), dim3(256), 0, (hipStream_t)stream, map, met, d_db, cells);
  HIPCHK(hipGetLastError());
  return BLAH2HIP_OK;
}

int blah2hip_amb_process_c32(blah2hip_amb_t h, const float *x, const float *y, uint32_t n,
                             float *map_out, double *metrics)
{
  if (!h || !x || !y) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n < h->dims.n_used) return fail(BLAH2HIP_ERR_UNDERFLOW, "Attempting to pop from an empty deque");
  HIPCHK(hipSetDevice(h->device));
  const size_t bytes = (size_t)n * sizeof(cf);
  int rc;
  if ((rc = ensure_staging(h, 2 * bytes))) return rc;
  char *dx = (char *)h->d_in, *dy = dx + bytes;
  HIPCHK(hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dy, y, bytes, hipMemcpyHostToDevice, h->stream));
  if ((rc = blah2hip_amb_process_dev(h, BLAH2HIP_FMT_C32, dx, dy, 1, n, nullptr, nullptr, h->stream))) return rc;
  return host_tail(h, map_out, metrics);
}

int blah2hip_amb_process_c64(blah2hip_amb_t h, const double *x, const double *y, uint32_t n,
                             float *map_out, double *metrics)
{
  if (!h || !x || !y) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n < h->dims.n_used) return fail(BLAH2HIP_ERR_UNDERFLOW, "Attempting to pop from an empty deque");
  HIPCHK(hipSetDevice(h->device));
  // IqData holds complex<double>; the kernels compute in fp32 (BASELINE.json).  Narrowed
  // into a pinned staging buffer that lives with the handle: no per-CPI allocation or page
  // faults, and the two H2D copies run at the full PCIe rate.  The channels are converted
  // and sent one after the other so that y's conversion overlaps x's copy.
  const size_t bytes = (size_t)n * sizeof(cf);
  if (h->h_pin_bytes < 2 * bytes) {
    if (h->h_pin) HIPCHK(hipHostFree(h->h_pin));
    h->h_pin = nullptr;
    h->h_pin_bytes = 0;
    HIPCHK(hipHostMalloc(&h->h_pin, 2 * bytes, hipHostMallocDefault));
    h->h_pin_bytes = 2 * bytes;
  }
  int rc;
  if ((rc = ensure_staging(h, 2 * bytes))) return rc;
  float *px = (float *)h->h_pin, *py = px + 2 * (size_t)n;
  char *dx = (char *)h->d_in, *dy = dx + bytes;
  for (size_t i = 0; i < 2 * (size_t)n; i++) px[i] = (float)x[i];
  HIPCHK(hipMemcpyAsync(dx, px, bytes, hipMemcpyHostToDevice, h->stream));
  for (size_t i = 0; i < 2 * (size_t)n; i++) py[i] = (float)y[i];
  HIPCHK(hipMemcpyAsync(dy, py, bytes, hipMemcpyHostToDevice, h->stream));
  if ((rc = blah2hip_amb_process_dev(h, BLAH2HIP_FMT_C32, dx, dy, 1, n, nullptr, nullptr, h->stream))) return rc;
  return host_tail(h, map_out, metrics);
}

int blah2hip_amb_process_i16(blah2hip_amb_t h, const int16_t *iq, uint32_t n, float *map_out,
                             double *metrics)
{
  if (!h || !iq) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n < h->dims.n_used) return fail(BLAH2HIP_ERR_UNDERFLOW, "Attempting to pop from an empty deque");
  HIPCHK(hipSetDevice(h->device));
  const size_t bytes = (size_t)n * 4 * sizeof(int16_t);
  int rc;
  if ((rc = ensure_staging(h, bytes))) return rc;
  HIPCHK(hipMemcpyAsync(h->d_in, iq, bytes, hipMemcpyHostToDevice, h->stream));
  if ((rc = blah2hip_amb_process_dev(h, BLAH2HIP_FMT_I16, h->d_in, nullptr, 1, n, nullptr, nullptr, h->stream))) return rc;
  return host_tail(h, map_out, metrics);
}

// ------------------------------------------------------------------- CFAR --
int blah2hip_cfar1d_dev(blah2hip_amb_t h, const void *d_map, const double *d_metrics,
                        uint32_t n_cpi, double pfa, int32_t n_guard, int32_t n_train,
                        int32_t min_delay, double min_doppler, blah2hip_hit_t *d_hits, uint32_t cap,
                        uint32_t *d_count, void *stream)
{
  if (!h || !d_hits || !d_count) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n_cpi == 0 || n_cpi > h->dims.max_batch) return fail(BLAH2HIP_ERR_INVALID, "n_cpi outside [1, max_batch]");
  // CfarDetector1D's ctor parameters are int8_t (CfarDetector1D.h:46)
  if (n_guard < 0 || n_guard > 127 || n_train < 0 || n_train > 127 || min_delay < -128 || min_delay > 127)
    return fail(BLAH2HIP_ERR_INVALID, "nGuard/nTrain/minDelay outside int8 range");
  HIPCHK(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  const double *d_alpha = nullptr;
  int rc;
  if ((rc = alpha_table(h, pfa, (size_t)(2 * n_train), &d_alpha, st))) return rc;
  HIPCHK(hipMemsetAsync(d_count, 0, n_cpi * sizeof(uint32_t), st));
  CfarArgs a;
  a.map = d_map ? (const cf *)d_map : h->d_map;
  a.metrics = d_metrics ? d_metrics : h->d_metrics;
  a.doppler = h->d_doppler;
  a.alpha = d_alpha;
  a.hits = d_hits;
  a.count = d_count;
  a.nD = (int32_t)h->dims.n_doppler_bins;
  a.nDelay = (int32_t)h->dims.n_delay_bins;
  a.delayMin = h->delayMin;
  a.delayAxis = nullptr; // the engine's own axis: delayMin + j
  a.nGuard = n_guard; a.nTrain = n_train; a.minDelay = min_delay;
  a.minDoppler = min_doppler;
  a.cap = cap;
  if ((rc = tic(h, BLAH2HIP_K_CFAR, st))) return rc;
  {
    // the row as fp64 |z|^2 in LDS while it fits (150 KB: 19 200 delay bins), straight from L2 beyond
    const size_t rowBytes = (size_t)a.nDelay * sizeof(double);
    if (rowBytes <= 150 * 1024) {
      if (rowBytes > 48 * 1024) LDSCFG(cfar1d_kernel<true>, 150 * 1024); // the attribute is set once per (device, kernel): its largest use
      hipLaunchKernelGGL(cfar1d_kernel<true>, dim3(a.nD, n_cpi), dim3(256), rowBytes, st, a);
    } else {
      hipLaunchKernelGGL(cfar1d_kernel<false>, dim3(a.nD, n_cpi), dim3(256), 0, st, a);
    }
  }
  HIPCHK(hipGetLastError());
  if ((rc = toc(h, BLAH2HIP_K_CFAR, st))) return rc;
  return BLAH2HIP_OK;
}

int blah2hip_cfar1d_process(blah2hip_amb_t h, uint32_t cpi, double pfa, int32_t n_guard,
                            int32_t n_train, int32_t min_delay, double min_doppler, double *delay,
                            double *doppler, double *snr, uint32_t cap, uint32_t *count)
{
  if (!h || !count) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (cpi >= h->dims.max_batch) return fail(BLAH2HIP_ERR_INVALID, "cpi index out of range");
  HIPCHK(hipSetDevice(h->device));
  const size_t cells = (size_t)h->dims.n_doppler_bins * h->dims.n_delay_bins;
  // worst case every cell fires; size the device list for that once
  if (h->hitCap < cells) {
    if (h->d_hits) HIPCHK(hipFree(h->d_hits));
    h->d_hits = nullptr;
    HIPCHK(hipMalloc(&h->d_hits, cells * sizeof(blah2hip_hit_t)));
    h->hitCap = (uint32_t)cells;
  }
  // run on a one-CPI view of the internal buffers
  const cf *m = h->d_map + cells * cpi;
  const double *met = h->d_metrics + 2 * cpi;
  int rc = blah2hip_cfar1d_dev(h, m, met, 1, pfa, n_guard, n_train, min_delay, min_doppler, h->d_hits,
                               h->hitCap, h->d_count, h->stream);
  if (rc) return rc;
  uint32_t n = 0;
  HIPCHK(hipMemcpyAsync(&n, h->d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  n = std::min(n, h->hitCap);
  std::vector<blah2hip_hit_t> hits(n);
  if (n) HIPCHK(hipMemcpy(hits.data(), h->d_hits, n * sizeof(blah2hip_hit_t), hipMemcpyDeviceToHost));
  // the reference emits row by row, then by delay (CfarDetector1D.cpp:36-92)
  std::sort(hits.begin(), hits.end(), [](const blah2hip_hit_t &a, const blah2hip_hit_t &b) {
    return a.row != b.row ? a.row < b.row : a.col < b.col;
  });
  *count = n;
  if (n > cap) return fail(BLAH2HIP_ERR_CAPACITY, "detection capacity too small");
  for (uint32_t i = 0; i < n; i++) {
    if (delay) delay[i] = (double)(hits[i].col + h->delayAxis[0]); // :88  j + x->delay[0]
    if (doppler) doppler[i] = h->dopplerAxis[hits[i].row];         // :89
    if (snr) snr[i] = hits[i].snr;                                 // :90
  }
  return BLAH2HIP_OK;
}

// CfarDetector1D::process on a map that lives on the HOST (any Map<complex<double>>, not only one
// the engine produced): uploads it, runs cfar1d_kernel, frees.  noise_power is Map::noisePower.
int blah2hip_cfar1d_map(const float *map, uint32_t n_doppler, uint32_t n_delay, const int32_t *delay_axis,
                        const double *doppler_axis, double noise_power, double pfa, int32_t n_guard,
                        int32_t n_train, int32_t min_delay, double min_doppler, int device, double *delay,
                        double *doppler, double *snr, uint32_t cap, uint32_t *count)
{
  if (!map || !delay_axis || !doppler_axis || !count) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n_doppler == 0 || n_delay == 0) return fail(BLAH2HIP_ERR_INVALID, "empty map");
  if (n_guard < 0 || n_guard > 127 || n_train < 0 || n_train > 127 || min_delay < -128 || min_delay > 127)
    return fail(BLAH2HIP_ERR_INVALID, "nGuard/nTrain/minDelay outside int8 range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(BLAH2HIP_ERR_NO_DEVICE, "no HIP device visible (the HIP path is the only path)");
  if (device < 0 || device >= ndev) return fail(BLAH2HIP_ERR_INVALID, "device index out of range");
  HIPCHK(hipSetDevice(device));
  const size_t cells = (size_t)n_doppler * n_delay;
  std::vector<double> alpha(2 * n_train + 1);
  alpha[0] = std::nan("");
  for (int n = 1; n <= 2 * n_train; n++) alpha[n] = n * (pow(pfa, -1.0 / n) - 1);
  const double met[2] = {noise_power, 0.0};
  char *pool = nullptr; // one allocation: map | hits | doppler | alpha | metrics | delay axis | count
  const size_t oMap = 0, oHits = oMap + cells * sizeof(cf), oDop = oHits + cells * sizeof(blah2hip_hit_t),
               oAlpha = oDop + n_doppler * sizeof(double), oMet = oAlpha + alpha.size() * sizeof(double),
               oAxis = oMet + 2 * sizeof(double), oCnt = oAxis + n_delay * sizeof(int32_t), total = oCnt + sizeof(uint32_t);
  HIPCHK(hipMalloc(&pool, total));
  std::vector<blah2hip_hit_t> hits;
  uint32_t n = 0;
  auto run = [&]() -> int {
    HIPCHK(hipMemcpy(pool + oMap, map, cells * sizeof(cf), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pool + oDop, doppler_axis, n_doppler * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pool + oAlpha, alpha.data(), alpha.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pool + oMet, met, sizeof met, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pool + oAxis, delay_axis, n_delay * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(pool + oCnt, 0, sizeof(uint32_t)));
    CfarArgs a;
    a.map = (const cf *)(pool + oMap);
    a.metrics = (const double *)(pool + oMet);
    a.doppler = (const double *)(pool + oDop);
    a.alpha = (const double *)(pool + oAlpha);
    a.hits = (blah2hip_hit_t *)(pool + oHits);
    a.count = (uint32_t *)(pool + oCnt);
    a.nD = (int32_t)n_doppler; a.nDelay = (int32_t)n_delay; a.delayMin = delay_axis[0];
    a.delayAxis = (const int32_t *)(pool + oAxis); // x->delay[j] of a caller-built map need not be delay[0] + j (CfarDetector1D.cpp:53)
    a.nGuard = n_guard; a.nTrain = n_train; a.minDelay = min_delay; a.minDoppler = min_doppler;
    a.cap = (uint32_t)cells;
    const size_t rowBytes = (size_t)a.nDelay * sizeof(double);
    if (rowBytes <= 150 * 1024) {
      if (rowBytes > 48 * 1024) HIPCHK(blah2hip_ensure_lds_((const void *)cfar1d_kernel<true>, 150 * 1024));
      hipLaunchKernelGGL(cfar1d_kernel<true>, dim3(a.nD, 1), dim3(256), rowBytes, 0, a);
    } else {
      hipLaunchKernelGGL(cfar1d_kernel<false>, dim3(a.nD, 1), dim3(256), 0, 0, a);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(&n, pool + oCnt, sizeof(uint32_t), hipMemcpyDeviceToHost)); // synchronises with the null stream
    hits.resize(n);
    if (n) HIPCHK(hipMemcpy(hits.data(), pool + oHits, n * sizeof(blah2hip_hit_t), hipMemcpyDeviceToHost));
    return BLAH2HIP_OK;
  };
  const int rc = run();
  (void)hipFree(pool);
  if (rc) return rc;
  std::sort(hits.begin(), hits.end(), [](const blah2hip_hit_t &a, const blah2hip_hit_t &b) {
    return a.row != b.row ? a.row < b.row : a.col < b.col;
  });
  *count = n;
  if (n > cap) return fail(BLAH2HIP_ERR_CAPACITY, "detection capacity too small");
  for (uint32_t i = 0; i < n; i++) {
    if (delay) delay[i] = (double)(hits[i].col + delay_axis[0]); // CfarDetector1D.cpp:88
    if (doppler) doppler[i] = doppler_axis[hits[i].row];         // :89
    if (snr) snr[i] = hits[i].snr;                               // :90
  }
  return BLAH2HIP_OK;
}

// ---------------------------------------------------------------- 2-D CFAR --
int blah2hip_cfar2d_dev(blah2hip_amb_t h, const void *d_map, const double *d_metrics, uint32_t n_cpi,
                        double pfa, int32_t ngd, int32_t ntd, int32_t ngf, int32_t ntf, int32_t min_delay,
                        double min_doppler, blah2hip_hit_t *d_hits, uint32_t cap, uint32_t *d_count,
                        void *stream)
{
  if (!h || !d_hits || !d_count) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n_cpi == 0 || n_cpi > h->dims.max_batch) return fail(BLAH2HIP_ERR_INVALID, "n_cpi outside [1, max_batch]");
  for (int32_t v : {ngd, ntd, ngf, ntf})
    if (v < 0 || v > 127) return fail(BLAH2HIP_ERR_INVALID, "guard/train sizes outside [0, 127]");
  if (min_delay < -128 || min_delay > 127) return fail(BLAH2HIP_ERR_INVALID, "minDelay outside int8 range");
  HIPCHK(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  const int nD = (int)h->dims.n_doppler_bins, nC = (int)h->dims.n_delay_bins;
  int rc;
  if (h->cfar2dForce == BLAH2HIP_CFAR2D_TILE && !cfar2d_use_tile(h, ngd, ntd, ngf, ntf))
    return fail(BLAH2HIP_ERR_UNSUPPORTED, "2-D window beyond the tile kernel's halo (nGf + nTf <= 24, nGd + nTd <= 40)");
  if (h->cfar2dForce == BLAH2HIP_CFAR2D_STREAM && !cfar2d_stream_shape(ngd, ntd, ngf, ntf))
    return fail(BLAH2HIP_ERR_UNSUPPORTED, "the stream kernel is not instantiated for this 2-D window (C2S_SHAPES in cfar_kernels.hpp)");
  {
    int lo, hi; // the handle's Doppler axis is monotonic (Ambiguity.cpp:60-66), so this cannot fail today; a forced kernel never runs another one
    if (h->cfar2dForce == BLAH2HIP_CFAR2D_STREAM && !cfar2d_dead_rows(h, min_doppler, &lo, &hi))
      return fail(BLAH2HIP_ERR_UNSUPPORTED, "the stream kernel needs the rows below minDoppler to be one interval of the Doppler axis");
  }
  if (!cfar2d_use_stream(h, ngd, ntd, ngf, ntf, min_doppler) && !cfar2d_use_tile(h, ngd, ntd, ngf, ntf) && (rc = ensure_sat(h))) return rc; // no-op once allocated
  const double *d_alpha = nullptr;
  if ((rc = alpha_table(h, pfa, (size_t)(2 * (ngd + ntd) + 1) * (2 * (ngf + ntf) + 1), &d_alpha, st))) return rc;
  HIPCHK(hipMemsetAsync(d_count, 0, n_cpi * sizeof(uint32_t), st));
  Cfar2dArgs a;
  a.map = d_map ? (const cf *)d_map : h->d_map;
  a.metrics = d_metrics ? d_metrics : h->d_metrics;
  a.doppler = h->d_doppler;
  a.alpha = d_alpha;
  a.sat = h->d_sat;
  a.hits = d_hits;
  a.count = d_count;
  a.nD = nD; a.nDelay = nC; a.delayMin = h->delayMin;
  a.ngD = ngd; a.ntD = ntd; a.ngF = ngf; a.ntF = ntf; a.minDelay = min_delay;
  a.minDoppler = min_doppler;
  a.cap = cap;
  if (cfar2d_use_stream(h, ngd, ntd, ngf, ntf, min_doppler)) {
    if ((rc = tic(h, BLAH2HIP_K_CFAR, st))) return rc;
    cfar2d_stream_launch(h, a, n_cpi, st);
    HIPCHK(hipGetLastError());
    if ((rc = toc(h, BLAH2HIP_K_CFAR, st))) return rc;
    return BLAH2HIP_OK;
  }
  if (cfar2d_use_tile(h, ngd, ntd, ngf, ntf)) {
    Cfar2dTileArgs ta;
    ta.d = a;
    ta.nCpi = (int32_t)n_cpi;
    ta.rowsOut = C2T_ROWS - 2 * (ngf + ntf);
    ta.tilesX = (nC + C2T_COLS - 1) / C2T_COLS;
    ta.tilesY = (nD + ta.rowsOut - 1) / ta.rowsOut;
    const int hC = ngd + ntd;
    // the whole threshold table in LDS when it fits beside the tile (the default 17 x 9 window: 154 entries)
    const int64_t tableN = (int64_t)(2 * (ngd + ntd) + 1) * (2 * (ngf + ntf) + 1) + 1;
    ta.alphaLds = (tableN <= 2048 && c2t_lds_bytes(hC, (int)tableN) <= 160 * 1024) ? (int32_t)tableN : 0;
    const size_t lds = c2t_lds_bytes(hC, ta.alphaLds);
    const int64_t nAll = (int64_t)ta.tilesX * ta.tilesY * n_cpi;
    const int grid = (int)std::max<int64_t>(8, (std::min<int64_t>(nAll, h->numCU) + 7) & ~7); // one persistent workgroup per CU (LDS)
    if ((rc = tic(h, BLAH2HIP_K_CFAR, st))) return rc;
    auto launch = [&](auto kern) -> int {
      LDSCFG(kern, lds);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * C2T_WAVES), lds, st, ta);
      return BLAH2HIP_OK;
    };
    const bool two = C2T_COLS + 2 * hC <= 128;
    if (two) rc = ta.alphaLds ? launch(cfar2d_tile_kernel<2, true>) : launch(cfar2d_tile_kernel<2, false>);
    else rc = ta.alphaLds ? launch(cfar2d_tile_kernel<3, true>) : launch(cfar2d_tile_kernel<3, false>);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    if ((rc = toc(h, BLAH2HIP_K_CFAR, st))) return rc;
    return BLAH2HIP_OK;
  }
  if ((rc = tic(h, BLAH2HIP_K_SAT_ROWS, st))) return rc;
  hipLaunchKernelGGL(sat_rows_kernel, dim3(nD, n_cpi), dim3(256), 0, st, a);
  if ((rc = toc(h, BLAH2HIP_K_SAT_ROWS, st))) return rc;
  if ((rc = tic(h, BLAH2HIP_K_SAT_COLS, st))) return rc;
  hipLaunchKernelGGL(sat_cols_kernel, dim3((nC + 63) / 64, n_cpi), dim3(64), 0, st, a);
  if ((rc = toc(h, BLAH2HIP_K_SAT_COLS, st))) return rc;
  if ((rc = tic(h, BLAH2HIP_K_CFAR, st))) return rc;
  hipLaunchKernelGGL(cfar2d_kernel, dim3((nC + 255) / 256, nD, n_cpi), dim3(256), 0, st, a);
  HIPCHK(hipGetLastError());
  if ((rc = toc(h, BLAH2HIP_K_CFAR, st))) return rc;
  return BLAH2HIP_OK;
}

int blah2hip_cfar1d_prepare(blah2hip_amb_t h, double pfa, int32_t n_train)
{
  if (!h) return fail(BLAH2HIP_ERR_INVALID, "NULL handle");
  if (n_train < 0 || n_train > 127) return fail(BLAH2HIP_ERR_INVALID, "nTrain outside int8 range");
  HIPCHK(hipSetDevice(h->device));
  const double *d = nullptr;
  return alpha_table(h, pfa, (size_t)(2 * n_train), &d, nullptr, true);
}

int blah2hip_cfar2d_prepare(blah2hip_amb_t h, double pfa, int32_t ngd, int32_t ntd, int32_t ngf, int32_t ntf)
{
  if (!h) return fail(BLAH2HIP_ERR_INVALID, "NULL handle");
  for (int32_t v : {ngd, ntd, ngf, ntf})
    if (v < 0 || v > 127) return fail(BLAH2HIP_ERR_INVALID, "guard/train sizes outside [0, 127]");
  HIPCHK(hipSetDevice(h->device));
  int rc;
  if (h->cfar2dForce == BLAH2HIP_CFAR2D_TILE && !cfar2d_use_tile(h, ngd, ntd, ngf, ntf))
    return fail(BLAH2HIP_ERR_UNSUPPORTED, "2-D window beyond the tile kernel's halo (nGf + nTf <= 24, nGd + nTd <= 40)");
  if (h->cfar2dForce == BLAH2HIP_CFAR2D_STREAM && !cfar2d_stream_shape(ngd, ntd, ngf, ntf))
    return fail(BLAH2HIP_ERR_UNSUPPORTED, "the stream kernel is not instantiated for this 2-D window (C2S_SHAPES in cfar_kernels.hpp)");
  if (!cfar2d_use_stream(h, ngd, ntd, ngf, ntf, 0.0) && !cfar2d_use_tile(h, ngd, ntd, ngf, ntf) && (rc = ensure_sat(h))) return rc;
  const double *d = nullptr;
  return alpha_table(h, pfa, (size_t)(2 * (ngd + ntd) + 1) * (2 * (ngf + ntf) + 1), &d, nullptr, true);
}

int blah2hip_cfar2d_process(blah2hip_amb_t h, uint32_t cpi, double pfa, int32_t ngd, int32_t ntd,
                            int32_t ngf, int32_t ntf, int32_t min_delay, double min_doppler, double *delay,
                            double *doppler, double *snr, uint32_t cap, uint32_t *count)
{
  if (!h || !count) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (cpi >= h->dims.max_batch) return fail(BLAH2HIP_ERR_INVALID, "cpi index out of range");
  HIPCHK(hipSetDevice(h->device));
  const size_t cells = (size_t)h->dims.n_doppler_bins * h->dims.n_delay_bins;
  if (h->hitCap < cells) {
    if (h->d_hits) HIPCHK(hipFree(h->d_hits));
    h->d_hits = nullptr;
    HIPCHK(hipMalloc(&h->d_hits, cells * sizeof(blah2hip_hit_t)));
    h->hitCap = (uint32_t)cells;
  }
  int rc = blah2hip_cfar2d_dev(h, h->d_map + cells * cpi, h->d_metrics + 2 * cpi, 1, pfa, ngd, ntd, ngf, ntf,
                               min_delay, min_doppler, h->d_hits, h->hitCap, h->d_count, h->stream);
  if (rc) return rc;
  uint32_t n = 0;
  HIPCHK(hipMemcpyAsync(&n, h->d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  n = std::min(n, h->hitCap);
  std::vector<blah2hip_hit_t> hits(n);
  if (n) HIPCHK(hipMemcpy(hits.data(), h->d_hits, n * sizeof(blah2hip_hit_t), hipMemcpyDeviceToHost));
  std::sort(hits.begin(), hits.end(), [](const blah2hip_hit_t &a, const blah2hip_hit_t &b) {
    return a.row != b.row ? a.row < b.row : a.col < b.col;
  });
  *count = n;
  if (n > cap) return fail(BLAH2HIP_ERR_CAPACITY, "detection capacity too small");
  for (uint32_t i = 0; i < n; i++) {
    if (delay) delay[i] = (double)(hits[i].col + h->delayAxis[0]);
    if (doppler) doppler[i] = h->dopplerAxis[hits[i].row];
    if (snr) snr[i] = hits[i].snr;
  }
  return BLAH2HIP_OK;
}

// ----------------------------------------------------------------- timing --
int blah2hip_amb_set_timing(blah2hip_amb_t h, int enable)
{
  if (!h) return fail(BLAH2HIP_ERR_INVALID, "NULL handle");
  h->timer.enabled = enable != 0;
  return BLAH2HIP_OK;
}

int blah2hip_amb_get_timing(blah2hip_amb_t h, double *ms_total, uint32_t *launches)
{
  if (!h || !ms_total || !launches) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(h->timer.collect(ms_total, launches));
  return BLAH2HIP_OK;
}

// ------------------------------------------------- centroid / interpolate --
// Host arithmetic on a handful of detections (O(n^2) over tens of items).
int blah2hip_centroid(const double *delay, const double *doppler, const double *snr, uint32_t count,
                      uint16_t n_delay, uint16_t n_doppler, double resolution_doppler,
                      double *delay_out, double *doppler_out, double *snr_out, uint32_t *count_out)
{
  if (!count_out || (count && (!delay || !doppler || !snr || !delay_out || !doppler_out || !snr_out)))
    return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  uint32_t kept = 0;
  for (uint32_t i = 0; i < count; i++) {
    // Centroid.cpp:36-39: uint16_t limits -> (int)delay - n_delay wraps when negative
    const uint16_t lo = (uint16_t)((int)delay[i] - (int)n_delay);
    const uint16_t hi = (uint16_t)((int)delay[i] + (int)n_delay);
    const double flo = doppler[i] - (n_doppler * resolution_doppler);
    const double fhi = doppler[i] + (n_doppler * resolution_doppler);
    bool keep = true;
    for (uint32_t j = 0; j < count && keep; j++) {
      if (j == i) continue;
      if (delay[j] > lo && delay[j] < hi && doppler[j] > flo && doppler[j] < fhi && snr[i] < snr[j]) keep = false;
    }
    if (keep) {
      delay_out[kept] = delay[i];
      doppler_out[kept] = doppler[i];
      snr_out[kept] = snr[i];
      kept++;
    }
  }
  *count_out = kept;
  return BLAH2HIP_OK;
}

int blah2hip_interpolate(const double *delay, const double *doppler, const double *snr, uint32_t count,
                         const float *map, uint32_t n_doppler, uint32_t n_delay, const int32_t *delay_axis,
                         const double *doppler_axis, double noise_power, int do_delay, int do_doppler,
                         double *delay_out, double *doppler_out, double *snr_out, uint32_t *count_out)
{
  if (!count_out || !map || !delay_axis || !doppler_axis) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (count && (!delay || !doppler || !snr || !delay_out || !doppler_out || !snr_out))
    return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  // 10*log10|z| - noisePower of one map cell (Interpolate.cpp:50-52)
  auto cell = [&](int64_t row, int64_t col) -> double {
    if (row < 0 || row >= (int64_t)n_doppler || col < 0 || col >= (int64_t)n_delay)
      return std::nan(""); // the reference would index out of bounds here
    const float *z = map + 2 * ((size_t)row * n_delay + (size_t)col);
    return 10.0 * std::log10(std::hypot((double)z[0], (double)z[1])) - noise_power;
  };
  // Map::doppler_hz_to_bin: exact match, 0 on a miss (Map.cpp:102-113)
  auto row_of = [&](double hz) -> int64_t {
    for (uint32_t r = 0; r < n_doppler; r++)
      if (doppler_axis[r] == hz) return r;
    return 0;
  };
  uint32_t kept = 0;
  for (uint32_t i = 0; i < count; i++) {
    double intDelay = delay[i], intDoppler = doppler[i], intSnrDelay = snr[i];
    const double intSnrDoppler = snr[i]; // never updated in the reference (:80 writes intSnrDelay)
    const int64_t row = row_of(doppler[i]);
    const int64_t col = (int64_t)(delay[i] - delay_axis[0]);
    if (do_delay) {
      if (delay[i] == delay_axis[0] || delay[i] == delay_axis[n_delay - 1]) continue; // :46-49
      const double s0 = cell(row, col - 1), s1 = cell(row, col), s2 = cell(row, col + 1);
      if (s1 < s0 || s1 < s2) continue; // :54-58 dropped (peak lower than a neighbour)
      double off = (s0 - s2) / (2 * (s0 - (2 * s1) + s2));
      intSnrDelay = s1 - (((s0 - s2) * off) / 4);
      intDelay = delay[i] + off;
    }
    if (do_doppler) {
      if (doppler[i] == doppler_axis[0] || doppler[i] == doppler_axis[n_doppler - 1]) continue; // :67-70
      const double s0 = cell(row - 1, col), s1 = cell(row, col), s2 = cell(row + 1, col);
      if (s1 < s0 || s1 < s2) continue;
      double off = (s0 - s2) / (2 * (s0 - (2 * s1) + s2));
      intSnrDelay = s1 - (((s0 - s2) * off) / 4); // sic, :80
      intDoppler = doppler[i] + ((doppler_axis[1] - doppler_axis[0]) * off);
    }
    delay_out[kept] = intDelay;
    doppler_out[kept] = intDoppler;
    snr_out[kept] = std::max(std::max(intSnrDelay, intSnrDoppler), snr[i]); // :88
    kept++;
  }
  *count_out = kept;
  return BLAH2HIP_OK;
}

// ---------------------------------------------------------- device context --
struct blah2hip_ctx_s {
  int device = 0;
  hipStream_t stream = nullptr;
};

int blah2hip_ctx_create(int device, blah2hip_ctx_t *out)
{
  if (!out) return fail(BLAH2HIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(BLAH2HIP_ERR_NO_DEVICE, "no HIP device visible (the HIP path is the only path)");
  if (device < 0 || device >= ndev) return fail(BLAH2HIP_ERR_INVALID, "device index out of range");
  HIPCHK(hipSetDevice(device));
  auto *c = new blah2hip_ctx_s;
  c->device = device;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return fail(BLAH2HIP_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
  *out = c;
  return BLAH2HIP_OK;
}

int blah2hip_ctx_destroy(blah2hip_ctx_t c)
{
  if (!c) return BLAH2HIP_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  delete c;
  return BLAH2HIP_OK;
}

void *blah2hip_ctx_stream(blah2hip_ctx_t c) { return c ? (void *)c->stream : nullptr; }

int blah2hip_ctx_sync(blah2hip_ctx_t c)
{
  if (!c) return fail(BLAH2HIP_ERR_INVALID, "NULL context");
  HIPCHK(hipStreamSynchronize(c->stream));
  return BLAH2HIP_OK;
}

int blah2hip_ctx_malloc(blah2hip_ctx_t c, size_t bytes, void **dptr)
{
  if (!c || !dptr) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMalloc(dptr, bytes));
  return BLAH2HIP_OK;
}

int blah2hip_ctx_free(blah2hip_ctx_t c, void *dptr)
{
  if (!c) return fail(BLAH2HIP_ERR_INVALID, "NULL context");
  if (dptr) { HIPCHK(hipSetDevice(c->device)); HIPCHK(hipFree(dptr)); }
  return BLAH2HIP_OK;
}

int blah2hip_ctx_malloc_host(blah2hip_ctx_t c, size_t bytes, void **hptr)
{
  if (!c || !hptr) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
  return BLAH2HIP_OK;
}

int blah2hip_ctx_free_host(blah2hip_ctx_t c, void *hptr)
{
  if (!c) return fail(BLAH2HIP_ERR_INVALID, "NULL context");
  if (hptr) HIPCHK(hipHostFree(hptr));
  return BLAH2HIP_OK;
}

int blah2hip_ctx_h2d(blah2hip_ctx_t c, void *dptr, const void *hptr, size_t bytes)
{
  if (!c || !dptr || !hptr) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, c->stream));
  return BLAH2HIP_OK;
}

int blah2hip_ctx_d2d(blah2hip_ctx_t c, void *dst, const void *src, size_t bytes)
{
  if (!c || !dst || !src) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return BLAH2HIP_OK;
}

int blah2hip_ctx_d2h(blah2hip_ctx_t c, void *hptr, const void *dptr, size_t bytes)
{
  if (!c || !dptr || !hptr) return fail(BLAH2HIP_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, c->stream));
  return BLAH2HIP_OK;
}

// A streaming read of `bytes` of device memory and nothing else (16-byte loads, four in flight per lane, the sum kept
// behind a condition that never holds): what the memory system delivers to a kernel that only reads, the ceiling the
// range kernel's loads are priced against beside the copy (bench.py `roofline.read_ceiling`).
typedef float v4f_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_read_kernel(const v4f_t *__restrict__ src, size_t n16, float *sink)
{
  // a workgroup takes contiguous 64 KB chunks (16 loads of 16 bytes per lane, all requested before the first is used:
  // the shape tools/membench measured at 6.3 TB/s), chunk c + k * gridDim.x
  constexpr size_t CH = 16 * 256;
  float acc = 0.f;
  const size_t nch = n16 / CH;
  for (size_t c = blockIdx.x; c < nch; c += gridDim.x) {
    const v4f_t *p = src + c * CH + threadIdx.x;
    v4f_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(p + 256 * k);
#pragma unroll
    for (int k = 0; k < 16; k++) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  for (size_t i = nch * CH + (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const v4f_t a = src[i];
    acc += a.x + a.y + a.z + a.w;
  }
  if (acc == 1.2345678e-30f && sink) *sink = acc;
}

int blah2hip_stream_read_dev(const void *d_src, size_t bytes, void *d_sink, void *stream)
{
  if (!d_src || ((uintptr_t)d_src & 15)) return fail(BLAH2HIP_ERR_INVALID, "stream_read_dev: source NULL or not 16-byte aligned");
  int dev = 0, cus = 256;
  HIPCHK(hipGetDevice(&dev));
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  stream_read_kernel<<<dim3(cus * 6), dim3(256), 0, (hipStream_t)stream>>>((const v4f_t *)d_src, bytes / 16, (float *)d_sink);
  HIPCHK(hipGetLastError());
  return BLAH2HIP_OK;
}

int blah2hip_amb_result_ptrs(blah2hip_amb_t h, const void **d_map, const double **d_metrics)
{
  if (!h) return fail(BLAH2HIP_ERR_INVALID, "NULL handle");
  if (d_map) *d_map = h->d_map;
  if (d_metrics) *d_metrics = h->d_metrics;
  return BLAH2HIP_OK;
}

// ---------------------------------------------------------------- clutter --
// implemented in clutter.hip; it reports errors through this internal hook so
// that blah2hip_last_error() covers both translation units
void blah2hip_set_error_(const char *msg) { g_err = msg ? msg : ""; }

hipError_t blah2hip_ensure_lds_(const void *kern, int bytes)
{
  static std::mutex mu;
  static std::set<std::pair<int, const void *>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(mu);
  if (done.count({dev, kern})) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.insert({dev, kern});
  return e;
}

} // extern "C"
