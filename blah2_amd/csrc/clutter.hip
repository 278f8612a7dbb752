// WienerHopf clutter filter on gfx950 (include/blah2hip.h, clutter section).
//
// Reference: /root/reference/src/process/clutter/WienerHopf.cpp:58-163.  With
//   xs[i] = x[(i - delayMin) mod N]                       (:67, uint32 arithmetic)
//   r[k]  = sum_n xs[(n+k) mod N] conj(xs[n])             (:76-84,  k < nBins)
//   b[k]  = sum_n  y[(n+k) mod N] conj(xs[n])             (:100-108)
//   A[i][j] = r[i-j]  (Hermitian Toeplitz, :85-97),  A w = b   (:111-122)
//   y_out[n] = y[n] - sum_k w[k] xs[n-k],  xs[m<0] = 0    (:125-160)
// nBins = delayMax - delayMin (no +1, :12).
//
// The reference does this with 4 FFTs + 2 IFFTs of length N and 3 of length
// N+nBins+1, plus a dense Cholesky.  None of those long transforms is needed:
//   * r and b are nBins lags of a circular correlation -> the same segmented
//     on-chip FFT correlation as the range kernel (clutter_corr_kernel),
//     partial sums per workgroup, reduced in fp64;
//   * A is Hermitian Toeplitz -> Levinson recursion in fp64, O(nBins^2), with Schur
//     residual recursions in place of its inner products (clutter_solve_kernel: one
//     workgroup per CPI, one barrier per order).  The reference's chol() fails exactly
//     when A is not positive definite; the recursion detects the same condition
//     (a prediction-error factor 1-|e|^2 <= 0 or r[0] <= 0) -> ok = 0;
//   * the FIR is an overlap-save convolution on the on-chip FFT
//     (clutter_fir_kernel), one pass over x and y.
// Results are mathematically identical to the reference's (same linear
// system, same linear convolution); arithmetic is fp32 for the transforms and
// fp64 for the reduction and the solve.
// the correlation kernels run at 244-256 VGPRs: keep the butterflies as single-instruction statements
// (fft_wg.hpp; the grouped form spills 5-33 registers here and costs 3.6 %, measured)
#define B2_SPLIT_BUTTERFLIES 1
#include <hip/hip_runtime.h>

#include "blah2hip.h"
#include "fft_wg.hpp"
#include "range_core.hpp"
#include "bufload.hpp"
#include "timing.hpp"

#include <algorithm>
#include <stdint.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace blah2;

namespace {

struct dcx {
  double x, y;
};

// WienerHopf.cpp:67: xs[i] = x[(i - delayMin) % N] with (i - delayMin) evaluated in
// uint32.  Without a division (i < N, |delayMin| < N):
//   delayMin <= 0: i + |delayMin| < 2N                     -> one conditional subtract
//   delayMin > 0 : i >= delayMin -> i - delayMin < N
//                  i <  delayMin -> the uint32 difference wraps to 2^32 + i - delayMin,
//                                   whose residue is (i + wrapC) mod N, wrapC = (2^32 - delayMin) mod N
// thresh = max(delayMin, 0), sub = delayMin as uint32, wrapC precomputed on the host.
struct XsMap {
  uint32_t N, thresh, sub, wrapC;
};
__device__ __forceinline__ uint32_t xs_index(uint32_t i, const XsMap &m)
{
  uint32_t j = (i >= m.thresh) ? i - m.sub : i + m.wrapC;
  return j >= m.N ? j - m.N : j;
}
// n < 2N -> n mod N
__device__ __forceinline__ uint32_t wrapN(uint32_t n, uint32_t N) { return n >= N ? n - N : n; }

// XCD-aware walk over the segments of a CPI.  Neighbouring segments read overlapping windows (F samples
// for segLen = F - nBins + 1 new ones: 2x at nBins = F/2), and workgroups are dispatched round-robin
// over the 8 XCDs, each with its own L2: with the natural map (workgroup b takes segments b, b + G, ...)
// the two readers of a shared half-window sit on different XCDs and both fetch it from HBM (measured,
// profiles/r02_cfg3_full: 10.3 GB fetched for 5.1 GB of input).  Here every XCD owns one contiguous
// eighth of the segments, and in each round its G/8 workgroups take G/8 CONSECUTIVE segments, so the
// overlaps are served by that XCD's L2.  gridDim.x must be a multiple of 8.
struct SegWalk {
  int base, step, first, count; // segments base + first + k*step, while first + k*step < count
};
__device__ __forceinline__ SegWalk seg_walk(int nSeg)
{
  const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int s8 = (nSeg + 7) >> 3;
  SegWalk w;
  w.base = xcd * s8;
  w.step = gx;
  w.first = l;
  w.count = min(s8, nSeg - w.base); // may be <= 0 for the last XCDs of a short CPI
  return w;
}

// In: InC32 (two complex fp32 planes) or InI16 (the .rspduo words, range_core.hpp); in.lx / in.ly read sample i of
// the reference / surveillance channel
// One channel of a CPI as something indexable by sample: a plain `const cf *` for the fp32 planes (the kernels are then
// token for token what they were: the correlation kernels sit at the 256-register cap, and reaching the same loads through
// an accessor object made the register allocator spill 34 values in the inner loop), a converting view for the .rspduo words.
struct I16Chan {
  const int16_t *p; // first int16 of this channel's (I, Q) pair in sample 0
  __device__ __forceinline__ cf operator[](uint32_t i) const
  {
    const uint32_t w = *reinterpret_cast<const uint32_t *>(p + 4 * (size_t)i); // (I, Q) as one 4-byte load
    return cmake((float)(int16_t)(w & 0xffffu), (float)(int16_t)(w >> 16));
  }
};
template <class In> struct ChanOf;
template <> struct ChanOf<InC32> {
  using type = const cf *;
  static __device__ __forceinline__ type x(const void *px, const void *, int64_t i) { return (const cf *)px + i; }
  static __device__ __forceinline__ type y(const void *, const void *py, int64_t i) { return (const cf *)py + i; }
};
template <> struct ChanOf<InI16> {
  using type = I16Chan;
  static __device__ __forceinline__ type x(const void *px, const void *, int64_t i) { return I16Chan{(const int16_t *)px + 4 * i}; }
  static __device__ __forceinline__ type y(const void *px, const void *, int64_t i) { return I16Chan{(const int16_t *)px + 4 * i + 2}; }
};

// one channel of a CPI as the raw-buffer channel type of bufload.hpp (sample stride, raw word, conversion)
template <class In> struct BufChanOf;
template <> struct BufChanOf<InC32> {
  using X = ChanC32;
  using Y = ChanC32;
  static __device__ __forceinline__ const void *x(const void *px, const void *, int64_t i) { return (const cf *)px + i; }
  static __device__ __forceinline__ const void *y(const void *, const void *py, int64_t i) { return (const cf *)py + i; }
};
template <> struct BufChanOf<InI16> {
  using X = ChanI16;
  using Y = ChanI16;
  static __device__ __forceinline__ const void *x(const void *px, const void *, int64_t i) { return (const int16_t *)px + 4 * i; }
  static __device__ __forceinline__ const void *y(const void *px, const void *, int64_t i) { return (const int16_t *)px + 4 * i + 2; }
};
// Does the window [src0, src0 + F) of the shifted reference channel map to ONE contiguous run of x, and where does it
// start?  xs_index is i - sub from `thresh` on (then mod N): contiguous unless the window starts before `thresh` (or
// before sample 0) or runs over the end of the CPI, which only the first and last windows of a CPI do.  *cnt = the samples
// of the window that exist (src < N); the rest is zero padding.
__device__ __forceinline__ bool xs_window_plain(int src0, int F, const XsMap &m, uint32_t *j0, int *cnt)
{
  const int64_t left = (int64_t)m.N - src0;
  *cnt = (int)(left < F ? (left < 0 ? 0 : left) : F);
  *j0 = (uint32_t)src0 - m.sub;
  return src0 >= 0 && (uint32_t)src0 >= m.thresh && (uint64_t)*j0 + (uint32_t)*cnt <= m.N;
}

struct CorrArgs {
  const void *x, *y; // InC32: the two planes; InI16: x = the interleaved buffer, y unused
  int64_t cpiStride;
  uint32_t N;
  XsMap xs;
  int32_t nBins, segLen, nSeg, nJobs;
  const cf *tw;
  cf *partial; // [nCpi][2][nJobs][nBins]
  float scale;
};

// grid (nJobs, nCpi): per segment FFT(x' = zero-padded xs segment) ONCE, then the
// xs window (-> r) and the y window (-> b) against it; both partial correlations
// accumulate in registers across the workgroup's segments.
// (F = 4096: built for ONE workgroup of 256 threads per SIMD set -- 257 registers are what the three live windows, two
// accumulators and two twiddle sets need there, and at the two-workgroup cap of 256 one value went to scratch in every
// build of rounds 2-5.  That instantiation only runs for filters of 2050 ... 4081 taps: the half-window form takes the rest.)
template <int R3, class In> __global__ __launch_bounds__(16 * R3, R3 == 16 ? 1 : 2) void clutter_corr_kernel(CorrArgs a)
{
  using W = WgFft<R3>;
  constexpr int T = W::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *P = reinterpret_cast<cf *>(smem);
  cf *Q = P + W::A_ELEMS;
  const int t = threadIdx.x;
  const int cpi = blockIdx.y;
  const typename ChanOf<In>::type X = ChanOf<In>::x(a.x, a.y, (int64_t)cpi * a.cpiStride);
  const typename ChanOf<In>::type Y = ChanOf<In>::y(a.x, a.y, (int64_t)cpi * a.cpiStride);
  cf tw1[15], tw3[16];
  W::load_twiddles(t, a.tw, tw1, tw3);

  cf accR[16], accB[16];
#pragma unroll
  for (int e = 0; e < 16; e++) { accR[e] = cmake(0.f, 0.f); accB[e] = cmake(0.f, 0.f); }
  const SegWalk sw = seg_walk(a.nSeg);
  for (int r = sw.first; r < sw.count; r += sw.step) {
    const int g = sw.base + r;
    const uint32_t n0 = (uint32_t)g * (uint32_t)a.segLen;
    // all 32 loads of the segment first (the y window is consumed last) -- except at F = 4096, where the kernel sits at the
    // 256-register cap: there the y window is requested inside the second transform (its 32 registers are free until
    // then), which removed the one spilled register (8 bytes of scratch per lane) of rounds 2-5; this instantiation
    // only runs for filters of 2050 ... 4081 taps (the half-window form takes the others)
    cf v[16], wv[16], yw[16];
    uint32_t j0;
    int xcnt;
    const bool whole = (uint64_t)n0 + 16 * T <= a.N; // the window does not run around the end of the CPI
    const bool plain = whole && xs_window_plain((int)n0, 16 * T, a.xs, &j0, &xcnt); // all but a CPI's first and last windows: immediates only
    auto load_y = [&]() {
      if (plain) {
        using CY = typename BufChanOf<In>::Y;
        const __amdgpu_buffer_rsrc_t yd = make_rsrc_b(BufChanOf<In>::y(a.x, a.y, (int64_t)cpi * a.cpiStride + n0), 16 * T * CY::STRIDE);
#pragma unroll
        for (int k = 0; k < 16; k++) yw[k] = RawBuiltin<CY>::cvt(RawBuiltin<CY>::ld(yd, (t + T * k) * CY::STRIDE, 0));
      } else {
#pragma unroll
        for (int k = 0; k < 16; k++) yw[k] = Y[wrapN(n0 + (uint32_t)(t + T * k), a.N)]; // y window (mode b)
      }
    };
    if (plain) {
      using CX = typename BufChanOf<In>::X;
      const __amdgpu_buffer_rsrc_t xd = make_rsrc_b(BufChanOf<In>::x(a.x, a.y, (int64_t)cpi * a.cpiStride + j0), 16 * T * CX::STRIDE);
#pragma unroll
      for (int k = 0; k < 16; k++) wv[k] = RawBuiltin<CX>::cvt(RawBuiltin<CX>::ld(xd, (t + T * k) * CX::STRIDE, 0));
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++) wv[k] = X[xs_index(wrapN(n0 + (uint32_t)(t + T * k), a.N), a.xs)]; // xs window (mode r): circular index, n0 + m < N + F
    }
    if (R3 < 16) load_y();
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int m = t + T * k;
      v[k] = (m < a.segLen && n0 + (uint32_t)m < a.N) ? wv[k] : cmake(0.f, 0.f); // x' = the same samples, cut to the segment
    }
    W::fwd_s1(t, v, tw1, P);
    __syncthreads();
    W::fwd_s2(t, v, P, Q);
    __syncthreads();
    W::fwd_s3(t, v, tw3, Q); // v = X' spectrum
    W::fwd_s1(t, wv, tw1, P);
    __syncthreads();
    W::fwd_s2(t, wv, P, Q);
    if (R3 == 16) load_y(); // behind the second transform's last exchange write: wv's 32 registers are about to be consumed
    __syncthreads();
    W::fwd_s3(t, wv, tw3, Q);
#pragma unroll
    for (int e = 0; e < 16; e++) accR[e] = cmacc(accR[e], wv[e], v[e]);
    W::fwd_s1(t, yw, tw1, P);
    __syncthreads();
    W::fwd_s2(t, yw, P, Q);
    __syncthreads();
    W::fwd_s3(t, yw, tw3, Q);
#pragma unroll
    for (int e = 0; e < 16; e++) accB[e] = cmacc(accB[e], yw[e], v[e]);
    __syncthreads();
  }
  // two explicit calls (a runtime-selected register array would be demoted to scratch)
  auto finish = [&](cf *acc, int mode) {
    W::inv_s1(t, acc, tw3, P);
    __syncthreads();
    W::inv_s2(t, acc, P, Q);
    __syncthreads();
    W::inv_s3(t, acc, tw1, Q);
    cf *dst = a.partial + (((size_t)cpi * 2 + mode) * a.nJobs + blockIdx.x) * a.nBins;
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const int k = t + T * c;
      if (k < a.nBins) dst[k] = cmake(acc[c].x * a.scale, acc[c].y * a.scale);
    }
    __syncthreads();
  };
  finish(accR, 0);
  finish(accB, 1);
}

// Half-window form of the same correlations, for filters with many taps (nBins - 1 <= F/2): cut the
// CPI into segments of exactly L = F/2 samples; with X_g, Y_g the transforms of the zero-padded
// segments, the window [segment g, segment g+1] has the spectrum X_g + (-1)^m X_{g+1} (a shift by F/2
// samples is the sign of the frequency index), so -- grouping every pair (n, n+k) by the segment that
// holds its LATER sample --
//     r = sum_g X_g conj(X_g + (-1)^m X_{g-1}),      b = sum_g Y_g conj(X_g + (-1)^m X_{g-1}):
// TWO transforms per L samples instead of three per F - nBins + 1, every sample read once, and only
// the previous segment's X to keep (the register budget of the windowed form).  A workgroup walks a
// contiguous run of segments (one extra transform for the segment in front of its run).  The
// sequence is treated as zero-extended to a whole number of segments; what that leaves out of the
// CIRCULAR correlations over N -- the pairs whose later sample wraps to the head of the CPI -- is one
// more product, tail (last nBins-1 samples of xs) against head (first nBins-1 of xs / y) shifted by
// tau = nBins - 1, i.e. times W_F^(tau m), done by the last workgroup.  Mathematically identical.
struct CorrHalfArgs {
  const void *x, *y;
  int64_t cpiStride;
  uint32_t N;
  XsMap xs;
  int32_t nBins, nSeg, per, nJobs; // nSeg = ceil(N / (F/2)), per = segments per workgroup
  const cf *tw;
  cf *partial; // [nCpi][2][nJobs][nBins]
  float scale;
};

template <int R3, class In> __global__ __launch_bounds__(16 * R3, 2) void clutter_corr_half_kernel(CorrHalfArgs a)
{
  using W = WgFft<R3>;
  constexpr int T = W::T, L = W::F / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *P = reinterpret_cast<cf *>(smem);
  cf *Q = P + W::A_ELEMS;
  const int t = threadIdx.x;
  const int cpi = blockIdx.y;
  const typename ChanOf<In>::type X = ChanOf<In>::x(a.x, a.y, (int64_t)cpi * a.cpiStride);
  const typename ChanOf<In>::type Y = ChanOf<In>::y(a.x, a.y, (int64_t)cpi * a.cpiStride);
  cf tw1[15], tw3[16];
  W::load_twiddles(t, a.tw, tw1, tw3);
  // (-1)^m for this thread's spectrum registers: m = q + 16 r + 256 s with q = (t + T j) / 16 and T a
  // multiple of 32, so the parity of m is that of t / 16
  const float sgn = ((t >> 4) & 1) ? -1.f : 1.f;

  auto fwd = [&](cf *v) {
    W::fwd_s1(t, v, tw1, P);
    __syncthreads();
    W::fwd_s2(t, v, P, Q);
    __syncthreads();
    W::fwd_s3(t, v, tw3, Q);
    __syncthreads();
  };
  // first L samples of a zero-padded window starting at sample n0 of xs / of y (zero beyond N)
  // Through buffer descriptors over exactly the `len` samples: the range check is the zero padding.  xs is read that
  // way where the stretch maps to one contiguous run of x (xs_window_plain: everywhere but at a CPI's ends).
  using CX = typename BufChanOf<In>::X;
  using CY = typename BufChanOf<In>::Y;
  auto load_xs = [&](uint32_t n0, uint32_t len, cf *v) {
    uint32_t j0;
    int cnt;
    if (xs_window_plain((int)n0, (int)len, a.xs, &j0, &cnt) && cnt == (int)len) {
      const __amdgpu_buffer_rsrc_t d = make_rsrc_b(BufChanOf<In>::x(a.x, a.y, (int64_t)cpi * a.cpiStride + j0), (int)len * CX::STRIDE);
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = RawBuiltin<CX>::cvt(RawBuiltin<CX>::ld(d, (t + T * k) * CX::STRIDE, 0));
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t m = (uint32_t)(t + T * k);
        const bool inr = m < len;
        const cf s = X[xs_index(inr ? n0 + m : 0u, a.xs)];
        v[k] = inr ? s : cmake(0.f, 0.f);
      }
    }
#pragma unroll
    for (int k = 8; k < 16; k++) v[k] = cmake(0.f, 0.f);
  };
  auto load_y = [&](uint32_t n0, uint32_t len, cf *v) {
    const __amdgpu_buffer_rsrc_t d = make_rsrc_b(BufChanOf<In>::y(a.x, a.y, (int64_t)cpi * a.cpiStride + n0), (int)len * CY::STRIDE);
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = RawBuiltin<CY>::cvt(RawBuiltin<CY>::ld(d, (t + T * k) * CY::STRIDE, 0));
#pragma unroll
    for (int k = 8; k < 16; k++) v[k] = cmake(0.f, 0.f);
  };
  auto seg_len = [&](int g) { return min((uint32_t)L, a.N - (uint32_t)g * (uint32_t)L); };

  cf accR[16], accB[16], prevX[16];
#pragma unroll
  for (int e = 0; e < 16; e++) { accR[e] = cmake(0.f, 0.f); accB[e] = cmake(0.f, 0.f); prevX[e] = cmake(0.f, 0.f); }
  const int g0 = blockIdx.x * a.per, g1 = min(g0 + a.per, a.nSeg);
  if (g0 > 0 && g0 < g1) { // the segment in front of the run (a full one)
    load_xs((uint32_t)(g0 - 1) * (uint32_t)L, (uint32_t)L, prevX);
    fwd(prevX);
  }
  for (int g = g0; g < g1; g++) {
    const uint32_t n0 = (uint32_t)g * (uint32_t)L, len = seg_len(g);
    cf v[16], yv[16];
    load_xs(n0, len, v);
    load_y(n0, len, yv);
    fwd(v);
    fwd(yv);
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const cf z = cmake(v[e].x + sgn * prevX[e].x, v[e].y + sgn * prevX[e].y); // X_g + (-1)^m X_{g-1}
      accR[e] = cmacc(accR[e], v[e], z);
      accB[e] = cmacc(accB[e], yv[e], z);
      prevX[e] = v[e];
    }
  }
  if (blockIdx.x == (unsigned)a.nJobs - 1 && a.nBins > 1) { // the wrap-around pairs
    const uint32_t tau = (uint32_t)a.nBins - 1;
    cf v[16], yv[16];
    load_xs(a.N - tau, tau, prevX); // tail of xs
    load_xs(0u, tau, v);            // head of xs
    load_y(0u, tau, yv);            // head of y
    fwd(prevX);
    fwd(v);
    fwd(yv);
#pragma unroll
    for (int j = 0; j < W::NP; j++)
#pragma unroll
      for (int sidx = 0; sidx < R3; sidx++) {
        const int e = j * R3 + sidx;
        const int p16 = t + T * j;                         // 16 q + r
        const int m = (p16 >> 4) + 16 * (p16 & 15) + 256 * sidx;
        const cf ph = a.tw[((uint32_t)m * tau) & (W::F - 1)]; // W_F^(tau m): the head sits tau samples behind the tail
        const cf z = cmulc(prevX[e], ph);                  // conj(phase) * X_tail
        accR[e] = cmacc(accR[e], v[e], z);
        accB[e] = cmacc(accB[e], yv[e], z);
      }
  }
  auto finish = [&](cf *acc, int mode) {
    W::inv_s1(t, acc, tw3, P);
    __syncthreads();
    W::inv_s2(t, acc, P, Q);
    __syncthreads();
    W::inv_s3(t, acc, tw1, Q);
    cf *dst = a.partial + (((size_t)cpi * 2 + mode) * a.nJobs + blockIdx.x) * a.nBins;
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const int k = t + T * c;
      if (k < a.nBins) dst[k] = cmake(acc[c].x * a.scale, acc[c].y * a.scale);
    }
    __syncthreads();
  };
  finish(accR, 0);
  finish(accB, 1);
}

// ---- reduction of the partials (fp64), then the Toeplitz solve -----------------
struct SolveArgs {
  const cf *partial; // [nCpi][2][nJobs][nBins]
  dcx *rb;           // [nCpi][2][nBins]: r then b, fp64
  cf *w;             // [nCpi][nBins]
  int32_t *ok;       // [nCpi]
  int32_t nBins, nJobs;
  uint32_t *epoch;   // launch epoch of the look-ahead solve's mailboxes (solve_la.hpp), bumped here
  // clutter_solve_kernel as the launch BEHIND the look-ahead solve: a workgroup runs only if its CPI's fault word carries
  // this launch's epoch (a bounded wait of the look-ahead form ran out), and counts itself in *retries
  const unsigned long long *gate = nullptr; // the CPI's fault word: gate[cpi * gateStride]
  int64_t gateStride = 0;
  uint32_t *retries = nullptr;
};

// grid (ceil(nBins/16), 2, nCpi) x 1024: fixed-order fp64 sum over the jobs.  A block takes 16 bins (128-byte rows) and cuts the
// jobs into 64 interleaved slices (thread = bin + 16 slice): a lone CPI has up to 512 partials per bin, which one thread per bin
// summed as 64 dependent rounds of L2 latency (27.8 us of the 171 us a lone CPI's full chain took at configs[1], round 4);
// the slices' sums meet in LDS and are added in slice order -- the same order every run.
constexpr int RED_SLICES = 64; // 1024 threads: a lone CPI's 512 partials per bin are 4 dependent rounds of 2 loads per thread (16 slices: 16 rounds, 9.8 us)
__global__ __launch_bounds__(16 * RED_SLICES) void clutter_reduce_kernel(SolveArgs a)
{
  __shared__ double sx[RED_SLICES][16], sy[RED_SLICES][16];
  const int bin = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + bin, mode = blockIdx.y, cpi = blockIdx.z;
  if (a.epoch && blockIdx.x == 0 && threadIdx.x == 0 && mode == 0 && cpi == 0) { // one thread per launch; the solve kernel behind the boundary reads it
    const uint32_t e = *a.epoch + 1u;
    *a.epoch = e ? e : 1u;
  }
  double ax = 0.0, ay = 0.0;
  if (k < a.nBins) {
    const cf *p = a.partial + ((size_t)cpi * 2 + mode) * a.nJobs * a.nBins + k;
    // two interleaved accumulators: the loads of a pair of rounds are independent
    double bx = 0.0, by = 0.0;
    int j = slice;
    for (; j + RED_SLICES < a.nJobs; j += 2 * RED_SLICES) {
      const cf u = p[(size_t)j * a.nBins], v = p[(size_t)(j + RED_SLICES) * a.nBins];
      ax += (double)u.x; ay += (double)u.y;
      bx += (double)v.x; by += (double)v.y;
    }
    if (j < a.nJobs) { const cf u = p[(size_t)j * a.nBins]; ax += (double)u.x; ay += (double)u.y; }
    ax += bx; ay += by;
  }
  sx[slice][bin] = ax;
  sy[slice][bin] = ay;
  __syncthreads();
  if (slice == 0 && k < a.nBins) {
    double tx = sx[0][bin], ty = sy[0][bin];
#pragma unroll 8
    for (int q = 1; q < RED_SLICES; q++) { tx += sx[q][bin]; ty += sy[q][bin]; } // in slice order: the same sum every run
    a.rb[((size_t)cpi * 2 + mode) * a.nBins + k] = {tx, ty};
  }
}

// Hermitian Toeplitz solve A w = b, A[i][j] = r[i-j], by the Levinson recursion with its inner
// products replaced by Schur-type residual recursions, so that one order is a purely ELEMENT-WISE
// update of length n with a few scalars broadcast -- no reduction, one workgroup barrier per order.
//
// Levinson (the round-1 kernel): forward vector f (T_m f = e_1), backward vector conj(rev f), solution
// x; per order m it needs ef = sum_i r[m-i] f[i] and ex = sum_i r[m-i] x[i]: two dot products, i.e. a
// wave reduction, an LDS exchange of partials and a second barrier per order (3.1 ms for 2047 orders).
// Apply T to the (zero-extended) vectors instead and carry the results along:
//     a_m[j] = (T f_m)[j],  c_m[j] = (T conj(rev f_m))[j],  g_m[j] = b[j] - (T x_m)[j]        (j >= m)
// then ef = a_m[m] and d = b[m] - ex = g_m[m] are simply the LEADING ELEMENTS, and with D = 1 - |ef|^2
//     a_{m+1}[j] = (a_m[j]   - ef       c_m[j-1]) / D        f_{m+1}[i]        = (f_m[i]         - ef       conj f_m[m-i]) / D
//     c_{m+1}[j] = (c_m[j-1] - conj(ef) a_m[j]  ) / D        conj f_{m+1}[m-i] = (conj f_m[m-i] - conj(ef) f_m[i]      ) / D
//     g_{m+1}[j] = g_m[j] - d c_{m+1}[j]                     x_{m+1}[i]        = x_m[i] + d conj f_{m+1}[m-i]
// Both columns are carried UNNORMALISED -- A = s a, C = s c, F = s f with s_{m+1} = s_m D, the
// prediction-error power relative to r[0] -- which removes every per-element division and makes the two
// columns literally the same update,
//     (u, v) -> (u - ef v, v - conj(ef) u),      acc += dt v',      ef = A_m[m] / s_m,  dt = d / s_{m+1},
// on different operands: index j > m carries (A[j], C[j-1], -g[j]), index j <= m carries
// (F[j], conj F[m-j], x[j]).  An index changes role once, at m = j, where A[m] and g[m] have been
// consumed as the scalars and F[m] = x[m] = 0 start.  Thread t owns the indices t + NT k (u and acc in
// registers); v comes from a neighbour (C[j-1]) or the mirror index (F[m-j]) through an LDS array that
// holds C[j] for j > m and F[j] for j <= m, double-buffered so that one barrier per order suffices.
// The critical path of an order is short: the owner of index m+1 forms ef' = A'[m+1] / s_{m+1} as soon
// as its own update is done -- 1/s_{m+1} = (1/s_m)(1/D) needs only THIS order's ef, so every thread
// computes it (one reciprocal) while the LDS operands are in flight -- and publishes (ef', 1/s_{m+1},
// d') for the next order.
// The matrix is positive definite iff r[0] > 0 and every D > 0 -- the condition under which the
// reference's chol() succeeds (WienerHopf.cpp:111) -- else ok = 0.  fp64 throughout.
__device__ __forceinline__ dcx dsub_mul(dcx u, dcx e, dcx v) // u - e*v, two dependent fmas per component
{
  return {__builtin_fma(e.y, v.y, __builtin_fma(-e.x, v.x, u.x)), __builtin_fma(-e.y, v.x, __builtin_fma(-e.x, v.y, u.y))};
}
__device__ __forceinline__ dcx dadd_mul(dcx u, dcx e, dcx v) // u + e*v
{
  return {__builtin_fma(-e.y, v.y, __builtin_fma(e.x, v.x, u.x)), __builtin_fma(e.y, v.x, __builtin_fma(e.x, v.y, u.y))};
}
// 1/d for d in (0, 1]: hardware estimate + two Newton steps (5 dependent instructions instead of the
// ~10 of the IEEE-exact division sequence; this reciprocal sits on the critical path of every order)
__device__ __forceinline__ double fast_rcp(double d)
{
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, e, r);
}

// scalars of one order, published through LDS by the owner of index m at the end of order m-1
struct SolveScal {
  dcx ef;    // reflection coefficient a_m[m] = A_m[m] / s_m
  dcx d;     // g_m[m]
  double rs; // 1 / s_m
  double pad;
};

template <int K>
__global__ __launch_bounds__(1024) void clutter_solve_kernel(SolveArgs a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = a.nBins;
  // two buffers of n + 1 entries (entry n stays zero: "F[m]" as seen by index 0), then the scalars
  dcx *cur = reinterpret_cast<dcx *>(smem);
  dcx *nxt = cur + (n + 1);
  SolveScal *scal = reinterpret_cast<SolveScal *>(nxt + (n + 1)); // [parity]
  const int cpi = blockIdx.x;
  const int t = threadIdx.x, NT = blockDim.x;
  if (a.gate) { // behind the look-ahead solve: only the CPIs it gave up on
    const unsigned long long g = __hip_atomic_load(a.gate + (size_t)cpi * a.gateStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)g != *a.epoch) return;
    if (t == 0) atomicAdd(a.retries, 1u);
  }
  const dcx *rg = a.rb + (size_t)cpi * 2 * n;
  const double r0 = rg[0].x;
  bool ok = (r0 > 0.0) && isfinite(r0);
  const double inv0 = ok ? 1.0 / r0 : 0.0;
  const dcx x0 = {rg[n].x * inv0, rg[n].y * inv0}; // x_1[0] = b[0] / r[0]
  dcx U[K], Acc[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int j = t + NT * k;
    U[k] = {0.0, 0.0};
    Acc[k] = {0.0, 0.0};
    if (j < n) {
      const dcx rj = rg[j], bj = rg[n + j];
      if (j == 0) {
        U[k] = {inv0, 0.0}; // F_1[0] = f_1[0] (s_1 = 1)
        Acc[k] = x0;
      } else {
        U[k] = {rj.x * inv0, rj.y * inv0};  // A_1[j] = r[j] / r[0]  (= C_1[j])
        const dcx g = dsub_mul(bj, rj, x0); // g_1[j] = b[j] - r[j] x_1[0]
        Acc[k] = {-g.x, -g.y};
      }
      cur[j] = U[k];
      if (j == 1) {
        scal[1] = SolveScal{U[k], {-Acc[k].x, -Acc[k].y}, 1.0, 0.0};
        U[k] = {0.0, 0.0};   // becomes F[1] = 0, x[1] = 0 at order 1
        Acc[k] = {0.0, 0.0};
      }
    }
  }
  if (t == 0) { cur[n] = {0.0, 0.0}; nxt[n] = {0.0, 0.0}; }
  __syncthreads();
  for (int m = 1; m < n && ok; m++) {
    const SolveScal q = scal[m & 1];
    // operands first: they are in flight while the scalars below are formed
    dcx v[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int j = t + NT * k;
      const int src = (j > m) ? j - 1 : (j == 0 ? n : m - j); // C[j-1], or F[m-j] (F[m] = 0 lives in entry n)
      const dcx s = cur[min(src, n)];
      v[k] = {s.x, (j > m) ? s.y : -s.y};
    }
    const double D = __builtin_fma(-q.ef.y, q.ef.y, __builtin_fma(-q.ef.x, q.ef.x, 1.0));
    if (!(D > 0.0) || !isfinite(D)) { ok = false; break; } // uniform: every thread reads the same scalars
    const double rsn = q.rs * fast_rcp(D); // 1 / s_{m+1}
    const dcx dt = {q.d.x * rsn, q.d.y * rsn};
    const dcx efc = {q.ef.x, -q.ef.y};
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int j = t + NT * k;
      if (j < n) {
        const dcx un = dsub_mul(U[k], q.ef, v[k]);
        const dcx vn = dsub_mul(v[k], efc, U[k]);
        const dcx an = dadd_mul(Acc[k], dt, vn);
        const bool hi = j > m;
        nxt[j] = {hi ? vn.x : un.x, hi ? vn.y : un.y};
        const bool owner = (j == m + 1); // this index turns into F[m+1] = 0, x[m+1] = 0 for the next order
        if (owner) scal[(m + 1) & 1] = SolveScal{{un.x * rsn, un.y * rsn}, {-an.x, -an.y}, rsn, 0.0};
        U[k] = {owner ? 0.0 : un.x, owner ? 0.0 : un.y};
        Acc[k] = {owner ? 0.0 : an.x, owner ? 0.0 : an.y};
      }
    }
    __syncthreads();
    dcx *tmp = cur; cur = nxt; nxt = tmp;
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int j = t + NT * k;
    if (j < n) a.w[(size_t)cpi * n + j] = ok ? cmake((float)Acc[k].x, (float)Acc[k].y) : cmake(0.f, 0.f);
  }
  if (t == 0) a.ok[cpi] = ok ? 1 : 0;
}

// The same recursion for ANY n (the long filters, blah2hip_clutter_create): every vector in global memory -- per CPI cur / nxt
// (n + 1 entries each, shared between the threads: written and read through L2, sc1, with the workgroup barrier between), U and
// Acc (n each, touched by their owner only) -- and a loop over the indices a thread owns instead of register arrays.  A few
// microseconds an order: built for coverage (the reference runs a dense nBins x nBins Cholesky there), not for speed.
__device__ __forceinline__ dcx ld_l2(const dcx *p)
{
  return {__hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
}
__device__ __forceinline__ void st_l2(dcx *p, dcx v)
{
  __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(1024) void clutter_solve_big_kernel(SolveArgs a, dcx *ws)
{
  __shared__ SolveScal scal[2];
  const int n = a.nBins, cpi = blockIdx.x, t = threadIdx.x, NT = blockDim.x;
  dcx *cur = ws + (size_t)cpi * (4 * (size_t)n + 2);
  dcx *nxt = cur + (n + 1), *U = nxt + (n + 1), *Acc = U + n;
  const dcx *rg = a.rb + (size_t)cpi * 2 * n;
  const double r0 = rg[0].x;
  bool ok = (r0 > 0.0) && isfinite(r0);
  const double inv0 = ok ? 1.0 / r0 : 0.0;
  const dcx x0 = {rg[n].x * inv0, rg[n].y * inv0};
  for (int j = t; j < n; j += NT) { // the first order, as clutter_solve_kernel sets it up
    const dcx rj = rg[j], bj = rg[n + j];
    dcx u, acc;
    if (j == 0) {
      u = {inv0, 0.0};
      acc = x0;
    } else {
      u = {rj.x * inv0, rj.y * inv0};
      const dcx g = dsub_mul(bj, rj, x0);
      acc = {-g.x, -g.y};
    }
    st_l2(cur + j, u);
    if (j == 1) {
      scal[1] = SolveScal{u, {-acc.x, -acc.y}, 1.0, 0.0};
      u = {0.0, 0.0};
      acc = {0.0, 0.0};
    }
    U[j] = u;
    Acc[j] = acc;
  }
  if (t == 0) { st_l2(cur + n, {0.0, 0.0}); st_l2(nxt + n, {0.0, 0.0}); }
  __syncthreads();
  for (int m = 1; m < n && ok; m++) {
    const SolveScal q = scal[m & 1];
    const double D = __builtin_fma(-q.ef.y, q.ef.y, __builtin_fma(-q.ef.x, q.ef.x, 1.0));
    if (!(D > 0.0) || !isfinite(D)) { ok = false; break; } // uniform: every thread reads the same scalars
    const double rsn = q.rs * fast_rcp(D);
    const dcx dt = {q.d.x * rsn, q.d.y * rsn};
    const dcx efc = {q.ef.x, -q.ef.y};
    for (int j = t; j < n; j += NT) {
      const int src = (j > m) ? j - 1 : (j == 0 ? n : m - j);
      const dcx s = ld_l2(cur + min(src, n));
      const dcx v = {s.x, (j > m) ? s.y : -s.y};
      const dcx u = U[j], acc = Acc[j];
      const dcx un = dsub_mul(u, q.ef, v);
      const dcx vn = dsub_mul(v, efc, u);
      const dcx an = dadd_mul(acc, dt, vn);
      const bool hi = j > m;
      st_l2(nxt + j, {hi ? vn.x : un.x, hi ? vn.y : un.y});
      const bool owner = (j == m + 1);
      if (owner) scal[(m + 1) & 1] = SolveScal{{un.x * rsn, un.y * rsn}, {-an.x, -an.y}, rsn, 0.0};
      U[j] = {owner ? 0.0 : un.x, owner ? 0.0 : un.y};
      Acc[j] = {owner ? 0.0 : an.x, owner ? 0.0 : an.y};
    }
    __syncthreads();
    dcx *tmp = cur; cur = nxt; nxt = tmp;
  }
  for (int j = t; j < n; j += NT) {
    const dcx acc = Acc[j];
    a.w[(size_t)cpi * n + j] = ok ? cmake((float)acc.x, (float)acc.y) : cmake(0.f, 0.f);
  }
  if (t == 0) a.ok[cpi] = ok ? 1 : 0;
}

#include "solve_la.hpp"

// ---- overlap-save FIR: y_out = y - (w * xs)[0..N) ----------------------------
struct FirArgs {
  const void *x, *y;
  cf *yout;
  int64_t cpiStride, outStride;
  uint32_t N;
  XsMap xs;
  int32_t nBins, segLen, nSeg;
  const cf *w;       // [nCpi][nBins]
  const int32_t *ok; // [nCpi]
  const cf *tw;
  float scale;
  int32_t carry;     // segLen = F/2: the upper half of a block's x window is the lower half of the next block's (see the kernel)
};

template <int R3, class In> __global__ __launch_bounds__(16 * R3, 2) void clutter_fir_kernel(FirArgs a)
{
  using W = WgFft<R3>;
  constexpr int T = W::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *P = reinterpret_cast<cf *>(smem);
  cf *Q = P + W::A_ELEMS;
  const int t = threadIdx.x;
  const int cpi = blockIdx.y;
  const typename ChanOf<In>::type X = ChanOf<In>::x(a.x, a.y, (int64_t)cpi * a.cpiStride);
  const typename ChanOf<In>::type Y = ChanOf<In>::y(a.x, a.y, (int64_t)cpi * a.cpiStride);
  cf *O = a.yout + (int64_t)cpi * a.outStride;
  const bool ok = a.ok[cpi] != 0;
  cf tw1[15], tw3[16];
  W::load_twiddles(t, a.tw, tw1, tw3);

  // spectrum of the taps (zero-padded to F), pre-scaled by 1/F, kept in registers -- all but the LARGEST tap, which is applied
  // in the time domain (below): with a direct path 58 dB above the noise, w * xs is a thousand times what is left of y, and
  // the transform's rounding of that one product (3.5 eps of it, incoherent) was the map's largest error behind a deep
  // cancellation (fixture deep_cancel: 0.0063 dB on a cell 19 dB under the mean level); one fma rounds it once
  cf ws[16];
  float bestMag = -1.f;
  int bestIdx = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int m = t + T * k;
    const cf wv = a.w[(size_t)cpi * a.nBins + min(m, a.nBins - 1)];
    const float keep = (m < a.nBins) ? a.scale : 0.f; // branch-free: the 16 loads go out together
    ws[k] = cmake(wv.x * keep, wv.y * keep);
    const float mag = (m < a.nBins) ? wv.x * wv.x + wv.y * wv.y : -1.f;
    if (mag > bestMag) { bestMag = mag; bestIdx = m; } // (ascending m: ties go to the lower index)
  }
  { // the workgroup's argmax: a butterfly over the lanes, then the waves' winners through the exchange buffer (free here); ties go
    // to the lower index, so every workgroup of the CPI picks the same tap
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float mu = __shfl_xor(bestMag, off);
      const int iu = __shfl_xor(bestIdx, off);
      if (mu > bestMag || (mu == bestMag && iu < bestIdx)) { bestMag = mu; bestIdx = iu; }
    }
    if (T > 64) {
      float *sm = reinterpret_cast<float *>(P);
      int *si = reinterpret_cast<int *>(P) + T / 64;
      if ((t & 63) == 0) { sm[t >> 6] = bestMag; si[t >> 6] = bestIdx; }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < T / 64; u++) {
        const float mu = sm[u];
        const int iu = si[u];
        if (mu > bestMag || (mu == bestMag && iu < bestIdx)) { bestMag = mu; bestIdx = iu; }
      }
      __syncthreads();
    }
  }
  const int k0 = bestIdx;
  const cf w0 = a.w[(size_t)cpi * a.nBins + k0];
#pragma unroll
  for (int k = 0; k < 16; k++)
    if (t + T * k == k0) ws[k] = cmake(0.f, 0.f);
  W::fwd_s1(t, ws, tw1, P);
  __syncthreads();
  W::fwd_s2(t, ws, P, Q);
  __syncthreads();
  W::fwd_s3(t, ws, tw3, Q);
  __syncthreads();

  const int hist = a.nBins - 1; // samples of history in front of each block
  const int N = (int)a.N;       // < 2^31 - 8192 (checked at create): 32-bit sample indices throughout
  // CARRY (a.carry, filters with nBins - 1 close below F/2, i.e. cfg 3's 2047 taps on F = 4096): the planner makes the blocks
  // exactly F/2 = 8 T samples long, a workgroup walks CONSECUTIVE blocks, and the upper half of a block's window
  // [n0 - hist, n0 - hist + F) is the lower half of the next block's, register for register (x'[t + T (k + 8)] -> x'[t + T k]):
  // eight loads per thread and block instead of sixteen, and x is read once.  (With blocks of F - hist samples every
  // sample of x was requested twice, and 43 % of the second requests went to HBM: 1.145 x the algorithmic bytes.)  It is a traffic
  // measure, not a speed-up: the kernel's time did not move (58.0 against 57.7 us/CPI at cfg 3) -- what bounds it is the
  // barrier-separated phases of two workgroups per CU, not the loads.  Otherwise: the XCD-aware strided walk, whole windows.
  const SegWalk sw = seg_walk(a.nSeg);
  int rFirst = sw.first, rStep = sw.step, rEnd = sw.count;
  if (a.carry) { // this workgroup's contiguous run inside its XCD's eighth
    const int per = (max(sw.count, 0) + sw.step - 1) / sw.step;
    rFirst = sw.first * per;
    rEnd = min(rFirst + per, sw.count);
    rStep = 1;
  }
  cf keep[8];
  bool have = false;
  // eight values v[k0 .. k0 + 7] of the window that starts at sample src0 of xs: x'[src0 + t + T k], zero outside [0, N)
  auto load_half = [&](cf *dst, int src0) {
    uint32_t j0;
    int xcnt;
    if (xs_window_plain(src0, 8 * T, a.xs, &j0, &xcnt)) {
      using CX = typename BufChanOf<In>::X;
      const __amdgpu_buffer_rsrc_t xd = make_rsrc_b(BufChanOf<In>::x(a.x, a.y, (int64_t)cpi * a.cpiStride + j0), xcnt * CX::STRIDE);
#pragma unroll
      for (int k = 0; k < 8; k++) dst[k] = RawBuiltin<CX>::cvt(RawBuiltin<CX>::ld(xd, (t + T * k) * CX::STRIDE, 0));
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int src = src0 + t + T * k;
        const bool inr = src >= 0 && src < N;
        const cf xv = X[xs_index(inr ? (uint32_t)src : 0u, a.xs)];
        dst[k] = inr ? xv : cmake(0.f, 0.f);
      }
    }
  };
  for (int r = rFirst; r < rEnd; r += rStep) {
    const int g = sw.base + r;
    const int n0 = g * a.segLen;
    cf v[16], yv[16];
    uint32_t j0;
    int xcnt;
    if (a.carry) {
      if (have) {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = keep[k];
      } else {
        load_half(v, n0 - hist);
      }
      load_half(v + 8, n0 - hist + 8 * T);
#pragma unroll
      for (int k = 0; k < 8; k++) keep[k] = v[8 + k];
      have = true;
      // (requesting the NEXT block's new half here, into eight more registers, so that it lands during this block's two
      // transforms, measured 60.8 against 58.0 us/CPI at cfg 3: not adopted)
    } else if (xs_window_plain(n0 - hist, 16 * T, a.xs, &j0, &xcnt)) { // all but the first and last windows of a CPI
      using CX = typename BufChanOf<In>::X;
      const __amdgpu_buffer_rsrc_t xd = make_rsrc_b(BufChanOf<In>::x(a.x, a.y, (int64_t)cpi * a.cpiStride + j0), xcnt * CX::STRIDE);
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = RawBuiltin<CX>::cvt(RawBuiltin<CX>::ld(xd, (t + T * k) * CX::STRIDE, 0));
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int m = t + T * k;
        const int src = n0 - hist + m;
        const bool inr = src >= 0 && src < N;
        const cf xv = X[xs_index(inr ? (uint32_t)src : 0u, a.xs)];
        v[k] = inr ? xv : cmake(0.f, 0.f);
      }
    }
    // y of the output samples, and later the stores: descriptors over exactly this block's new samples [n0, n0 + min(segLen,
    // N - n0)), register k at offset m - hist -- history rows (negative) and rows beyond the block are outside the range, so
    // the loads return zeros that are never stored and the stores are dropped: no predicate, no branch per element
    const int cnt = min(a.segLen, N - n0);
    using CY = typename BufChanOf<In>::Y;
    const __amdgpu_buffer_rsrc_t yd = make_rsrc_b(BufChanOf<In>::y(a.x, a.y, (int64_t)cpi * a.cpiStride + n0), cnt * CY::STRIDE);
    const __amdgpu_buffer_rsrc_t od = make_rsrc_b(O + n0, cnt * 8);
#pragma unroll
    for (int k = 0; k < 16; k++) yv[k] = RawBuiltin<CY>::cvt(RawBuiltin<CY>::ld(yd, (t + T * k - hist) * CY::STRIDE, 0));
    cf xdir[16]; // the window k0 samples earlier: register c holds xs[n - k0] of the output sample n that register c writes
    if (ok) {
      W::fwd_s1(t, v, tw1, P);
      __syncthreads();
      W::fwd_s2(t, v, P, Q);
      __syncthreads();
      W::fwd_s3(t, v, tw3, Q);
#pragma unroll
      for (int e = 0; e < 16; e++) v[e] = cmul(v[e], ws[e]);
      // (requested here: the samples arrive during the inverse transform; most of their lines are in L2 from the window above)
      if (xs_window_plain(n0 - hist - k0, 16 * T, a.xs, &j0, &xcnt)) {
        using CX = typename BufChanOf<In>::X;
        const __amdgpu_buffer_rsrc_t xd = make_rsrc_b(BufChanOf<In>::x(a.x, a.y, (int64_t)cpi * a.cpiStride + j0), xcnt * CX::STRIDE);
#pragma unroll
        for (int k = 0; k < 16; k++) xdir[k] = RawBuiltin<CX>::cvt(RawBuiltin<CX>::ld(xd, (t + T * k) * CX::STRIDE, 0));
      } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const int src = n0 - hist - k0 + t + T * k;
          const bool inr = src >= 0 && src < N;
          const cf xv = X[xs_index(inr ? (uint32_t)src : 0u, a.xs)];
          xdir[k] = inr ? xv : cmake(0.f, 0.f);
        }
      }
      __syncthreads();
      W::inv_s1(t, v, tw3, P);
      __syncthreads();
      W::inv_s2(t, v, P, Q);
      __syncthreads();
      W::inv_s3(t, v, tw1, Q);
      __syncthreads();
    }
    // the largest tap in the time domain: y - w0 xs[n - k0] - (the other taps' convolution); xs[i < 0] = 0 -- the convolution
    // is linear, WienerHopf.cpp:125-160 -- is the window's own zero fill
#pragma unroll
    for (int c = 0; c < 16; c++)
      bufstore_c32(od, (t + T * c - hist) * 8, ok ? csub(csub(yv[c], cmul(w0, xdir[c])), v[c]) : yv[c]); // not PD: surveillance channel passes through
  }
}


} // namespace

// The error string is shared with capi.hip through this hook.
extern "C" void blah2hip_set_error_(const char *msg);
extern "C" hipError_t blah2hip_ensure_lds_(const void *kern, int bytes);

#define CHIP(expr)                                                                        \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      blah2hip_set_error_((std::string(#expr) + ": " + hipGetErrorString(e_)).c_str());   \
      return BLAH2HIP_ERR_HIP;                                                            \
    }                                                                                     \
  } while (0)
#define CFAIL(code, msg)            \
  do {                              \
    blah2hip_set_error_(msg);       \
    return code;                    \
  } while (0)

struct blah2hip_clutter_s {
  int device = 0;
  int32_t delayMin = 0, delayMax = 0;
  uint32_t N = 0, maxBatch = 1;
  int32_t nBins = 0;
  int r3 = 8;
  int F = 2048, segLen = 0, nSeg = 0, nJobs = 0, firGrid = 0, numCU = 256;
  bool firCarry = false;     // blocks of F/2 samples, the window overlap carried in registers (plan)
  bool firNoCarry = false;   // BLAH2HIP_CLUTTER_OPT_FIR_CARRY = 0
  hipStream_t stream = nullptr;
  cf *d_tw = nullptr;
  cf *d_partial = nullptr;
  dcx *d_rb = nullptr;
  cf *d_w = nullptr;
  int32_t *d_ok = nullptr;
  cf *d_stage = nullptr; // host entry points: x, y, y_out planes
  size_t stageElems = 0;
  int solveK = 0;            // indices per thread of the one-workgroup Toeplitz solve (0 = by size)
  int solveForm = 0;         // BLAH2HIP_CLUTTER_OPT_SOLVE_FORM
  int solveE = 0;            // BLAH2HIP_CLUTTER_OPT_SOLVE_E (0 = by launch size)
  sla::u64 *d_mail = nullptr; // mailboxes of the look-ahead solve (solve_la.hpp), sized for max_batch
  int64_t mailWords = 0;
  uint32_t *d_epoch = nullptr; // [0] launch epoch, [1] sticky fault word (a bounded spin ran out), [2] CPIs solved again by the gated launch
  uint32_t spinLimit = sla::SPIN_LIMIT; // BLAH2HIP_CLUTTER_OPT_SOLVE_SPIN_LIMIT
  int laCap[4][2] = {};        // workgroups of clutter_solve_la_kernel<kE[i], 4 | 8> the chip holds at once
  int lastForm = 0, lastE = 0, lastG = 0; // what the last launch ran
  bool corrHalf = false;     // half-window correlation (2 transforms per F/2 samples) instead of the windowed one
  int fftLenForce = 0;       // BLAH2HIP_CLUTTER_OPT_FFT_LEN (0 = planner)
  int corrForce = 0;         // BLAH2HIP_CLUTTER_OPT_CORR
  int32_t *lastOk = nullptr; // where the last process call wrote its flags
  hipStream_t lastStream = nullptr; // ... and the stream it was enqueued on (blah2hip_clutter_get_info waits for that one only)
  KernelTimer<BLAH2HIP_CK_COUNT> timer;
  int stages = 7;            // launch_clutter runs: 1 correlations + reduction, 2 solve, 4 FIR (the long form drives its children piecewise)
  // LONG filters (more than F - 15 = 4081 taps; see long_process): two children of LONG_C taps run the
  // correlations (first lag = this filter's) and the FIR (first lag 0, on pre-shifted planes) chunk by chunk
  blah2hip_clutter_s *subCorr = nullptr, *subFir = nullptr;
  int nChunks = 0;
  cf *d_long = nullptr;      // [3][maxBatch][N]: private copies of x and y (the output is built in the y copy) + one work plane
  dcx *d_solveWs = nullptr;  // [maxBatch][4 nBins + 2]: clutter_solve_big_kernel's vectors
};

namespace {

hipError_t solve_la_capacity(blah2hip_clutter_s *h);

// Mailboxes of the look-ahead Toeplitz solve (solve_la.hpp): sized for the largest launch each plan can be chosen for
// (create, and again when BLAH2HIP_CLUTTER_OPT_SOLVE_E changes).  Zeroed once: tags are launch epochs >= 1.
int solve_la_alloc(blah2hip_clutter_s *h)
{
  // the mailbox layout depends on the slice width only; the largest batch a width can be chosen for is the handle's
  int64_t need = 0;
  for (int E : sla::kE) {
    if (h->solveE && E != h->solveE) continue;
    const sla::Plan p4 = sla::make_plan(h->nBins, E, 4), p8 = sla::make_plan(h->nBins, E, 8);
    int64_t batch = h->maxBatch;
    if (!h->solveE && E != 12) batch = std::min<int64_t>(batch, 8 * (h->numCU / (8 * std::min(p4.G, p8.G))));
    need = std::max(need, p4.stride * batch);
  }
  if (need > h->mailWords) {
    if (h->d_mail) CHIP(hipFree(h->d_mail));
    h->d_mail = nullptr; h->mailWords = 0;
    CHIP(hipMalloc(&h->d_mail, (size_t)need * sizeof(sla::u64)));
    CHIP(hipMemset(h->d_mail, 0, (size_t)need * sizeof(sla::u64)));
    h->mailWords = need;
  }
  if (!h->d_epoch) {
    CHIP(hipMalloc(&h->d_epoch, 4 * sizeof(uint32_t)));
    CHIP(hipMemset(h->d_epoch, 0, 4 * sizeof(uint32_t)));
  }
  if (!h->laCap[0][0]) CHIP(solve_la_capacity(h));
  return BLAH2HIP_OK;
}

// Resident workgroups per instantiation (occupancy x CUs): what sla::choose_plan may count on for plans whose workgroups
// wait for each other.
template <int E> hipError_t solve_la_capacity_e(blah2hip_clutter_s *h, int i)
{
  int b4 = 0, b8 = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b4, sla::clutter_solve_la_kernel<E, 4>, 256, 0);
  if (e != hipSuccess) return e;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b8, sla::clutter_solve_la_kernel<E, 8>, 512, 0);
  if (e != hipSuccess) return e;
  h->laCap[i][0] = std::max(1, b4) * h->numCU;
  h->laCap[i][1] = std::max(1, b8) * h->numCU;
  return hipSuccess;
}
hipError_t solve_la_capacity(blah2hip_clutter_s *h)
{
  hipError_t e = solve_la_capacity_e<2>(h, 0);
  if (e == hipSuccess) e = solve_la_capacity_e<3>(h, 1);
  if (e == hipSuccess) e = solve_la_capacity_e<6>(h, 2);
  if (e == hipSuccess) e = solve_la_capacity_e<12>(h, 3);
  return e;
}
int solve_la_cap(const blah2hip_clutter_s *h, int E, int NW)
{
  const int i = E == 2 ? 0 : (E == 3 ? 1 : (E == 6 ? 2 : 3));
  return h->laCap[i][NW == 8 ? 1 : 0];
}

template <int E> void launch_solve_la(const sla::Args &a, int grid, int nw, hipStream_t st)
{
  if (nw == 4) hipLaunchKernelGGL((sla::clutter_solve_la_kernel<E, 4>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((sla::clutter_solve_la_kernel<E, 8>), dim3(grid), dim3(512), 0, st, a);
}

// Transform length, segmentation, correlation form and the buffers sized by them (create, and again when
// BLAH2HIP_CLUTTER_OPT_FFT_LEN / _CORR re-plan).  F - nBins + 1 useful samples per F log F work.
int clutter_plan(blah2hip_clutter_s *h)
{
  const int nBins = h->nBins;
  int bestR3 = 0;
  double best = 1e300;
  for (int r3 : {4, 8, 16}) {
    const int F = 256 * r3;
    if (h->fftLenForce && F != h->fftLenForce) continue;
    const int L = F - nBins + 1;
    if (L < 16) continue;
    // measured per-point speed of the three transform kernels (tools/gpu_diag.py)
    const double cost = (double)F * std::log2((double)F) / (double)L * (r3 == 16 ? 1.4 : (r3 == 4 ? 1.08 : 1.0));
    if (cost < best) { best = cost; bestR3 = r3; }
  }
  if (!bestR3) CFAIL(BLAH2HIP_ERR_UNSUPPORTED, "nBins too large for the on-chip transform lengths (<= 4096)");
  h->r3 = bestR3; h->F = 256 * bestR3;
  h->segLen = h->F - nBins + 1;
  // a filter whose history is just under half the transform: blocks of exactly F/2 samples, so that consecutive windows
  // overlap by whole registers and the FIR kernel carries the overlap instead of reading it again (clutter_fir_kernel)
  h->firCarry = h->segLen >= h->F / 2 && h->segLen <= h->F / 2 + h->F / 64 && !h->firNoCarry;
  if (h->firCarry) h->segLen = h->F / 2;
  h->nSeg = (int)((h->N + (uint32_t)h->segLen - 1) / (uint32_t)h->segLen);
  // correlations: windowed form 3 transforms per segLen samples, half-window form 2 per F/2
  // (needs nBins - 1 <= F/2 and at least nBins samples)
  const bool halfOk = (nBins - 1 <= h->F / 2) && ((uint32_t)nBins <= h->N);
  h->corrHalf = halfOk && (4.0 / h->F < 3.0 / h->segLen);
  if (h->corrForce == BLAH2HIP_CLUTTER_CORR_HALF) {
    if (!halfOk) CFAIL(BLAH2HIP_ERR_UNSUPPORTED, "the half-window correlation needs nBins - 1 <= F/2 and nBins <= nSamples");
    h->corrHalf = true;
  }
  if (h->corrForce == BLAH2HIP_CLUTTER_CORR_WINDOW) h->corrHalf = false;
  auto up8 = [](int v) { return std::max(8, (v + 7) & ~7); };
  h->nJobs = up8(std::min(h->nSeg, 2 * h->numCU)); // partial correlations per CPI (x2 modes in one workgroup)
  h->firGrid = up8(std::min(h->nSeg, 8 * h->numCU));
  std::vector<cf> tw(h->F);
  for (int k = 0; k < h->F; k++) {
    const double a = -2.0 * M_PI * (double)k / (double)h->F;
    tw[k] = cmake((float)std::cos(a), (float)std::sin(a));
  }
  if (h->d_tw) CHIP(hipFree(h->d_tw));
  h->d_tw = nullptr;
  if (h->d_partial) CHIP(hipFree(h->d_partial));
  h->d_partial = nullptr;
  CHIP(hipMalloc(&h->d_tw, h->F * sizeof(cf)));
  CHIP(hipMemcpy(h->d_tw, tw.data(), h->F * sizeof(cf), hipMemcpyHostToDevice));
  CHIP(hipMalloc(&h->d_partial, (size_t)h->maxBatch * 2 * h->nJobs * nBins * sizeof(cf)));
  return BLAH2HIP_OK;
}

// clutter_solve_kernel: one index per thread up to 1024 taps (measured: more, smaller threads win while the recursion is
// latency-bound), 2 up to 2048, 4 above
void launch_solve_stepwise(blah2hip_clutter_s *h, const SolveArgs &sa, uint32_t nCpi, hipStream_t st)
{
  if (h->d_solveWs) { // the long filters: the recursion on vectors in global memory
    hipLaunchKernelGGL(clutter_solve_big_kernel, dim3(nCpi), dim3(1024), 0, st, sa, h->d_solveWs);
    return;
  }
  const size_t sl = ((size_t)2 * (h->nBins + 1)) * sizeof(dcx) + 2 * sizeof(SolveScal);
  const int kper = h->solveK ? h->solveK : (h->nBins > 2048 ? 4 : (h->nBins > 1024 ? 2 : 1));
  const int nt = std::min(1024, 64 * ((h->nBins + 64 * kper - 1) / (64 * kper)));
  if (kper == 1) hipLaunchKernelGGL(clutter_solve_kernel<1>, dim3(nCpi), dim3(nt), sl, st, sa);
  else if (kper == 2) hipLaunchKernelGGL(clutter_solve_kernel<2>, dim3(nCpi), dim3(nt), sl, st, sa);
  else hipLaunchKernelGGL(clutter_solve_kernel<4>, dim3(nCpi), dim3(nt), sl, st, sa);
}

// The Toeplitz solve of nCpi systems whose r, b sit in h->d_rb (the epoch of the look-ahead form's mailboxes has been
// bumped by the launch in front: clutter_reduce_kernel, or solve_epoch_kernel)
int launch_solve(blah2hip_clutter_s *h, const SolveArgs &sa, uint32_t nCpi, hipStream_t st)
{
  int32_t *ok = sa.ok;
  CHIP(blah2hip_ensure_lds_((const void *)clutter_solve_kernel<1>, 160 * 1024 - 2048));
  CHIP(blah2hip_ensure_lds_((const void *)clutter_solve_kernel<2>, 160 * 1024 - 2048));
  CHIP(blah2hip_ensure_lds_((const void *)clutter_solve_kernel<4>, 160 * 1024 - 2048));
  CHIP(h->timer.tic(BLAH2HIP_CK_SOLVE, st));
  if (h->solveForm != BLAH2HIP_CLUTTER_SOLVE_STEPWISE && h->d_mail) {
    // blocks of 32 orders on several workgroups per CPI (solve_la.hpp): as many CUs per CPI as the launch leaves free
    const sla::Plan p = sla::choose_plan(h->nBins, (int)nCpi, h->numCU, h->solveE, [h](int E, int NW) { return solve_la_cap(h, E, NW); });
    sla::Args la;
    la.rb = sa.rb; la.w = sa.w; la.ok = ok; la.mail = h->d_mail; la.epoch = h->d_epoch; la.fault = h->d_epoch + 1;
    la.spinLimit = h->spinLimit;
    la.n = h->nBins; la.NB = p.NB; la.nbulk = p.nbulk; la.G = p.G; la.nCpi = (int)nCpi; la.mailStride = p.stride;
    la.offHalo = p.offHalo; la.offFeed = p.offFeed; la.offHaloFlag = p.offHaloFlag; la.offFeedFlag = p.offFeedFlag;
    la.offStatus = p.offStatus;
    if (p.stride * (int64_t)nCpi > h->mailWords) CFAIL(BLAH2HIP_ERR_INVALID, "internal: solve mailbox too small for this launch");
    const int grid = ((int)nCpi + 7) / 8 * 8 * p.G;
    switch (p.E) {
    case 2: launch_solve_la<2>(la, grid, p.NW, st); break;
    case 3: launch_solve_la<3>(la, grid, p.NW, st); break;
    case 6: launch_solve_la<6>(la, grid, p.NW, st); break;
    default: launch_solve_la<12>(la, grid, p.NW, st); break;
    }
    h->lastForm = BLAH2HIP_CLUTTER_SOLVE_LOOKAHEAD; h->lastE = p.E; h->lastG = p.G;
    CHIP(hipGetLastError());
    // ... and behind it the one-workgroup kernel, gated per CPI on the fault word of this launch: workgroups that a busy
    // chip (another handle's kernels, a chip-filling launch on a second stream) dispatched so late that a bounded wait ran
    // out leave their CPI to it.  No CPI faulted: every workgroup reads one word and leaves (2 us for the launch).
    SolveArgs ga = sa;
    ga.gate = h->d_mail + p.offStatus + 1; ga.gateStride = p.stride; ga.retries = h->d_epoch + 2; ga.epoch = h->d_epoch;
    launch_solve_stepwise(h, ga, nCpi, st);
  } else {
    launch_solve_stepwise(h, sa, nCpi, st);
    h->lastForm = BLAH2HIP_CLUTTER_SOLVE_STEPWISE; h->lastE = 0; h->lastG = 1;
  }
  CHIP(hipGetLastError());
  CHIP(h->timer.toc(BLAH2HIP_CK_SOLVE, st));
  return BLAH2HIP_OK;
}

__global__ void solve_epoch_kernel(uint32_t *epoch)
{
  const uint32_t e = *epoch + 1u;
  *epoch = e ? e : 1u;
}

template <int R3, class In> int launch_clutter(blah2hip_clutter_s *h, const void *px, const void *py, uint32_t nCpi,
                                               int64_t stride, cf *yout, int64_t outStride, int32_t *ok, hipStream_t st)
{
  using W = WgFft<R3>;
  const size_t lds = (size_t)(W::A_ELEMS + W::B_ELEMS) * sizeof(cf);
  // once per (device, kernel), see capi.hip
  CHIP(blah2hip_ensure_lds_((const void *)clutter_corr_kernel<R3, In>, (int)lds));
  CHIP(blah2hip_ensure_lds_((const void *)clutter_fir_kernel<R3, In>, (int)lds));
  XsMap xs;
  xs.N = h->N;
  xs.thresh = h->delayMin > 0 ? (uint32_t)h->delayMin : 0u;
  xs.sub = (uint32_t)h->delayMin;
  xs.wrapC = (uint32_t)(((1ull << 32) - (uint64_t)(h->delayMin > 0 ? h->delayMin : 0)) % h->N);
  // Workgroups per CPI shrink as the batch grows: a correlation workgroup ends with two
  // inverse transforms and a partial write, a FIR workgroup starts with the transform of
  // the taps -- per-workgroup costs that a long walk over segments amortises.  Two
  // resident generations' worth keeps the tail short.
  const int slots = 4 * h->numCU; // residency of these kernels: 4 workgroups per CU (LDS)
  // (multiples of 8: the segment walk is XCD-aware, see seg_walk)
  auto up8 = [](int v) { return std::max(8, (v + 7) & ~7); };
  const int nJobs = std::min(h->nJobs, up8((2 * slots + (int)nCpi - 1) / (int)nCpi));
  const int firGrid = std::min(h->firGrid, up8((2 * slots + (int)nCpi - 1) / (int)nCpi));
  if (h->stages & 1) {
  CHIP(h->timer.tic(BLAH2HIP_CK_CORR, st));
  if (h->corrHalf) {
    CHIP(blah2hip_ensure_lds_((const void *)clutter_corr_half_kernel<R3, In>, (int)lds));
    CorrHalfArgs ha;
    ha.x = px; ha.y = py; ha.cpiStride = stride; ha.N = h->N; ha.xs = xs;
    ha.nBins = h->nBins; ha.nSeg = (int)((h->N + (uint32_t)(h->F / 2) - 1) / (uint32_t)(h->F / 2));
    ha.per = (ha.nSeg + nJobs - 1) / nJobs; ha.nJobs = nJobs;
    ha.tw = h->d_tw; ha.partial = h->d_partial; ha.scale = 1.0f / (float)h->F;
    hipLaunchKernelGGL((clutter_corr_half_kernel<R3, In>), dim3(nJobs, nCpi), dim3(W::T), lds, st, ha);
  } else {
    CorrArgs ca;
    ca.x = px; ca.y = py; ca.cpiStride = stride; ca.N = h->N; ca.xs = xs;
    ca.nBins = h->nBins; ca.segLen = h->segLen; ca.nSeg = h->nSeg; ca.nJobs = nJobs;
    ca.tw = h->d_tw; ca.partial = h->d_partial; ca.scale = 1.0f / (float)h->F;
    hipLaunchKernelGGL((clutter_corr_kernel<R3, In>), dim3(nJobs, nCpi), dim3(W::T), lds, st, ca);
  }
  CHIP(hipGetLastError());
  CHIP(h->timer.toc(BLAH2HIP_CK_CORR, st));

  SolveArgs sa;
  sa.partial = h->d_partial; sa.rb = h->d_rb; sa.w = h->d_w; sa.ok = ok; sa.nBins = h->nBins; sa.nJobs = nJobs;
  sa.epoch = h->d_epoch;
  CHIP(h->timer.tic(BLAH2HIP_CK_REDUCE, st));
  hipLaunchKernelGGL(clutter_reduce_kernel, dim3((h->nBins + 15) / 16, 2, nCpi), dim3(16 * RED_SLICES), 0, st, sa);
  CHIP(h->timer.toc(BLAH2HIP_CK_REDUCE, st));
  if (h->stages & 2) { const int rc_ = launch_solve(h, sa, nCpi, st); if (rc_) return rc_; }
  }

  FirArgs fa;
  fa.x = px; fa.y = py; fa.yout = yout; fa.cpiStride = stride; fa.outStride = outStride; fa.N = h->N; fa.xs = xs;
  fa.nBins = h->nBins; fa.segLen = h->segLen; fa.nSeg = h->nSeg; fa.w = h->d_w; fa.ok = ok; fa.tw = h->d_tw;
  fa.scale = 1.0f / (float)h->F;
  fa.carry = h->firCarry ? 1 : 0;
  if (yout && (h->stages & 4)) { // nullptr: the taps only (blah2hip_clutter_estimate_dev_fmt: the FIR runs fused into the range kernel)
    CHIP(h->timer.tic(BLAH2HIP_CK_FIR, st));
    hipLaunchKernelGGL((clutter_fir_kernel<R3, In>), dim3(firGrid, nCpi), dim3(W::T), lds, st, fa);
    CHIP(hipGetLastError());
    CHIP(h->timer.toc(BLAH2HIP_CK_FIR, st));
  }
  h->lastOk = ok;
  h->lastStream = st;
  return BLAH2HIP_OK;
}

// ---- LONG filters: more taps than one transform holds --------------------------------------------------------------
// WienerHopf.cpp takes any nBins (a dense nBins x nBins Cholesky and FFTs of the whole CPI).  The kernels above hold
// nBins <= F - 15 = 4081.  Beyond that -- any nBins <= nSamples -- the same
// kernels run chunk by chunk of LONG_C = 2048 lags / taps on rotated or shifted copies of the channels:
//   b[cC + j] = sum_n y[n] conj(xs[n - cC - j]) = the child's b[j] with y rotated by cC            (circular, :100-108)
//   r[cC + j] = the same with xs in y's place                                                       (:76-84)
//   y - (w * xs) = y - sum_c (w[cC ...] * xs delayed by cC, zeros shifted in)                       (linear, :125-160)
// with xs[i] = x[(uint32(i) - uint32(delayMin)) mod N] (:61-70, the reference's own unsigned arithmetic).  One child handle
// (first lag = this filter's) does the correlations, a second (first lag 0: its xs IS its x) the FIR passes; the solve is
// clutter_solve_kernel<11, true, 768>.  fp32 planes only.  A rarely used form, built for coverage rather than speed: 2 nChunks - 1
// correlation passes and nChunks FIR passes over the CPI, ~1 us per order of the solve.
constexpr int LONG_C = 2048;

// planes [nCpi][N] out of planes [nCpi][N] (every index below 2^31: no overflow in uint32):
// MODE 0: dst[m] = src[(m + off) mod N]                      (y rotated)
// MODE 1: dst[m] = xs[(m + off) mod N]                       (xs rotated; src = x)
// MODE 2: dst[m] = m >= off ? xs[m - off] : 0                (xs delayed, zeros shifted in; src = x)
// xs[i] = x[(uint32(i) - uint32(delayMin)) mod N]: the reference's own unsigned arithmetic (WienerHopf.cpp:61-70)
// (A first version with the mode as a run-time argument and a grid-stride loop faulted on gfx950 -- in a stand-alone
// program too; this one is one sample per thread and a compile-time mode.)
template <int MODE>
__global__ __launch_bounds__(256) void long_plane_kernel(const cf *src, cf *dst, uint32_t N, uint32_t off, uint32_t dmin)
{
  const cf *s = src + (size_t)blockIdx.y * N;
  cf *d = dst + (size_t)blockIdx.y * N;
  const uint32_t m = blockIdx.x * 256u + threadIdx.x;
  if (m >= N) return;
  cf v = cmake(0.f, 0.f);
  if (MODE == 0) v = s[(m + off) % N];
  if (MODE == 1) v = s[(((m + off) % N) - dmin) % N];
  if (MODE == 2 && m >= off) v = s[((m - off) - dmin) % N];
  d[m] = v;
}

// which 0: the child's b -> b[cC + j]; 1: the child's r -> r[cC + j] (chunk 0); 2: the child's b -> r[cC + j]
__global__ __launch_bounds__(256) void long_gather_kernel(const dcx *sub, dcx *big, int C, int n, int c, int which)
{
  const int j = blockIdx.x * 256 + threadIdx.x, cpi = blockIdx.y;
  const int k = c * C + j;
  if (j >= C || k >= n) return;
  const dcx v = sub[((size_t)cpi * 2 + (which == 1 ? 0 : 1)) * C + j];
  big[((size_t)cpi * 2 + (which == 0 ? 1 : 0)) * n + k] = v;
}

// the FIR child's taps of chunk c: w[cC + j], zeros behind the filter's last tap
__global__ __launch_bounds__(256) void long_taps_kernel(const cf *w, cf *wsub, int C, int n, int c)
{
  const int j = blockIdx.x * 256 + threadIdx.x, cpi = blockIdx.y;
  if (j >= C) return;
  const int k = c * C + j;
  wsub[(size_t)cpi * C + j] = k < n ? w[(size_t)cpi * n + k] : cmake(0.f, 0.f);
}

int long_process(blah2hip_clutter_s *h, const cf *d_x, const cf *d_y, uint32_t nCpi, int64_t stride, cf *yout, int64_t outStride,
                 int32_t *ok, hipStream_t st)
{
  const uint32_t N = h->N, dmin = (uint32_t)h->delayMin;
  const int n = h->nBins, C = LONG_C;
  const size_t plane = (size_t)h->maxBatch * N;
  cf *xp = h->d_long, *yp = xp + plane, *wp = yp + plane;
  const dim3 pg((N + 255) / 256, nCpi), gg((C + 255) / 256, nCpi);
  blah2hip_clutter_s *sc = h->subCorr, *sf = h->subFir;
  // private copies with the children's stride (the caller's may differ from N); the output is built in the copy of y
  CHIP(hipMemcpy2DAsync(xp, (size_t)N * sizeof(cf), d_x, (size_t)stride * sizeof(cf), (size_t)N * sizeof(cf), nCpi, hipMemcpyDeviceToDevice, st));
  CHIP(hipMemcpy2DAsync(yp, (size_t)N * sizeof(cf), d_y, (size_t)stride * sizeof(cf), (size_t)N * sizeof(cf), nCpi, hipMemcpyDeviceToDevice, st));
  sc->stages = 1;
  for (int c = 0; c < h->nChunks; c++) {
    const cf *yin = yp;
    if (c > 0) {
      long_plane_kernel<0><<<pg, 256, 0, st>>>(yp, wp, N, (uint32_t)(c * C), 0u);
      yin = wp;
    }
    int rc = launch_clutter<16, InC32>(sc, xp, yin, nCpi, (int64_t)N, nullptr, 0, ok, st);
    if (rc) return rc;
    long_gather_kernel<<<gg, 256, 0, st>>>(sc->d_rb, h->d_rb, C, n, c, 0);
    if (c == 0) {
      long_gather_kernel<<<gg, 256, 0, st>>>(sc->d_rb, h->d_rb, C, n, c, 1);
    } else {
      long_plane_kernel<1><<<pg, 256, 0, st>>>(xp, wp, N, (uint32_t)(c * C), dmin);
      rc = launch_clutter<16, InC32>(sc, xp, wp, nCpi, (int64_t)N, nullptr, 0, ok, st);
      if (rc) return rc;
      long_gather_kernel<<<gg, 256, 0, st>>>(sc->d_rb, h->d_rb, C, n, c, 2);
    }
    CHIP(hipGetLastError());
  }
  SolveArgs sa;
  sa.partial = nullptr; sa.rb = h->d_rb; sa.w = h->d_w; sa.ok = ok; sa.nBins = n; sa.nJobs = 0; sa.epoch = h->d_epoch;
  { const int rc = launch_solve(h, sa, nCpi, st); if (rc) return rc; }
  if (yout) {
    sf->stages = 4;
    CHIP(h->timer.tic(BLAH2HIP_CK_FIR, st));
    for (int c = 0; c < h->nChunks; c++) {
      long_taps_kernel<<<gg, 256, 0, st>>>(h->d_w, sf->d_w, C, n, c);
      long_plane_kernel<2><<<pg, 256, 0, st>>>(xp, wp, N, (uint32_t)(c * C), dmin);
      CHIP(hipGetLastError());
      const int rc = launch_clutter<16, InC32>(sf, wp, yp, nCpi, (int64_t)N, yp, (int64_t)N, ok, st); // in place: y -= w_c * xs_c
      if (rc) return rc;
    }
    CHIP(hipMemcpy2DAsync(yout, (size_t)outStride * sizeof(cf), yp, (size_t)N * sizeof(cf), (size_t)N * sizeof(cf), nCpi, hipMemcpyDeviceToDevice, st));
    CHIP(h->timer.toc(BLAH2HIP_CK_FIR, st));
  }
  h->lastOk = ok;
  h->lastStream = st;
  return BLAH2HIP_OK;
}

} // namespace

extern "C" {

int blah2hip_clutter_create(int32_t delay_min, int32_t delay_max, uint32_t n_samples, int device,
                            uint32_t max_batch, blah2hip_clutter_t *out)
{
  if (!out) CFAIL(BLAH2HIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (delay_max <= delay_min) CFAIL(BLAH2HIP_ERR_INVALID, "clutter filter needs delayMax > delayMin");
  if (n_samples == 0) CFAIL(BLAH2HIP_ERR_INVALID, "nSamples must be positive");
  if (n_samples > 0x7fffffffu - 8192u) CFAIL(BLAH2HIP_ERR_UNSUPPORTED, "nSamples >= 2^31 - 8192 (32-bit sample indices)");
  if ((uint64_t)(delay_min < 0 ? -(int64_t)delay_min : (int64_t)delay_min) >= n_samples)
    CFAIL(BLAH2HIP_ERR_INVALID, "|delayMin| >= nSamples");
  const int nBins = delay_max - delay_min;
  if (max_batch == 0) max_batch = 1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    CFAIL(BLAH2HIP_ERR_NO_DEVICE, "no HIP device visible (the HIP path is the only path)");
  if (device < 0 || device >= ndev) CFAIL(BLAH2HIP_ERR_INVALID, "device index out of range");
  CHIP(hipSetDevice(device));
  // beyond one transform (nBins > F - 15 = 4081): the long form, chunks of LONG_C taps on two child handles (long_process);
  // its one-workgroup solve holds one fp64 vector of nBins in LDS, eight indices per thread
  const bool isLong = nBins > 4096 - 15;
  // (more taps than samples: the reference reads its nSamples correlation lags out of bounds, WienerHopf.cpp:76-108)
  if (isLong && (uint32_t)nBins > n_samples) CFAIL(BLAH2HIP_ERR_UNSUPPORTED, "a long filter (more than 4081 taps) needs nBins <= nSamples");
  auto *h = new blah2hip_clutter_s;
  // everything that can fail runs inside `build`; a partially built handle is torn down by destroy()
  auto build = [&]() -> int {
  h->device = device;
  h->delayMin = delay_min; h->delayMax = delay_max;
  h->N = n_samples; h->maxBatch = max_batch; h->nBins = nBins;
  hipDeviceProp_t prop;
  CHIP(hipGetDeviceProperties(&prop, device));
  h->numCU = prop.multiProcessorCount;
  CHIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  CHIP(hipMalloc(&h->d_rb, (size_t)max_batch * 2 * nBins * sizeof(dcx)));
  CHIP(hipMalloc(&h->d_w, (size_t)max_batch * nBins * sizeof(cf)));
  CHIP(hipMalloc(&h->d_ok, max_batch * sizeof(int32_t)));
  if (isLong) {
    h->nChunks = (nBins + LONG_C - 1) / LONG_C;
    h->solveForm = BLAH2HIP_CLUTTER_SOLVE_STEPWISE;
    { const int rc_ = blah2hip_clutter_create(delay_min, delay_min + LONG_C, n_samples, device, max_batch, &h->subCorr); if (rc_) return rc_; }
    { const int rc_ = blah2hip_clutter_create(0, LONG_C, n_samples, device, max_batch, &h->subFir); if (rc_) return rc_; }
    h->r3 = h->subFir->r3; h->F = h->subFir->F; h->segLen = h->subFir->segLen; h->nSeg = h->subFir->nSeg;
    CHIP(hipMalloc(&h->d_long, (size_t)3 * max_batch * n_samples * sizeof(cf)));
    CHIP(hipMalloc(&h->d_solveWs, (size_t)max_batch * (4 * (size_t)nBins + 2) * sizeof(dcx)));
    CHIP(hipMalloc(&h->d_epoch, 4 * sizeof(uint32_t)));
    CHIP(hipMemset(h->d_epoch, 0, 4 * sizeof(uint32_t)));
    return BLAH2HIP_OK;
  }
  { const int rc_ = clutter_plan(h); if (rc_) return rc_; }
  { const int rc_ = solve_la_alloc(h); if (rc_) return rc_; }
  return BLAH2HIP_OK;
  };
  const int rc = build();
  if (rc != BLAH2HIP_OK) {
    blah2hip_clutter_destroy(h);
    return rc;
  }
  *out = h;
  return BLAH2HIP_OK;
}

int blah2hip_clutter_destroy(blah2hip_clutter_t h)
{
  if (!h) return BLAH2HIP_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->subCorr) (void)blah2hip_clutter_destroy(h->subCorr);
  if (h->subFir) (void)blah2hip_clutter_destroy(h->subFir);
  for (void *p : {(void *)h->d_tw, (void *)h->d_partial, (void *)h->d_rb, (void *)h->d_w, (void *)h->d_ok,
                  (void *)h->d_stage, (void *)h->d_mail, (void *)h->d_epoch, (void *)h->d_long, (void *)h->d_solveWs})
    if (p) (void)hipFree(p);
  h->timer.destroy();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return BLAH2HIP_OK;
}

int blah2hip_clutter_get_dims(blah2hip_clutter_t h, uint32_t *n_bins, uint32_t *fft_len, uint32_t *seg_len)
{
  if (!h) CFAIL(BLAH2HIP_ERR_INVALID, "NULL handle");
  if (n_bins) *n_bins = (uint32_t)h->nBins;
  if (fft_len) *fft_len = (uint32_t)h->F;
  if (seg_len) *seg_len = (uint32_t)h->segLen;
  return BLAH2HIP_OK;
}

int blah2hip_clutter_get_info(blah2hip_clutter_t h, int what, int64_t *value)
{
  if (!h || !value) CFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  switch (what) {
  case BLAH2HIP_CLUTTER_INFO_SOLVE_FORM: *value = h->lastForm; return BLAH2HIP_OK;
  case BLAH2HIP_CLUTTER_INFO_SOLVE_E: *value = h->lastE; return BLAH2HIP_OK;
  case BLAH2HIP_CLUTTER_INFO_SOLVE_G: *value = h->lastG; return BLAH2HIP_OK;
  case BLAH2HIP_CLUTTER_INFO_SOLVE_FAULT:
  case BLAH2HIP_CLUTTER_INFO_SOLVE_RETRIES: {
    // the stream the handle's last call ran on, not the device (a poll in a pipeline must not stall other streams and
    // handles); both words in one copy
    uint32_t v[2] = {0, 0};
    CHIP(hipSetDevice(h->device));
    CHIP(hipStreamSynchronize(h->lastStream));
    if (h->d_epoch) CHIP(hipMemcpy(v, h->d_epoch + 1, sizeof v, hipMemcpyDeviceToHost));
    *value = what == BLAH2HIP_CLUTTER_INFO_SOLVE_FAULT ? v[0] : v[1];
    return BLAH2HIP_OK;
  }
  default: CFAIL(BLAH2HIP_ERR_INVALID, "unknown info");
  }
}

int blah2hip_clutter_read_last(blah2hip_clutter_t h, uint32_t cpi, float *w, double *rb, int *ok)
{
  if (!h) CFAIL(BLAH2HIP_ERR_INVALID, "NULL handle");
  if (cpi >= h->maxBatch) CFAIL(BLAH2HIP_ERR_INVALID, "cpi index out of range");
  CHIP(hipSetDevice(h->device));
  CHIP(hipDeviceSynchronize());
  if (w) CHIP(hipMemcpy(w, h->d_w + (size_t)cpi * h->nBins, (size_t)h->nBins * sizeof(cf), hipMemcpyDeviceToHost));
  if (rb) CHIP(hipMemcpy(rb, h->d_rb + (size_t)cpi * 2 * h->nBins, (size_t)2 * h->nBins * sizeof(dcx), hipMemcpyDeviceToHost));
  if (ok) {
    int32_t v = 0;
    if (!h->lastOk) CFAIL(BLAH2HIP_ERR_INVALID, "no process call yet");
    CHIP(hipMemcpy(&v, h->lastOk + cpi, sizeof(int32_t), hipMemcpyDeviceToHost));
    *ok = v;
  }
  return BLAH2HIP_OK;
}

int blah2hip_clutter_set_option(blah2hip_clutter_t h, int option, int64_t value)
{
  if (!h) CFAIL(BLAH2HIP_ERR_INVALID, "NULL handle");
  if (h->subCorr) CFAIL(BLAH2HIP_ERR_UNSUPPORTED, "a long filter (more than 4081 taps) has one plan: no options");
  switch (option) {
  case BLAH2HIP_CLUTTER_OPT_SOLVE_K:
    if (value != 0 && value != 1 && value != 2 && value != 4) CFAIL(BLAH2HIP_ERR_INVALID, "indices per thread: 0 (auto), 1, 2 or 4");
    if (value && (int64_t)h->nBins > 1024 * value) CFAIL(BLAH2HIP_ERR_UNSUPPORTED, "nBins needs more indices per thread");
    h->solveK = (int)value;
    return BLAH2HIP_OK;
  case BLAH2HIP_CLUTTER_OPT_SOLVE_FORM:
    if (value < BLAH2HIP_CLUTTER_SOLVE_AUTO || value > BLAH2HIP_CLUTTER_SOLVE_LOOKAHEAD)
      CFAIL(BLAH2HIP_ERR_INVALID, "solve form: BLAH2HIP_CLUTTER_SOLVE_AUTO, _STEPWISE or _LOOKAHEAD");
    h->solveForm = (int)value;
    return BLAH2HIP_OK;
  case BLAH2HIP_CLUTTER_OPT_SOLVE_E: {
    if (value != 0 && value != 2 && value != 3 && value != 6 && value != 12)
      CFAIL(BLAH2HIP_ERR_INVALID, "indices per lane of the look-ahead solve: 0 (by launch size), 2, 3, 6 or 12");
    const int prev = h->solveE;
    h->solveE = (int)value;
    CHIP(hipSetDevice(h->device));
    CHIP(hipDeviceSynchronize()); // the mailboxes may still be in use by enqueued work
    const int rc = solve_la_alloc(h);
    if (rc) h->solveE = prev;
    return rc;
  }
  case BLAH2HIP_CLUTTER_OPT_SOLVE_SPIN_LIMIT:
    if (value < 0 || value > (int64_t)0x7fffffff) CFAIL(BLAH2HIP_ERR_INVALID, "polls per bounded wait: 0 (default) ... 2^31 - 1");
    h->spinLimit = value ? (uint32_t)value : sla::SPIN_LIMIT;
    return BLAH2HIP_OK;
  case BLAH2HIP_CLUTTER_OPT_FIR_CARRY: {
    if (value != 0 && value != 1) CFAIL(BLAH2HIP_ERR_INVALID, "FIR carry: 0 or 1");
    const bool prev = h->firNoCarry;
    h->firNoCarry = value == 0;
    CHIP(hipSetDevice(h->device));
    CHIP(hipDeviceSynchronize()); // the buffers may still be in use by enqueued work
    const int rc = clutter_plan(h);
    if (rc) { h->firNoCarry = prev; (void)clutter_plan(h); }
    return rc;
  }
  case BLAH2HIP_CLUTTER_OPT_FFT_LEN:
  case BLAH2HIP_CLUTTER_OPT_CORR: {
    const bool isLen = option == BLAH2HIP_CLUTTER_OPT_FFT_LEN;
    if (isLen && value != 0 && value != 1024 && value != 2048 && value != 4096)
      CFAIL(BLAH2HIP_ERR_INVALID, "clutter transform length: 0 (planner), 1024, 2048 or 4096");
    if (!isLen && (value < BLAH2HIP_CLUTTER_CORR_AUTO || value > BLAH2HIP_CLUTTER_CORR_WINDOW))
      CFAIL(BLAH2HIP_ERR_INVALID, "correlation form: BLAH2HIP_CLUTTER_CORR_AUTO, _HALF or _WINDOW");
    const int prevLen = h->fftLenForce, prevCorr = h->corrForce;
    (isLen ? h->fftLenForce : h->corrForce) = (int)value;
    CHIP(hipSetDevice(h->device));
    CHIP(hipDeviceSynchronize()); // the buffers may still be in use by enqueued work
    const int rc = clutter_plan(h);
    if (rc) {
      const std::string msg = blah2hip_last_error();
      h->fftLenForce = prevLen; h->corrForce = prevCorr;
      (void)clutter_plan(h); // back to the previous, valid plan
      blah2hip_set_error_(msg.c_str());
    }
    return rc;
  }
  default: CFAIL(BLAH2HIP_ERR_INVALID, "unknown option");
  }
}

int blah2hip_clutter_solve(blah2hip_clutter_t h, const double *rb, uint32_t n_cpi, float *w, int32_t *ok)
{
  if (!h || !rb || !w || !ok) CFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n_cpi == 0 || n_cpi > h->maxBatch) CFAIL(BLAH2HIP_ERR_INVALID, "n_cpi outside [1, max_batch]");
  CHIP(hipSetDevice(h->device));
  const size_t n = (size_t)h->nBins;
  CHIP(hipMemcpyAsync(h->d_rb, rb, (size_t)n_cpi * 2 * n * sizeof(dcx), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(solve_epoch_kernel, dim3(1), dim3(1), 0, h->stream, h->d_epoch);
  SolveArgs sa;
  sa.partial = nullptr; sa.rb = h->d_rb; sa.w = h->d_w; sa.ok = h->d_ok; sa.nBins = h->nBins; sa.nJobs = 0; sa.epoch = h->d_epoch;
  { const int rc_ = launch_solve(h, sa, n_cpi, h->stream); if (rc_) return rc_; }
  h->lastOk = h->d_ok;
  h->lastStream = h->stream;
  CHIP(hipMemcpyAsync(w, h->d_w, (size_t)n_cpi * n * sizeof(cf), hipMemcpyDeviceToHost, h->stream));
  CHIP(hipMemcpyAsync(ok, h->d_ok, (size_t)n_cpi * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  CHIP(hipStreamSynchronize(h->stream));
  return BLAH2HIP_OK;
}

int blah2hip_clutter_solve_dev(blah2hip_clutter_t h, const double *d_rb, uint32_t n_cpi, float *d_w, int32_t *d_ok, void *stream)
{
  if (!h || !d_rb || !d_w || !d_ok) CFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n_cpi == 0 || n_cpi > h->maxBatch) CFAIL(BLAH2HIP_ERR_INVALID, "n_cpi outside [1, max_batch]");
  CHIP(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(solve_epoch_kernel, dim3(1), dim3(1), 0, st, h->d_epoch);
  SolveArgs sa;
  sa.partial = nullptr; sa.rb = (dcx *)d_rb; sa.w = (cf *)d_w; sa.ok = d_ok; sa.nBins = h->nBins; sa.nJobs = 0; sa.epoch = h->d_epoch;
  h->lastOk = d_ok;
  h->lastStream = (hipStream_t)stream;
  return launch_solve(h, sa, n_cpi, st);
}

int blah2hip_clutter_set_timing(blah2hip_clutter_t h, int enable)
{
  if (!h) CFAIL(BLAH2HIP_ERR_INVALID, "NULL handle");
  h->timer.enabled = enable != 0;
  if (h->subCorr) h->subCorr->timer.enabled = enable != 0; // the long form: correlations and reduction are the child's launches
  return BLAH2HIP_OK;
}

int blah2hip_clutter_get_timing(blah2hip_clutter_t h, double *ms_total, uint32_t *launches)
{
  if (!h || !ms_total || !launches) CFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  CHIP(hipSetDevice(h->device));
  CHIP(hipDeviceSynchronize());
  CHIP(h->timer.collect(ms_total, launches));
  if (h->subCorr) { // correlations + reduction of the long form (its FIR passes and solve are bracketed on this handle)
    double ms[BLAH2HIP_CK_COUNT];
    uint32_t nl[BLAH2HIP_CK_COUNT];
    CHIP(h->subCorr->timer.collect(ms, nl));
    for (int k : {BLAH2HIP_CK_CORR, BLAH2HIP_CK_REDUCE}) { ms_total[k] += ms[k]; launches[k] += nl[k]; }
  }
  return BLAH2HIP_OK;
}

int blah2hip_clutter_process_dev_fmt(blah2hip_clutter_t h, int fmt, const void *d_x, const void *d_y, uint32_t n_cpi,
                                     uint64_t cpi_stride, void *d_y_out, uint64_t out_stride, int32_t *d_ok, void *stream)
{
  if (!h || !d_x || !d_y_out) CFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (fmt != BLAH2HIP_FMT_C32 && fmt != BLAH2HIP_FMT_I16) CFAIL(BLAH2HIP_ERR_INVALID, "clutter filter input: BLAH2HIP_FMT_C32 or BLAH2HIP_FMT_I16");
  if (fmt == BLAH2HIP_FMT_C32 && !d_y) CFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n_cpi == 0 || n_cpi > h->maxBatch) CFAIL(BLAH2HIP_ERR_INVALID, "n_cpi outside [1, max_batch]");
#ifndef B2_EXPERIMENT_ALIASED_CPIS // tools/gpu_cfg3_bytes.py
  if (n_cpi > 1 && (cpi_stride < h->N || out_stride < h->N)) CFAIL(BLAH2HIP_ERR_INVALID, "cpi_stride / out_stride < nSamples");
#endif
  CHIP(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  int32_t *ok = d_ok ? d_ok : h->d_ok;
  // in-place operation (FMT_C32, d_y_out == d_y) is safe: every output sample is read (as y) by the one
  // thread that writes it, and x is never written
  const int64_t cs = (int64_t)cpi_stride, os = (int64_t)out_stride;
  cf *yo = (cf *)d_y_out;
  if (h->subCorr) {
    if (fmt != BLAH2HIP_FMT_C32) CFAIL(BLAH2HIP_ERR_UNSUPPORTED, "a long filter (more than 4081 taps) takes fp32 planes only");
    return long_process(h, (const cf *)d_x, (const cf *)d_y, n_cpi, cs, yo, os, ok, st);
  }
  if (fmt == BLAH2HIP_FMT_I16) {
    switch (h->r3) {
    case 4: return launch_clutter<4, InI16>(h, d_x, nullptr, n_cpi, cs, yo, os, ok, st);
    case 8: return launch_clutter<8, InI16>(h, d_x, nullptr, n_cpi, cs, yo, os, ok, st);
    default: return launch_clutter<16, InI16>(h, d_x, nullptr, n_cpi, cs, yo, os, ok, st);
    }
  }
  switch (h->r3) {
  case 4: return launch_clutter<4, InC32>(h, d_x, d_y, n_cpi, cs, yo, os, ok, st);
  case 8: return launch_clutter<8, InC32>(h, d_x, d_y, n_cpi, cs, yo, os, ok, st);
  default: return launch_clutter<16, InC32>(h, d_x, d_y, n_cpi, cs, yo, os, ok, st);
  }
}

int blah2hip_clutter_estimate_dev_fmt(blah2hip_clutter_t h, int fmt, const void *d_x, const void *d_y, uint32_t n_cpi,
                                      uint64_t cpi_stride, int32_t *d_ok, void *stream)
{
  if (!h || !d_x) CFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (fmt != BLAH2HIP_FMT_C32 && fmt != BLAH2HIP_FMT_I16) CFAIL(BLAH2HIP_ERR_INVALID, "clutter filter input: BLAH2HIP_FMT_C32 or BLAH2HIP_FMT_I16");
  if (fmt == BLAH2HIP_FMT_C32 && !d_y) CFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n_cpi == 0 || n_cpi > h->maxBatch) CFAIL(BLAH2HIP_ERR_INVALID, "n_cpi outside [1, max_batch]");
  if (n_cpi > 1 && cpi_stride < h->N) CFAIL(BLAH2HIP_ERR_INVALID, "cpi_stride < nSamples");
  CHIP(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  int32_t *ok = d_ok ? d_ok : h->d_ok;
  const int64_t cs = (int64_t)cpi_stride;
  if (h->subCorr) {
    if (fmt != BLAH2HIP_FMT_C32) CFAIL(BLAH2HIP_ERR_UNSUPPORTED, "a long filter (more than 4081 taps) takes fp32 planes only");
    return long_process(h, (const cf *)d_x, (const cf *)d_y, n_cpi, cs, nullptr, 0, ok, st);
  }
  if (fmt == BLAH2HIP_FMT_I16) {
    switch (h->r3) {
    case 4: return launch_clutter<4, InI16>(h, d_x, nullptr, n_cpi, cs, nullptr, 0, ok, st);
    case 8: return launch_clutter<8, InI16>(h, d_x, nullptr, n_cpi, cs, nullptr, 0, ok, st);
    default: return launch_clutter<16, InI16>(h, d_x, nullptr, n_cpi, cs, nullptr, 0, ok, st);
    }
  }
  switch (h->r3) {
  case 4: return launch_clutter<4, InC32>(h, d_x, d_y, n_cpi, cs, nullptr, 0, ok, st);
  case 8: return launch_clutter<8, InC32>(h, d_x, d_y, n_cpi, cs, nullptr, 0, ok, st);
  default: return launch_clutter<16, InC32>(h, d_x, d_y, n_cpi, cs, nullptr, 0, ok, st);
  }
}

int blah2hip_clutter_taps_dev(blah2hip_clutter_t h, const float **d_w, uint32_t *n_bins, int32_t *delay_min)
{
  if (!h) CFAIL(BLAH2HIP_ERR_INVALID, "NULL handle");
  if (d_w) *d_w = reinterpret_cast<const float *>(h->d_w);
  if (n_bins) *n_bins = (uint32_t)h->nBins;
  if (delay_min) *delay_min = h->delayMin;
  return BLAH2HIP_OK;
}

int blah2hip_clutter_process_dev(blah2hip_clutter_t h, const void *d_x, const void *d_y, uint32_t n_cpi,
                                 uint64_t cpi_stride, void *d_y_out, int32_t *d_ok, void *stream)
{
  return blah2hip_clutter_process_dev_fmt(h, BLAH2HIP_FMT_C32, d_x, d_y, n_cpi, cpi_stride, d_y_out, cpi_stride, d_ok, stream);
}

int blah2hip_clutter_process_c32(blah2hip_clutter_t h, const float *x, const float *y, uint32_t n, float *y_out, int *ok)
{
  if (!h || !x || !y || !ok) CFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n != h->N) CFAIL(BLAH2HIP_ERR_INVALID, "n differs from the nSamples the filter was created for");
  CHIP(hipSetDevice(h->device));
  const size_t plane = (size_t)h->N;
  if (h->stageElems < 3 * plane) {
    if (h->d_stage) CHIP(hipFree(h->d_stage));
    h->d_stage = nullptr;
    CHIP(hipMalloc(&h->d_stage, 3 * plane * sizeof(cf)));
    h->stageElems = 3 * plane;
  }
  cf *dx = h->d_stage, *dy = dx + plane, *dout = dy + plane;
  CHIP(hipMemcpyAsync(dx, x, plane * sizeof(cf), hipMemcpyHostToDevice, h->stream));
  CHIP(hipMemcpyAsync(dy, y, plane * sizeof(cf), hipMemcpyHostToDevice, h->stream));
  int rc = blah2hip_clutter_process_dev(h, dx, dy, 1, plane, dout, h->d_ok, h->stream);
  if (rc) return rc;
  int32_t okv = 0;
  CHIP(hipMemcpyAsync(&okv, h->d_ok, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  CHIP(hipStreamSynchronize(h->stream));
  *ok = okv;
  if (okv && y_out) CHIP(hipMemcpy(y_out, dout, plane * sizeof(cf), hipMemcpyDeviceToHost));
  return BLAH2HIP_OK;
}

int blah2hip_clutter_process_c64(blah2hip_clutter_t h, const double *x, const double *y, uint32_t n, double *y_out, int *ok)
{
  if (!h || !x || !y || !ok) CFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  std::vector<float> fx(2 * (size_t)n), fy(2 * (size_t)n), fo(y_out ? 2 * (size_t)n : 0);
  for (size_t i = 0; i < 2 * (size_t)n; i++) { fx[i] = (float)x[i]; fy[i] = (float)y[i]; }
  int rc = blah2hip_clutter_process_c32(h, fx.data(), fy.data(), n, y_out ? fo.data() : nullptr, ok);
  if (rc) return rc;
  if (*ok && y_out)
    for (size_t i = 0; i < 2 * (size_t)n; i++) y_out[i] = (double)fo[i];
  return BLAH2HIP_OK;
}

} // extern "C"
