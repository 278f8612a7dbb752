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
//   * A is Hermitian Toeplitz -> Levinson recursion in fp64, O(nBins^2), one
//     workgroup (clutter_solve_kernel).  The reference's chol() fails exactly
//     when A is not positive definite; Levinson detects the same condition
//     (a prediction-error factor 1-|e|^2 <= 0 or r[0] <= 0) -> ok = 0;
//   * the FIR is an overlap-save convolution on the on-chip FFT
//     (clutter_fir_kernel), one pass over x and y.
// Results are mathematically identical to the reference's (same linear
// system, same linear convolution); arithmetic is fp32 for the transforms and
// fp64 for the reduction and the solve.
#include <hip/hip_runtime.h>

#include "blah2hip.h"
#include "fft_wg.hpp"

#include <algorithm>
#include <stdint.h>

#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

using namespace blah2;

namespace {

struct dcx {
  double x, y;
};

// WienerHopf.cpp:67: (i - delayMin) evaluated in uint32, then mod N
__device__ __forceinline__ uint32_t xs_index(uint32_t i, uint32_t dMinU32, uint32_t N) { return (i - dMinU32) % N; }

struct CorrArgs {
  const cf *x, *y;
  int64_t cpiStride;
  uint32_t N, dMinU32;
  int32_t nBins, segLen, nSeg, nJobs;
  const cf *tw;
  cf *partial; // [nCpi][2][nJobs][nBins]
  float scale;
};

// grid (nJobs, 2, nCpi): blockIdx.y = 0 -> r (window = xs), 1 -> b (window = y)
template <int R3> __global__ __launch_bounds__(16 * R3, 2) void clutter_corr_kernel(CorrArgs a)
{
  using W = WgFft<R3>;
  constexpr int T = W::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *P = reinterpret_cast<cf *>(smem);
  cf *Q = P + W::A_ELEMS;
  const int t = threadIdx.x;
  const int mode = blockIdx.y;
  const int cpi = blockIdx.z;
  const cf *X = a.x + (int64_t)cpi * a.cpiStride;
  const cf *Y = a.y + (int64_t)cpi * a.cpiStride;
  cf tw1[15], tw3[16];
  W::load_twiddles(t, a.tw, tw1, tw3);

  cf acc[16];
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = cmake(0.f, 0.f);
  for (int g = blockIdx.x; g < a.nSeg; g += a.nJobs) {
    const uint32_t n0 = (uint32_t)g * (uint32_t)a.segLen;
    cf v[16], yv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int m = t + T * k;
      const uint32_t n = n0 + (uint32_t)m;
      const uint32_t nc = n < a.N ? n : a.N - 1;
      const cf xv = X[xs_index(nc, a.dMinU32, a.N)];
      v[k] = (m < a.segLen && n < a.N) ? xv : cmake(0.f, 0.f);
      const uint32_t nw = n % a.N; // circular window
      yv[k] = mode == 0 ? X[xs_index(nw, a.dMinU32, a.N)] : Y[nw];
    }
    W::fwd_s1(t, v, tw1, P);
    __syncthreads();
    W::fwd_s2(t, v, P, Q);
    __syncthreads();
    W::fwd_s3(t, v, tw3, Q);
    W::fwd_s1(t, yv, tw1, P);
    __syncthreads();
    W::fwd_s2(t, yv, P, Q);
    __syncthreads();
    W::fwd_s3(t, yv, tw3, Q);
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = cmacc(acc[e], yv[e], v[e]);
    __syncthreads();
  }
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
}

// ---- reduction of the partials (fp64) + Levinson solve, one workgroup/CPI ----
struct SolveArgs {
  const cf *partial; // [nCpi][2][nJobs][nBins]
  dcx *rb;           // [nCpi][2][nBins] scratch
  cf *w;             // [nCpi][nBins]
  int32_t *ok;       // [nCpi]
  int32_t nBins, nJobs;
};

__device__ __forceinline__ dcx dmul(dcx a, dcx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ dcx dconj(dcx a) { return {a.x, -a.y}; }

__device__ dcx block_sum(dcx v, dcx *red)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v.x += __shfl_xor(v.x, off);
    v.y += __shfl_xor(v.y, off);
  }
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads(); // red[] may still be read from the previous call
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  dcx s = {0.0, 0.0};
  for (int w = 0; w < nw; w++) { s.x += red[w].x; s.y += red[w].y; }
  return s;
}

__global__ __launch_bounds__(256) void clutter_solve_kernel(SolveArgs a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = a.nBins;
  dcx *f = reinterpret_cast<dcx *>(smem); // forward vector  (T_m f = e_1)
  dcx *xv = f + n;                        // running solution
  dcx *red = xv + n;                      // [8] reduction scratch, then 2 broadcast slots
  const int cpi = blockIdx.x;
  const int t = threadIdx.x, nt = blockDim.x;
  dcx *r = a.rb + (size_t)cpi * 2 * n;
  dcx *b = r + n;
  for (int k = t; k < n; k += nt) {
    dcx sr = {0.0, 0.0}, sb = {0.0, 0.0};
    const cf *pr = a.partial + ((size_t)cpi * 2 + 0) * a.nJobs * n + k;
    const cf *pb = a.partial + ((size_t)cpi * 2 + 1) * a.nJobs * n + k;
    for (int j = 0; j < a.nJobs; j++) {
      sr.x += (double)pr[(size_t)j * n].x; sr.y += (double)pr[(size_t)j * n].y;
      sb.x += (double)pb[(size_t)j * n].x; sb.y += (double)pb[(size_t)j * n].y;
    }
    r[k] = sr;
    b[k] = sb;
    f[k] = {0.0, 0.0};
    xv[k] = {0.0, 0.0};
  }
  __syncthreads();
  const double r0 = r[0].x;
  bool ok = (r0 > 0.0) && isfinite(r0); // a zero/negative diagonal is not positive definite
  if (ok && t == 0) {
    f[0] = {1.0 / r0, 0.0};
    xv[0] = {b[0].x / r0, b[0].y / r0};
  }
  __syncthreads();
  for (int m = 1; m < n && ok; m++) {
    // ef = sum_i r[m-i] f[i],  ex = sum_i r[m-i] x[i],  i < m
    dcx ef = {0.0, 0.0}, ex = {0.0, 0.0};
    for (int i = t; i < m; i += nt) {
      const dcx rr = r[m - i];
      const dcx p1 = dmul(rr, f[i]), p2 = dmul(rr, xv[i]);
      ef.x += p1.x; ef.y += p1.y;
      ex.x += p2.x; ex.y += p2.y;
    }
    ef = block_sum(ef, red);
    ex = block_sum(ex, red);
    const double denom = 1.0 - (ef.x * ef.x + ef.y * ef.y);
    if (!(denom > 0.0) || !isfinite(denom)) { ok = false; break; } // uniform: every thread sees the same sums
    // f_new[i] = (f[i] - ef*conj(f[m-i])) / denom, i = 0..m with f[m] = 0; pairs (i, m-i) are independent
    const double inv = 1.0 / denom;
    for (int i = t; 2 * i <= m; i += nt) {
      const int j = m - i;
      const dcx fi = f[i], fj = (j < m) ? f[j] : dcx{0.0, 0.0};
      const dcx ti = dmul(ef, dconj(fj)), tj = dmul(ef, dconj(fi));
      f[i] = {(fi.x - ti.x) * inv, (fi.y - ti.y) * inv};
      if (j != i) f[j] = {(fj.x - tj.x) * inv, (fj.y - tj.y) * inv};
    }
    __syncthreads();
    // x_new[i] = x[i] + (b[m] - ex) * conj(f_new[m-i]), i = 0..m with x[m] = 0
    const dcx d = {b[m].x - ex.x, b[m].y - ex.y};
    for (int i = t; i <= m; i += nt) {
      const dcx g = dconj(f[m - i]);
      const dcx p = dmul(d, g);
      xv[i] = {xv[i].x + p.x, xv[i].y + p.y};
    }
    __syncthreads();
  }
  for (int k = t; k < n; k += nt) a.w[(size_t)cpi * n + k] = ok ? cmake((float)xv[k].x, (float)xv[k].y) : cmake(0.f, 0.f);
  if (t == 0) a.ok[cpi] = ok ? 1 : 0;
}

// ---- overlap-save FIR: y_out = y - (w * xs)[0..N) ----------------------------
struct FirArgs {
  const cf *x, *y;
  cf *yout;
  int64_t cpiStride, outStride;
  uint32_t N, dMinU32;
  int32_t nBins, segLen, nSeg;
  const cf *w;       // [nCpi][nBins]
  const int32_t *ok; // [nCpi]
  const cf *tw;
  float scale;
};

template <int R3> __global__ __launch_bounds__(16 * R3, 2) void clutter_fir_kernel(FirArgs a)
{
  using W = WgFft<R3>;
  constexpr int T = W::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *P = reinterpret_cast<cf *>(smem);
  cf *Q = P + W::A_ELEMS;
  const int t = threadIdx.x;
  const int cpi = blockIdx.y;
  const cf *X = a.x + (int64_t)cpi * a.cpiStride;
  const cf *Y = a.y + (int64_t)cpi * a.cpiStride;
  cf *O = a.yout + (int64_t)cpi * a.outStride;
  const bool ok = a.ok[cpi] != 0;
  cf tw1[15], tw3[16];
  W::load_twiddles(t, a.tw, tw1, tw3);

  // spectrum of the taps (zero-padded to F), pre-scaled by 1/F, kept in registers
  cf ws[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int m = t + T * k;
    const cf wv = a.w[(size_t)cpi * a.nBins + (m < a.nBins ? m : 0)];
    ws[k] = (m < a.nBins) ? cmake(wv.x * a.scale, wv.y * a.scale) : cmake(0.f, 0.f);
  }
  W::fwd_s1(t, ws, tw1, P);
  __syncthreads();
  W::fwd_s2(t, ws, P, Q);
  __syncthreads();
  W::fwd_s3(t, ws, tw3, Q);
  __syncthreads();

  const int hist = a.nBins - 1; // samples of history in front of each block
  for (int g = blockIdx.x; g < a.nSeg; g += gridDim.x) {
    const int64_t n0 = (int64_t)g * a.segLen;
    cf v[16], yv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int m = t + T * k;
      const int64_t src = n0 - hist + m;
      const bool in = src >= 0 && src < (int64_t)a.N;
      const uint32_t sc = in ? (uint32_t)src : 0u;
      const cf xv = X[xs_index(sc, a.dMinU32, a.N)];
      v[k] = in ? xv : cmake(0.f, 0.f);
      // y of the output sample this register will end up holding
      const int64_t n = n0 + (m - hist);
      const bool outv = (m >= hist) && (m < hist + a.segLen) && (n < (int64_t)a.N);
      yv[k] = Y[outv ? n : 0];
    }
    if (ok) {
      W::fwd_s1(t, v, tw1, P);
      __syncthreads();
      W::fwd_s2(t, v, P, Q);
      __syncthreads();
      W::fwd_s3(t, v, tw3, Q);
#pragma unroll
      for (int e = 0; e < 16; e++) v[e] = cmul(v[e], ws[e]);
      __syncthreads();
      W::inv_s1(t, v, tw3, P);
      __syncthreads();
      W::inv_s2(t, v, P, Q);
      __syncthreads();
      W::inv_s3(t, v, tw1, Q);
      __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const int m = t + T * c;
      const int64_t n = n0 + (m - hist);
      if (m >= hist && m < hist + a.segLen && n < (int64_t)a.N)
        O[n] = ok ? csub(yv[c], v[c]) : yv[c]; // not PD: surveillance channel passes through
    }
  }
}

thread_local std::string g_cerr;
int cfail(int code, const std::string &m)
{
  g_cerr = m;
  return code;
}

} // namespace

// The error string is shared with capi.hip through this hook.
extern "C" void blah2hip_set_error_(const char *msg);

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
  int F = 2048, segLen = 0, nSeg = 0, nJobs = 0, firGrid = 0;
  hipStream_t stream = nullptr;
  cf *d_tw = nullptr;
  cf *d_partial = nullptr;
  dcx *d_rb = nullptr;
  cf *d_w = nullptr;
  int32_t *d_ok = nullptr;
  cf *d_stage = nullptr; // host entry points: x, y, y_out planes
  size_t stageElems = 0;
};

namespace {

template <int R3> int launch_clutter(blah2hip_clutter_s *h, const cf *x, const cf *y, uint32_t nCpi,
                                     int64_t stride, cf *yout, int64_t outStride, int32_t *ok, hipStream_t st)
{
  using W = WgFft<R3>;
  const size_t lds = (size_t)(W::A_ELEMS + W::B_ELEMS) * sizeof(cf);
  static thread_local bool configured = false;
  if (!configured) {
    CHIP(hipFuncSetAttribute((const void *)clutter_corr_kernel<R3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHIP(hipFuncSetAttribute((const void *)clutter_fir_kernel<R3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHIP(hipFuncSetAttribute((const void *)clutter_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    configured = true;
  }
  const uint32_t dMinU32 = (uint32_t)h->delayMin;
  CorrArgs ca;
  ca.x = x; ca.y = y; ca.cpiStride = stride; ca.N = h->N; ca.dMinU32 = dMinU32;
  ca.nBins = h->nBins; ca.segLen = h->segLen; ca.nSeg = h->nSeg; ca.nJobs = h->nJobs;
  ca.tw = h->d_tw; ca.partial = h->d_partial; ca.scale = 1.0f / (float)h->F;
  hipLaunchKernelGGL(clutter_corr_kernel<R3>, dim3(h->nJobs, 2, nCpi), dim3(W::T), lds, st, ca);
  CHIP(hipGetLastError());

  SolveArgs sa;
  sa.partial = h->d_partial; sa.rb = h->d_rb; sa.w = h->d_w; sa.ok = ok; sa.nBins = h->nBins; sa.nJobs = h->nJobs;
  const size_t sl = ((size_t)2 * h->nBins + 16) * sizeof(dcx);
  hipLaunchKernelGGL(clutter_solve_kernel, dim3(nCpi), dim3(256), sl, st, sa);
  CHIP(hipGetLastError());

  FirArgs fa;
  fa.x = x; fa.y = y; fa.yout = yout; fa.cpiStride = stride; fa.outStride = outStride; fa.N = h->N; fa.dMinU32 = dMinU32;
  fa.nBins = h->nBins; fa.segLen = h->segLen; fa.nSeg = h->nSeg; fa.w = h->d_w; fa.ok = ok; fa.tw = h->d_tw;
  fa.scale = 1.0f / (float)h->F;
  hipLaunchKernelGGL(clutter_fir_kernel<R3>, dim3(h->firGrid, nCpi), dim3(W::T), lds, st, fa);
  CHIP(hipGetLastError());
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
  const int nBins = delay_max - delay_min;
  if (max_batch == 0) max_batch = 1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    CFAIL(BLAH2HIP_ERR_NO_DEVICE, "no HIP device visible (the HIP path is the only path)");
  if (device < 0 || device >= ndev) CFAIL(BLAH2HIP_ERR_INVALID, "device index out of range");
  CHIP(hipSetDevice(device));
  // transform length: F - nBins + 1 useful samples per F log F work
  int bestR3 = 0;
  double best = 1e300;
  int forced = 0;
  if (const char *e = std::getenv("BLAH2HIP_CLUTTER_FFT_LEN")) forced = std::atoi(e);
  for (int r3 : {4, 8, 16}) {
    const int F = 256 * r3;
    if (forced && F != forced) continue;
    const int L = F - nBins + 1;
    if (L < 16) continue;
    const double cost = (double)F * std::log2((double)F) / (double)L;
    if (cost < best) { best = cost; bestR3 = r3; }
  }
  if (!bestR3) CFAIL(BLAH2HIP_ERR_UNSUPPORTED, "nBins too large for the on-chip transform lengths (<= 4096)");
  // the solve keeps two fp64 vectors of nBins in LDS
  if (((size_t)2 * nBins + 16) * sizeof(dcx) > 160 * 1024 - 64)
    CFAIL(BLAH2HIP_ERR_UNSUPPORTED, "nBins too large for the on-chip Toeplitz solve");
  auto *h = new blah2hip_clutter_s;
  h->device = device;
  h->delayMin = delay_min; h->delayMax = delay_max;
  h->N = n_samples; h->maxBatch = max_batch; h->nBins = nBins;
  h->r3 = bestR3; h->F = 256 * bestR3;
  h->segLen = h->F - nBins + 1;
  h->nSeg = (int)((n_samples + (uint32_t)h->segLen - 1) / (uint32_t)h->segLen);
  hipDeviceProp_t prop;
  CHIP(hipGetDeviceProperties(&prop, device));
  h->nJobs = std::min(h->nSeg, 4 * prop.multiProcessorCount);
  h->firGrid = std::min(h->nSeg, 8 * prop.multiProcessorCount);
  CHIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  std::vector<cf> tw(h->F);
  for (int k = 0; k < h->F; k++) {
    const double a = -2.0 * M_PI * (double)k / (double)h->F;
    tw[k] = cmake((float)std::cos(a), (float)std::sin(a));
  }
  CHIP(hipMalloc(&h->d_tw, h->F * sizeof(cf)));
  CHIP(hipMemcpy(h->d_tw, tw.data(), h->F * sizeof(cf), hipMemcpyHostToDevice));
  CHIP(hipMalloc(&h->d_partial, (size_t)max_batch * 2 * h->nJobs * nBins * sizeof(cf)));
  CHIP(hipMalloc(&h->d_rb, (size_t)max_batch * 2 * nBins * sizeof(dcx)));
  CHIP(hipMalloc(&h->d_w, (size_t)max_batch * nBins * sizeof(cf)));
  CHIP(hipMalloc(&h->d_ok, max_batch * sizeof(int32_t)));
  *out = h;
  return BLAH2HIP_OK;
}

int blah2hip_clutter_destroy(blah2hip_clutter_t h)
{
  if (!h) return BLAH2HIP_OK;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (void *p : {(void *)h->d_tw, (void *)h->d_partial, (void *)h->d_rb, (void *)h->d_w, (void *)h->d_ok,
                  (void *)h->d_stage})
    if (p) hipFree(p);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return BLAH2HIP_OK;
}

int blah2hip_clutter_process_dev(blah2hip_clutter_t h, const void *d_x, const void *d_y, uint32_t n_cpi,
                                 uint64_t cpi_stride, void *d_y_out, int32_t *d_ok, void *stream)
{
  if (!h || !d_x || !d_y || !d_y_out) CFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n_cpi == 0 || n_cpi > h->maxBatch) CFAIL(BLAH2HIP_ERR_INVALID, "n_cpi outside [1, max_batch]");
  if (n_cpi > 1 && cpi_stride < h->N) CFAIL(BLAH2HIP_ERR_INVALID, "cpi_stride < nSamples");
  CHIP(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  int32_t *ok = d_ok ? d_ok : h->d_ok;
  // in-place operation is safe: every output sample is read (as y) by the one
  // thread that writes it, and x is never written
  switch (h->r3) {
  case 4: return launch_clutter<4>(h, (const cf *)d_x, (const cf *)d_y, n_cpi, (int64_t)cpi_stride, (cf *)d_y_out, (int64_t)cpi_stride, ok, st);
  case 8: return launch_clutter<8>(h, (const cf *)d_x, (const cf *)d_y, n_cpi, (int64_t)cpi_stride, (cf *)d_y_out, (int64_t)cpi_stride, ok, st);
  default: return launch_clutter<16>(h, (const cf *)d_x, (const cf *)d_y, n_cpi, (int64_t)cpi_stride, (cf *)d_y_out, (int64_t)cpi_stride, ok, st);
  }
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
