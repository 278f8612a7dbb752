// SpectrumAnalyser on gfx950 (include/blah2hip.h, spectrum section).
//
// Reference: /root/reference/src/process/spectrum/SpectrumAnalyser.cpp:9-71,
// called once per CPI on the reference channel (blah2.cpp:264) before the
// clutter filter.  With D = decimation = uint32(n / bandwidth) (:16),
// nS = nSpectrum = n / D (:17) and N = nfft = nS * D (:18):
//   X = FFT_N(x[0..N))                                   (:33-40)
//   spectrum[k] = X[(k*D + N/2 + 1) mod N],  k in [0,nS)  (:43-54, shift then every D-th bin)
// The frequency axis loop `for (i = -nSpectrum/2; i < nSpectrum/2; i++)` runs on a
// uint32_t (:64): -nSpectrum/2 is (2^32 - nS)/2, never below nS/2, so the axis is
// always EMPTY; the host class reproduces that.
//
// Only nS of the N bins are kept, all congruent to c = (N/2 + 1) mod D modulo D, so
// the N-point transform is never formed.  With h = N/2 + 1 = hq*D + c and
// n = p + nS*j (p < nS, j < D):
//   X[m*D + c] = sum_p W_nS^(p*m) * u[p],
//   u[p]       = W_N^(p*c) * sum_j x[p + nS*j] * W_D^(j*c)
//   spectrum[k] = X[((k + hq) mod nS)*D + c]
// i.e. one pass over the samples (spectrum_fold_kernel, HBM-bound: N*8 bytes
// read once, coalesced along p), a small reduction (spectrum_reduce_kernel) and an
// nS-point DFT of the folded sequence (spectrum_dft_kernel; nS ~ 2000, direct
// evaluation from an exact root table).
// Everything after the sample load is fp64: the result is the reference's up to
// fp64 rounding of a differently-ordered sum (the IQ samples are int16-valued, so
// their fp32 storage is exact).
#include <hip/hip_runtime.h>

#include "blah2hip.h"
#include "range_core.hpp"

#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

using namespace blah2;

namespace {

struct dcx { double x, y; };
__device__ __forceinline__ dcx dmake(double a, double b) { dcx r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ dcx dmul(dcx a, dcx b) { return dmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

struct SpecArgs {
  const dcx *wD;    // exp(-2 pi i m / D), m in [0, D)
  const dcx *wS;    // exp(-2 pi i m / nS), m in [0, nS)
  dcx *part;        // [nCpi][nJ][nS] partial folds
  dcx *u;           // [nCpi][nS] folded sequence
  dcx *out;         // [nCpi][nS]
  int64_t cpiStride;
  uint32_t D, nS, c, hq, nJ;
  uint64_t N;
};

constexpr int FOLD_WAVES = 4;

// grid (ceil(nS/64), nJ, nCpi), 256 threads: lane <-> p (coalesced), the j range
// of this block's chunk is split over the 4 waves.
template <class In>
__global__ __launch_bounds__(64 * FOLD_WAVES) void spectrum_fold_kernel(SpecArgs a, In in)
{
  __shared__ dcx red[FOLD_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t p = blockIdx.x * 64 + lane;
  const uint32_t cpi = blockIdx.z;
  const uint32_t per = (a.D + a.nJ - 1) / a.nJ;
  const uint32_t j0 = blockIdx.y * per;
  const uint32_t j1 = min(a.D, j0 + per);
  const bool pok = p < a.nS;
  const int64_t base = (int64_t)cpi * a.cpiStride + (pok ? p : 0);
  dcx acc = dmake(0.0, 0.0);
  // (j*c) mod D advances by c per j and by FOLD_WAVES*c per step of this wave
  uint32_t idx = (uint32_t)(((uint64_t)(j0 + wave) * a.c) % a.D);
  const uint32_t step = (uint32_t)(((uint64_t)FOLD_WAVES * a.c) % a.D);
  for (uint32_t j = j0 + wave; j < j1; j += FOLD_WAVES) {
    const cf s = in.lx(base + (int64_t)a.nS * j);
    const dcx w = a.wD[idx];
    acc.x += (double)s.x * w.x - (double)s.y * w.y;
    acc.y += (double)s.x * w.y + (double)s.y * w.x;
    idx += step;
    if (idx >= a.D) idx -= a.D;
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && pok) {
    dcx s = red[0][lane];
#pragma unroll
    for (int w = 1; w < FOLD_WAVES; w++) { s.x += red[w][lane].x; s.y += red[w][lane].y; }
    a.part[((size_t)cpi * a.nJ + blockIdx.y) * a.nS + p] = s;
  }
}

// grid (ceil(nS/256), nCpi): u[p] = W_N^(p*c) * (chunk sums in index order); the
// phase is reduced exactly in integers and evaluated in fp64.
__global__ __launch_bounds__(256) void spectrum_reduce_kernel(SpecArgs a)
{
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t cpi = blockIdx.y;
  if (p >= a.nS) return;
  const dcx *part = a.part + (size_t)cpi * a.nJ * a.nS;
  dcx s = dmake(0.0, 0.0);
  for (uint32_t q = 0; q < a.nJ; q++) { const dcx v = part[(size_t)q * a.nS + p]; s.x += v.x; s.y += v.y; }
  const uint64_t ph = ((uint64_t)p * a.c) % a.N;
  double sn, cs;
  sincospi(-2.0 * (double)ph / (double)a.N, &sn, &cs);
  a.u[(size_t)cpi * a.nS + p] = dmul(s, dmake(cs, sn));
}

// nS-point DFT of u, direct from the exact root table: grid (ceil(nS/32), nCpi),
// 256 threads = 32 output bins x 8 slices of the input index (p = slice, slice+8, ...),
// u and the roots staged in LDS (2*nS*16 bytes), slices folded with DPP-free
// shuffles at the end.  2000^2 complex fp64 MACs per CPI: ~1 us of fp64 vector rate,
// the point of the layout is to keep the serial chain per thread at nS/8 steps.
constexpr int DFT_SLICES = 8;
constexpr int DFT_BINS = 32;

// STAGED = false (nS > 4096, two tables no longer fit in LDS): the same sums straight from the L2-resident tables.
template <bool STAGED>
__global__ __launch_bounds__(DFT_SLICES * DFT_BINS) void spectrum_dft_kernel(SpecArgs a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t cpi = blockIdx.y;
  const dcx *u = a.u + (size_t)cpi * a.nS;
  const dcx *w = a.wS;
  if (STAGED) {
    dcx *ul = reinterpret_cast<dcx *>(smem);
    dcx *wl = ul + a.nS;
    for (uint32_t p = threadIdx.x; p < a.nS; p += blockDim.x) {
      ul[p] = u[p];
      wl[p] = w[p];
    }
    __syncthreads();
    u = ul;
    w = wl;
  }
  // lanes of a wave: 8 bins x 8 slices, slices in the low bits so that the fold is
  // three xor-shuffles inside a wave
  const uint32_t slice = threadIdx.x & (DFT_SLICES - 1);
  const uint32_t m = blockIdx.x * DFT_BINS + (threadIdx.x >> 3);
  const uint32_t mm = m < a.nS ? m : 0;
  dcx acc = dmake(0.0, 0.0);
  uint32_t idx = (uint32_t)(((uint64_t)slice * mm) % a.nS);
  const uint32_t step = (uint32_t)(((uint64_t)DFT_SLICES * mm) % a.nS);
#pragma unroll 4
  for (uint32_t p = slice; p < a.nS; p += DFT_SLICES) {
    const dcx t = dmul(u[p], w[idx]);
    acc.x += t.x;
    acc.y += t.y;
    idx += step;
    if (idx >= a.nS) idx -= a.nS;
  }
#pragma unroll
  for (int off = 1; off < DFT_SLICES; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off);
    acc.y += __shfl_xor(acc.y, off);
  }
  if (slice == 0 && m < a.nS) {
    // spectrum[k] = Xd[(k + hq) mod nS]  <=>  k = (m - hq) mod nS
    const uint32_t hqm = a.hq % a.nS;
    const uint32_t k = m >= hqm ? m - hqm : m + a.nS - hqm;
    a.out[(size_t)cpi * a.nS + k] = acc;
  }
}

// ---- nS > 65536: the nS-point DFT of the folded sequence as a chirp-z (Bluestein) product on a power-of-two fp64 transform ----
// The direct evaluation above is quadratic in nS (1.7e10 complex multiply-adds at the old cap of 65536; the reference
// accepts any bandwidth, SpectrumAnalyser.cpp:9-30).  With ch[p] = exp(-i pi p^2 / nS) (the angle reduced exactly:
// p^2 mod 2 nS in integers):  X[m] = ch[m] sum_p (u[p] ch[p]) conj(ch[m - p])  -- a convolution, run as three M-point
// transforms (M = 2^k >= 2 nS - 1; the kernel spectrum is formed once per handle).  A radix-2 autosort (Stockham) pass per
// launch between two global buffers: log2 M launches of M / 2 butterflies, every twiddle from one exact table.  This is an
// envelope path (a bandwidth above 65 kHz of spectrum bins), written for correctness in fp64, not for speed: 20-odd
// launches of a few MB each.
struct BluArgs {
  const dcx *u;      // [nCpi][nS]
  const dcx *chirp;  // [nS]
  const dcx *kspec;  // [M]: transform of the wrapped conj chirp, times 1 / M
  const dcx *tw;     // [M / 2]: exp(-2 pi i k / M)
  dcx *a, *b;        // [nCpi][M] ping-pong
  dcx *out;          // [nCpi][nS]
  uint32_t nS, M, hq;
};

// a[p] = u[p] ch[p] (p < nS), 0 beyond
__global__ void blu_load_kernel(BluArgs g)
{
  const uint32_t cpi = blockIdx.y;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < g.M; p += gridDim.x * blockDim.x)
    g.a[(size_t)cpi * g.M + p] = p < g.nS ? dmul(g.u[(size_t)cpi * g.nS + p], g.chirp[p]) : dmake(0.0, 0.0);
}

// one Stockham radix-2 pass, src -> dst: butterflies of span `half` (1, 2, 4, ...), INV: conjugate twiddles
template <bool INV>
__global__ void blu_pass_kernel(const dcx *src, dcx *dst, const dcx *tw, uint32_t M, uint32_t half)
{
  const uint32_t cpi = blockIdx.y;
  src += (size_t)cpi * M;
  dst += (size_t)cpi * M;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < M / 2; i += gridDim.x * blockDim.x) {
    // Stockham: x_out[q + 2 half p'] ... with j = i / half (block), k = i % half
    const uint32_t k = i & (half - 1), j = i - k; // j = block * half
    const dcx a0 = src[i], a1 = src[i + M / 2];
    dcx w = tw[k * (M / 2 / half)];
    if (INV) w.y = -w.y;
    const dcx t = dmul(a1, w);
    dst[2 * j + k] = dmake(a0.x + t.x, a0.y + t.y);
    dst[2 * j + k + half] = dmake(a0.x - t.x, a0.y - t.y);
  }
}

__global__ void blu_mul_kernel(dcx *a, const dcx *kspec, uint32_t M)
{
  const uint32_t cpi = blockIdx.y;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < M; p += gridDim.x * blockDim.x)
    a[(size_t)cpi * M + p] = dmul(a[(size_t)cpi * M + p], kspec[p]);
}

// spectrum[k] = Xd[(k + hq) mod nS],  Xd[m] = conv[m] ch[m]
__global__ void blu_store_kernel(BluArgs g, const dcx *conv)
{
  const uint32_t cpi = blockIdx.y;
  const uint32_t hqm = g.hq % g.nS;
  for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < g.nS; m += gridDim.x * blockDim.x) {
    const uint32_t k = m >= hqm ? m - hqm : m + g.nS - hqm;
    g.out[(size_t)cpi * g.nS + k] = dmul(conv[(size_t)cpi * g.M + m], g.chirp[m]);
  }
}

} // namespace

extern "C" void blah2hip_set_error_(const char *msg);
extern "C" hipError_t blah2hip_ensure_lds_(const void *kern, int bytes);

#define SHIP(expr)                                                                        \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      blah2hip_set_error_((std::string(#expr) + ": " + hipGetErrorString(e_)).c_str());   \
      return BLAH2HIP_ERR_HIP;                                                            \
    }                                                                                     \
  } while (0)
#define SFAIL(code, msg)            \
  do {                              \
    blah2hip_set_error_(msg);       \
    return code;                    \
  } while (0)

struct blah2hip_spectrum_s {
  int device = 0;
  uint32_t n = 0, D = 0, nS = 0, c = 0, hq = 0, nJ = 0, maxBatch = 1;
  uint64_t N = 0;
  double bandwidth = 0;
  hipStream_t stream = nullptr;
  dcx *d_wD = nullptr, *d_wS = nullptr, *d_part = nullptr, *d_u = nullptr, *d_out = nullptr;
  cf *d_stage = nullptr;
  // nS > DIRECT_MAX: the chirp-z path
  uint32_t M = 0;
  dcx *d_chirp = nullptr, *d_kspec = nullptr, *d_btw = nullptr, *d_ba = nullptr, *d_bb = nullptr;
};
constexpr uint32_t DIRECT_MAX = 65536; // bins evaluated directly (quadratic); beyond: chirp-z on a power-of-two transform

// the M-point forward (INV = false) or unnormalised inverse transform of every CPI's row of `a` (ping-pong with `b`);
// returns the buffer that holds the result
template <bool INV> dcx *blu_transform(blah2hip_spectrum_s *h, dcx *a, dcx *b, uint32_t n_cpi, hipStream_t st)
{
  const dim3 grid(std::min<uint32_t>((h->M / 2 + 255) / 256, 4096), n_cpi);
  for (uint32_t half = 1; half < h->M; half <<= 1) {
    hipLaunchKernelGGL(blu_pass_kernel<INV>, grid, dim3(256), 0, st, a, b, h->d_btw, h->M, half);
    std::swap(a, b);
  }
  return a;
}

extern "C" {

int blah2hip_spectrum_create(uint32_t n_samples, double bandwidth, int device, uint32_t max_batch, blah2hip_spectrum_t *out)
{
  if (!out) SFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (!(bandwidth > 0) || max_batch == 0) SFAIL(BLAH2HIP_ERR_INVALID, "bandwidth and max_batch must be positive");
  const uint32_t D = (uint32_t)((double)n_samples / bandwidth); // SpectrumAnalyser.cpp:16
  if (D == 0) SFAIL(BLAH2HIP_ERR_INVALID, "nSamples < bandwidth: the reference divides by a zero decimation here");
  const uint32_t nS = n_samples / D;                             // :17
  // the kept bins come from a direct nS-point DFT (nS ~ bandwidth in Hz / 1, 2000 in the reference's configs): quadratic, so
  // bounded where it would take seconds
  if (nS > (1u << 26)) SFAIL(BLAH2HIP_ERR_UNSUPPORTED, "nSpectrum > 2^26 (the chirp-z buffers: 3 x 2^28 bytes per CPI)");
  int count = 0;
  SHIP(hipGetDeviceCount(&count));
  if (device < 0 || device >= count) SFAIL(BLAH2HIP_ERR_NO_DEVICE, "no such HIP device");
  SHIP(hipSetDevice(device));
  auto *h = new blah2hip_spectrum_s;
  // everything that can fail runs inside `build`; a partially built handle is torn down by destroy()
  auto build = [&]() -> int {
  h->device = device;
  h->n = n_samples;
  h->bandwidth = bandwidth;
  h->D = D;
  h->nS = nS;
  h->N = (uint64_t)nS * D;                                       // :18
  const uint64_t hh = h->N / 2 + 1;                              // :46
  h->c = (uint32_t)(hh % D);
  h->hq = (uint32_t)(hh / D);
  h->maxBatch = max_batch;
  hipDeviceProp_t prop;
  SHIP(hipGetDeviceProperties(&prop, device));
  const uint32_t bx = (nS + 63) / 64;
  h->nJ = std::max<uint32_t>(1, std::min<uint32_t>((D + FOLD_WAVES - 1) / FOLD_WAVES, (4u * prop.multiProcessorCount + bx - 1) / bx));
  SHIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  std::vector<dcx> wD(D), wS(nS);
  for (uint32_t m = 0; m < D; m++) { wD[m].x = std::cos(-2.0 * M_PI * (double)m / D); wD[m].y = std::sin(-2.0 * M_PI * (double)m / D); }
  for (uint32_t m = 0; m < nS; m++) { wS[m].x = std::cos(-2.0 * M_PI * (double)m / nS); wS[m].y = std::sin(-2.0 * M_PI * (double)m / nS); }
  if (nS > DIRECT_MAX) { // chirp-z: the chirp, the transform's root table, and the kernel spectrum (formed below, on the device)
    h->M = 1;
    while (h->M < 2 * nS - 1) h->M <<= 1;
    const uint32_t M = h->M;
    std::vector<dcx> ch(nS), btw(M / 2), kb(M);
    for (uint32_t p = 0; p < nS; p++) {
      const uint64_t q = ((uint64_t)p * p) % (2ull * nS); // exp(-i pi p^2 / nS) has period 2 nS in p^2
      ch[p].x = std::cos(-M_PI * (double)q / nS); ch[p].y = std::sin(-M_PI * (double)q / nS);
    }
    for (uint32_t k = 0; k < M / 2; k++) { btw[k].x = std::cos(-2.0 * M_PI * (double)k / M); btw[k].y = std::sin(-2.0 * M_PI * (double)k / M); }
    for (auto &v : kb) v.x = v.y = 0.0;
    for (uint32_t p = 0; p < nS; p++) { // conj chirp, wrapped: index p and M - p
      kb[p].x = ch[p].x; kb[p].y = -ch[p].y;
      if (p) { kb[M - p].x = ch[p].x; kb[M - p].y = -ch[p].y; }
    }
    SHIP(hipMalloc(&h->d_chirp, nS * sizeof(dcx)));
    SHIP(hipMalloc(&h->d_btw, (M / 2) * sizeof(dcx)));
    SHIP(hipMalloc(&h->d_kspec, (size_t)M * sizeof(dcx)));
    SHIP(hipMalloc(&h->d_ba, (size_t)max_batch * M * sizeof(dcx)));
    SHIP(hipMalloc(&h->d_bb, (size_t)max_batch * M * sizeof(dcx)));
    SHIP(hipMemcpy(h->d_chirp, ch.data(), nS * sizeof(dcx), hipMemcpyHostToDevice));
    SHIP(hipMemcpy(h->d_btw, btw.data(), (M / 2) * sizeof(dcx), hipMemcpyHostToDevice));
    SHIP(hipMemcpy(h->d_ba, kb.data(), (size_t)M * sizeof(dcx), hipMemcpyHostToDevice));
    dcx *res = blu_transform<false>(h, h->d_ba, h->d_bb, 1, h->stream);
    SHIP(hipGetLastError());
    SHIP(hipStreamSynchronize(h->stream));
    std::vector<dcx> ks(M);
    SHIP(hipMemcpy(ks.data(), res, (size_t)M * sizeof(dcx), hipMemcpyDeviceToHost));
    for (auto &v : ks) { v.x /= (double)M; v.y /= (double)M; } // the inverse below is unnormalised
    SHIP(hipMemcpy(h->d_kspec, ks.data(), (size_t)M * sizeof(dcx), hipMemcpyHostToDevice));
  }
  SHIP(hipMalloc(&h->d_wD, D * sizeof(dcx)));
  SHIP(hipMalloc(&h->d_wS, nS * sizeof(dcx)));
  SHIP(hipMalloc(&h->d_part, (size_t)max_batch * h->nJ * nS * sizeof(dcx)));
  SHIP(hipMalloc(&h->d_out, (size_t)max_batch * nS * sizeof(dcx)));
  SHIP(hipMalloc(&h->d_u, (size_t)max_batch * nS * sizeof(dcx)));
  SHIP(hipMemcpy(h->d_wD, wD.data(), D * sizeof(dcx), hipMemcpyHostToDevice));
  SHIP(hipMemcpy(h->d_wS, wS.data(), nS * sizeof(dcx), hipMemcpyHostToDevice));
  SHIP(blah2hip_ensure_lds_((const void *)spectrum_dft_kernel<true>, (int)(2 * 4096 * sizeof(dcx))));
  return BLAH2HIP_OK;
  };
  const int rc = build();
  if (rc != BLAH2HIP_OK) {
    blah2hip_spectrum_destroy(h);
    return rc;
  }
  *out = h;
  return BLAH2HIP_OK;
}

int blah2hip_spectrum_destroy(blah2hip_spectrum_t h)
{
  if (!h) return BLAH2HIP_OK;
  (void)hipSetDevice(h->device);
  (void)hipFree(h->d_wD);
  (void)hipFree(h->d_wS);
  (void)hipFree(h->d_part);
  (void)hipFree(h->d_out);
  (void)hipFree(h->d_u);
  for (dcx *p : {h->d_chirp, h->d_kspec, h->d_btw, h->d_ba, h->d_bb})
    if (p) (void)hipFree(p);
  if (h->d_stage) (void)hipFree(h->d_stage);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return BLAH2HIP_OK;
}

int blah2hip_spectrum_get_dims(blah2hip_spectrum_t h, uint32_t *decimation, uint32_t *n_spectrum, uint64_t *nfft)
{
  if (!h) SFAIL(BLAH2HIP_ERR_INVALID, "NULL handle");
  if (decimation) *decimation = h->D;
  if (n_spectrum) *n_spectrum = h->nS;
  if (nfft) *nfft = h->N;
  return BLAH2HIP_OK;
}

int blah2hip_spectrum_process_dev(blah2hip_spectrum_t h, int fmt, const void *d_x, uint32_t n_cpi, uint64_t cpi_stride,
                                  double *d_out, void *stream)
{
  if (!h || !d_x || !d_out) SFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  if (n_cpi == 0 || n_cpi > h->maxBatch) SFAIL(BLAH2HIP_ERR_INVALID, "n_cpi outside [1, max_batch]");
  if (n_cpi > 1 && cpi_stride < h->N) SFAIL(BLAH2HIP_ERR_INVALID, "cpi_stride < nfft");
  SHIP(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  SpecArgs a;
  a.wD = h->d_wD;
  a.wS = h->d_wS;
  a.part = h->d_part;
  a.u = h->d_u;
  a.out = reinterpret_cast<dcx *>(d_out);
  a.cpiStride = (int64_t)cpi_stride;
  a.D = h->D;
  a.nS = h->nS;
  a.c = h->c;
  a.hq = h->hq;
  a.nJ = h->nJ;
  a.N = h->N;
  const dim3 g1((h->nS + 63) / 64, h->nJ, n_cpi);
  switch (fmt) {
  case BLAH2HIP_FMT_C32: {
    InC32 in{(const cf *)d_x, (const cf *)d_x};
    hipLaunchKernelGGL(spectrum_fold_kernel<InC32>, g1, dim3(64 * FOLD_WAVES), 0, st, a, in);
    break;
  }
  case BLAH2HIP_FMT_I16: {
    InI16 in{(const int16_t *)d_x};
    hipLaunchKernelGGL(spectrum_fold_kernel<InI16>, g1, dim3(64 * FOLD_WAVES), 0, st, a, in);
    break;
  }
  case BLAH2HIP_FMT_F16: {
    InF16 in{(const _Float16 *)d_x, (const _Float16 *)d_x};
    hipLaunchKernelGGL(spectrum_fold_kernel<InF16>, g1, dim3(64 * FOLD_WAVES), 0, st, a, in);
    break;
  }
  default: SFAIL(BLAH2HIP_ERR_INVALID, "unknown sample format");
  }
  SHIP(hipGetLastError());
  hipLaunchKernelGGL(spectrum_reduce_kernel, dim3((h->nS + 255) / 256, n_cpi), dim3(256), 0, st, a);
  if (h->nS > DIRECT_MAX) {
    BluArgs g;
    g.u = h->d_u; g.chirp = h->d_chirp; g.kspec = h->d_kspec; g.tw = h->d_btw; g.a = h->d_ba; g.b = h->d_bb;
    g.out = reinterpret_cast<dcx *>(d_out); g.nS = h->nS; g.M = h->M; g.hq = h->hq;
    const dim3 gm(std::min<uint32_t>((h->M + 255) / 256, 4096), n_cpi);
    hipLaunchKernelGGL(blu_load_kernel, gm, dim3(256), 0, st, g);
    dcx *f = blu_transform<false>(h, h->d_ba, h->d_bb, n_cpi, st);
    dcx *other = f == h->d_ba ? h->d_bb : h->d_ba;
    hipLaunchKernelGGL(blu_mul_kernel, gm, dim3(256), 0, st, f, h->d_kspec, h->M);
    dcx *conv = blu_transform<true>(h, f, other, n_cpi, st);
    hipLaunchKernelGGL(blu_store_kernel, gm, dim3(256), 0, st, g, (const dcx *)conv);
    SHIP(hipGetLastError());
    return BLAH2HIP_OK;
  }
  const dim3 dgrid((h->nS + DFT_BINS - 1) / DFT_BINS, n_cpi);
  if (h->nS <= 4096) hipLaunchKernelGGL(spectrum_dft_kernel<true>, dgrid, dim3(DFT_SLICES * DFT_BINS), 2 * (size_t)h->nS * sizeof(dcx), st, a);
  else hipLaunchKernelGGL(spectrum_dft_kernel<false>, dgrid, dim3(DFT_SLICES * DFT_BINS), 0, st, a);
  SHIP(hipGetLastError());
  return BLAH2HIP_OK;
}

int blah2hip_spectrum_process_c32(blah2hip_spectrum_t h, const float *x, uint32_t n, double *spectrum_out)
{
  if (!h || !x || !spectrum_out) SFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  if ((uint64_t)n < h->N) SFAIL(BLAH2HIP_ERR_INVALID, "fewer than nfft samples");
  SHIP(hipSetDevice(h->device));
  if (!h->d_stage) SHIP(hipMalloc(&h->d_stage, h->N * sizeof(cf)));
  SHIP(hipMemcpyAsync(h->d_stage, x, h->N * sizeof(cf), hipMemcpyHostToDevice, h->stream));
  int rc = blah2hip_spectrum_process_dev(h, BLAH2HIP_FMT_C32, h->d_stage, 1, h->N, reinterpret_cast<double *>(h->d_out), h->stream);
  if (rc) return rc;
  SHIP(hipMemcpyAsync(spectrum_out, h->d_out, (size_t)h->nS * sizeof(dcx), hipMemcpyDeviceToHost, h->stream));
  SHIP(hipStreamSynchronize(h->stream));
  return BLAH2HIP_OK;
}

int blah2hip_spectrum_process_c64(blah2hip_spectrum_t h, const double *x, uint32_t n, double *spectrum_out)
{
  if (!h || !x || !spectrum_out) SFAIL(BLAH2HIP_ERR_INVALID, "NULL argument");
  if ((uint64_t)n < h->N) SFAIL(BLAH2HIP_ERR_INVALID, "fewer than nfft samples");
  std::vector<float> fx(2 * (size_t)h->N);
  for (size_t i = 0; i < fx.size(); i++) fx[i] = (float)x[i];
  return blah2hip_spectrum_process_c32(h, fx.data(), (uint32_t)h->N, spectrum_out);
}

} // extern "C"
