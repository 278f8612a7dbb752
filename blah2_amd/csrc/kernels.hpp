// HIP kernels of the blah2 cross-ambiguity engine (gfx950 / MI355X only).
//
//   range_kernel        Hot loop A  Ambiguity.cpp:106-149  (segmented on-chip FFT correlation)
//   doppler_dft_kernel  Hot loop B  Ambiguity.cpp:152-169  (+ partial sums of Map::set_metrics)
//   metrics_kernel      Map::set_metrics                    Map.cpp:187-206
//   cfar1d_kernel       CfarDetector1D::process             CfarDetector1D.cpp:23-100
//   rotate_kernel       Doppler-centre shift                Ambiguity.cpp:95-102
//
// All paths are relative to /root/reference/src.
#pragma once

#include <hip/hip_runtime.h>

#include "blah2hip.h"
#include "range_core.hpp"

namespace blah2 {

// --------------------------------------------------------------------------
// Range kernel.  One workgroup of T = 16*R3 threads per pulse (grid-stride over
// the nCpi*nDoppler pulses of the batch), 16 points per thread in registers,
// transform length F = 256*R3.  Per pulse: nSeg x { FFT(x segment), FFT(y
// window), acc += Y*conj(X) } then one inverse FFT and a coalesced store of the
// nDelay wanted lags.  Every input sample is read from HBM once (the y windows
// of neighbouring segments overlap by nDelay-1 samples, served by L2).
//
// HBM traffic per pulse: 2*nCorr*8 B in (C32) or nCorr*8 B in (I16), nDelay*8 B out.
// LDS: A and B exchange buffers, (16*PA + 16*PB)*8 B  (19 KB / 36 KB / 70 KB for
// F = 1024 / 2048 / 4096).
struct RangeArgs {
  RangePlan plan;
  const cf *tw;        // exp(-2 pi i k / F), k in [0, F)
  cf *out;             // [nCpi*nDoppler][nDelay]
  int64_t cpiStride;   // samples between consecutive CPIs of the batch
  int32_t nPulses;     // nCpi * nDoppler
};

template <int R3, class In>
__global__ __launch_bounds__(16 * R3, 2) void range_kernel(RangeArgs a, In in)
{
  using W = WgFft<R3>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *A = reinterpret_cast<cf *>(smem);
  cf *B = A + W::A_ELEMS;
  const int t = threadIdx.x;

  cf tw1[15], tw3[16];
  W::load_twiddles(t, a.tw, tw1, tw3);

  const RangePlan p = a.plan;
  for (int pulse = blockIdx.x; pulse < a.nPulses; pulse += gridDim.x) {
    const int cpi = pulse / p.nDoppler;
    const int i = pulse - cpi * p.nDoppler;
    const int64_t base = (int64_t)cpi * a.cpiStride + (int64_t)i * p.nCorr;

    cf v[16], xs[16], acc[16];
    for (int s = 0; s < p.nSeg; s++) {
      load_seg_x<R3>(in, p, base, s, t, v);
      W::fwd_s1(t, v, tw1, A);
      __syncthreads();
      W::fwd_s2(t, v, A, B);
      __syncthreads();
      W::fwd_s3(t, v, tw3, B);
#pragma unroll
      for (int e = 0; e < 16; e++) xs[e] = v[e];

      load_seg_y<R3>(in, p, base, s, t, v);
      W::fwd_s1(t, v, tw1, A);
      __syncthreads();
      W::fwd_s2(t, v, A, B);
      __syncthreads();
      W::fwd_s3(t, v, tw3, B);
      if (s == 0) {
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = cmulc(v[e], xs[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = cmacc(acc[e], v[e], xs[e]);
      }
    }
    __syncthreads(); // B is still being read by fwd_s3 of slower waves
    W::inv_s1(t, acc, tw3, B);
    __syncthreads();
    W::inv_s2(t, acc, B, A);
    __syncthreads();
    W::inv_s3(t, acc, tw1, A);
    store_lags<R3>(a.out, p, pulse, t, acc);
    __syncthreads(); // A is rewritten by the next pulse's fwd_s1
  }
}

// --------------------------------------------------------------------------
// Doppler-centre shift, Ambiguity.cpp:95-102:  x[i] *= exp(+j 2 pi fMid i / fs),
// i = index inside the CPI buffer.  fMid = m2/2 with m2 = dopplerMin+dopplerMax
// an integer, so the phase is (m2*i mod 2fs)/(2fs) turns exactly; evaluated in
// fp64 and rounded once.  Writes complex fp32 planes (also converts int16).
template <class In>
__global__ void rotate_kernel(In in, cf *xo, cf *yo, int64_t cpiStride, int64_t outStride,
                              uint32_t n, int32_t m2, uint32_t fs)
{
  const int cpi = blockIdx.y;
  const int64_t twofs = 2 * (int64_t)fs;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int64_t rem = ((int64_t)m2 * (int64_t)i) % twofs;
    if (rem < 0) rem += twofs;
    double s, c;
    sincospi(2.0 * (double)rem / (double)twofs, &s, &c);
    const cf xv = in.lx(cpi * cpiStride + i);
    const double re = (double)xv.x * c - (double)xv.y * s;
    const double im = (double)xv.x * s + (double)xv.y * c;
    xo[cpi * outStride + i] = cmake((float)re, (float)im);
    if (yo) yo[cpi * outStride + i] = in.ly(cpi * cpiStride + i);
  }
}

// --------------------------------------------------------------------------
// Doppler kernel v0: direct DFT down the pulse axis (works for any nDoppler,
// which is always odd in the reference, Ambiguity.cpp:26-36).
//   D[k][j] = sum_i R[i][j] * exp(-2 pi i * i*k / nD)
//   M[o][j] = D[(o + nD/2 + 1) % nD][j]                          (Ambiguity.cpp:165)
// lane <-> delay column j (coalesced), each thread accumulates KPT output rows,
// the 4 waves of a workgroup split the pulse axis and reduce through LDS.
// The roots come from an exact table W[k] = exp(-2 pi i k/nD) (fp64 -> fp32),
// indexed by (i*k mod nD), which is wave-uniform -> scalar loads.
// Epilogue: writes the map tile and one (sum of 10 log10|z|, max) partial per
// workgroup for Map::set_metrics.
constexpr int DOP_KPT = 8;
constexpr int DOP_WAVES = 4;

struct DopplerArgs {
  const cf *R;      // [nCpi][nD][nDelay]
  cf *map;          // [nCpi][nD][nDelay]
  const cf *W;      // [nD]
  double *partSum;  // [nCpi][nTilesPerCpi]
  float *partMax;   // [nCpi][nTilesPerCpi]
  int32_t nD, nDelay;
};

__global__ __launch_bounds__(64 * DOP_WAVES) void doppler_dft_kernel(DopplerArgs a)
{
  __shared__ cf red[DOP_WAVES][DOP_KPT][64];
  __shared__ double wsum[DOP_WAVES];
  __shared__ float wmax[DOP_WAVES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nD = a.nD, nDelay = a.nDelay;
  const int j = blockIdx.x * 64 + lane;
  const int o0 = blockIdx.y * DOP_KPT;
  const int cpi = blockIdx.z;
  const cf *R = a.R + (size_t)cpi * nD * nDelay;
  const bool jok = j < nDelay;
  const int chunk = (nD + DOP_WAVES - 1) / DOP_WAVES;
  const int i0 = wave * chunk;
  const int i1 = min(nD, i0 + chunk);

  int src[DOP_KPT], idx[DOP_KPT];
  cf acc[DOP_KPT];
#pragma unroll
  for (int kk = 0; kk < DOP_KPT; kk++) {
    int o = o0 + kk;
    if (o >= nD) o = nD - 1; // clamped duplicate, masked at the store
    src[kk] = (o + nD / 2 + 1) % nD;
    idx[kk] = (int)(((int64_t)src[kk] * i0) % nD);
    acc[kk] = cmake(0.f, 0.f);
  }
  for (int i = i0; i < i1; i++) {
    const cf r = jok ? R[(size_t)i * nDelay + j] : cmake(0.f, 0.f);
#pragma unroll
    for (int kk = 0; kk < DOP_KPT; kk++) {
      const cf w = a.W[idx[kk]];
      acc[kk].x += r.x * w.x - r.y * w.y;
      acc[kk].y += r.x * w.y + r.y * w.x;
      idx[kk] += src[kk];
      if (idx[kk] >= nD) idx[kk] -= nD;
    }
  }
#pragma unroll
  for (int kk = 0; kk < DOP_KPT; kk++) red[wave][kk][lane] = acc[kk];
  __syncthreads();
  // wave w finalises output rows kk = 2w, 2w+1
  double lsum = 0.0;
  float lmax = 0.f; // Map.cpp:193: the running max starts at 0
#pragma unroll
  for (int h = 0; h < DOP_KPT / DOP_WAVES; h++) {
    const int kk = wave * (DOP_KPT / DOP_WAVES) + h;
    cf s = red[0][kk][lane];
#pragma unroll
    for (int w = 1; w < DOP_WAVES; w++) s = cadd(s, red[w][kk][lane]);
    const int o = o0 + kk;
    if (jok && o < nD) {
      a.map[(size_t)cpi * nD * nDelay + (size_t)o * nDelay + j] = s;
      // 10*log10|z| = 5*log10(re^2+im^2)
      const float v = 1.50514997831990597607f * log2f(s.x * s.x + s.y * s.y);
      lsum += (double)v;
      lmax = fmaxf(lmax, v);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lsum += __shfl_xor(lsum, off);
    lmax = fmaxf(lmax, __shfl_xor(lmax, off));
  }
  if (lane == 0) { wsum[wave] = lsum; wmax[wave] = lmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    float m = 0.f;
    for (int w = 0; w < DOP_WAVES; w++) { s += wsum[w]; m = fmaxf(m, wmax[w]); }
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    const int nTiles = gridDim.x * gridDim.y;
    a.partSum[(size_t)cpi * nTiles + tile] = s;
    a.partMax[(size_t)cpi * nTiles + tile] = m;
  }
}

// Map::set_metrics (Map.cpp:187-206): noisePower = mean(10 log10|z|),
// maxPower = max(0, max 10 log10|z|) - noisePower.  One workgroup per CPI sums
// the per-tile partials in a fixed order (deterministic).
__global__ void metrics_kernel(const double *partSum, const float *partMax, int nTiles,
                               double cells, double *metrics)
{
  __shared__ double ssum[256];
  __shared__ float smax[256];
  const int cpi = blockIdx.x;
  double s = 0.0;
  float m = 0.f;
  for (int i = threadIdx.x; i < nTiles; i += blockDim.x) {
    s += partSum[(size_t)cpi * nTiles + i];
    m = fmaxf(m, partMax[(size_t)cpi * nTiles + i]);
  }
  ssum[threadIdx.x] = s;
  smax[threadIdx.x] = m;
  __syncthreads();
  for (int off = blockDim.x / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      ssum[threadIdx.x] += ssum[threadIdx.x + off];
      smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double noise = ssum[0] / cells;
    metrics[2 * cpi + 0] = noise;
    metrics[2 * cpi + 1] = (double)smax[0] - noise;
  }
}

// --------------------------------------------------------------------------
// CfarDetector1D::process (CfarDetector1D.cpp:23-100): cell-averaging CFAR
// along delay for each Doppler row with |doppler| >= minDoppler.  One
// workgroup per row; |z|^2 of the row is staged in LDS as fp64 and the window
// sum runs in the reference's index order (leading cells need k > 0, trailing
// k >= 0, :61,:68).  alpha[n] = n*(pfa^(-1/n)-1) is tabulated on the host with
// the same libm pow the reference calls (:76).  Hits are appended through a
// per-CPI atomic counter; the host API sorts them into row-major order.
struct CfarArgs {
  const cf *map;         // [nCpi][nD][nDelay]
  const double *metrics; // [nCpi][2]
  const double *doppler; // [nD] Hz
  const double *alpha;   // [2*nTrain+1]
  blah2hip_hit_t *hits;  // [nCpi][cap]
  uint32_t *count;       // [nCpi]
  int32_t nD, nDelay, delayMin;
  int32_t nGuard, nTrain, minDelay;
  double minDoppler;
  uint32_t cap;
};

__global__ void cfar1d_kernel(CfarArgs a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sq = reinterpret_cast<double *>(smem);
  const int row = blockIdx.x, cpi = blockIdx.y;
  if (fabs(a.doppler[row]) < a.minDoppler) return; // :40
  const cf *z = a.map + ((size_t)cpi * a.nD + row) * a.nDelay;
  for (int j = threadIdx.x; j < a.nDelay; j += blockDim.x) {
    const cf c = z[j];
    sq[j] = (double)c.x * (double)c.x + (double)c.y * (double)c.y; // |z*z| (:47)
  }
  __syncthreads();
  const double noisePower = a.metrics[2 * cpi];
  for (int j = threadIdx.x; j < a.nDelay; j += blockDim.x) {
    if (j + a.delayMin < a.minDelay) continue; // :53  x->delay[j] < minDelay
    int n = 0;
    double tot = 0.0;
    for (int k = j - a.nGuard - a.nTrain; k < j - a.nGuard; k++)
      if (k > 0 && k < a.nDelay) { tot += sq[k]; n++; }
    for (int k = j + a.nGuard + 1; k < j + a.nGuard + a.nTrain + 1; k++)
      if (k >= 0 && k < a.nDelay) { tot += sq[k]; n++; }
    if (n == 0) continue; // alpha = 0*inf = NaN in the reference: never exceeds
    const double thr = a.alpha[n] * (tot / n);
    if (sq[j] > thr) {
      const uint32_t slot = atomicAdd(&a.count[cpi], 1u);
      if (slot < a.cap) {
        blah2hip_hit_t h;
        h.row = row;
        h.col = j;
        h.snr = 5.0 * log10(sq[j]) - noisePower; // 10 log10|z| - noisePower (:48)
        a.hits[(size_t)cpi * a.cap + slot] = h;
      }
    }
  }
}

} // namespace blah2
