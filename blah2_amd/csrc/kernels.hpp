// HIP kernels of the blah2 cross-ambiguity engine (gfx950 / MI355X only).
//
//   range_kernel / range8_kernel   Hot loop A  Ambiguity.cpp:106-149  (segmented on-chip FFT correlation;
//                                  16 points per thread for F = 2048 / 4096, 8 for F = 1024)
//   doppler_tile_kernel            Hot loop B  Ambiguity.cpp:152-169  nD <= 513, batched launches
//   doppler_tilem_kernel                                              513 < nD <= 2049
//   doppler_fft_kernel                                                nD <= 2049, one column per workgroup
//   doppler_dft_kernel                                                direct fallback
//                                  (all with the per-workgroup partial sums of Map::set_metrics)
//   metrics_kernel                 Map::set_metrics                   Map.cpp:187-206
//   (the detector kernels live in cfar_kernels.hpp)
//   rotate_kernel                  Doppler-centre shift               Ambiguity.cpp:95-102
// The clutter filter and the spectrum analyser live in clutter.hip / spectrum.hip.
//
// All paths are relative to /root/reference/src.
#pragma once

#include <hip/hip_runtime.h>

#include "blah2hip.h"
#include "range_core.hpp"
#include "fft_wg8.hpp"
#include "fft_wave.hpp"
#include "fft_wave2.hpp"
#include "fft_wave1k.hpp"
#include "bufload.hpp"
#include "trace.hpp"

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
// 3 waves per SIMD: 138 VGPRs, no scratch; at 4 (128 VGPRs) the kernel spills 14-16 registers for the
// same speed (measured, `small` config: 1.55 M vs 1.53 M CPIs/s)
#ifndef RANGE8_WAVES_PER_SIMD
#define RANGE8_WAVES_PER_SIMD 3
#endif
struct RangeArgs {
  RangePlan plan;
  const cf *tw;        // exp(-2 pi i k / F), k in [0, F)
  cf *out;             // tiled range map, see rmap_index()
  int64_t cpiStride;   // samples between consecutive CPIs of the batch
  int32_t nPulses;     // nCpi * nDoppler
};

// segment s of the pulse at sample index pulseBase: v[k] = x'[t + T*k], yv[k] = y'[t + T*k]
// EX < E: only the first EX loads of the x segment are issued (the rest of x' is zero padding)
template <int T, int E, class In, int EX = E>
__device__ __forceinline__ void bufload_seg(const In &in, const RangePlan &p, int64_t pulseBase, int s, int t, cf *v, cf *yv)
{
  using B = BufLoad<In>;
  using CX = typename B::X;
  using CY = typename B::Y;
  constexpr int STEPX = T * CX::STRIDE, STEPY = T * CY::STRIDE;
  constexpr int NV = ((E - 1) * STEPY >> 12) + 1;
  const int s0 = s * p.segLen;
  const int cnt = min(p.segLen, p.nCorr - s0);
  const b2_v4i xd = make_rsrc(B::xp(in, pulseBase + s0), cnt * CX::STRIDE);
  const b2_v4i yd = make_rsrc(B::yp(in, pulseBase), p.nCorr * CY::STRIDE);
  int vx[1] = {t * CX::STRIDE};
  int vy[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) vy[j] = (s0 + p.delayMin + t) * CY::STRIDE + j * 4096; // may be negative: reads as zero
  typename CX::raw xr[EX];
  typename CY::raw yr[E];
  bufload_chan<CX, STEPX, EX, true>(xr, xd, vx);
  bufload_chan<CY, STEPY, E, false>(yr, yd, vy);
  bufwait<E, EX>(xr);
#pragma unroll
  for (int k = 0; k < EX; k++) v[k] = CX::cvt(xr[k]);
  bufwait<0, E>(yr);
#pragma unroll
  for (int k = 0; k < E; k++) yv[k] = CY::cvt(yr[k]);
}

// ILV: the x and y transforms advance together through their own exchange buffers
// (+3..6 % for F <= 2048, neutral at 4096 where it costs registers; measured).
template <int R3, class In>
__global__ __launch_bounds__(16 * R3, 2) void range_kernel(RangeArgs a, In in)
{
  using W = WgFft<R3>;
  constexpr bool ILV = (R3 <= 8);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *P = reinterpret_cast<cf *>(smem);
  cf *Q = P + W::A_ELEMS;
  const int t = threadIdx.x;

  cf tw1[15], tw3[16];
  W::load_twiddles(t, a.tw, tw1, tw3);

  const RangePlan p = a.plan;
  for (int pulse = blockIdx.x; pulse < a.nPulses; pulse += gridDim.x) {
    const int cpi = pulse / p.nDoppler;
    const int i = pulse - cpi * p.nDoppler;
    const int64_t base = (int64_t)cpi * a.cpiStride + (int64_t)i * p.nCorr;

    cf v[16], yv[16], acc[16];
    // zero-padded reference segments that end below 9*T samples (cfg 3: 2049 of 4096) have x'[t + T*k] = 0
    // for k >= 9: those loads are not issued and the first 16-point step skips the zero inputs
    const bool xshort = !ILV && p.segLen <= 9 * 16 * R3;
    for (int s = 0; s < p.nSeg; s++) {
      // both channels' loads go out first: 32 requests in flight per thread
      if (xshort) bufload_seg<16 * R3, 16, In, 9>(in, p, base, s, t, v, yv);
      else bufload_seg<16 * R3, 16>(in, p, base, s, t, v, yv);
      if (ILV) {
        // The x and y transforms advance together, each through its own exchange
        // buffer (P for x, Q for y, used first in the A layout and then in the B
        // layout): every barrier interval holds two independent instruction
        // streams, so LDS latency of one overlaps butterflies of the other.
        W::fwd_s1(t, v, tw1, P);
        W::fwd_s1(t, yv, tw1, Q);
        __syncthreads();
        W::fwd_s2_load(t, v, P);
        W::fwd_s2_load(t, yv, Q);
        dft16<-1>(v);
        dft16<-1>(yv);
        __syncthreads(); // every thread has read its A-layout values
        W::fwd_s2_store(t, v, P);
        W::fwd_s2_store(t, yv, Q);
        __syncthreads();
        W::fwd_s3(t, v, tw3, P);  // v  = X spectrum
        W::fwd_s3(t, yv, tw3, Q); // yv = Y spectrum
      } else {
        // one transform at a time, P in the A layout and Q in the B layout
        if (xshort) W::fwd_s1_nz9(t, v, tw1, P);
        else W::fwd_s1(t, v, tw1, P);
        __syncthreads();
        W::fwd_s2(t, v, P, Q);
        __syncthreads();
        W::fwd_s3(t, v, tw3, Q);
        W::fwd_s1(t, yv, tw1, P);
        __syncthreads();
        W::fwd_s2(t, yv, P, Q);
        __syncthreads();
        W::fwd_s3(t, yv, tw3, Q);
      }
      if (s == 0) {
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = cmulc(yv[e], v[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = cmacc(acc[e], yv[e], v[e]);
      }
      __syncthreads(); // P/Q are rewritten by the next segment (or the inverse)
    }
    W::inv_s1(t, acc, tw3, P);
    __syncthreads();
    W::inv_s2(t, acc, P, Q);
    __syncthreads();
    W::inv_s3(t, acc, tw1, Q);
    store_lags<R3>(a.out, p, cpi, i, t, acc);
    __syncthreads(); // Q is rewritten by the next pulse
  }
}

// --------------------------------------------------------------------------
// FUSED clutter FIR + range correlation at F = 4096 (round 6; WienerHopf.cpp:124-160 feeding Ambiguity.cpp:106-149 without
// the filtered channel ever crossing HBM).  tools/proto/fir_range_fusion_model.py is this kernel's transform sequence in
// NumPy, exact against the two-stage path.  A pulse is cut into segments of L = F/2 reference samples on the PULSE's own
// grid; with X_g the transform of the zero-padded segment g and H the taps' spectrum,
//     the window [segment g-1 | segment g]  has the spectrum  X_(g-1) + (-1)^m X_g                 (a shift by F/2 is a sign)
//     (w * xs) on the L samples that start at  g L + delayMin  = the last L outputs of IFFT(H (X_(g-1) + (-1)^m X_g))
// (xs[i] = x[i - delayMin]: the filter's own shift moves its OUTPUT grid, not the segments), and with the clutter window's
// first lag equal to the map's (config.yml; the launcher checks) the correlation's y' window of segment g is exactly the
// filter's output blocks g and g + 1: register for register, no exchange.  So ONE forward transform per segment serves the
// filter's window spectrum AND the correlation's x': per pulse 1 (history block) + S + S (filter inverses) + S (y' forward)
// + 1 (correlation inverse) + 1 (the ragged last segment twice: masked to the pulse for the correlation, with the filter's
// |delayMin| samples of look-ahead for the filter) = 3 S + 3 transforms -- 18 at configs[2] where clutter_fir_kernel +
// range_kernel run 20.5 -- and x, y are read once: 177 instead of 417 MB per CPI.  Registers: previous and current
// segment spectrum, accumulator, one work array, the previous output block (4 x 32 + 16) + the twiddles (62).
// Requirements (launcher): F = 4096, nBins <= L + 1, nDelay <= L + 1, clutter delayMin == map delayMin <= 0,
// nCorr >= L - delayMin, the CPI's last used sample + |delayMin| inside the CPI (no circular wrap of xs in reach).
struct RangeFirArgs {
  RangePlan plan;
  const cf *tw;       // exp(-2 pi i k / F)
  cf *out;            // tiled range map
  int64_t cpiStride;
  int32_t nPulses;
  const cf *H;        // [nCpi][16][256]: the taps' spectrum / F in the transform's register layout (taps_spectrum_kernel)
  const cf *w;        // [nCpi][nBins]: the taps themselves (the few direct products at a pulse's edges, and the largest tap)
  int32_t nBins;
  const int32_t *k0;  // [nCpi]: the largest tap's index -- left out of H and applied in the time domain (clutter_fir_kernel does the same)
  uint32_t N;         // samples per CPI
};

// grid nCpi x 256: H[cpi][e][t] = register e of thread t of FFT_4096(w zero-padded, its LARGEST tap left out) / F; k0[cpi] = that
// tap's index (ties to the lower index).  The largest tap is applied in the time domain by range_fir_kernel: with a direct path
// far above the noise its product is nearly all of w * xs, and one fma rounds it once where the transform rounds it 3.5 eps.
__global__ __launch_bounds__(256) void taps_spectrum_kernel(const cf *w, int nBins, const cf *tw, cf *H, int32_t *k0)
{
  using W = WgFft<16>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *P = reinterpret_cast<cf *>(smem);
  cf *Q = P + W::A_ELEMS;
  const int t = threadIdx.x, cpi = blockIdx.x;
  cf tw1[15], tw3[16];
  W::load_twiddles(t, tw, tw1, tw3);
  cf v[16];
  const float sc = 1.0f / (float)W::F;
  float bestMag = -1.f;
  int bestIdx = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int m = t + 256 * k;
    const cf x = w[(size_t)cpi * nBins + min(m, nBins - 1)];
    v[k] = m < nBins ? cmake(x.x * sc, x.y * sc) : cmake(0.f, 0.f);
    const float mag = m < nBins ? x.x * x.x + x.y * x.y : -1.f;
    if (mag > bestMag) { bestMag = mag; bestIdx = m; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float mu = __shfl_xor(bestMag, off);
    const int iu = __shfl_xor(bestIdx, off);
    if (mu > bestMag || (mu == bestMag && iu < bestIdx)) { bestMag = mu; bestIdx = iu; }
  }
  {
    float *sm = reinterpret_cast<float *>(P);
    int *si = reinterpret_cast<int *>(P) + 4;
    if ((t & 63) == 0) { sm[t >> 6] = bestMag; si[t >> 6] = bestIdx; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const float mu = sm[u];
      const int iu = si[u];
      if (mu > bestMag || (mu == bestMag && iu < bestIdx)) { bestMag = mu; bestIdx = iu; }
    }
    __syncthreads();
  }
  if (t == 0) k0[cpi] = bestIdx;
#pragma unroll
  for (int k = 0; k < 16; k++)
    if (t + 256 * k == bestIdx) v[k] = cmake(0.f, 0.f);
  W::fwd_s1(t, v, tw1, P);
  __syncthreads();
  W::fwd_s2(t, v, P, Q);
  __syncthreads();
  W::fwd_s3(t, v, tw3, Q);
#pragma unroll
  for (int e = 0; e < 16; e++) H[((size_t)cpi * 16 + e) * 256 + t] = v[e];
}

template <class In>
__global__ __launch_bounds__(256, 2) void range_fir_kernel(RangeFirArgs a, In in)
{
  using W = WgFft<16>;
  using CX = typename BufLoad<In>::X;
  using CY = typename BufLoad<In>::Y;
  using RX = RawBuiltin<CX>;
  using RY = RawBuiltin<CY>;
  constexpr int T = 256, L = 2048;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *P = reinterpret_cast<cf *>(smem);
  cf *Q = P + W::A_ELEMS;
  const int t = threadIdx.x;
  // the stage-3 twiddles depend on t % 16 only: a 16 x 15 table behind the exchange buffers (2 KB) instead of 30 registers --
  // the kernel's time does not move (1601-1646 against 1623-1649 us per 16 CPIs, same box), its spilled registers go from 28 to 8
  // (fp32 planes; none with the int16 words), saved once per kernel and read back five times a pulse
  cf tw1[15];
  W::load_tw1(t, a.tw, tw1);
  cf *T3 = Q + W::B_ELEMS;
  if (t < 240) T3[t] = a.tw[(16 * (t / 15) * (t % 15 + 1)) & (W::F - 1)];
  __syncthreads();
  const cf *tw3 = T3 + (t & 15) * 15;
  const float sgn = ((t >> 4) & 1) ? -1.f : 1.f; // (-1)^m of this thread's spectrum registers (m = q + 16 r + 256 s, q = t >> 4)
  const RangePlan p = a.plan;
  const int dmin = p.delayMin; // = the clutter window's first lag (launcher), <= 0
  const int SB = (p.nCorr - dmin + L - 1) / L;  // blocks of filter output that reach into the pulse (segments: S = SB or SB - 1)
  auto fwd_half = [&](cf *v) { // inputs v[8..15] are zero
    W::fwd_s1_nz9(t, v, tw1, P);
    __syncthreads();
    W::fwd_s2(t, v, P, Q);
    __syncthreads();
    W::fwd_s3(t, v, tw3, Q);
    __syncthreads();
  };
  for (int pulse = blockIdx.x; pulse < a.nPulses; pulse += gridDim.x) {
    const int cpi = pulse / p.nDoppler;
    const int i = pulse - cpi * p.nDoppler;
    const int p0 = i * p.nCorr;
    // x of the whole CPI: sample u of the stream at offset u (beyond the CPI: zeros -- out of reach by the launcher's check)
    const __amdgpu_buffer_rsrc_t xd = make_rsrc_b(BufLoad<In>::xp(in, (int64_t)cpi * a.cpiStride), (int)a.N * CX::STRIDE);
    // y of this pulse: outside it the range check returns the zeros the correlation's mask wants
    const __amdgpu_buffer_rsrc_t yd = make_rsrc_b(BufLoad<In>::yp(in, (int64_t)cpi * a.cpiStride + p0), p.nCorr * CY::STRIDE);
    // ... and x of this pulse for the windows: beyond the pulse's end the range check IS the mask (no fetch, no select)
    const __amdgpu_buffer_rsrc_t xpd = make_rsrc_b(BufLoad<In>::xp(in, (int64_t)cpi * a.cpiStride + p0), p.nCorr * CX::STRIDE);
    const cf *Hc = a.H + (size_t)cpi * 16 * 256 + t;
    const cf *wc = a.w + (size_t)cpi * a.nBins;
    const int k0 = __builtin_amdgcn_readfirstlane(a.k0[cpi]); // the largest tap: in the time domain, on the filter's TRUE stream
    const cf w0 = wc[k0];
    // Three arrays (+ the twiddles): what range_kernel holds.  V = the spectrum of the filter's WINDOW
    // [segment g-1 | segment g] of x masked to the pulse, wk = the work array, acc = the correlation's accumulator.
    // Where the filter's stream differs from the masked x -- it looks |delayMin| samples past the pulse's end, and it is zero
    // on the CPI's first |delayMin| samples (xs[m < 0] = 0, WienerHopf.cpp:125-160) -- the <= |delayMin| products a sample is
    // off by are added directly (`edges`): a few hundred MACs a pulse instead of a second transform.
    cf V[16], wk[16], acc[16];
    // the filter's output block g from wk = IFFT(H V_g): samples n = g L + delayMin + [0, L) of the pulse are wk[8..15];
    // leaves y' = y - (w * xs) of the block, masked to the pulse, in wk[0..7] and zeros in wk[8..15]
    // B2_FIR_YPOS (experiments, tools/prof_fused_ab.sh): where the block's y is requested -- 1 (default) behind the inverse's
    // first exchange; 2 behind its second; 0 before the window's transform (see profiles/r06_ab_experiments.json)
#ifndef B2_FIR_YPOS
#define B2_FIR_YPOS 1
#endif
    typename RY::raw yr[8];
    typename RX::raw xr0[8]; // x[n - delayMin - k0] of the block's samples n: the largest tap's operand
    auto y_request = [&](int g, int ka = 0, int kb = 8) {
#pragma unroll
      for (int k = ka; k < kb; k++) {
        int vo = (g * L + dmin + t + T * k) * CY::STRIDE; // may be negative: the whole offset in the VGPR (bufload.hpp)
        asm volatile("" : "+v"(vo));
        yr[k] = RY::ld(yd, vo, 0);
        int uo = (p0 + g * L + t + T * k - k0) * CX::STRIDE; // CPI index n - delayMin - k0 + p0; negative (pulse 0): reads as zero
        asm volatile("" : "+v"(uo));
        xr0[k] = RX::ld(xd, uo, 0);
      }
    };
    auto block_out = [&](int g) {
      W::inv_s1(t, wk, tw3, Q);
      __syncthreads();
      if (B2_FIR_YPOS == 1) y_request(g); // y of the block: requested here, used after the last stage
      if (B2_FIR_YPOS == 3) y_request(g, 0, 4);
      W::inv_s2(t, wk, Q, P);
      __syncthreads();
      if (B2_FIR_YPOS == 2) y_request(g);
      if (B2_FIR_YPOS == 3) y_request(g, 4, 8);
      W::inv_s3(t, wk, tw1, P);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 8; k++) { // the largest tap: w0 xs[n - k0], xs zero on the CPI's first |delayMin| samples (WienerHopf.cpp:125-160)
        const int u = p0 + g * L + t + T * k - k0;
        const cf x = RX::cvt(xr0[k]);
        wk[8 + k] = cadd(wk[8 + k], u >= -dmin ? cmul(w0, x) : cmake(0.f, 0.f));
      }
      const bool tail = (g + 1) * L > p.nCorr;                      // the block holds the pulse's last |delayMin| samples
      const bool head = i == 0 && g * L + dmin < a.nBins + (-dmin); // pulse 0: samples the CPI's first |delayMin| reach
      if (tail || head) { // (workgroup-uniform) the edges, through LDS so that the register arrays keep static indices
        cf *E = P + t;    // P is free: its last readers passed the barrier above
#pragma unroll
        for (int k = 0; k < 8; k++) E[k * T] = wk[8 + k];
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
          const int n = g * L + dmin + t + T * k; // sample of the pulse
          cf c = E[k * T];
          if (tail) // past the pulse's end: the taps kk <= n - delayMin - nCorr reach x[p0 + nCorr ...], masked out of the window
            for (int kk = 0; kk <= min(n - dmin - p.nCorr, a.nBins - 1) && n < p.nCorr; kk++) {
              const cf x = RX::cvt(RX::ld(xd, (p0 + n - dmin - kk) * CX::STRIDE, 0));
              if (kk != k0) c = cadd(c, cmul(wc[kk], x)); // (the largest tap already runs on the true stream)
            }
          if (head && n >= 0) // the CPI's first |delayMin| samples: in the window of pulse 0, zero in the filter's stream
            for (int kk = max(n + 1, 0); kk <= n - dmin && kk < a.nBins; kk++) {
              const cf x = RX::cvt(RX::ld(xd, (n - dmin - kk) * CX::STRIDE, 0));
              if (kk != k0) c = csub(c, cmul(wc[kk], x));
            }
          E[k * T] = c;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) wk[8 + k] = E[k * T];
        __syncthreads(); // P is written again by the next transform
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int n = g * L + dmin + t + T * k;
        const cf yv = RY::cvt(yr[k]);
        wk[k] = (n >= 0 && n < p.nCorr) ? csub(yv, wk[8 + k]) : cmake(0.f, 0.f);
      }
#pragma unroll
      for (int k = 8; k < 16; k++) wk[k] = cmake(0.f, 0.f);
    };
    // ---- block 0: the history block X_(-1) and segment 0 separately (the history is not a segment of the correlation);
    //      X_0 sits in the accumulator's registers, which have nothing to hold yet
    if (B2_FIR_YPOS == 0) y_request(0);
#pragma unroll
    for (int e = 0; e < 16; e++) wk[e] = Hc[e * 256];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = RX::cvt(RX::ld(xpd, (t + T * k) * CX::STRIDE, 0));
#pragma unroll
    for (int k = 8; k < 16; k++) acc[k] = cmake(0.f, 0.f);
#ifdef B2_FIR_CARRY // experiment: the upper half of a block's window is the lower half of the next block's -- carried in 16 registers
    cf keep[8];
#pragma unroll
    for (int k = 0; k < 8; k++) keep[k] = acc[k];
#endif
    if (i > 0) { // (pulse 0: the stream is zero in front of the CPI)
#pragma unroll
      for (int k = 0; k < 8; k++) V[k] = RX::cvt(RX::ld(xd, (p0 - L + t + T * k) * CX::STRIDE, 0));
#pragma unroll
      for (int k = 8; k < 16; k++) V[k] = cmake(0.f, 0.f);
      fwd_half(V);
    } else {
#pragma unroll
      for (int e = 0; e < 16; e++) V[e] = cmake(0.f, 0.f);
    }
    fwd_half(acc);
#pragma unroll
    for (int e = 0; e < 16; e++) wk[e] = cmul(wk[e], cmake(V[e].x + sgn * acc[e].x, V[e].y + sgn * acc[e].y));
    block_out(0);
    // Z_g = FFT([block g | 0]); the y' window of segment g is [block g | block g + 1] = Z_g + (-1)^m Z_(g+1), so
    //   sum_g Y'_g conj(X_g) = Z_0 conj(X_0) + sum_(g >= 1) Z_g conj(X_g + (-1)^m X_(g-1)) = ... + sum Z_g conj((-1)^m V_g)
    fwd_half(wk);
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = cmulc(wk[e], acc[e]);
    for (int g = 1; g < SB; g++) {
      // (H into the work array, which is free until the product below: its L2 latency hides behind the window's transform)
      if (B2_FIR_YPOS == 0) y_request(g);
#pragma unroll
      for (int e = 0; e < 16; e++) wk[e] = Hc[e * 256];
#ifdef B2_FIR_CARRY
#pragma unroll
      for (int k = 0; k < 8; k++) {
        V[k] = keep[k];
        V[8 + k] = RX::cvt(RX::ld(xpd, (g * L + t + T * k) * CX::STRIDE, 0));
        keep[k] = V[8 + k];
      }
#else
#pragma unroll
      for (int k = 0; k < 16; k++) V[k] = RX::cvt(RX::ld(xpd, ((g - 1) * L + t + T * k) * CX::STRIDE, 0));
#endif
      W::fwd_s1(t, V, tw1, P);
      __syncthreads();
      W::fwd_s2(t, V, P, Q);
      __syncthreads();
      W::fwd_s3(t, V, tw3, Q);
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; e++) wk[e] = cmul(wk[e], V[e]);
      block_out(g);
      fwd_half(wk);
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = cmacc(acc[e], wk[e], cmake(sgn * V[e].x, sgn * V[e].y));
    }
    W::inv_s1(t, acc, tw3, P);
    __syncthreads();
    W::inv_s2(t, acc, P, Q);
    __syncthreads();
    W::inv_s3(t, acc, tw1, Q);
    store_lags<16>(a.out, p, cpi, i, t, acc);
    __syncthreads(); // P, Q are rewritten by the next pulse
  }
}

// --------------------------------------------------------------------------
// Range kernel on the 8-points-per-thread transform (fft_wg8.hpp): identical mathematics and
// interface, T = F/8 threads per pulse, half the registers per thread (3 waves per SIMD, see RANGE8_WAVES_PER_SIMD).  Stage 4 of
// the transform runs across lanes (fwd_s3_lanes / inv_s4_lanes), so a transform has two LDS exchanges
// and two barriers.  Buffer schedule (A = E1 layout, B = E2 layout, every transform the same):
//   forward  s1 -> A | barrier | s2: A -> B | barrier | s3 + s4: B -> registers
//   inverse  s4 + s3: registers -> A | barrier | s2: A -> B | barrier | s1: B -> registers
// A buffer is rewritten only after a barrier that follows its last read: A's readers (s2 loads) finish
// before the barrier between s2 and s3, B's readers (s3 loads) before the next transform's first
// barrier, and the inverse's last loads (from B) precede the next pulse's first barrier.
template <int R4, class In>
__global__ __launch_bounds__(64 * R4, RANGE8_WAVES_PER_SIMD) void range8_kernel(RangeArgs a, In in)
{
  using W = WgFft8<R4>;
  constexpr int T = W::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *A = reinterpret_cast<cf *>(smem);
  cf *B = A + W::BUF_ELEMS;
  const int t = threadIdx.x;
  cf tw1[7], tw2[7], tw3[7];
  W::load_twiddles(t, a.tw, tw1, tw2, tw3);
  const typename W::LaneConst lc = W::lane_constants(t, a.tw);

  const RangePlan p = a.plan;
  for (int pulse = blockIdx.x; pulse < a.nPulses; pulse += gridDim.x) {
    const int cpi = pulse / p.nDoppler;
    const int i = pulse - cpi * p.nDoppler;
    const int64_t base = (int64_t)cpi * a.cpiStride + (int64_t)i * p.nCorr;
    cf acc[8];
    for (int s = 0; s < p.nSeg; s++) {
      cf v[8], yv[8];
      bufload_seg<T, 8>(in, p, base, s, t, v, yv);
      W::fwd_s1(t, v, tw1, A);
      __syncthreads();
      W::fwd_s2_load(t, v, A);
      W::fwd_s2_store(t, v, tw2, B);
      __syncthreads();
      W::fwd_s3_lanes(t, v, tw3, lc, B); // v = X spectrum

      W::fwd_s1(t, yv, tw1, A);
      __syncthreads();
      W::fwd_s2_load(t, yv, A);
      W::fwd_s2_store(t, yv, tw2, B);
      __syncthreads();
      W::fwd_s3_lanes(t, yv, tw3, lc, B); // yv = Y spectrum
      if (s == 0) {
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = cmulc(yv[e], v[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = cmacc(acc[e], yv[e], v[e]);
      }
    }
    W::inv_s4_lanes(t, acc, tw3, lc);
    W::inv_s3_store(t, acc, A);
    __syncthreads();
    W::inv_s2_load(t, acc, tw2, A);
    W::inv_s2_store(t, acc, B);
    __syncthreads();
    W::inv_s1(t, acc, tw1, B);
    store_lags_g<T, 8>(a.out, p, cpi, i, t, acc);
  }
}

// --------------------------------------------------------------------------
// Range kernel on the one-wave transform (fft_wave.hpp, F = 2048): identical mathematics and
// interface, ONE wave64 per pulse, 32 points per lane, one LDS exchange per transform through a
// wave-private region and no workgroup barrier anywhere.  The lags come out in natural order
// z[t + 64*c], so the store is the same 128-byte-run pattern as the other kernels'.
#ifndef RANGEW_WAVES_PER_SIMD
#define RANGEW_WAVES_PER_SIMD 2
#endif
// segment s of the pulse: v[k] = x'[t + 64*k], k < NX, yv[k] = y'[t + 64*k], k < NY (the rest of the
// windows is zero padding, or -- y' beyond segLen + nDelay - 2 -- never meets a wanted lag)
template <class In, int NX, int NY>
__device__ __forceinline__ void bufload_seg_w(const In &in, const RangePlan &p, int64_t pulseBase, int s, int t, cf *v, cf *yv)
{
  static_assert((NX == 24 || NX == 32) && (NY == 28 || NY == 32), "");
  using B = BufLoad<In>;
  using CX = typename B::X;
  using CY = typename B::Y;
  constexpr int STEPX = 64 * CX::STRIDE, STEPY = 64 * CY::STRIDE;
  constexpr int NV = (31 * STEPY >> 12) + 1;
  const int s0 = s * p.segLen;
  const int cnt = min(p.segLen, p.nCorr - s0);
  const b2_v4i xd = make_rsrc(B::xp(in, pulseBase + s0), cnt * CX::STRIDE);
  const b2_v4i yd = make_rsrc(B::yp(in, pulseBase), p.nCorr * CY::STRIDE);
  int vx[1] = {t * CX::STRIDE};
  int vy[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) vy[j] = (s0 + p.delayMin + t) * CY::STRIDE + j * 4096; // may be negative: reads as zero
  typename CX::raw xr[NX];
  typename CY::raw yr[NY];
  bufload_chan<CX, STEPX, NX, true>(xr, xd, vx);
  bufload_chan<CY, STEPY, NY, false>(yr, yd, vy);
  bufwait<NY + NX - 16, 16>(xr);
  if constexpr (NX == 32) bufwait<NY, 16>(xr + 16);
  else bufwait<NY, 8>(xr + 16);
#pragma unroll
  for (int k = 0; k < NX; k++) v[k] = CX::cvt(xr[k]);
  bufwait<NY - 16, 16>(yr);
  if constexpr (NY == 32) bufwait<0, 16>(yr + 16);
  else { bufwait<4, 8>(yr + 16); bufwait<0, 4>(yr + 24); }
#pragma unroll
  for (int k = 0; k < NY; k++) yv[k] = CY::cvt(yr[k]);
}

// lags z[t + 64*c] of one pulse into the tiled range map: lane t owns position t & 15 of tile
// (t >> 4) + 4*c, so consecutive c are a constant stride apart
template <int NC>
__device__ __forceinline__ void store_lags_w(cf *out, const RangePlan &p, int cpi, int pulse, int t, const cf *v)
{
  // column j + colOff of the map; colOff is a multiple of 16 except for the (rare) chunks behind an aliasing boundary,
  // where the lane's position inside its tile shifts: those go through the general index
  const int jt = t + (p.colOff & 15);
  cf *o = out + (((int64_t)cpi * p.nTilesOut + (p.colOff >> 4) + (jt >> 4)) * p.nDoppler + pulse) * 16 + (jt & 15);
  const int64_t step = (int64_t)p.nDoppler * 64; // four tiles
  int rem = p.nDelay - t;                        // lane t stores register c iff 64*c < rem
  int nd = p.nDelay;
  // opaque per call: otherwise the 32 lane masks and 32 uniform conditions are hoisted out of the
  // pulse loop and live (spilled) across the whole kernel
  asm volatile("" : "+v"(rem), "+s"(nd));
#pragma unroll
  for (int c = 0; c < NC; c++) {
    if (64 * c >= nd) break; // wave-uniform
#if defined(RW_NO_STORE) // timing experiment: everything but the stores
    if (64 * c < rem && p.scale == 12345.f) *o = cmake(v[c].x * p.scale, v[c].y * p.scale);
#else
    if (64 * c < rem) *o = cmake(v[c].x * p.scale, v[c].y * p.scale);
#endif
    o += step;
  }
}

#ifdef RANGEW_TRACE
#define RW_T(k) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); tr[k] += now_ - t0_; t0_ = now_; }
#else
#define RW_T(k)
#endif

// RANGEW_WAVES independent waves per workgroup share the stage-twiddle table; each has its own
// exchange region and walks its own pulses (no barrier after the table is filled).
#ifndef RANGEW_WAVES
#define RANGEW_WAVES 8
#endif
// SHORTW: windows short enough for the pruned form, segLen <= 24*64 and segLen + nDelay - 1 <= 28*64 (cfg 2:
// x' has 1300 and y' needs 1709 of 2048 samples): x' = 0 from 24*64 on, y' is not needed from 28*64 on --
// 12 of 64 loads are not issued and the first 32-point step of both transforms skips the zero inputs.
// A template parameter, not a branch: both forms in one loop body spill.
// OUT7: nDelay <= 7*64 -- the inverse transform computes only the 7 wanted outputs per lane (always with SHORTW, whose
// window bound implies it; also for long segments with few lags, cfg 5: segLen 1627, nDelay 411).
template <class In, bool SHORTW, bool OUT7 = SHORTW>
__global__ __launch_bounds__(64 * RANGEW_WAVES, RANGEW_WAVES_PER_SIMD) void rangew_kernel(RangeArgs a, In in)
{
  using W = WaveFft;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *table = reinterpret_cast<cf *>(smem);
  const int t = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform: the pulse loop stays scalar
  cf *X = table + W::TW_ELEMS + wave * W::X_ELEMS;
  W::fill_table(threadIdx.x, 64 * RANGEW_WAVES, a.tw, table);
  __syncthreads();
  W::Tw w;
  W::load_twiddles(t, a.tw, table, w);
  const RangePlan p = a.plan;
#ifdef RANGEW_TRACE
  uint64_t tr[6] = {0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime();
#endif
  for (int pulse = blockIdx.x * RANGEW_WAVES + wave; pulse < a.nPulses; pulse += gridDim.x * RANGEW_WAVES) {
    const int cpi = pulse / p.nDoppler;
    const int i = pulse - cpi * p.nDoppler;
    const int64_t base = (int64_t)cpi * a.cpiStride + (int64_t)i * p.nCorr;
    cf acc[32];
#pragma unroll
    for (int e = 0; e < 32; e++) acc[e] = cmake(0.f, 0.f);
    for (int s = 0; s < p.nSeg; s++) {
      cf v[32], yv[32];
      constexpr int NX = SHORTW ? 24 : 32, NY = SHORTW ? 28 : 32;
      RW_T(0)
      bufload_seg_w<In, NX, NY>(in, p, base, s, t, v, yv);
#ifdef RANGEW_TRACE
      asm volatile("" : "+v"(v[0].x), "+v"(yv[NY - 1].y));
#endif
      RW_T(1)
#if defined(RW_NO_COMPUTE) // timing experiment: the kernel's memory operations alone
#pragma unroll
      for (int e = 0; e < NX; e++) acc[e] = cadd(acc[e], cadd(v[e], yv[e]));
#pragma unroll
      for (int e = NX; e < NY; e++) acc[e] = cadd(acc[e], yv[e]);
#else
      W::transform<-1, NX>(t, v, w, X);  // v  = X spectrum
#ifdef RANGEW_TRACE
      asm volatile("" : "+v"(v[0].x));
#endif
      RW_T(2)
      W::transform<-1, NY>(t, yv, w, X); // yv = Y spectrum
#ifdef RANGEW_TRACE
      asm volatile("" : "+v"(yv[0].x));
#endif
      RW_T(3)
#pragma unroll
      for (int e = 0; e < 32; e++) acc[e] = cmacc(acc[e], yv[e], v[e]);
#endif
    }
    RW_T(0)
#if !defined(RW_NO_COMPUTE)
    W::template transform<+1, 32, OUT7>(t, acc, w, X);
#endif
#ifdef RANGEW_TRACE
    asm volatile("" : "+v"(acc[0].x));
#endif
    RW_T(4)
    store_lags_w<OUT7 ? 7 : 32>(a.out, p, cpi, i, t, acc);
    RW_T(5)
  }
#ifdef RANGEW_TRACE // buckets: other, load, X, Y, inverse, store
  if (t == 0) trace_finish("rangew", tr, blockIdx.x == 0 && threadIdx.x == 0);
#endif
}

// --------------------------------------------------------------------------
// Range kernel on the one-wave 1024-point transform (fft_wave1k.hpp): the kernel above at 16 points per lane, with the
// loads of the NEXT segment in flight while this one is transformed.
//
// Why.  The 2048-point kernel's memory operations alone (-DRW_NO_COMPUTE) take 84 % of its time, its loads alone run
// at the streaming rate (5.9 TB/s), and doubling the waves per SIMD without prefetching changes nothing (measured,
// DESIGN.md section 4): a wave has loads in flight only while it waits for them, a quarter of its time, so a CU
// averages ~50 KB in flight whatever the occupancy -- the kernel is bound by bytes in flight, not by issue slots.
// Prefetching needs registers the 2048-point kernel does not have (3 x 64 + the next x and y: 320).  At 16 points
// per lane x, y and the accumulator are 3 x 32, and the raw samples of the next segment need not be a fourth set:
//     x_{s+1} is requested as soon as x_s has been copied out of its 18 (SHORTX; else 32) registers -- a whole
//             segment ahead;
//     y_{s+1} is requested into y_s's own registers after the spectrum product, and is in flight during the
//             x transform of the next segment.
// Peak: 32 (X) + 32 (y / Y) + 32 (accumulator) + 18 (next x) = 114 registers + the 16-point kernel's temporaries.
// The loads and stores are compiler builtins here (not the asm of bufload.hpp): the waits between issue and use are
// the compiler's, counted in program order across the loop.  The lag stores of a pulse are issued behind the next
// pulse's first loads.  Out-of-range offsets (negative for y, beyond the segment for x, everything once the wave
// has run out of pulses) read as zero without a branch, as in the other kernels.
#ifndef RANGEW1K_WAVES_PER_SIMD
#define RANGEW1K_WAVES_PER_SIMD 3
#endif
#ifndef RANGEW1K_WAVES
#define RANGEW1K_WAVES 12
#endif
// x' of segment s: rx[k] = x'[t + 64*k], k < NX; `live` false: nothing is read (zero records)
template <class In, int NX>
__device__ __forceinline__ void w1k_issue_x(const In &in, const RangePlan &p, int64_t pulseBase, int s, int t, bool live,
                                            typename RawBuiltin<typename BufLoad<In>::X>::raw *rx)
{
  using CX = typename BufLoad<In>::X;
  const int s0 = s * p.segLen;
  const int cnt = live ? min(p.segLen, p.nCorr - s0) : 0;
  const __amdgpu_buffer_rsrc_t d = make_rsrc_b(BufLoad<In>::xp(in, pulseBase + s0), cnt * CX::STRIDE);
#pragma unroll
  for (int k = 0; k < NX; k++) rx[k] = RawBuiltin<CX>::ld(d, t * CX::STRIDE, k * 64 * CX::STRIDE);
}
// y' of segment s: ry[k] = y'[t + 64*k], k < 16; the offset may be negative (reads as zero): all of it in voffset
// K0 > 0: only registers K0..15 (the first K0 are carried over from the previous window, see REUSE below)
template <class In, int K0 = 0>
__device__ __forceinline__ void w1k_issue_y(const In &in, const RangePlan &p, int64_t pulseBase, int s, int t, bool live,
                                            typename RawBuiltin<typename BufLoad<In>::Y>::raw *ry)
{
  using CY = typename BufLoad<In>::Y;
  const __amdgpu_buffer_rsrc_t d = make_rsrc_b(BufLoad<In>::yp(in, pulseBase), live ? p.nCorr * CY::STRIDE : 0);
  // one opaque VGPR base per 4 KiB of offset: a constant the compiler can see beyond the 12-bit immediate would be split
  // into soffset, and a negative voffset stays out of range whatever soffset is (bufload.hpp)
  constexpr int STEP = 64 * CY::STRIDE, NV = (15 * STEP >> 12) + 1;
  int vb[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) {
    vb[j] = (s * p.segLen + p.delayMin + t) * CY::STRIDE + j * 4096;
    asm volatile("" : "+v"(vb[j]));
  }
#pragma unroll
  for (int k = K0; k < 16; k++) ry[k] = RawBuiltin<CY>::ld(d, vb[(k * STEP) >> 12] + ((k * STEP) & 4095), 0);
}

// SHORTX: segLen <= 9*64 (cfg 2: 557): x' = 0 from 9*64 on -- 7 of 16 loads are not issued and the first 16-point
// step skips the zero inputs.  OUT7: nDelay <= 7*64 -- the inverse computes only the 7 wanted outputs per lane.
// REUSE: segLen = 9*64 exactly.  Consecutive y' windows of a pulse then overlap by WHOLE registers: y'_{s+1}[t + 64 k] =
// y'_s[t + 64 (k + 9)], so registers 9..15 of one window are registers 0..6 of the next and only nine are loaded.  The
// overlap (nDelay - 1 of every segLen + nDelay - 1 samples) is otherwise read again, and at this kernel's streaming
// rate an XCD's L2 has turned over between two segments of a wave: the re-read comes from HBM (PMC: 1.08-1.14 x the
// algorithmic bytes at F = 2048, more at F = 1024 with its shorter segments).
template <class In, bool SHORTX, bool OUT7, bool REUSE = false>
__global__ __launch_bounds__(64 * RANGEW1K_WAVES, RANGEW1K_WAVES_PER_SIMD) void rangew1k_kernel(RangeArgs a, In in)
{
  static_assert(!REUSE || SHORTX, "");
  using W = Wave1kFft;
  using RX = RawBuiltin<typename BufLoad<In>::X>;
  using RY = RawBuiltin<typename BufLoad<In>::Y>;
  constexpr int NX = SHORTX ? 9 : 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *table = reinterpret_cast<cf *>(smem);
  const int t = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  cf *X = table + W::TW_ELEMS + wave * W::X_ELEMS;
  W::fill_table(threadIdx.x, 64 * RANGEW1K_WAVES, a.tw, table);
  __syncthreads(); // the only barrier: waves without a pulse leave after it
  W::Tw w;
  W::load_twiddles(t, a.tw, table, w);
  const RangePlan p = a.plan;
  const int stride = gridDim.x * RANGEW1K_WAVES;
  int pulse = blockIdx.x * RANGEW1K_WAVES + wave;
  if (pulse >= a.nPulses) return;
  int cpi = pulse / p.nDoppler;
  int i = pulse - cpi * p.nDoppler;
  int64_t base = (int64_t)cpi * a.cpiStride + (int64_t)i * p.nCorr;
  int s = 0;
  typename RX::raw rx[NX];
  typename RY::raw ry[16];
  cf acc[16];
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = cmake(0.f, 0.f);
  w1k_issue_x<In, NX>(in, p, base, 0, t, true, rx);
  w1k_issue_y<In>(in, p, base, 0, t, true, ry);
#ifdef RANGEW_TRACE // buckets: 0 loop bookkeeping, 1 wait x, 2 issue x + X transform, 3 wait y, 4 Y transform + product + issue y, 5 inverse + stores
  uint64_t tr[6] = {0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime();
#endif
  for (;;) {
    // the segment after this one (wave-uniform)
    int ns = s + 1, npulse = pulse, ncpi = cpi, ni = i;
    int64_t nbase = base;
    if (ns == p.nSeg) {
      ns = 0;
      npulse = pulse + stride;
      ncpi = npulse / p.nDoppler;
      ni = npulse - ncpi * p.nDoppler;
      nbase = (int64_t)ncpi * a.cpiStride + (int64_t)ni * p.nCorr;
    }
    const bool more = npulse < a.nPulses;
    cf v[16];
    RW_T(0)
#pragma unroll
    for (int k = 0; k < NX; k++) v[k] = RX::cvt(rx[k]);
#ifdef RANGEW_TRACE
    asm volatile("" : "+v"(v[0].x), "+v"(v[NX - 1].y)); // the wait for x ends here
#endif
    RW_T(1)
    // the first steps of the transform read the raw registers and write new ones; only then is the next x requested into
    // them -- requested before, the compiler has to copy the 18 registers out of the loads' way
    W::s1<-1, NX>(v, w);
    __builtin_amdgcn_sched_barrier(0);
    w1k_issue_x<In, NX>(in, p, nbase, ns, t, more, rx);
    __builtin_amdgcn_sched_barrier(0);
    W::finish<-1>(t, v, w, X); // v = X spectrum
#ifdef RANGEW_TRACE
    asm volatile("" : "+v"(v[0].x));
#endif
    RW_T(2)
    cf yin[16], yv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) yin[k] = RY::cvt(ry[k]);
#ifdef RANGEW_TRACE
    asm volatile("" : "+v"(yin[0].x), "+v"(yin[15].y)); // the wait for y ends here
#endif
    RW_T(3)
    // out of place: the raw registers 9..15 are the next window's 0..6 (REUSE) and stay where they are
    W::s1_nd<-1>(yv, yin, w);
    W::finish<-1>(t, yv, w, X); // yv = Y spectrum
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = cmacc(acc[e], yv[e], v[e]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (REUSE) {
      if (ns != 0) {
#pragma unroll
        for (int k = 0; k < 7; k++) ry[k] = ry[9 + k];
        w1k_issue_y<In, 7>(in, p, nbase, ns, t, more, ry);
      } else {
        w1k_issue_y<In>(in, p, nbase, ns, t, more, ry);
      }
    } else {
      w1k_issue_y<In>(in, p, nbase, ns, t, more, ry);
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef RANGEW_TRACE
    asm volatile("" : "+v"(acc[0].x));
#endif
    RW_T(4)
    if (ns == 0) { // the pulse is complete
      W::template transform<+1, 16, OUT7>(t, acc, w, X);
      store_lags_w<OUT7 ? 7 : 16>(a.out, p, cpi, i, t, acc);
#ifdef RANGEW_TRACE
      RW_T(5)
      if (!more) {
        if (t == 0) trace_finish("rangew1k", tr, blockIdx.x == 0 && threadIdx.x == 0);
        break;
      }
#else
      if (!more) break;
#endif
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = cmake(0.f, 0.f);
      pulse = npulse;
      cpi = ncpi;
      i = ni;
    }
    s = ns;
    base = nbase;
  }
}

#ifdef B2_RANGEW1K_GLDS
// EXPERIMENT (round 6; built only with -DB2_RANGEW1K_GLDS, tools/build_variant.sh): rangew1k_kernel
// with a segment's x' and y' windows landing in the wave's exchange region by LDS-DMA (`buffer_load_dwordx4 ... lds`)
// instead of in 18 + 32 registers requested a segment ahead.  156 -> <= 128 VGPRs: FOUR waves per SIMD, 16 waves per
// workgroup, LDS = table + 16 x 8.5 KB = 147 KB.  There is no room for a second landing zone (a prefetched y' window of
// 8 KB per wave would need 128 KB more), so each window lands JUST IN TIME -- x' before the segment's first transform, y'
// between its two -- and its round trip is covered by the SIMD's other three waves, not by this wave's own arithmetic.
// (First form: x' kept its 18 prefetch registers, only y' by LDS-DMA: 128 VGPRs with 27 spilled.)
// A window of 128 n samples = n pieces of 1 KiB (lane L of piece j: samples 128 j + 2 L, + 1, lane-linear: the LDS image
// IS the window); the descriptor's range check zero-pads a window's END dword by dword, but a NEGATIVE offset drops all
// 16 bytes (tools/membench/gldsprobe.hip), so the y' window must start on an even sample of the pulse: delayMin even,
// checked by the launcher (only segment 0 reaches below the pulse's first sample).
// No whole-register reuse of the y' windows' overlap (REUSE): every y' sample is requested nDelay / segLen more often.
constexpr int RANGEG_WAVES = 16;
typedef __attribute__((address_space(3))) void b2_lds_void;
template <bool SHORTX, bool OUT7>
__global__ __launch_bounds__(64 * RANGEG_WAVES, 4) void rangew1k_glds_kernel(RangeArgs a, InC32 in)
{
  using W = Wave1kFft;
  constexpr int NX = SHORTX ? 9 : 16, PX = SHORTX ? 5 : 8; // x' values per lane; 1 KiB pieces of its window
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *table = reinterpret_cast<cf *>(smem);
  const int t = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  cf *X = table + W::TW_ELEMS + wave * W::X_ELEMS;
  W::fill_table(threadIdx.x, 64 * RANGEG_WAVES, a.tw, table);
  __syncthreads(); // the only barrier: waves without a pulse leave after it
  W::Tw w;
  W::load_twiddles(t, a.tw, table, w);
  const RangePlan p = a.plan;
  const int stride = gridDim.x * RANGEG_WAVES;
  int pulse = blockIdx.x * RANGEG_WAVES + wave;
  if (pulse >= a.nPulses) return;
  int cpi = pulse / p.nDoppler;
  int i = pulse - cpi * p.nDoppler;
  int64_t base = (int64_t)cpi * a.cpiStride + (int64_t)i * p.nCorr;
  int s = 0;
  cf acc[16];
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = cmake(0.f, 0.f);
  for (;;) {
    cf v[16];
    {
      // x' of this segment: sample m of the segment at X[m], zeros from the segment's (or the pulse's) end on
      const int s0 = s * p.segLen;
      const __amdgpu_buffer_rsrc_t d = make_rsrc_b(in.x + base + s0, min(p.segLen, p.nCorr - s0) * 8);
#pragma unroll
      for (int j = 0; j < PX; j++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (b2_lds_void *)(X + 128 * j), 16, 16 * t + 1024 * j, 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < NX; k++) v[k] = X[t + 64 * k];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // in registers before the transform's first exchange overwrites the window
    }
    W::s1<-1, NX>(v, w);
    W::finish<-1>(t, v, w, X); // v = X spectrum; the exchange region is free again
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // ... once the transform's last reads have returned: nothing else orders an LDS-DMA behind them
    cf yv[16];
    {
      // y' of this segment: sample m of the window at X[m]
      const __amdgpu_buffer_rsrc_t d = make_rsrc_b(in.y + base, p.nCorr * 8);
      const int voff = (s * p.segLen + p.delayMin + 2 * t) * 8; // negative below the pulse's first sample: zeros
#pragma unroll
      for (int j = 0; j < 8; j++) {
        // the whole offset in the VGPR, opaque: split into voffset + immediate, a negative voffset is out of range whatever
        // the immediate adds (bufload.hpp) -- the lanes below the pulse's first sample would lose their later pieces
        int vo = voff + 1024 * j;
        asm volatile("" : "+v"(vo));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (b2_lds_void *)(X + 128 * j), 16, vo, 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; k++) yv[k] = X[t + 64 * k];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    W::template transform<-1, 16, false>(t, yv, w, X);
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = cmacc(acc[e], yv[e], v[e]);
    s++;
    if (s == p.nSeg) { // the pulse is complete
      W::template transform<+1, 16, OUT7>(t, acc, w, X);
      store_lags_w<OUT7 ? 7 : 16>(a.out, p, cpi, i, t, acc);
      pulse += stride;
      if (pulse >= a.nPulses) break;
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = cmake(0.f, 0.f);
      cpi = pulse / p.nDoppler;
      i = pulse - cpi * p.nDoppler;
      base = (int64_t)cpi * a.cpiStride + (int64_t)i * p.nCorr;
      s = 0;
    }
  }
}
#endif

// --------------------------------------------------------------------------
// Range kernel for SMALL launches at F = 1024 (a lone CPI, the real-time shape of blah2.cpp:245-289: 513 pulses): the
// one-wave kernel above gives a pulse to ONE wave, its segments in series -- with fewer pulses than wave slots the
// chip is half empty and a pulse takes seven segments' time (41.8 us for a lone CPI at cfg 2); the workgroup kernels
// put a pulse's segments in series too (22 us).  Here a pulse is a WORKGROUP of FOUR waves: wave q takes segments q,
// q + 4, ... (cfg 2, seven segments: two each, the last wave one), exactly the loop body of rangew1k_kernel -- the next
// segment's x' requested as soon as this one's has left its registers, its y' after the spectrum product -- with the
// sum over ITS segments in registers; the four partial sums meet in LDS and the last wave, which had the fewest
// segments, adds them (linearity: the inverse transform of the sum is the sum of the segments' correlations), runs the
// one inverse and stores the lags.  Two barriers per pulse.
//
// Round 4's form had EIGHT waves per pulse (a wave per segment + a finisher), 77 KB of LDS, two workgroups per CU at 128
// VGPRs -- and spilled (12-52 bytes of scratch per lane), and a lone CPI's 513 pulses were one more than its 512
// resident workgroups: the last pulse ran alone after everything else (4 of 16 us).  Four waves need 41.5 KB: THREE
// workgroups per CU at 168 VGPRs (no scratch), 768 resident workgroups, a lone CPI in one round; the work per SIMD is
// the same 3.5 segments.
constexpr int RANGEPS_WAVES = 4;
// timing experiments: -DRANGEPS_ABLATE=1 the kernel's memory operations alone (no transforms),
// =2 its transforms alone (no loads: zero records)
#if defined(RANGEPS_ABLATE) && RANGEPS_ABLATE == 1
#define PS_TR(...)
#else
#define PS_TR(...) __VA_ARGS__
#endif
#if defined(RANGEPS_ABLATE) && RANGEPS_ABLATE == 2
#define PS_LIVE(c) false
#else
#define PS_LIVE(c) (c)
#endif
// (one instantiation -- fp32 planes, long segments, all sixteen outputs of the inverse -- is a register over three waves
// per SIMD; it is built for two: its third workgroup per CU queues)
template <class In, bool SHORTX, bool OUT7>
__global__ __launch_bounds__(64 * RANGEPS_WAVES, (!SHORTX && !OUT7 && sizeof(typename RawBuiltin<typename BufLoad<In>::X>::raw) == 8) ? 2 : 3)
void rangeps_kernel(RangeArgs a, In in)
{
  using W = Wave1kFft;
  using RX = RawBuiltin<typename BufLoad<In>::X>;
  using RY = RawBuiltin<typename BufLoad<In>::Y>;
  constexpr int NX = SHORTX ? 9 : 16;
  constexpr int FIN = RANGEPS_WAVES - 1; // the wave with the fewest segments finishes the pulse
  // sixteen x' loads of eight bytes are 32 registers in flight through both transforms: with them the kernel is 2-4 registers
  // over three waves per SIMD (scratch); that shape requests the next x' with the next y', after the spectrum product
  constexpr bool XLATE = !SHORTX && sizeof(typename RX::raw) == 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *table = reinterpret_cast<cf *>(smem);
  const int t = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  cf *regions = table + W::TW_ELEMS;
  cf *X = regions + wave * W::X_ELEMS;
  W::fill_table(threadIdx.x, 64 * RANGEPS_WAVES, a.tw, table);
  __syncthreads();
  W::Tw w;
  W::load_twiddles(t, a.tw, table, w);
  const RangePlan p = a.plan;
  const int stride = gridDim.x;
  int pulse = blockIdx.x;
  if (pulse >= a.nPulses) return; // the whole workgroup
  int cpi = pulse / p.nDoppler;
  int i = pulse - cpi * p.nDoppler;
  int64_t base = (int64_t)cpi * a.cpiStride + (int64_t)i * p.nCorr;
  // the walk of rangew1k_kernel with a stride of four segments; a wave whose index is beyond the pulse's segments (fewer
  // than four of them) transforms zero records and parks a zero sum: the barriers below are reached by every wave
  int s = wave;
  typename RX::raw rx[NX];
  typename RY::raw ry[16];
  cf acc[16];
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = cmake(0.f, 0.f);
  w1k_issue_x<In, NX>(in, p, base, s, t, PS_LIVE(s < p.nSeg), rx);
  w1k_issue_y<In>(in, p, base, s, t, PS_LIVE(s < p.nSeg), ry);
  for (;;) {
    // the segment after this one (wave-uniform)
    int ns = s + RANGEPS_WAVES, npulse = pulse, ncpi = cpi, ni = i;
    int64_t nbase = base;
    const bool done = ns >= p.nSeg; // this wave's share of the pulse is complete
    if (done) {
      ns = wave;
      npulse = pulse + stride;
      ncpi = npulse / p.nDoppler;
      ni = npulse - ncpi * p.nDoppler;
      nbase = (int64_t)ncpi * a.cpiStride + (int64_t)ni * p.nCorr;
    }
    const bool more = npulse < a.nPulses;
    const bool live = PS_LIVE(more && ns < p.nSeg);
    cf v[16];
#pragma unroll
    for (int k = 0; k < NX; k++) v[k] = RX::cvt(rx[k]);
    PS_TR(W::s1<-1, NX>(v, w));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!XLATE) w1k_issue_x<In, NX>(in, p, nbase, ns, t, live, rx);
    __builtin_amdgcn_sched_barrier(0);
    PS_TR(W::finish<-1>(t, v, w, X)); // v = X spectrum
    cf yin[16], yv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) yin[k] = RY::cvt(ry[k]);
#if defined(RANGEPS_ABLATE) && RANGEPS_ABLATE == 1
#pragma unroll
    for (int k = 0; k < 16; k++) yv[k] = yin[k];
#else
    W::s1_nd<-1>(yv, yin, w);
    W::finish<-1>(t, yv, w, X); // yv = Y spectrum
#endif
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = cmacc(acc[e], yv[e], v[e]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (XLATE) w1k_issue_x<In, NX>(in, p, nbase, ns, t, live, rx);
    w1k_issue_y<In>(in, p, nbase, ns, t, live, ry);
    __builtin_amdgcn_sched_barrier(0);
    if (done) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int e = 0; e < 16; e++) X[e * 64 + t] = acc[e]; // this wave's share of sum_s Y_s conj X_s, parked
      __syncthreads(); // the four partial sums of this pulse are in the regions
      if (wave == FIN) {
#pragma unroll
        for (int q = 0; q < FIN; q++) {
          const cf *r = regions + q * W::X_ELEMS;
#pragma unroll
          for (int e = 0; e < 16; e++) { const cf u = r[e * 64 + t]; acc[e] = cmake(acc[e].x + u.x, acc[e].y + u.y); }
        }
      }
      __syncthreads(); // the regions are free for the next pulse's transforms
      if (wave == FIN) {
        PS_TR(W::template transform<+1, 16, OUT7>(t, acc, w, X));
        store_lags_w<OUT7 ? 7 : 16>(a.out, p, cpi, i, t, acc);
      }
      if (!more) break;
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = cmake(0.f, 0.f);
      pulse = npulse;
      cpi = ncpi;
      i = ni;
    }
    s = ns;
    base = nbase;
  }
}
#undef PS_TR
#undef PS_LIVE

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
// Doppler kernel (Hot loop B, Ambiguity.cpp:152-169): for every delay column
//   D[k] = sum_i R[i] * exp(-2 pi i * i*k / nD)        forward DFT over the pulses
//   M[o] = D[(o + nD/2 + 1) % nD]                      (Ambiguity.cpp:165)
// nD is always odd and rarely smooth (127, 301, 513 = 27*19, 1025, 2049 = 3*683),
// so the DFT is evaluated as a chirp-z (Bluestein) convolution on the power-of-
// two workgroup FFT:   D[k] = c[k] * sum_n (R[n] c[n]) * conj(c[k-n]),
// c[n] = exp(-i pi n^2 / nD).  M = 16*T >= 2*nD - 2 suffices because c is even:
// the only lags that alias, +(nD-1) and -(nD-1), carry the same kernel value.
// The spectrum of the kernel (times 1/M) is precomputed in fp64 on the host, in
// the register layout the forward transform leaves its output in.
//
// Accuracy: the zero-delay columns hold the direct-path peak in every pulse, a
// DC term ~1e2..1e4 times larger than what the other Doppler bins of that
// column contain.  DFT(R - r0)[k] = DFT(R)[k] for k != 0 for ANY constant r0,
// so the first pulse's value is subtracted before the transform (an exact
// fp32 subtraction by Sterbenz for the dominant part) and nD*r0 is added back
// to bin 0.  This removes the fp32 cancellation error of the DC term.
//
// One column per T threads, 256/T columns per workgroup; columns come out of
// the tiled range map with 128-byte-stride gathers (the 16 columns of a tile
// are handled by workgroups that the block->tile map keeps on one XCD, so the
// lines are fetched from HBM once and the partial row writes of the final map
// merge in that XCD's L2).  Epilogue: Map::set_metrics partials.
struct DopplerArgs {
  const cf *R;      // tiled range map
  cf *map;          // [nCpi][nD][nDelay]
  const cf *W;      // direct kernel: exp(-2 pi i k/nD), [nD]
  const cf *tw;     // fft kernel: exp(-2 pi i k/M), [M]
  const cf *chirp;  // fft kernel: exp(-i pi n^2/nD), [nD]
  const cf *bf;     // fft kernel: kernel spectrum / M, [16][T]
  const cf *bfn;    // the same in natural order [M] (M = 2048: doppler_tilew_kernel)
  double *partSum;  // [nCpi][partsPerCpi]
  float *partMax;   // [nCpi][partsPerCpi]
  int32_t nD, nDelay, nTiles;
  // small launches (doppler_sub1k_kernel): Map::set_metrics finished by the CPI's last workgroup, no third launch
  uint32_t *tickets = nullptr; // [nCpi], zero between launches
  double *metrics = nullptr;   // [nCpi][2]
};

// 10*log10|z| = 5*log10(re^2+im^2) = 5*log10(2) * log2(re^2+im^2)
__device__ __forceinline__ float db_of(cf z) { return 1.50514997831990597607f * log2f(z.x * z.x + z.y * z.y); }

// Map::set_metrics partial of one wave: the lanes' (sum of dB values, largest dB value) folded by a butterfly over the
// lane index, every lane ending with the wave's result.  One definition for every kernel that fuses the metrics, so
// the order of the additions -- and with it the bits of noisePower -- is the same whichever Doppler kernel ran.
__device__ __forceinline__ void wave_sum_max(double &lsum, float &lmax)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lsum += __shfl_xor(lsum, off);
    lmax = fmaxf(lmax, __shfl_xor(lmax, off));
  }
}

// sum / max over the 256 threads of a workgroup -> one partial per workgroup
__device__ __forceinline__ void block_metrics_partial(double lsum, float lmax, double *dst_sum, float *dst_max)
{
  __shared__ double wsum[4];
  __shared__ float wmax[4];
  wave_sum_max(lsum, lmax);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { wsum[wave] = lsum; wmax[wave] = lmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    float m = 0.f; // Map.cpp:193: the running max starts at 0
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) { s += wsum[w]; m = fmaxf(m, wmax[w]); }
    *dst_sum = s;
    *dst_max = m;
  }
}

// One column per workgroup of T = 16*R3 threads (for nD <= 513 that is a single
// wave: its barriers cost nothing and the two exchange buffers can alias).
// Epilogue: per-workgroup (sum, max) partial of Map::set_metrics.
template <int R3>
__global__ __launch_bounds__(16 * R3) void doppler_fft_kernel(DopplerArgs a)
{
  using W = WgFft<R3>;
  constexpr int T = W::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  cf *A = reinterpret_cast<cf *>(smem);
  cf *B = (R3 == 4) ? A : A + W::A_ELEMS; // single wave: in-order LDS, no cross-wave hazard
  const int nD = a.nD;
  const int cpi = blockIdx.y;
  // block -> column: the 16 columns of one tile get block ids that are congruent
  // mod 8 (same XCD) and adjacent in dispatch order
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tileIdx = (slot >> 4) * 8 + xcd;
  const int col = tileIdx * 16 + (slot & 15);
  const bool colok = tileIdx < a.nTiles && col < a.nDelay;

  cf tw1[15], tw3[16];
  W::load_twiddles(t, a.tw, tw1, tw3);

  const cf *Rc = a.R + rmap_index(nD, a.nTiles, cpi, 0, colok ? col : 0);
  const cf r0 = Rc[0];
  cf v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int i = t + T * k;
    const cf rv = Rc[(size_t)(i < nD ? i : 0) * 16];
    const cf ch = a.chirp[i < nD ? i : 0];
    v[k] = (i < nD) ? cmul(csub(rv, r0), ch) : cmake(0.f, 0.f);
  }
  W::fwd_s1(t, v, tw1, A);
  __syncthreads();
  W::fwd_s2(t, v, A, B);
  __syncthreads();
  W::fwd_s3(t, v, tw3, B);
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = cmul(v[e], a.bf[e * T + t]);
  __syncthreads();
  W::inv_s1(t, v, tw3, B);
  __syncthreads();
  W::inv_s2(t, v, B, A);
  __syncthreads();
  W::inv_s3(t, v, tw1, A);

  double lsum = 0.0;
  float lmax = 0.f;
  cf *mapc = a.map + (size_t)cpi * nD * a.nDelay + col;
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const int k = t + T * c;
    if (colok && k < nD) {
      cf d = cmul(v[c], a.chirp[k]);
      if (k == 0) d = cmake(d.x + (float)nD * r0.x, d.y + (float)nD * r0.y);
      int o = k - (nD / 2 + 1);
      if (o < 0) o += nD;
      mapc[(size_t)o * a.nDelay] = d;
      const float db = db_of(d);
      lsum += (double)db;
      lmax = fmaxf(lmax, db);
    }
  }
  const int nParts = gridDim.x;
  double *pS = a.partSum + (size_t)cpi * nParts;
  float *pM = a.partMax + (size_t)cpi * nParts;

  // one (sum, max) partial per workgroup; metrics_kernel folds them in index order.
  // (A last-arriver reduction inside this kernel was tried: one ticket atomic per
  // workgroup on a per-CPI counter serialises at ~12 ns each, 5 us per CPI.)
  block_metrics_partial(lsum, lmax, pS + blockIdx.x, pM + blockIdx.x);
}

// Tile variant for nD <= 513 (M = 1024, one wave per column): a workgroup of 16
// waves owns one 16-column tile of the range map.  The tile (nD x 128 bytes,
// contiguous in HBM) is read with fully coalesced loads and transposed through
// LDS, each wave runs its column's chirp-z transform in its own LDS region (the
// staging area of a column IS that wave's exchange buffer, so no cross-wave
// hazard exists after the fill), the results are transposed back through LDS
// and the final map is written as 128-byte row segments.  This replaces the
// 128-byte-stride gathers/scatters of doppler_fft_kernel, which spent 57 % of
// its wave cycles waiting on memory (profiles/r01_pmc.csv).
constexpr int DOPT_PITCH = WgFft<4>::A_ELEMS + 1; // 1089: odd pitch -> 16 columns hit 16 different bank pairs
constexpr int DOPT_CHIRP = 264;                    // HALF the chirp: c[nD - n] = (-1)^nD c[n], n <= nD/2 <= 256
template <int NCOL> constexpr int dopt_lds_elems() { return NCOL * DOPT_PITCH + 1024 + DOPT_CHIRP; }

__device__ __forceinline__ int relaunder(int v);

// NCOL = 16: whole 128-byte lines per row, 147 KB of LDS, one workgroup per CU;
// NCOL = 8: half lines (the sibling workgroup takes the other half out of L2),
// 75 KB, two workgroups per CU so that one's memory phases overlap the other's math.
// Workgroups are PERSISTENT (grid = resident workgroups, each walks its tiles): twiddles (31
// gathered loads per thread), chirp and kernel spectrum are set up once per workgroup instead of
// once per tile, the next tile's loads are issued as soon as this tile's column is in registers and
// land during the transforms, and a tile's row stores drain while the next tile is filled.  The
// kernel spectrum and the chirp live in LDS (nothing queues behind the tile loads: loads return
// in order; the chirp as its first half only -- exp(-i pi n^2/nD) at nD - n is (-1)^nD times its value
// at n -- because two 8-column workgroups per CU leave 3.8 KB), per-phase address arithmetic is
// recomputed instead of carried across the loop.
template <int NCOL>
__global__ __launch_bounds__(64 * NCOL) void doppler_tile_kernel(DopplerArgs a, int nCpi)
{
  using W = WgFft<4>;
  constexpr int T = 64;
  constexpr int NR = 9; // nD <= 513: rows t + 64*k, k < 9, are the only ones that exist
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double wsum[16];
  __shared__ float wmax[16];
  cf *lds = reinterpret_cast<cf *>(smem);
  const int tid = threadIdx.x;
  const int w = tid >> 6, t = tid & 63; // wave = column of the tile
  const int nD = a.nD;
  constexpr int NT = 64 * NCOL;     // threads
  constexpr int SH = (NCOL == 16) ? 4 : 3;
  cf *region = lds + w * DOPT_PITCH;
  cf *bfL = lds + NCOL * DOPT_PITCH;
  cf *chirpL = bfL + 1024;
  const int tilesPerCpi = (a.nDelay + NCOL - 1) / NCOL;
  const int nTilesAll = tilesPerCpi * nCpi;
  const int cells = nD * NCOL;

  // once per workgroup: tables
#pragma unroll
  for (int j = 0; j < 1024 / NT; j++) bfL[tid + NT * j] = a.bf[tid + NT * j];
  for (int i = tid; i < DOPT_CHIRP; i += NT) chirpL[i] = a.chirp[min(i, nD - 1)];
  const float csign = (nD & 1) ? -1.f : 1.f;
  // chirp value of row n < nD from the half table
  auto chirp_at = [&](int n) {
    const bool hi = 2 * n > nD;
    const cf c = chirpL[hi ? nD - n : n];
    return cmake(hi ? csign * c.x : c.x, hi ? csign * c.y : c.y);
  };
  cf tw1[15], tw3[16];
  W::load_twiddles(t, a.tw, tw1, tw3);

  // coalesced read of one (half) tile: all loads in flight before the first use
  cf nt[NR];
  auto tile_load = [&](int it) {
    const int cpi = it / tilesPerCpi, sub = it - cpi * tilesPerCpi;
    const cf *Rt = a.R + rmap_index(nD, a.nTiles, cpi, 0, sub * NCOL);
    const int tl = relaunder(tid);
#pragma unroll
    for (int j = 0; j < NR; j++) {
      const int idx = tl + NT * j;
      const int c = idx & (NCOL - 1), row = idx >> SH;
      nt[j] = Rt[idx < cells ? row * 16 + c : 0];
    }
  };
  int it = blockIdx.x;
  if (it < nTilesAll) tile_load(it);
  for (; it < nTilesAll; it += gridDim.x) {
    const int cpi = it / tilesPerCpi, sub = it - cpi * tilesPerCpi;
    const int col0 = sub * NCOL;
    // phase 1: the tile, transposed into the per-column regions
    {
      const int tl = relaunder(tid);
#pragma unroll
      for (int j = 0; j < NR; j++) {
        const int idx = tl + NT * j;
        const int c = idx & (NCOL - 1), row = idx >> SH;
        if (idx < cells) lds[c * DOPT_PITCH + row] = nt[j];
      }
    }
    __syncthreads();

    // phase 2: this wave's column -> registers (DC removal + chirp), next tile's loads, then the
    // transform.  From here on `region` is this wave's private exchange buffer (A and B alias: a
    // single wave's LDS operations execute in order; __builtin_amdgcn_wave_barrier() is no
    // instruction, it only stops the compiler from moving LDS accesses across the stage boundaries)
    cf v[16];
    const cf r0 = region[0];
    {
      const int t2 = relaunder(t);
#pragma unroll
      for (int k = 0; k < NR; k++) {
        const int i = t2 + T * k;
        const cf rv = region[min(i, nD - 1)];
        const cf p = cmul(csub(rv, r0), chirp_at(min(i, nD - 1)));
        v[k] = cmake(i < nD ? p.x : 0.f, i < nD ? p.y : 0.f);
      }
    }
#pragma unroll
    for (int k = NR; k < 16; k++) v[k] = cmake(0.f, 0.f);
    if (it + (int)gridDim.x < nTilesAll) tile_load(it + gridDim.x);
    __builtin_amdgcn_wave_barrier();
    W::fwd_s1(t, v, tw1, region);
    __builtin_amdgcn_wave_barrier();
    W::fwd_s2_load(t, v, region);
    __builtin_amdgcn_wave_barrier();
    dft16<-1>(v);
    W::fwd_s2_store(t, v, region);
    __builtin_amdgcn_wave_barrier();
    W::fwd_s3(t, v, tw3, region);
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = cmul(v[e], bfL[e * T + t]);
    __builtin_amdgcn_wave_barrier();
    W::inv_s1(t, v, tw3, region);
    __builtin_amdgcn_wave_barrier();
    W::inv_s2_load(t, v, region);
    __builtin_amdgcn_wave_barrier();
    dft16<+1>(v);
    W::inv_s2_store(t, v, region);
    __builtin_amdgcn_wave_barrier();
    W::inv_s3(t, v, tw1, region);
    __builtin_amdgcn_wave_barrier();

    // phase 3: rotate rows by nD/2+1 and park the column back in its region
    {
      const int t3 = relaunder(t);
#pragma unroll
      for (int c = 0; c < NR; c++) {
        const int k = t3 + T * c;
        cf d = cmul(v[c], chirp_at(min(k, nD - 1)));
        if (c == 0 && t3 == 0) d = cmake(d.x + (float)nD * r0.x, d.y + (float)nD * r0.y);
        int o = k - (nD / 2 + 1);
        if (o < 0) o += nD;
        region[k < nD ? o : DOPT_PITCH - 1] = d; // rows beyond nD go to a spare slot: no branch per row
      }
    }
    __syncthreads();

    // phase 4: coalesced row-segment stores + Map::set_metrics partials
    double lsum = 0.0;
    float lmax = 0.f;
    cf *mapb = a.map + (size_t)cpi * nD * a.nDelay + col0;
    const int ncol = min(NCOL, a.nDelay - col0);
    {
      const int tl = relaunder(tid);
#pragma unroll
      for (int j = 0; j < NR; j++) {
        const int idx = tl + NT * j;
        const int c = idx & (NCOL - 1), o = idx >> SH;
        const bool ok = idx < cells && c < ncol;
        const cf d = lds[c * DOPT_PITCH + min(o, nD - 1)];
        if (ok) mapb[(size_t)o * a.nDelay + c] = d;
        const float db = db_of(d);
        lsum += ok ? (double)db : 0.0;
        lmax = ok ? fmaxf(lmax, db) : lmax;
      }
    }
    wave_sum_max(lsum, lmax);
    if (t == 0) { const int wl = relaunder(tid) >> 6; wsum[wl] = lsum; wmax[wl] = lmax; }
    __syncthreads(); // also: every thread has taken its rows out of the regions
    if (tid == 0) {
      double sacc = 0.0;
      float m = 0.f; // Map.cpp:193: the running max starts at 0
      for (int i = 0; i < NCOL; i++) { sacc += wsum[i]; m = fmaxf(m, wmax[i]); }
      const size_t part = (size_t)cpi * tilesPerCpi + sub;
      a.partSum[part] = sacc;
      a.partMax[part] = m;
    }
  }
}

// doppler_tile_kernel<16> on the one-wave 1024-point transform of fft_wave1k.hpp (round 3).  Same phases, same tile, same
// results to rounding; what changes is what a thread carries.  The workgroup transform keeps its 31 stage twiddles in 62
// registers per thread, which at four waves per SIMD (a 1024-thread workgroup) leaves room for nothing else: chirp
// values, rotation targets and transposition addresses were recomputed from lane indices in every phase of every tile
// -- 670 of the 1240 vector instructions per column were index arithmetic and selects, on a kernel whose SIMDs are 68 %
// busy.  Wave1kFft reads its stage twiddles from a 7.5 KB table in LDS (fits beside the sixteen regions: 157 KB), so the
// per-lane constants of the walk live in registers across the tiles: the chirp of this lane's nine rows (zero for rows
// beyond nD, which also zeroes the padding rows without a select) and where each output row goes after the rotation by
// nD/2 + 1.  One exchange per transform instead of two; the kernel spectrum in natural order (a.bfn).
// NCOL = 8 (round 5): HALF tiles, 512 threads.  8 regions + the stage-twiddle table are 75.6 KB; the kernel spectrum is the
// transform of an EVEN sequence (b[m] = b[1024 - m], capi.hip), so it is even itself and entries 0 ... 512 (4 KB) serve
// all 1024: 79.7 KB, TWO workgroups per CU, and the barrier-separated phases of one (transposes, row stores) run under
// the transforms of the other; the sibling half of a tile is read by the next workgroup of the walk (same 128-byte lines).
template <int NCOL> constexpr int dopt1k_lds_elems() { return NCOL * DOPT_PITCH + Wave1kFft::TW_ELEMS + (NCOL == 16 ? 1024 : 513); }
constexpr int DOPT1K_LDS_ELEMS = dopt1k_lds_elems<16>();
template <int NCOL> __global__ __launch_bounds__(64 * NCOL, 4) void doppler_tile1k_kernel(DopplerArgs a, int nCpi)
{
#ifdef DOPW_TRACE
  uint64_t tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime();
#define D1_T(k) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); tr[k] += now_ - t0_; t0_ = now_; }
#else
#define D1_T(k)
#endif
  using K = Wave1kFft;
  static_assert(NCOL == 16 || NCOL == 8, "");
  constexpr int T = 64, NR = 9, NT = 64 * NCOL, SH = NCOL == 16 ? 4 : 3;
  constexpr bool BF_FULL = NCOL == 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double wsum[NCOL];
  __shared__ float wmax[NCOL];
  cf *lds = reinterpret_cast<cf *>(smem);
  const int tid = threadIdx.x;
  const int w = tid >> 6, t = tid & 63; // wave = column of the tile
  const int nD = a.nD;
  cf *region = lds + w * DOPT_PITCH;
  cf *table = lds + NCOL * DOPT_PITCH;
  cf *bfL = table + K::TW_ELEMS;
  const int tilesPerCpi = (a.nDelay + NCOL - 1) / NCOL;
  const int nTilesAll = tilesPerCpi * nCpi;
  const int cells = nD * NCOL;

  // once per workgroup: tables, and this lane's constants
  K::fill_table(tid, NT, a.tw, table);
  if (BF_FULL || tid <= 512) bfL[tid] = a.bfn[tid]; // NCOL = 8: 512 threads -> entries 0 ... 511, and 512 below
  if (!BF_FULL && tid == 0) bfL[512] = a.bfn[512];
  K::Tw tw;
  K::load_twiddles(t, a.tw, table, tw);
  cf ch[NR];  // chirp of row t + 64 k; 0 beyond nD
  int oidx[NR]; // where output row t + 64 k goes in the region: rotated by nD/2 + 1, or the spare slot
  int ridx[NR]; // the input row, clamped (nD < 513: rows of the 9 x 64 that do not exist)
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const int i = t + T * k;
    ridx[k] = min(i, nD - 1);
    const cf c = a.chirp[min(i, nD - 1)];
    ch[k] = cmake(i < nD ? c.x : 0.f, i < nD ? c.y : 0.f);
    int o = i - (nD / 2 + 1);
    if (o < 0) o += nD;
    oidx[k] = i < nD ? o : DOPT_PITCH - 1;
  }

  cf nt[NR];
  // NCOL = 8: the last four of a thread's nine cells are requested at the top of the tile's own iteration, not a tile
  // ahead -- held through the transforms they were the registers that spilled (and a spill behind a load waits for it)
  constexpr int PRE = NCOL == 16 ? NR : NR - 4;
  auto tile_load = [&](int it, int j0, int j1) {
    const int cpi = it / tilesPerCpi, sub = it - cpi * tilesPerCpi;
    const cf *Rt = a.R + rmap_index(nD, a.nTiles, cpi, 0, sub * NCOL);
    const int tl = relaunder(tid);
#pragma unroll
    for (int j = j0; j < j1; j++) {
      const int idx = tl + NT * j; // the 16-column tile is contiguous: cell = row * 16 + column (a half tile: 64-byte row pieces)
      nt[j] = Rt[idx < cells ? (NCOL == 16 ? idx : (idx >> SH) * 16 + (idx & (NCOL - 1))) : 0];
    }
  };
  int it = blockIdx.x;
  if (it < nTilesAll) tile_load(it, 0, PRE);
  __syncthreads(); // tables
  for (; it < nTilesAll; it += gridDim.x) {
    const int cpi = it / tilesPerCpi, sub = it - cpi * tilesPerCpi;
    const int col0 = sub * NCOL;
    // phase 1: the tile, transposed into the per-column regions
    if constexpr (PRE < NR) tile_load(it, PRE, NR);
    {
      const int tl = relaunder(tid);
      cf *dst = lds + (tl & (NCOL - 1)) * DOPT_PITCH + (tl >> SH); // idx + 1024 j: same column, row + 64 j
#pragma unroll
      for (int j = 0; j < NR; j++)
        if (tl + NT * j < cells) dst[T * j] = nt[j];
    }
    D1_T(0)
    __syncthreads();
    D1_T(1)

    // phase 2: this wave's column -> registers (DC removal + chirp), next tile's loads, both transforms
    cf v[16];
    const cf r0 = region[0];
#pragma unroll
    for (int k = 0; k < NR; k++) // rows beyond nD: a valid cell times 0 (NCOL = 8 recomputes the clamped row: nine registers it needs elsewhere)
      v[k] = cmul(csub(region[NCOL == 16 ? ridx[k] : min(t + T * k, nD - 1)], r0), ch[k]);
    if (it + (int)gridDim.x < nTilesAll) tile_load(it + gridDim.x, 0, PRE);
    __builtin_amdgcn_wave_barrier();
    D1_T(2)
    K::transform<-1, 9>(t, v, tw, region);
    D1_T(3)
    if constexpr (BF_FULL) {
#pragma unroll
      for (int e = 0; e < 16; e++) v[e] = cmul(v[e], bfL[e * T + t]);
    } else { // B[m] = B[1024 - m]: entries 0 ... 512 only
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = cmul(v[e], bfL[e * T + t]);
#pragma unroll
      for (int e = 8; e < 16; e++) v[e] = cmul(v[e], bfL[(16 - e) * T - t]);
    }
    __builtin_amdgcn_wave_barrier();
    D1_T(4)
    K::transform<+1>(t, v, tw, region);
    __builtin_amdgcn_wave_barrier();
    D1_T(3)

    // phase 3: chirp, rotate rows by nD/2 + 1, park the column back in its region
    {
#pragma unroll
      for (int c = 0; c < NR; c++) {
        cf d = cmul(v[c], ch[c]);
        if (c == 0 && t == 0) d = cmake(d.x + (float)nD * r0.x, d.y + (float)nD * r0.y);
        if constexpr (NCOL == 16) {
          region[oidx[c]] = d;
        } else { // recomputed (registers): rotated by nD/2 + 1, rows that do not exist go to the spare slot
          const int i = t + T * c;
          int o = i - (nD / 2 + 1);
          o = o < 0 ? o + nD : o;
          region[i < nD ? o : DOPT_PITCH - 1] = d;
        }
      }
    }
    D1_T(5)
    __syncthreads();
    D1_T(1)

    // phase 4: coalesced row-segment stores + Map::set_metrics partials
    double lsum = 0.0;
    float lmax = 0.f;
    cf *mapb = a.map + (size_t)cpi * nD * a.nDelay + col0;
    const int ncol = min(NCOL, a.nDelay - col0);
    {
      const int tl = relaunder(tid);
      const int c = tl & (NCOL - 1), o0 = tl >> SH;
      const cf *src = lds + c * DOPT_PITCH + o0;
      cf *dstg = mapb + (size_t)o0 * a.nDelay + c;
      const size_t gstep = (size_t)T * a.nDelay;
#pragma unroll
      for (int j = 0; j < NR; j++) {
        const bool ok = tl + NT * j < cells && c < ncol;
        const cf d = src[min(T * j, nD - 1 - o0)];
        if (ok) dstg[gstep * j] = d;
        const float db = db_of(d);
        lsum += ok ? (double)db : 0.0;
        lmax = ok ? fmaxf(lmax, db) : lmax;
      }
    }
    wave_sum_max(lsum, lmax);
    if (t == 0) { const int wl = relaunder(tid) >> 6; wsum[wl] = lsum; wmax[wl] = lmax; }
    D1_T(6)
    __syncthreads(); // also: every thread has taken its rows out of the regions
    if (tid == 0) {
      double sacc = 0.0;
      float m = 0.f; // Map.cpp:193: the running max starts at 0
      for (int i = 0; i < NCOL; i++) { sacc += wsum[i]; m = fmaxf(m, wmax[i]); }
      const size_t part = (size_t)cpi * tilesPerCpi + sub;
      a.partSum[part] = sacc;
      a.partMax[part] = m;
    }
    D1_T(1)
  }
#ifdef DOPW_TRACE // buckets: fill, barriers, column read + next tile's requests, transforms, kernel-spectrum product, park, stores + metrics
  if (t == 0) trace_finish("dop1k", tr, blockIdx.x == 0);
#endif
}
#undef D1_T

// The reduction of metrics_kernel (below) by the 256 threads of a workgroup: the same order of operations, so that the
// fused finish and the separate launch give the same bits.
__device__ __forceinline__ void metrics_finish_256(const double *partSum, const float *partMax, int nParts, int cpi, double cells,
                                                   double *metrics, double *ssum, float *smax)
{
  const int tid = threadIdx.x;
  double s = 0.0;
  float m = 0.f;
  for (int i = tid; i < nParts; i += 256) { // the partials were stored write-through (sc1): L2-served loads see them, no fence
    s += __hip_atomic_load(partSum + (size_t)cpi * nParts + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    m = fmaxf(m, __hip_atomic_load(partMax + (size_t)cpi * nParts + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  ssum[tid] = s;
  smax[tid] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      ssum[tid] += ssum[tid + off];
      smax[tid] = fmaxf(smax[tid], smax[tid + off]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double noise = ssum[0] / cells;
    metrics[2 * cpi + 0] = noise;
    metrics[2 * cpi + 1] = (double)smax[0] - noise;
  }
}

// The Doppler stage of a SMALL launch (a single CPI, the real-time shape of blah2.cpp:245-289; nD <= 513): the 16-column
// tile kernel above keeps a CU busy for ~12 us per tile and a lone CPI has 26 tiles, the one-column-per-workgroup kernel
// runs five barrier-separated phases per column behind 128-byte-stride gathers (16 us).  Here a workgroup is FOUR waves
// = four columns (a quarter of a 16-column tile: 32-byte row pieces, the other three quarters are read by its siblings
// out of the same lines), one wave per SIMD, no persistence: 103 workgroups at cfg 2, every CU busy, each wave alone
// on its SIMD through both one-wave 1024-point transforms.  Map::set_metrics' partials are finished by the CPI's LAST
// workgroup (one ticket per workgroup behind write-through partials, MI355X_MICROARCH.md "inter-workgroup visibility"),
// so the chain of a lone CPI is two launches instead of three.
constexpr int DOPS_NCOL = 4;
constexpr int DOPS_LDS_ELEMS = DOPS_NCOL * DOPT_PITCH + Wave1kFft::TW_ELEMS + 1024;
__global__ __launch_bounds__(64 * DOPS_NCOL) void doppler_sub1k_kernel(DopplerArgs a, int nCpi)
{
#ifdef DOPW_TRACE
  uint64_t tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime();
#define DS_T(k) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); tr[k] += now_ - t0_; t0_ = now_; }
#else
#define DS_T(k)
#endif
  using K = Wave1kFft;
  constexpr int NCOL = DOPS_NCOL, T = 64, NR = 9, NT = 64 * NCOL, SH = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double ssum[256];
  __shared__ float smax[256];
  __shared__ int lastWg;
  cf *lds = reinterpret_cast<cf *>(smem);
  const int tid = threadIdx.x;
  const int w = tid >> 6, t = tid & 63; // wave = column of the sub-tile
  const int nD = a.nD;
  cf *region = lds + w * DOPT_PITCH;
  cf *table = lds + NCOL * DOPT_PITCH;
  cf *bfL = table + K::TW_ELEMS;
  const int subsPerCpi = (a.nDelay + NCOL - 1) / NCOL;
  const int cells = nD * NCOL;
  const int it = blockIdx.x;
  const int cpi = it / subsPerCpi, sub = it - cpi * subsPerCpi;
  if (cpi >= nCpi) return;
  const int col0 = sub * NCOL;

  // the sub-tile: rows of four columns out of the 16-column tile's rows (requested before the tables are filled)
  cf nt[NR];
  {
    const cf *Rt = a.R + rmap_index(nD, a.nTiles, cpi, 0, col0);
#pragma unroll
    for (int j = 0; j < NR; j++) {
      const int idx = tid + NT * j;
      const int c = idx & (NCOL - 1), row = idx >> SH;
      nt[j] = Rt[idx < cells ? row * 16 + c : 0];
    }
  }
  K::fill_table(tid, NT, a.tw, table);
#pragma unroll
  for (int j = 0; j < 1024 / NT; j++) bfL[tid + NT * j] = a.bfn[tid + NT * j];
  cf ch[NR];
  int oidx[NR], ridx[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const int i = t + T * k;
    ridx[k] = min(i, nD - 1);
    const cf c = a.chirp[min(i, nD - 1)];
    ch[k] = cmake(i < nD ? c.x : 0.f, i < nD ? c.y : 0.f);
    int o = i - (nD / 2 + 1);
    if (o < 0) o += nD;
    oidx[k] = i < nD ? o : DOPT_PITCH - 1;
  }
  // phase 1: the sub-tile, transposed into the per-column regions
  {
    cf *dst = lds + (tid & (NCOL - 1)) * DOPT_PITCH + (tid >> SH); // idx + 256 j: same column, row + 64 j
#pragma unroll
    for (int j = 0; j < NR; j++)
      if (tid + NT * j < cells) dst[T * j] = nt[j];
  }
  DS_T(0)
  __syncthreads(); // tables and tile
  DS_T(1)
  K::Tw tw;
  K::load_twiddles(t, a.tw, table, tw);

  // phase 2: this wave's column (DC removal + chirp), both transforms
  cf v[16];
  const cf r0 = region[0];
#pragma unroll
  for (int k = 0; k < NR; k++) v[k] = cmul(csub(region[ridx[k]], r0), ch[k]);
  __builtin_amdgcn_wave_barrier();
  DS_T(2)
  K::transform<-1, 9>(t, v, tw, region);
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = cmul(v[e], bfL[e * T + t]);
  __builtin_amdgcn_wave_barrier();
  K::transform<+1>(t, v, tw, region);
  __builtin_amdgcn_wave_barrier();
  DS_T(3)

  // phase 3: chirp, rotate rows by nD/2 + 1, park the column back in its region
#pragma unroll
  for (int c = 0; c < NR; c++) {
    cf d = cmul(v[c], ch[c]);
    if (c == 0 && t == 0) d = cmake(d.x + (float)nD * r0.x, d.y + (float)nD * r0.y);
    region[oidx[c]] = d;
  }
  DS_T(4)
  __syncthreads();
  DS_T(1)

  // phase 4: row-piece stores (32 bytes) + Map::set_metrics partials
  double lsum = 0.0;
  float lmax = 0.f;
  cf *mapb = a.map + (size_t)cpi * nD * a.nDelay + col0;
  const int ncol = min(NCOL, a.nDelay - col0);
  {
    const int c = tid & (NCOL - 1), o0 = tid >> SH;
    const cf *src = lds + c * DOPT_PITCH + o0;
    cf *dstg = mapb + (size_t)o0 * a.nDelay + c;
    const size_t gstep = (size_t)T * a.nDelay;
#pragma unroll
    for (int j = 0; j < NR; j++) {
      const bool ok = tid + NT * j < cells && c < ncol;
      const cf d = src[min(T * j, nD - 1 - o0)];
      if (ok) dstg[gstep * j] = d;
      const float db = db_of(d);
      lsum += ok ? (double)db : 0.0;
      lmax = ok ? fmaxf(lmax, db) : lmax;
    }
  }
  wave_sum_max(lsum, lmax);
  if (t == 0) { ssum[w] = lsum; smax[w] = lmax; }
  DS_T(5)
  __syncthreads();
  DS_T(1)
  if (tid == 0) {
    double sacc = 0.0;
    float m = 0.f; // Map.cpp:193: the running max starts at 0
    for (int i = 0; i < NCOL; i++) { sacc += ssum[i]; m = fmaxf(m, smax[i]); }
    const size_t part = (size_t)cpi * subsPerCpi + sub;
    // write-through (sc1) stores, the store queue drained, then the ticket: the form of the guide's hand-off that needs no
    // release fence here (a fence writes back every dirty line of the XCD's L2 -- the map this launch has just stored)
    // and no acquire on the reading side, whose loads are sc1 too
    __hip_atomic_store(a.partSum + part, sacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.partMax + part, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int last = 0;
    if (a.tickets) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned tk = __hip_atomic_fetch_add(a.tickets + cpi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = tk == (unsigned)subsPerCpi - 1u;
      if (last) __hip_atomic_store(a.tickets + cpi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // for the next launch
    }
    lastWg = last;
  }
  __syncthreads();
  if (lastWg) metrics_finish_256(a.partSum, a.partMax, subsPerCpi, cpi, (double)nD * (double)a.nDelay, a.metrics, ssum, smax);
  DS_T(6)
#ifdef DOPW_TRACE // buckets: requests + tables + tile fill, barriers, column read, transforms + product, park, stores + metrics, ticket + finish
  if (t == 0) trace_finish("dops", tr, blockIdx.x == 0);
#endif
}
#undef DS_T

// Tile variant for 513 < nD <= 1025 on the ONE-WAVE 2048-point transform (fft_wave.hpp): the phases of
// doppler_tile_kernel<8> with one wave per column doing both transforms of the chirp-z convolution in
// its own exchange region (which is also the column's staging area), no barrier between the tile
// fill and the write-back.  Eight columns per 512-thread workgroup; LDS = the stage-twiddle table +
// 8 regions = 148 KB, so ONE workgroup per CU (two waves per SIMD) -- a workgroup is therefore
// PERSISTENT and software-pipelined over its tiles: the next tile's loads are issued (into 17
// registers per thread) right after this tile's column has been taken out of LDS and land during the
// two transforms; this tile's row stores drain while the next tile is filled.  The kernel spectrum
// (natural order, 16 KB, shared by every workgroup) comes from L2 right before its use.
// Replaces doppler_tilem_kernel<8> (two-wave columns, workgroup barriers at every stage, one tile
// per workgroup) in the automatic choice; numbers in DESIGN.md.
constexpr int DOPW_NCOL = 8;
constexpr int DOPW_MAX_ND = 1025;
// region stride = 2 (mod 8): the transposing accesses of phases 1 and 4 (a 16-lane group touches 4
// column pairs x 4 rows, one column of each pair per instruction) hit 16 distinct bank pairs
// (at the stride 2112 = 0 (mod 16) all columns fell on the same banks)
constexpr int DOPW_RS = WaveFft::X_ELEMS + 2;
static_assert(DOPW_RS % 8 == 2, "");
constexpr int DOPW_CHIRP_ELEMS = 1088; // rows t + 64*k, k < 17
constexpr int DOPW_LDS_ELEMS = WaveFft::TW_ELEMS + DOPW_NCOL * DOPW_RS + DOPW_CHIRP_ELEMS;
__global__ __launch_bounds__(64 * DOPW_NCOL, 2) void doppler_tilew_kernel(DopplerArgs a, int nCpi)
{
#ifdef DOPW_TRACE
  uint64_t tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime();
#define DW_T(k) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); tr[k] += now_ - t0_; t0_ = now_; }
#else
#define DW_T(k)
#endif
  using W = WaveFft;
  constexpr int NCOL = DOPW_NCOL, NT = 64 * NCOL, SH = 3;
  constexpr int NR = 17;  // rows t + 64*k, k < 17, cover nD <= 1025 + 62
  constexpr int NRP = 9;  // tile cell PAIRS (two neighbouring columns, 16 bytes) per thread: nD * 4 / 512 <= 8.01
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double wsum[NCOL];
  __shared__ float wmax[NCOL];
  cf *table = reinterpret_cast<cf *>(smem);
  cf *regions = table + W::TW_ELEMS;
  const int tid = threadIdx.x;
  const int w = tid >> 6, t = tid & 63; // wave = column of the half tile
  const int nD = a.nD;
  const int tilesPerCpi = (a.nDelay + NCOL - 1) / NCOL;
  const int nTilesAll = tilesPerCpi * nCpi;
  cf *region = regions + w * DOPW_RS;

  // the chirp (the same for every column) lives in LDS: 34 registers less across the loop
  cf *chirpL = regions + NCOL * DOPW_RS;
  // zero beyond nD: the padding rows of a column (which the tile fill leaves at zero, see below) then need no select
  for (int i = tid; i < DOPW_CHIRP_ELEMS; i += NT) {
    const cf c = a.chirp[min(i, nD - 1)];
    chirpL[i] = cmake(i < nD ? c.x : 0.f, i < nD ? c.y : 0.f);
  }
  W::fill_table(tid, NT, a.tw, table);
  W::Tw tw;
  W::load_twiddles(t, a.tw, table, tw);

  // coalesced read of a half tile (NCOL columns x nD pulses, 64 bytes per pulse row)
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 nt[NRP];
  const int pairs = nD * (NCOL / 2);
  // Through a buffer descriptor over the half tile's nD row pieces (64 of every 128 bytes): pair idx = tid + 512 j sits
  // 16 KiB behind pair tid + 512 (j - 1), so the loads differ only in soffset; pieces of rows >= nD are beyond the
  // descriptor and read as zeros -- which is what the padding rows of the columns have to hold
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  auto tile_load = [&](int it) {
    const int cpi = it / tilesPerCpi, sub = it - cpi * tilesPerCpi;
    const cf *Rt = a.R + rmap_index(nD, a.nTiles, cpi, 0, sub * NCOL);
    const __amdgpu_buffer_rsrc_t d = make_rsrc_b(Rt, (nD - 1) * 128 + NCOL * 8);
    const int tl = relaunder(tid);
    const int voff = (tl >> (SH - 1)) * 128 + (tl & (NCOL / 2 - 1)) * 16;
#pragma unroll
    for (int j = 0; j < NRP; j++) {
      const u4 r = __builtin_amdgcn_raw_buffer_load_b128(d, voff, j * (NT >> (SH - 1)) * 128, 0);
      nt[j] = f4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
    }
  };
  int it = blockIdx.x;
  if (it < nTilesAll) tile_load(it);
  for (; it < nTilesAll; it += gridDim.x) {
    const int cpi = it / tilesPerCpi, sub = it - cpi * tilesPerCpi;
    const int col0 = sub * NCOL;
    // phase 1: the tile, transposed into the per-column regions (the previous tile's row stores
    // have read the regions: barrier at the end of the loop body)
    {
      const int tl = relaunder(tid);
      cf *dst = regions + (2 * (tl & (NCOL / 2 - 1))) * DOPW_RS + (tl >> (SH - 1)); // pair idx + 512 j: same columns, row + 128 j
#pragma unroll
      for (int j = 0; j < NRP; j++) { // every row up to 9 * 128 - 1, the zeros of rows >= nD included: no predicate
        dst[(NT >> (SH - 1)) * j] = cmake(nt[j].x, nt[j].y);
        dst[DOPW_RS + (NT >> (SH - 1)) * j] = cmake(nt[j].z, nt[j].w);
      }
    }
    DW_T(0)
    __syncthreads();
    DW_T(1)

    // phase 2: this wave's column -> registers (DC removal + chirp); then the next tile's loads go
    // out; transform, x kernel spectrum, inverse.  From here to the next barrier `region` is this
    // wave's private exchange buffer (a single wave's LDS operations execute in order).
    cf v[32];
    const cf r0 = region[0];
    const int t2 = relaunder(t);
#pragma unroll
    for (int k = 0; k < NR; k++) v[k] = cmul(csub(region[t2 + 64 * k], r0), chirpL[t2 + 64 * k]); // rows >= nD: (0 - r0) * 0
#pragma unroll
    for (int k = NR; k < 32; k++) v[k] = cmake(0.f, 0.f);
    // The kernel spectrum (16 KB, L2-resident) is requested here and lands during the forward
    // transform; the next tile is requested only AFTER the spectrum has been used and lands during the
    // inverse transform: loads return in order, so a spectrum load behind the tile's HBM loads waits
    // for the tile (measured: 36 % of the kernel), and both sets in flight at once do not fit the
    // registers.  Fetched every time: held across the loop the spectrum would cost 64 registers.
    cf bf[32];
    {
      const cf *bfp = a.bfn + relaunder(t);
#pragma unroll
      for (int e = 0; e < 32; e++) bf[e] = bfp[64 * e];
    }
    __builtin_amdgcn_wave_barrier();
    DW_T(2)
    W::transform<-1>(t, v, tw, region);
    DW_T(3)
#pragma unroll
    for (int e = 0; e < 32; e++) v[e] = cmul(v[e], bf[e]);
    if (it + (int)gridDim.x < nTilesAll) tile_load(it + gridDim.x);
    __builtin_amdgcn_wave_barrier();
    DW_T(4)
    W::transform<+1>(t, v, tw, region);
    __builtin_amdgcn_wave_barrier();
    DW_T(3)

    // phase 3: rotate rows by nD/2+1 and park the column back in its region
    const int t3 = relaunder(t); // the 17 rotated row indices are recomputed, not carried across the loop
#pragma unroll
    for (int c = 0; c < NR; c++) {
      const int k = t3 + 64 * c;
      cf d = cmul(v[c], chirpL[k]);
      if (c == 0 && t3 == 0) d = cmake(d.x + (float)nD * r0.x, d.y + (float)nD * r0.y);
      int o = k - (nD / 2 + 1);
      if (o < 0) o += nD;
      region[k < nD ? o : DOPW_RS - 1] = d; // rows beyond nD go to a spare slot: no branch per row
    }
    DW_T(5)
    __syncthreads();
    DW_T(1)

    // phase 4: coalesced row-segment stores + Map::set_metrics partials
    double lsum = 0.0;
    float lmax = 0.f;
    cf *mapb = a.map + (size_t)cpi * nD * a.nDelay + col0;
    const int ncol = min(NCOL, a.nDelay - col0);
    const int tl4 = relaunder(tid);
    const bool wide = (a.nDelay & 1) == 0; // 16-byte row pieces need the rows to start 16-byte aligned
    {
      // stores through a descriptor over the CPI's map: rows >= nD fall behind its end, columns beyond the map's last
      // one (the last tile of a row) get offset -1; pair idx + 512 j is 128 rows further down: soffset
      const __amdgpu_buffer_rsrc_t md = make_rsrc_b(a.map + (size_t)cpi * nD * a.nDelay, nD * a.nDelay * 8);
      const int pc = tl4 & (NCOL / 2 - 1), o0 = tl4 >> (SH - 1);
      const cf *src = regions + (2 * pc) * DOPW_RS + o0;
      const bool c0 = 2 * pc < ncol, c1 = 2 * pc + 1 < ncol;
      const int off = (o0 * a.nDelay + col0 + 2 * pc) * 8;
      const int rstep = (NT >> (SH - 1)) * a.nDelay * 8;
#pragma unroll
      for (int j = 0; j < NRP; j++) {
        const cf d0 = src[(NT >> (SH - 1)) * j], d1 = src[DOPW_RS + (NT >> (SH - 1)) * j];
        if (wide) { // ncol is even with nDelay: both columns or none
          const u4 q = {__float_as_uint(d0.x), __float_as_uint(d0.y), __float_as_uint(d1.x), __float_as_uint(d1.y)};
          __builtin_amdgcn_raw_buffer_store_b128(q, md, c1 ? off : -1, j * rstep, 0);
        } else {
          bufstore_c32(md, (c0 ? off : -1), d0, j * rstep);
          bufstore_c32(md, (c1 ? off + 8 : -1), d1, j * rstep);
        }
        const bool inr = o0 + (NT >> (SH - 1)) * j < nD;
        const bool ok0 = inr && c0, ok1 = inr && c1;
        const float db0 = db_of(d0), db1 = db_of(d1);
        lsum += (ok0 ? (double)db0 : 0.0) + (ok1 ? (double)db1 : 0.0);
        lmax = ok0 ? fmaxf(lmax, db0) : lmax;
        lmax = ok1 ? fmaxf(lmax, db1) : lmax;
      }
    }
    wave_sum_max(lsum, lmax);
    if (t == 0) { const int wl = relaunder(tid) >> 6; wsum[wl] = lsum; wmax[wl] = lmax; }
    DW_T(6)
    __syncthreads(); // also: every thread has taken its rows out of the regions
    if (tid == 0) {
      double sacc = 0.0;
      float m = 0.f; // Map.cpp:193: the running max starts at 0
      for (int i = 0; i < NCOL; i++) { sacc += wsum[i]; m = fmaxf(m, wmax[i]); }
      const size_t part = (size_t)cpi * tilesPerCpi + sub;
      a.partSum[part] = sacc;
      a.partMax[part] = m;
    }
    DW_T(1)
  }
#ifdef DOPW_TRACE // buckets: fill, barriers, column read + issue, transforms, kernel-spectrum product, park, stores
  if (t == 0) trace_finish("dopw", tr, blockIdx.x == 0);
#endif
}

// Tile variant for 1025 < nD <= 2049 on the TWO-WAVE 4096-point transform (fft_wave2.hpp): the structure of
// doppler_tilew_kernel one size up.  Four columns per 512-thread workgroup, a pair of waves per column; LDS = the two
// stage-twiddle factor tables + 4 exchange regions (which are also the columns' staging areas) + the chirp = 158.5 KB,
// one PERSISTENT workgroup per CU, software-pipelined over its quarter tiles: the kernel spectrum (natural order,
// 32 KB, shared by every workgroup) is requested from L2 before the forward transform, the next tile's cells after
// the spectrum product, this tile's row stores drain while the next is filled.  Barriers are workgroup-wide (the four
// columns advance together), two per transform + four around the transposes.
// Replaces doppler_tilem_kernel<16> (four-wave columns on the workgroup transform, one tile per workgroup, nothing
// overlapped: 9 % of the HBM peak at cfg 5, and 1.89 x the algorithmic traffic -- its four quarter-tile workgroups of
// a 128-byte line ran on four different XCDs).  Here the tile walk is XCD-aware: the four quarters of a 16-column tile
// go to four workgroups of ONE XCD that are dispatched together, so a line is fetched from HBM once and the 32-byte
// row pieces of the final map merge in that XCD's L2.
constexpr int DOPW2_NCOL = 4;
constexpr int DOPW2_MAX_ND = 2049;
constexpr int DOPW2_RS = Wave2Fft::X_ELEMS + 2; // region stride: the transposing accesses of phases 1 and 4 spread over the banks
constexpr int DOPW2_CHIRP_ELEMS = 17 * 128;     // rows T + 128*k, k < 17
constexpr int DOPW2_LDS_ELEMS = Wave2Fft::TW_ELEMS + DOPW2_NCOL * DOPW2_RS + DOPW2_CHIRP_ELEMS;
__global__ __launch_bounds__(128 * DOPW2_NCOL, 2) void doppler_tilew2_kernel(DopplerArgs a, int nCpi)
{
  using W = Wave2Fft;
  constexpr int NCOL = DOPW2_NCOL, NT = 128 * NCOL;
  constexpr int NR = 17;  // rows T + 128*k, k < 17, cover nD <= 2049 + 126
  constexpr int NRP = 9;  // tile cell PAIRS (two neighbouring columns, 16 bytes) per thread: nD * 2 / 512 <= 8.01
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double wsum[2 * NCOL];
  __shared__ float wmax[2 * NCOL];
  cf *table = reinterpret_cast<cf *>(smem);
  cf *regions = table + W::TW_ELEMS;
  const int tid = threadIdx.x;
  const int col = __builtin_amdgcn_readfirstlane(tid >> 7);      // column of the quarter tile = wave pair
  const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 1); // wave inside the pair
  const int lane = tid & 63;
  const int T = W::logical(wave, lane);
  const int nD = a.nD;
  const int tilesPerCpi = (a.nDelay + NCOL - 1) / NCOL;
  const int groupsPerCpi = (a.nDelay + 15) / 16; // 16-column tiles = groups of four quarter tiles
  const int nGroupsAll = groupsPerCpi * nCpi;
  cf *region = regions + col * DOPW2_RS;

  cf *chirpL = regions + NCOL * DOPW2_RS;
  for (int i = tid; i < DOPW2_CHIRP_ELEMS; i += NT) { // zero beyond nD, like doppler_tilew_kernel
    const cf c = a.chirp[min(i, nD - 1)];
    chirpL[i] = cmake(i < nD ? c.x : 0.f, i < nD ? c.y : 0.f);
  }
  W::fill_table(tid, NT, a.tw, table);
  W::Tw tw;
  W::load_twiddles(wave, lane, a.tw, table, tw);

  // XCD-aware walk: workgroup b = xcd + 8 (4 j + r) takes quarter r of group (j 8 + xcd), then every (gridDim/32)*8-th group
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int quarter = slot & 3;
  const int gStep = (gridDim.x >> 5) * 8;
  int grp = (slot >> 2) * 8 + xcd;

  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 nt[NRP];
  const int pairs = nD * (NCOL / 2);
  auto tile_of = [&](int g, int &cpi, int &sub) {
    cpi = g / groupsPerCpi;
    sub = (g - cpi * groupsPerCpi) * 4 + quarter; // quarter-tile index inside the CPI (may lie beyond the last column)
  };
  auto tile_load = [&](int g) {
    int cpi, sub;
    tile_of(g, cpi, sub);
    const cf *Rt = a.R + rmap_index(nD, a.nTiles, cpi, 0, min(sub, tilesPerCpi - 1) * NCOL);
    // buffer loads over the quarter tile's nD row pieces (32 of every 128 bytes): pair idx + 512 j is 256 rows further
    // down (soffset), rows >= nD read as zeros (see doppler_tilew_kernel)
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t d = make_rsrc_b(Rt, (nD - 1) * 128 + NCOL * 8);
    const int tl = relaunder(tid);
    const int voff = (tl >> 1) * 128 + (tl & 1) * 16;
#pragma unroll
    for (int j = 0; j < NRP; j++) {
      const u4 r = __builtin_amdgcn_raw_buffer_load_b128(d, voff, j * (NT >> 1) * 128, 0);
      nt[j] = f4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
    }
  };
  if (grp < nGroupsAll) tile_load(grp);
  for (; grp < nGroupsAll; grp += gStep) {
    int cpi, sub;
    tile_of(grp, cpi, sub);
    const bool live = sub < tilesPerCpi; // the last group of a CPI may have fewer than four quarters
    const int col0 = sub * NCOL;
    // phase 1: the tile, transposed into the per-column regions
    {
      const int tl = relaunder(tid);
      cf *dst = regions + (2 * (tl & 1)) * DOPW2_RS + (tl >> 1);
#pragma unroll
      for (int j = 0; j < NRP; j++) { // every row up to 9 * 256 - 1, the zeros of rows >= nD included
        dst[(NT >> 1) * j] = cmake(nt[j].x, nt[j].y);
        dst[DOPW2_RS + (NT >> 1) * j] = cmake(nt[j].z, nt[j].w);
      }
    }
    __syncthreads();

    // phase 2: this pair's column -> registers (DC removal + chirp), the kernel spectrum requested, forward transform,
    // x spectrum, the next tile requested, inverse transform
    cf v[32];
    const cf r0 = region[0];
    {
      const int t2 = relaunder(T);
#pragma unroll
      for (int k = 0; k < NR; k++) v[k] = cmul(csub(region[t2 + 128 * k], r0), chirpL[t2 + 128 * k]); // rows >= nD: (0 - r0) * 0
    }
#pragma unroll
    for (int k = NR; k < 32; k++) v[k] = cmake(0.f, 0.f);
    cf bf[32];
    {
      const cf *bfp = a.bfn + relaunder(T);
#pragma unroll
      for (int e = 0; e < 32; e++) bf[e] = bfp[128 * e];
    }
    __syncthreads(); // both waves of a pair have taken their rows out of the region: it becomes the exchange buffer
    W::transform<-1>(wave, lane, v, tw, region);
#pragma unroll
    for (int e = 0; e < 32; e++) v[e] = cmul(v[e], bf[e]);
    if (grp + gStep < nGroupsAll) tile_load(grp + gStep);
    W::transform<+1>(wave, lane, v, tw, region);

    // phase 3: rotate rows by nD/2+1 and park the column back in its region
    {
      const int t3 = relaunder(T);
#pragma unroll
      for (int c = 0; c < NR; c++) {
        const int k = t3 + 128 * c;
        cf d = cmul(v[c], chirpL[k]);
        if (c == 0 && t3 == 0) d = cmake(d.x + (float)nD * r0.x, d.y + (float)nD * r0.y);
        int o = k - (nD / 2 + 1);
        if (o < 0) o += nD;
        region[k < nD ? o : DOPW2_RS - 1] = d; // rows beyond nD go to a spare slot: no branch per row
      }
    }
    __syncthreads();

    // phase 4: coalesced row-segment stores + Map::set_metrics partials
    double lsum = 0.0;
    float lmax = 0.f;
    cf *mapb = a.map + (size_t)cpi * nD * a.nDelay + col0;
    const int ncol = live ? min(NCOL, a.nDelay - col0) : 0;
    {
      const int tl4 = relaunder(tid);
      const bool wide = (a.nDelay & 1) == 0; // 16-byte row pieces need the rows to start 16-byte aligned
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t md = make_rsrc_b(a.map + (size_t)cpi * nD * a.nDelay, nD * a.nDelay * 8);
      const int pc = tl4 & 1, o0 = tl4 >> 1;
      const cf *src = regions + (2 * pc) * DOPW2_RS + o0;
      const bool c0 = 2 * pc < ncol, c1 = 2 * pc + 1 < ncol;
      const int off = (o0 * a.nDelay + col0 + 2 * pc) * 8;
      const int rstep = (NT >> 1) * a.nDelay * 8;
#pragma unroll
      for (int j = 0; j < NRP; j++) {
        const cf d0 = src[(NT >> 1) * j], d1 = src[DOPW2_RS + (NT >> 1) * j];
        if (wide) {
          const u4 q = {__float_as_uint(d0.x), __float_as_uint(d0.y), __float_as_uint(d1.x), __float_as_uint(d1.y)};
          __builtin_amdgcn_raw_buffer_store_b128(q, md, c1 ? off : -1, j * rstep, 0);
        } else {
          bufstore_c32(md, (c0 ? off : -1), d0, j * rstep);
          bufstore_c32(md, (c1 ? off + 8 : -1), d1, j * rstep);
        }
        const bool inr = o0 + (NT >> 1) * j < nD;
        const bool ok0 = inr && c0, ok1 = inr && c1;
        const float db0 = db_of(d0), db1 = db_of(d1);
        lsum += (ok0 ? (double)db0 : 0.0) + (ok1 ? (double)db1 : 0.0);
        lmax = ok0 ? fmaxf(lmax, db0) : lmax;
        lmax = ok1 ? fmaxf(lmax, db1) : lmax;
      }
    }
    wave_sum_max(lsum, lmax);
    if (lane == 0) { const int wl = relaunder(tid) >> 6; wsum[wl] = lsum; wmax[wl] = lmax; }
    __syncthreads(); // also: every thread has taken its rows out of the regions
    if (tid == 0 && live) {
      double sacc = 0.0;
      float m = 0.f; // Map.cpp:193: the running max starts at 0
      for (int i = 0; i < 2 * NCOL; i++) { sacc += wsum[i]; m = fmaxf(m, wmax[i]); }
      const size_t part = (size_t)cpi * tilesPerCpi + sub;
      a.partSum[part] = sacc;
      a.partMax[part] = m;
    }
  }
}

// Tile variant for 1025 < nD <= 2049 with ONE wave per column (round 5): the 4096-point transform as FOUR one-wave
// 1024-point transforms (fft_wave1k.hpp) of the samples 4 n + r, r = 0 ... 3, and a radix-4 step across them that never
// leaves the lane -- the wave holds x[4 (T + 64 k) + r] in v[r][k], the sub-transforms are self-sorting (in T + 64 k, out
// T + 64 a), so X[(T + 64 a) + 1024 q] = sum_r (-i)^(r q) W_4096^(r (T + 64 a)) X_r[T + 64 a] combines v[0..3][a] in
// registers; the inverse is its mirror image and ends in the layout the forward transform started from.  No barrier and no
// second wave inside a column: the structure of doppler_tilew_kernel, eight columns per 512-thread workgroup, instead of
// doppler_tilew2_kernel's four pairs of waves behind workgroup-wide barriers.  MEASURED (round 5, cfg 5): parity-green at
// the first run, and no faster -- 11.4 against 11.0 us/CPI at batch 8, 9.2 against 9.3 at batch 32:
// the transforms are 30 % of its time, the row-piece stores and tile requests of a half tile (64-byte pieces: sixteen
// cache lines per 1 KB instruction) 25 %, the three barriers 19 %.  Selectable (BLAH2HIP_DOP_TILEW4), not the default.
// The price of one wave per column is registers and LDS: 128 data
// registers per lane, so the kernel spectrum is read from L2 a quarter at a time, the stage twiddles W_4096^(r (T + 64 a)) are one product of a per-lane base and a constant, and the next
// tile's cells are requested in two halves -- the upper rows after the spectrum product (they land during the inverse
// transform), the lower ones after the column has been parked.  A column's region (2050 entries, stride = 2 mod 8) is its staging area and, in
// its first 1088 entries, its exchange buffer: 8 x 16.0 KB + the 7.5 KB stage-twiddle table + the 18 KB chirp = 153.5 KB,
// one workgroup per CU.
constexpr int DOPW4_NCOL = 8;
constexpr int DOPW4_MAX_ND = 2049;
constexpr int DOPW4_RS = 2050; // 2049 rows + a spare slot; = 2 (mod 8) like doppler_tilew_kernel's
constexpr int DOPW4_CHIRP_ELEMS = 2304; // rows 4 (t + 64 k) + r, k < 9
static_assert(DOPW4_RS % 8 == 2 && DOPW4_RS > DOPW4_MAX_ND && Wave1kFft::X_ELEMS <= DOPW4_RS, "");
constexpr int DOPW4_LDS_ELEMS = Wave1kFft::TW_ELEMS + DOPW4_NCOL * DOPW4_RS + DOPW4_CHIRP_ELEMS;
// exp(-2 pi i j / 64), j = 0 ... 45 (r a, r <= 3, a <= 15)
constexpr float DOPW4_W64[46][2] = {
    {1.000000000f, -0.000000000f}, {0.995184727f, -0.098017140f}, {0.980785280f, -0.195090322f}, {0.956940336f, -0.290284677f},
    {0.923879533f, -0.382683432f}, {0.881921264f, -0.471396737f}, {0.831469612f, -0.555570233f}, {0.773010453f, -0.634393284f},
    {0.707106781f, -0.707106781f}, {0.634393284f, -0.773010453f}, {0.555570233f, -0.831469612f}, {0.471396737f, -0.881921264f},
    {0.382683432f, -0.923879533f}, {0.290284677f, -0.956940336f}, {0.195090322f, -0.980785280f}, {0.098017140f, -0.995184727f},
    {0.000000000f, -1.000000000f}, {-0.098017140f, -0.995184727f}, {-0.195090322f, -0.980785280f}, {-0.290284677f, -0.956940336f},
    {-0.382683432f, -0.923879533f}, {-0.471396737f, -0.881921264f}, {-0.555570233f, -0.831469612f}, {-0.634393284f, -0.773010453f},
    {-0.707106781f, -0.707106781f}, {-0.773010453f, -0.634393284f}, {-0.831469612f, -0.555570233f}, {-0.881921264f, -0.471396737f},
    {-0.923879533f, -0.382683432f}, {-0.956940336f, -0.290284677f}, {-0.980785280f, -0.195090322f}, {-0.995184727f, -0.098017140f},
    {-1.000000000f, -0.000000000f}, {-0.995184727f, 0.098017140f}, {-0.980785280f, 0.195090322f}, {-0.956940336f, 0.290284677f},
    {-0.923879533f, 0.382683432f}, {-0.881921264f, 0.471396737f}, {-0.831469612f, 0.555570233f}, {-0.773010453f, 0.634393284f},
    {-0.707106781f, 0.707106781f}, {-0.634393284f, 0.773010453f}, {-0.555570233f, 0.831469612f}, {-0.471396737f, 0.881921264f},
    {-0.382683432f, 0.923879533f}, {-0.290284677f, 0.956940336f}};
__global__ __launch_bounds__(64 * DOPW4_NCOL, 2) void doppler_tilew4_kernel(DopplerArgs a, int nCpi)
{
#ifdef DOPW_TRACE
  uint64_t tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime();
#define D4_T(k) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); tr[k] += now_ - t0_; t0_ = now_; }
#else
#define D4_T(k)
#endif
  using K = Wave1kFft;
  constexpr int NCOL = DOPW4_NCOL, NT = 64 * NCOL, SH = 3;
  constexpr int NK = 9;    // v[r][k], k < 9: rows 4 (t + 64 k) + r < 2304 cover nD <= 2049
  constexpr int NRP = 17;  // tile cell PAIRS (two neighbouring columns, 16 bytes) per thread: nD * 4 / 512 <= 16.01
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double wsum[NCOL];
  __shared__ float wmax[NCOL];
  cf *table = reinterpret_cast<cf *>(smem);
  cf *regions = table + K::TW_ELEMS;
  const int tid = threadIdx.x;
  const int w = tid >> 6, t = tid & 63; // wave = column of the half tile
  const int nD = a.nD;
  const int tilesPerCpi = (a.nDelay + NCOL - 1) / NCOL;
  const int nTilesAll = tilesPerCpi * nCpi;
  cf *region = regions + w * DOPW4_RS;

  // a.tw = exp(-2 pi i k / 4096): the 1024-point roots are every fourth entry
  for (int e = tid; e < K::TW_ELEMS; e += NT) table[e] = a.tw[4 * ((((e >> 6) + 1) * (e & 63)) & 1023)];
  // The chirp (the same for every column) in LDS, zero beyond nD.  The interleaved column read touches rows up to 2303 of a
  // 2050-entry region: what it finds there -- the next column's region, or the chirp behind the last one -- is finite and
  // meets a zero chirp.  The regions start out finite too.
  cf *chirpL = regions + NCOL * DOPW4_RS;
  for (int i = tid; i < DOPW4_CHIRP_ELEMS; i += NT) {
    const cf c = a.chirp[min(i, nD - 1)];
    chirpL[i] = cmake(i < nD ? c.x : 0.f, i < nD ? c.y : 0.f);
  }
  for (int i = tid; i < NCOL * DOPW4_RS; i += NT) regions[i] = cmake(0.f, 0.f);
  K::Tw tw;
  tw.tab = table + t;
  tw.w64 = a.tw[4 * 16 * (t & 31)];
  tw.w32 = a.tw[4 * 32 * (t & 15)];
  cf base[3]; // W_4096^(r t), r = 1, 2, 3
#pragma unroll
  for (int r = 1; r < 4; r++) base[r - 1] = a.tw[r * t];
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  f4 nt[NRP];
  auto tile_load = [&](int it, int j0, int j1) {
    const int cpi = it / tilesPerCpi, sub = it - cpi * tilesPerCpi;
    const cf *Rt = a.R + rmap_index(nD, a.nTiles, cpi, 0, sub * NCOL);
    const __amdgpu_buffer_rsrc_t d = make_rsrc_b(Rt, (nD - 1) * 128 + NCOL * 8);
    const int tl = relaunder(tid);
    const int voff = (tl >> (SH - 1)) * 128 + (tl & (NCOL / 2 - 1)) * 16;
#pragma unroll
    for (int j = j0; j < j1; j++) {
      const u4 r = __builtin_amdgcn_raw_buffer_load_b128(d, voff, j * (NT >> (SH - 1)) * 128, 0);
      nt[j] = f4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
    }
  };
  int it = blockIdx.x;
  if (it < nTilesAll) tile_load(it, 0, NRP);
  __syncthreads(); // table, zeroed regions
  for (; it < nTilesAll; it += gridDim.x) {
    const int cpi = it / tilesPerCpi, sub = it - cpi * tilesPerCpi;
    const int col0 = sub * NCOL;
    // phase 1: the tile, transposed into the per-column regions
    {
      const int tl = relaunder(tid);
      cf *dst = regions + (2 * (tl & (NCOL / 2 - 1))) * DOPW4_RS + (tl >> (SH - 1)); // pair idx + 512 j: same columns, row + 128 j
#pragma unroll
      for (int j = 0; j < NRP; j++) { // rows up to 2049 (the zeros of rows >= nD included); the last pair's rows beyond: dropped
        if (j < NRP - 1 || (tl >> (SH - 1)) + (NT >> (SH - 1)) * j < DOPW4_RS) {
          dst[(NT >> (SH - 1)) * j] = cmake(nt[j].x, nt[j].y);
          dst[DOPW4_RS + (NT >> (SH - 1)) * j] = cmake(nt[j].z, nt[j].w);
        }
      }
    }
    D4_T(0)
    __syncthreads();
    D4_T(1)

    // phase 2: this wave's column -> registers, rows 4 (t + 64 k) + r (DC removal + chirp)
    cf v[4][16];
    const cf r0 = region[0];
    {
      const int t2 = relaunder(t);
#pragma unroll
      for (int k = 0; k < NK; k++) { // (a row group at a time: all 36 chirp values requested at once are 72 registers)
#pragma unroll
        for (int r = 0; r < 4; r++) v[r][k] = cmul(csub(region[4 * t2 + 256 * k + r], r0), chirpL[4 * t2 + 256 * k + r]); // rows >= nD: (finite - r0) * 0
        if (k & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    cf bfa[16], bfb[16];
    const cf *bfp = a.bfn + relaunder(t);
    auto bf_load = [&](cf *dst, int q) {
#pragma unroll
      for (int e = 0; e < 16; e++) dst[e] = bfp[64 * e + 1024 * q];
    };
    bf_load(bfa, 0);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
    D4_T(2)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      K::transform<-1, 9>(t, v[r], tw, region);
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    D4_T(3)
    // radix 4 across the sub-transforms: v[q][e] = X[(t + 64 e) + 1024 q]
    // (The 45 constants W_64^(r e) must not be materialised ahead of their use: as operands of the packed-arithmetic asm they
    // are two VGPRs each, loop-invariant, and hoisted out of the tile loop they are 90 registers.  Each is a literal plus a
    // zero the compiler cannot see through, produced next to its one use.)
    float zf = 0.f;
    asm volatile("" : "+v"(zf));
    auto w64 = [&](int j) -> cf { return cmake(DOPW4_W64[j][0] + zf, DOPW4_W64[j][1] + zf); };
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const cf t0 = v[0][e];
      const cf t1 = cmul(v[1][e], e ? cmul(base[0], w64(e)) : base[0]);
      const cf t2 = cmul(v[2][e], e ? cmul(base[1], w64(2 * e)) : base[1]);
      const cf t3 = cmul(v[3][e], e ? cmul(base[2], w64(3 * e)) : base[2]);
      const cf s02 = cadd(t0, t2), d02 = csub(t0, t2), s13 = cadd(t1, t3), d13 = csub(t1, t3);
      v[0][e] = cadd(s02, s13);
      v[2][e] = csub(s02, s13);
      v[1][e] = cadd_i<-1>(d02, d13); // d02 - i d13
      v[3][e] = csub_i<-1>(d02, d13); // d02 + i d13
      if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    // x kernel spectrum (natural order, 32 KB shared by every workgroup: L2), a quarter at a time through two buffers of
    // sixteen: the first quarter was requested before the forward transforms, each next one before the previous product
    __builtin_amdgcn_sched_barrier(0);
    bf_load(bfb, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; e++) v[0][e] = cmul(v[0][e], bfa[e]);
    __builtin_amdgcn_sched_barrier(0);
    bf_load(bfa, 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; e++) v[1][e] = cmul(v[1][e], bfb[e]);
    __builtin_amdgcn_sched_barrier(0);
    bf_load(bfb, 3);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; e++) v[2][e] = cmul(v[2][e], bfa[e]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; e++) v[3][e] = cmul(v[3][e], bfb[e]);
    __builtin_amdgcn_sched_barrier(0);
    D4_T(4)
    const bool more = it + (int)gridDim.x < nTilesAll;
    // the inverse: radix 4 across q, conjugate twiddles, four inverse sub-transforms
    asm volatile("" : "+v"(zf));
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const cf y0 = v[0][e], y1 = v[1][e], y2 = v[2][e], y3 = v[3][e];
      const cf s02 = cadd(y0, y2), d02 = csub(y0, y2), s13 = cadd(y1, y3), d13 = csub(y1, y3);
      v[0][e] = cadd(s02, s13);
      const cf z1 = cadd_i<+1>(d02, d13); // d02 + i d13
      const cf z2 = csub(s02, s13);
      const cf z3 = csub_i<+1>(d02, d13); // d02 - i d13
      v[1][e] = cmulc(z1, e ? cmul(base[0], w64(e)) : base[0]);
      v[2][e] = cmulc(z2, e ? cmul(base[1], w64(2 * e)) : base[1]);
      v[3][e] = cmulc(z3, e ? cmul(base[2], w64(3 * e)) : base[2]);
      if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // the four inverse sub-transforms; the rows are parked only after the last one (a parked row's slot can lie in the
    // exchange area, the region's first 1088 entries)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      K::transform<+1>(t, v[r], tw, region);
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }

    D4_T(5)
    // phase 3: chirp, rotate rows by nD/2 + 1, park the column back in its region
    {
      const int t3 = relaunder(t);
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int k = 0; k < NK; k++) {
          const int i = 4 * t3 + 256 * k + r;
          cf d = cmul(v[r][k], chirpL[i]);
          if (k == 0 && r == 0 && t3 == 0) d = cmake(d.x + (float)nD * r0.x, d.y + (float)nD * r0.y);
          int o = i - (nD / 2 + 1);
          if (o < 0) o += nD;
          region[i < nD ? o : DOPW4_RS - 1] = d; // rows beyond nD go to a spare slot: no branch per row
          if (k == 4) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next tile: requested once the column's 128 data registers are free (68 registers of cells held through the
    // inverse transform spill), in flight during the barrier and the row stores
    // (unconditionally -- a workgroup's last iteration requests its own tile again: behind `if (more)` the cells of the
    // previous request would count as live through the whole loop body, 68 registers the transforms do not have)
    tile_load(more ? it + (int)gridDim.x : it, 0, NRP);
    D4_T(6)
    __syncthreads();
    D4_T(1)

    // phase 4: coalesced row-segment stores + Map::set_metrics partials
    double lsum = 0.0;
    float lmax = 0.f;
    const int ncol = min(NCOL, a.nDelay - col0);
    const int tl4 = relaunder(tid);
    const bool wide = (a.nDelay & 1) == 0; // 16-byte row pieces need the rows to start 16-byte aligned
    {
      const __amdgpu_buffer_rsrc_t md = make_rsrc_b(a.map + (size_t)cpi * nD * a.nDelay, nD * a.nDelay * 8);
      const int pc = tl4 & (NCOL / 2 - 1), o0 = tl4 >> (SH - 1);
      const cf *src = regions + (2 * pc) * DOPW4_RS + o0;
      const bool c0 = 2 * pc < ncol, c1 = 2 * pc + 1 < ncol;
      const int off = (o0 * a.nDelay + col0 + 2 * pc) * 8;
      const int rstep = (NT >> (SH - 1)) * a.nDelay * 8;
#pragma unroll
      for (int j = 0; j < NRP; j++) {
        const int rj = min((NT >> (SH - 1)) * j, DOPW4_RS - 1 - o0); // (the last pair's rows beyond the region: any slot, not stored)
        const cf d0 = src[rj], d1 = src[DOPW4_RS + rj];
        if (wide) { // ncol is even with nDelay: both columns or none
          const u4 q = {__float_as_uint(d0.x), __float_as_uint(d0.y), __float_as_uint(d1.x), __float_as_uint(d1.y)};
          __builtin_amdgcn_raw_buffer_store_b128(q, md, c1 ? off : -1, j * rstep, 0);
        } else {
          bufstore_c32(md, (c0 ? off : -1), d0, j * rstep);
          bufstore_c32(md, (c1 ? off + 8 : -1), d1, j * rstep);
        }
        const bool inr = o0 + (NT >> (SH - 1)) * j < nD;
        const bool ok0 = inr && c0, ok1 = inr && c1;
        const float db0 = db_of(d0), db1 = db_of(d1);
        lsum += (ok0 ? (double)db0 : 0.0) + (ok1 ? (double)db1 : 0.0);
        lmax = ok0 ? fmaxf(lmax, db0) : lmax;
        lmax = ok1 ? fmaxf(lmax, db1) : lmax;
      }
    }
    wave_sum_max(lsum, lmax);
    if (t == 0) { const int wl = relaunder(tid) >> 6; wsum[wl] = lsum; wmax[wl] = lmax; }
    D4_T(7)
    __syncthreads(); // also: every thread has taken its rows out of the regions
    if (tid == 0) {
      double sacc = 0.0;
      float m = 0.f; // Map.cpp:193: the running max starts at 0
      for (int i = 0; i < NCOL; i++) { sacc += wsum[i]; m = fmaxf(m, wmax[i]); }
      const size_t part = (size_t)cpi * tilesPerCpi + sub;
      a.partSum[part] = sacc;
      a.partMax[part] = m;
    }
    D4_T(1)
  }
#ifdef DOPW_TRACE // buckets: fill, barriers, column read, forward transforms, combine + spectrum, inverse, park + next tile's requests, stores
  if (t == 0) trace_finish("dopw4", tr, blockIdx.x == 0);
#endif
}
#undef D4_T

// Tile variant for multi-wave columns: 513 < nD <= 1025 (M = 2048, R3 = 8: a column is a
// 128-thread, two-wave transform, 8 columns per 1024-thread workgroup) and 1025 < nD <= 2049
// (M = 4096, R3 = 16: 256 threads per column, 4 columns per workgroup).  Same phases as
// doppler_tile_kernel.  A column's exchange buffer is ONE region (stage 2 works inside rows
// of R3 consecutive threads, i.e. inside a wave, so it runs in place); the stage boundaries
// are workgroup barriers, all columns advance in lockstep.  LDS: 8 x 17 KB + the kernel
// spectrum (16 KB) = 155 KB, or 4 x 34 KB = 139 KB (R3 = 16 reads the kernel spectrum, which
// every workgroup shares, from L2 instead): one workgroup per CU, 4 waves per SIMD.
template <int R3> struct DopM {
  using W = WgFft<R3>;
  static constexpr int NCOL = 1024 / W::T;          // 8 or 4
  static constexpr int SH = (NCOL == 8) ? 3 : 2;
  static constexpr int REGION = W::A_ELEMS;         // complex values per column
  static constexpr bool BF_LDS = (R3 == 8);
  static constexpr int LDS_ELEMS = NCOL * REGION + (BF_LDS ? W::F : 0);
  static constexpr int MAX_ND = 9 * W::T - (W::T - 1); // rows t + T*k, k < 9: 1025 / 2049
};

// hides a value's provenance from the optimiser, so that addresses derived from it are
// recomputed where they are used instead of being kept live (and spilled) from their first use
__device__ __forceinline__ int relaunder(int v)
{
  asm volatile("" : "+v"(v));
  return v;
}

template <int R3>
__global__ __launch_bounds__(1024) void doppler_tilem_kernel(DopplerArgs a)
{
  using D = DopM<R3>;
  using W = WgFft<R3>;
  constexpr int T = W::T;   // threads per column
  constexpr int NCOL = D::NCOL;
  constexpr int NT = 1024;
  constexpr int NR = 9;     // rows t + T*k, k < 9, cover nD <= MAX_ND
  constexpr int NRT = 9;    // tile cells per thread: nD * NCOL / 1024 <= 8.01
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf *lds = reinterpret_cast<cf *>(smem);
  const int tid = threadIdx.x;
  const int w = tid / T, t = tid % T; // column of the tile, thread inside the column's transform
  const int nD = a.nD;
  const int sub = blockIdx.x, cpi = blockIdx.y;
  const int col0 = sub * NCOL;
  cf *region = lds + w * D::REGION;
  cf *bfL = lds + NCOL * D::REGION;

  // phase 1: coalesced tile read, all loads in flight before the first use
  const cf *Rt = a.R + rmap_index(nD, a.nTiles, cpi, 0, col0);
  const int cells = nD * NCOL;
  cf v[16];
#pragma unroll
  for (int j = 0; j < NRT; j++) {
    const int idx = tid + NT * j;
    const int c = idx & (NCOL - 1), row = idx >> D::SH;
    v[j] = Rt[idx < cells ? row * 16 + c : 0];
  }
  cf bfs[W::F / NT];
  if (D::BF_LDS) {
#pragma unroll
    for (int j = 0; j < W::F / NT; j++) bfs[j] = a.bf[tid + NT * j];
  }
  // R3 = 16 carries 15 stage-3 twiddles (7 for R3 = 8) and would spill at the 128 VGPRs a
  // 1024-thread workgroup leaves: there each twiddle set and the chirp are fetched (L1/L2)
  // right before the stage that uses them instead of being held across the whole transform
  constexpr bool RELOAD = true; // (R3 = 8 held them until the grouped butterflies took two more registers: 6 spills)
  cf tw1[15], tw3[16], ch[NR];
  if (RELOAD) W::load_tw1(t, a.tw, tw1);
  else W::load_twiddles(t, a.tw, tw1, tw3);
#pragma unroll
  for (int k = 0; k < NR; k++) ch[k] = a.chirp[min(t + T * k, nD - 1)];
#pragma unroll
  for (int j = 0; j < NRT; j++) {
    const int idx = tid + NT * j;
    const int c = idx & (NCOL - 1), row = idx >> D::SH;
    if (idx < cells) lds[c * D::REGION + row] = v[j];
  }
  if (D::BF_LDS) {
#pragma unroll
    for (int j = 0; j < W::F / NT; j++) bfL[tid + NT * j] = bfs[j];
  }
  __syncthreads();

  // phase 2: column -> registers (DC removal + chirp), transform, x kernel spectrum, inverse
  const cf r0 = region[0];
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const int i = t + T * k;
    const cf rv = region[min(i, nD - 1)];
    const cf p = cmul(csub(rv, r0), ch[k]);
    v[k] = cmake(i < nD ? p.x : 0.f, i < nD ? p.y : 0.f);
  }
#pragma unroll
  for (int k = NR; k < 16; k++) v[k] = cmake(0.f, 0.f);
  __syncthreads(); // every thread of the column has taken its rows out of the region
  W::fwd_s1(t, v, tw1, region);
  __syncthreads();
  W::fwd_s2_load(t, v, region);
  dft16<-1>(v);
  W::fwd_s2_store(t, v, region);
  __syncthreads();
  if (RELOAD) W::load_tw3(relaunder(t), a.tw, tw3);
  W::fwd_s3(t, v, tw3, region);
#pragma unroll
  for (int e = 0; e < 16; e++) v[e] = cmul(v[e], D::BF_LDS ? bfL[e * T + t] : a.bf[e * T + t]);
  __syncthreads();
  W::inv_s1(t, v, tw3, region);
  __syncthreads();
  W::inv_s2_load(t, v, region);
  dft16<+1>(v);
  W::inv_s2_store(t, v, region);
  __syncthreads();
  if (RELOAD) W::load_tw1(relaunder(t), a.tw, tw1);
  W::inv_s3(t, v, tw1, region);
  __syncthreads();

  // phase 3: rotate rows by nD/2+1 and park the column back in its region
  if (RELOAD) {
    const int tl = relaunder(t);
#pragma unroll
    for (int k = 0; k < NR; k++) ch[k] = a.chirp[min(tl + T * k, nD - 1)];
  }
#pragma unroll
  for (int c = 0; c < NR; c++) {
    const int k = t + T * c;
    cf d = cmul(v[c], ch[c]);
    if (c == 0 && t == 0) d = cmake(d.x + (float)nD * r0.x, d.y + (float)nD * r0.y);
    int o = k - (nD / 2 + 1);
    if (o < 0) o += nD;
    if (k < nD) region[o] = d;
  }
  __syncthreads();

  // phase 4: coalesced row-segment stores + Map::set_metrics partials
  double lsum = 0.0;
  float lmax = 0.f;
  cf *mapb = a.map + (size_t)cpi * nD * a.nDelay + col0;
  const int ncol = min(NCOL, a.nDelay - col0);
#pragma unroll
  for (int j = 0; j < NRT; j++) {
    const int idx = tid + NT * j;
    const int c = idx & (NCOL - 1), o = idx >> D::SH;
    const bool ok = idx < cells && c < ncol;
    const cf d = lds[c * D::REGION + min(o, nD - 1)];
    if (ok) mapb[(size_t)o * a.nDelay + c] = d;
    const float db = db_of(d);
    lsum += ok ? (double)db : 0.0;
    lmax = ok ? fmaxf(lmax, db) : lmax;
  }
  const size_t part = (size_t)cpi * gridDim.x + blockIdx.x;
  __shared__ double wsum[16];
  __shared__ float wmax[16];
  wave_sum_max(lsum, lmax);
  if ((tid & 63) == 0) { wsum[tid >> 6] = lsum; wmax[tid >> 6] = lmax; }
  __syncthreads();
  if (tid == 0) {
    double sacc = 0.0;
    float m = 0.f; // Map.cpp:193: the running max starts at 0
    for (int i = 0; i < NT / 64; i++) { sacc += wsum[i]; m = fmaxf(m, wmax[i]); }
    a.partSum[part] = sacc;
    a.partMax[part] = m;
  }
}

// Fallback for nD > 2049 (transform longer than the on-chip FFT covers):
// direct DFT, lane <-> delay column, KPT output rows per thread, the 4 waves
// split the pulse axis and reduce through LDS.  Same DC handling as above.
constexpr int DOP_KPT = 8;
constexpr int DOP_WAVES = 4;

__global__ __launch_bounds__(64 * DOP_WAVES) void doppler_dft_kernel(DopplerArgs a)
{
  __shared__ cf red[DOP_WAVES][DOP_KPT][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nD = a.nD, nDelay = a.nDelay;
  const int j = blockIdx.x * 64 + lane;
  const int o0 = blockIdx.y * DOP_KPT;
  const int cpi = blockIdx.z;
  const bool jok = j < nDelay;
  const cf *Rc = a.R + rmap_index(nD, a.nTiles, cpi, 0, jok ? j : 0);
  const cf r0 = Rc[0];
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
    const cf r = csub(Rc[(size_t)i * 16], r0);
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
  double lsum = 0.0;
  float lmax = 0.f;
#pragma unroll
  for (int h = 0; h < DOP_KPT / DOP_WAVES; h++) {
    const int kk = wave * (DOP_KPT / DOP_WAVES) + h;
    cf s = red[0][kk][lane];
#pragma unroll
    for (int w = 1; w < DOP_WAVES; w++) s = cadd(s, red[w][kk][lane]);
    const int o = o0 + kk;
    if (jok && o < nD) {
      if ((o + nD / 2 + 1) % nD == 0) s = cmake(s.x + (float)nD * r0.x, s.y + (float)nD * r0.y);
      a.map[(size_t)cpi * nD * nDelay + (size_t)o * nDelay + j] = s;
      const float db = db_of(s);
      lsum += (double)db;
      lmax = fmaxf(lmax, db);
    }
  }
  const size_t part = (size_t)cpi * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x;
  block_metrics_partial(lsum, lmax, a.partSum + part, a.partMax + part);
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
// Map::to_json's cell values (Map.cpp:115-185): db[i] = 10*log10|z| - noisePower as fp32,
// half the bytes of the complex map for a front-end that only plots.
__global__ __launch_bounds__(256) void db_map_kernel(const cf *map, const double *metrics, float *db, uint32_t cells)
{
  const uint32_t cpi = blockIdx.y;
  const float noise = (float)metrics[2 * cpi];
  const cf *z = map + (size_t)cpi * cells;
  float *o = db + (size_t)cpi * cells;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += gridDim.x * blockDim.x) o[i] = db_of(z[i]) - noise;
}

} // namespace blah2
