// Workgroup FFT, 8 points per thread: F = 8 * 8 * 8 * R4 points (R4 = 2, 4, 8 ->
// F = 1024, 2048, 4096) by T = F/8 threads, radix 8-8-8-R4 with three LDS
// exchanges.  Same contract as WgFft (fft_wg.hpp) -- forward DIF leaves the
// spectrum in a permuted register layout, the inverse is the mirror image and
// ends in natural order -- but with half the registers per thread, so that the
// correlation kernels fit 4 waves per SIMD instead of 2 and LDS / barrier /
// HBM latencies are hidden by other waves (profiles/r01_pmc.csv: at 2 waves per
// SIMD a wave issued a VALU instruction every ~8 cycles, ideal 4).
//
// Stage 4 has two forms.  The LDS form (fwd_s3_store + fwd_s4 / inv_s4 + inv_s3_load) is a third
// exchange through E3.  The lane form (fwd_s3_lanes / inv_s4_lanes, device only) keeps the data where
// stage 3 left it: the R4 inputs of a stage-4 DFT sit in R4 CONSECUTIVE LANES (t3 = t mod R4), so the
// DFT runs across lanes with DPP moves (quad_perm for lane^1 and lane^2, row_half_mirror + a quad
// reversal for lane^4) as a decimation-in-frequency butterfly network with per-lane signs and
// twiddles -- two LDS exchanges and two barriers per transform instead of three.  The spectrum then
// stays in thread (q1,q2,t3), register q3, at frequency index q4 = bitrev(t3); X and Y share the
// layout and the inverse mirrors it, so nothing else changes.
//
// Index algebra, T2 = T/8 = 8*R4:
//   n = t + T*k1,            t  = t2 + T2*k2,        t2 = t3 + R4*k3
//   m = q1 + 8*q2 + 64*q3 + 512*q4
//   S1 thread t            : DFT8 over k1 -> q1, times W_F^(t*q1)      -> E1[q1][t]
//   S2 thread (q1, t2)     : DFT8 over k2 -> q2, times W_T^(t2*q2)     -> E2[q1,q2][t2]
//   S3 thread (q1,q2,t3)   : DFT8 over k3 -> q3, times W_T2^(t3*q3)    -> E3[q1,q2,q3][t3]
//   S4 thread (q1,q2,q3)x(8/R4) : DFT_R4 over t3 -> q4
// Each twiddle set depends only on the role of the thread that applies it
// (t, t2, t3) and the register index, and the inverse stage with the same role
// applies the conjugate before its inverse DFT, so 3 x 7 complex registers serve
// both directions.
//
// LDS (complex fp32; accesses are paired by hipcc into ds_*2_b64, conflict-free
// iff the 16 lanes of a group hit 16 distinct complex indices mod 16):
//   E1[q1*T + t]                    S1 writes / S2 reads 16 consecutive t (t2)
//   E2[(q1*8+q2)*P2 + t2]           P2 = T2 + R4: S3 groups are (16/R4) rows x R4 columns
//   E3[t3*P3 + q3*64 + (q1*8+q2)]   P3 = 512 + 16/R4: S3 groups are R4 rows x (16/R4) columns,
//                                   S4 reads 16 consecutive columns of one row
#pragma once

#include "fft_wg.hpp"

namespace blah2 {

template <int R4> struct WgFft8 {
  static constexpr int F = 512 * R4;
  static constexpr int T = 64 * R4;   // threads
  static constexpr int T2 = 8 * R4;   // size of the stage-3/4 sub-transform
  static constexpr int NP = 8 / R4;   // (q1,q2,q3) triples per thread in S4
  static constexpr int P1 = T;
  static constexpr int P2 = T2 + R4;
  static constexpr int P3 = 512 + 16 / R4;
  static constexpr int E1_ELEMS = 8 * P1;
  static constexpr int E2_ELEMS = 64 * P2;
  static constexpr int E3_ELEMS = R4 * P3;
  static constexpr int BUF_ELEMS = (E1_ELEMS > E2_ELEMS ? (E1_ELEMS > E3_ELEMS ? E1_ELEMS : E3_ELEMS)
                                                        : (E2_ELEMS > E3_ELEMS ? E2_ELEMS : E3_ELEMS));

  // tw1[q-1] = W_F^(t*q); tw2[q-1] = W_T^(t2*q), t2 = t % T2; tw3[q-1] = W_T2^(t3*q), t3 = t % R4
  template <class TW> B2_HD static void load_twiddles(int t, const TW *tw, cf *tw1, cf *tw2, cf *tw3)
  {
    const int t2 = t % T2, t3 = t % R4;
#pragma unroll
    for (int q = 1; q < 8; q++) {
      tw1[q - 1] = tw[(t * q) & (F - 1)];
      tw2[q - 1] = tw[(8 * t2 * q) & (F - 1)];   // W_T = W_F^8
      tw3[q - 1] = tw[(64 * t3 * q) & (F - 1)];  // W_T2 = W_F^64
    }
  }

  // ---- forward ---------------------------------------------------------------
  // v[k1] = in[t + T*k1]
  B2_HD static void fwd_s1(int t, cf *v, const cf *tw1, cf *E)
  {
    dft8<-1>(v);
    E[t] = v[0];
#pragma unroll
    for (int q = 1; q < 8; q++) E[q * P1 + t] = cmul(v[q], tw1[q - 1]);
  }
  B2_HD static void fwd_s2_load(int t, cf *v, const cf *E)
  {
    const int q1 = t / T2, t2 = t % T2;
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = E[q1 * P1 + t2 + T2 * k];
  }
  B2_HD static void fwd_s2_store(int t, cf *v, const cf *tw2, cf *E)
  {
    const int q1 = t / T2, t2 = t % T2;
    dft8<-1>(v);
    E[(q1 * 8) * P2 + t2] = v[0];
#pragma unroll
    for (int q = 1; q < 8; q++) E[(q1 * 8 + q) * P2 + t2] = cmul(v[q], tw2[q - 1]);
  }
  B2_HD static void fwd_s3_load(int t, cf *v, const cf *E)
  {
    const int q12 = t / R4, t3 = t % R4;
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = E[q12 * P2 + t3 + R4 * k];
  }
  B2_HD static void fwd_s3_store(int t, cf *v, const cf *tw3, cf *E)
  {
    const int q12 = t / R4, t3 = t % R4;
    dft8<-1>(v);
    E[t3 * P3 + q12] = v[0];
#pragma unroll
    for (int q = 1; q < 8; q++) E[t3 * P3 + q * 64 + q12] = cmul(v[q], tw3[q - 1]);
  }
  // leaves X[q1 + 8*q2 + 64*q3 + 512*q4] in v[j*R4 + q4] where q3*64 + (q1*8+q2) = t + T*j
  B2_HD static void fwd_s4(int t, cf *v, const cf *E)
  {
#pragma unroll
    for (int j = 0; j < NP; j++) {
      const int rho = t + T * j;
      cf *w = v + j * R4;
#pragma unroll
      for (int u = 0; u < R4; u++) w[u] = E[u * P3 + rho];
      small_dft<-1>(w);
    }
  }

  // ---- inverse (unnormalised) --------------------------------------------------
  B2_HD static void inv_s4(int t, cf *v, cf *E)
  {
#pragma unroll
    for (int j = 0; j < NP; j++) {
      const int rho = t + T * j;
      cf *w = v + j * R4;
      small_dft<+1>(w);
#pragma unroll
      for (int u = 0; u < R4; u++) E[u * P3 + rho] = w[u];
    }
  }
  B2_HD static void inv_s3_load(int t, cf *v, const cf *tw3, const cf *E)
  {
    const int q12 = t / R4, t3 = t % R4;
    v[0] = E[t3 * P3 + q12];
#pragma unroll
    for (int q = 1; q < 8; q++) v[q] = cmulc(E[t3 * P3 + q * 64 + q12], tw3[q - 1]);
    dft8<+1>(v);
  }
  B2_HD static void inv_s3_store(int t, const cf *v, cf *E)
  {
    const int q12 = t / R4, t3 = t % R4;
#pragma unroll
    for (int k = 0; k < 8; k++) E[q12 * P2 + t3 + R4 * k] = v[k];
  }
  B2_HD static void inv_s2_load(int t, cf *v, const cf *tw2, const cf *E)
  {
    const int q1 = t / T2, t2 = t % T2;
    v[0] = E[(q1 * 8) * P2 + t2];
#pragma unroll
    for (int q = 1; q < 8; q++) v[q] = cmulc(E[(q1 * 8 + q) * P2 + t2], tw2[q - 1]);
    dft8<+1>(v);
  }
  B2_HD static void inv_s2_store(int t, const cf *v, cf *E)
  {
    const int q1 = t / T2, t2 = t % T2;
#pragma unroll
    for (int k = 0; k < 8; k++) E[q1 * P1 + t2 + T2 * k] = v[k];
  }
  // leaves z[t + T*c] in v[c]
  B2_HD static void inv_s1(int t, cf *v, const cf *tw1, const cf *E)
  {
    v[0] = E[t];
#pragma unroll
    for (int q = 1; q < 8; q++) v[q] = cmulc(E[q * P1 + t], tw1[q - 1]);
    dft8<+1>(v);
  }

#if defined(__HIPCC__)
  // ---- stage 4 across lanes ------------------------------------------------------
  // per-lane constants of the butterfly network: sgn[b] = -1 on lanes whose bit b is set, and the
  // twiddle a lane applies after (forward) / before (inverse) the level on bit b > 0:
  // W_{2^(b+1)}^(lane mod 2^b) on lanes with bit b set, 1 elsewhere.  `tw` is exp(-2 pi i k/F).
  struct LaneConst {
    float sgn[3];
    cf w[3]; // w[b] for b = 1, 2 (w[0] unused)
  };
  template <class TW> __device__ static LaneConst lane_constants(int t, const TW *tw)
  {
    LaneConst c;
#pragma unroll
    for (int b = 0; b < 3; b++) {
      const bool set = (t >> b) & 1;
      c.sgn[b] = set ? -1.f : 1.f;
      const int e = (t & ((1 << b) - 1)) * (F >> (b + 1)); // W_{2^(b+1)}^k = W_F^(k F / 2^(b+1))
      c.w[b] = (b > 0 && set) ? tw[e & (F - 1)] : cmake(1.f, 0.f);
    }
    return c;
  }
  template <int MASK> __device__ __forceinline__ static float lane_xor(float v)
  {
    const int i = __builtin_bit_cast(int, v);
    int r;
    if (MASK == 1) r = __builtin_amdgcn_mov_dpp(i, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
    else if (MASK == 2) r = __builtin_amdgcn_mov_dpp(i, 0x4E, 0xf, 0xf, true); // quad_perm [2,3,0,1]
    else {                                                                      // lane ^ 4
      r = __builtin_amdgcn_mov_dpp(i, 0x141, 0xf, 0xf, true);                   // row_half_mirror: l -> 7 - l
      r = __builtin_amdgcn_mov_dpp(r, 0x1B, 0xf, 0xf, true);                    // quad_perm [3,2,1,0]: -> l ^ 4
    }
    return __builtin_bit_cast(float, r);
  }
  template <int B> __device__ __forceinline__ static cf lane_bfly(cf z, const LaneConst &c)
  {
    // lanes with bit B clear: z + partner; set: partner - z
    return cmake(__builtin_fmaf(z.x, c.sgn[B], lane_xor<(1 << B)>(z.x)), __builtin_fmaf(z.y, c.sgn[B], lane_xor<(1 << B)>(z.y)));
  }
  // forward: stage-3 DFT in registers, then the R4-point DFT across the lanes t3 = t mod R4
  __device__ static void fwd_s3_lanes(int t, cf *v, const cf *tw3, const LaneConst &c, const cf *E)
  {
    fwd_s3_load(t, v, E);
    dft8<-1>(v);
#pragma unroll
    for (int q = 1; q < 8; q++) v[q] = cmul(v[q], tw3[q - 1]);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      cf z = v[q];
      if (R4 == 8) { z = lane_bfly<2>(z, c); z = cmul(z, c.w[2]); }
      if (R4 >= 4) { z = lane_bfly<1>(z, c); z = cmul(z, c.w[1]); }
      z = lane_bfly<0>(z, c);
      v[q] = z;
    }
  }
  // inverse (unnormalised): the mirror image, ends where inv_s3_load would have left v
  __device__ static void inv_s4_lanes(int t, cf *v, const cf *tw3, const LaneConst &c)
  {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      cf z = v[q];
      z = lane_bfly<0>(z, c);
      if (R4 >= 4) { z = cmulc(z, c.w[1]); z = lane_bfly<1>(z, c); }
      if (R4 == 8) { z = cmulc(z, c.w[2]); z = lane_bfly<2>(z, c); }
      v[q] = z;
    }
#pragma unroll
    for (int q = 1; q < 8; q++) v[q] = cmulc(v[q], tw3[q - 1]);
    dft8<+1>(v);
  }
#endif

private:
  template <int SIGN> B2_HD static void small_dft(cf *w)
  {
    if (R4 == 8) dft8<SIGN>(w);
    else if (R4 == 4) dft4<SIGN>(w[0], w[1], w[2], w[3]);
    else { const cf a = w[0], b = w[1]; w[0] = cadd(a, b); w[1] = csub(a, b); }
  }
};

} // namespace blah2
