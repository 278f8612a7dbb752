// One-wave 1024-point complex FFT for gfx950: ONE wave64, 16 points per lane in registers, two lane-swap
// butterflies, ONE exchange through a wave-private LDS region, no barrier -- the one-wave 2048-point transform
// of fft_wave.hpp at HALF the registers per lane, so that a range kernel built on it (x, y and an accumulator:
// 3 x 32 registers + the 16-point kernel's temporaries) fits four waves per SIMD where the 2048-point one fits two.
// The butterfly count per point is the same (a 32-point step per lane and one lane stage there, a 16-point step
// and two lane stages here).
//
// Input index n = T + 64 k1 (lane T, register k1), output index m = R + 64 a (lane R, register a): self-sorting
// in this layout, so the inverse is the same code with conjugated roots (unnormalised) and a cross spectrum is a
// register-wise product.  With T = t1 + 16 t0 + 32 t2 (t1 = T & 15, t0 = lane bit 4, t2 = lane bit 5):
//
//   S1   lane T           : 16-point DFT over k1 -> q, times W_1024^(T q)
//   SW1  lanes T, T +- 32 : v_permlane32_swap on the register pairs (2p, 2p+1): lane t2 = h then holds z_q[t] and
//                           z_q[t + 32], t = T mod 32, for the 8 values q = 2p + h
//   B1                    : u0 = z[t] + z[t+32],  u1 = (z[t] - z[t+32]) W_64^t               (e = 0, 1)
//   SW2  lanes T, T +- 16 : v_permlane16_swap on the pairs (u0, u1): lane t0 = g then holds u_e[t1] and u_e[t1 + 16]
//                           for e = g
//   B2                    : s0 = u[t1] + u[t1+16],  s1 = (u[t1] - u[t1+16]) W_32^t1          (b = 0, 1);
//                           write row R = q + 16 e + 32 b, column t1 of the exchange region
//   S3   lane R           : read row R (16 values over t1), 16-point DFT over t1 -> a
// because for the 64-point DFT across the lanes, m2 = 4a + 2b + e gives
//   W_64^(T m2) = W_16^(t1 a) W_32^(t1 b) W_64^((t1 + 16 t0) e) (-1)^(t0 b) (-1)^(t2 e),
// and m = q + 16 m2 = (q + 16 e + 32 b) + 64 a = R + 64 a.
//
// LDS.  Exchange region: 64 rows of 16 complex values at a pitch of 17 (8.5 KB per wave): a reader's 32 lanes of a
// ds_read_b64 group hit 32 distinct bank pairs (17 R + t1 mod 32), a writer's address is a per-lane constant
// ((h + 16 g) 17 + t1) plus a compile-time one ((2p + 32 b) 17), and the two rows a 32-lane write group touches
// (R, R + 16) sit 16 bank pairs apart.  Stage twiddles W_1024^(T q), q = 1..15: a [q][T] table, 7.5 KB per workgroup.
//
// Every stage is a pure per-lane function of (lane, registers, LDS) except the two lane swaps, which the host
// emulation (tests/host/emulate_fft.cpp) performs on the two lanes' arrays.
#pragma once

#include "fft_wave.hpp"

#include <stddef.h>

namespace blah2 {

// 16-point DFT of which only the outputs v[0..6] are wanted (the inverse transform of a lag window of at most
// 7*64 lags).  k = k1 + 4 k0: both k0 = 0, 1 for k1 = 0, 1, 2 and k0 = 0 for k1 = 3.  v[7..15] are left undefined.
template <int SIGN> B2_HD void dft16_out7(cf *v)
{
  dft4<SIGN>(v[0], v[4], v[8], v[12]);
  dft4<SIGN>(v[1], v[5], v[9], v[13]);
  dft4<SIGN>(v[2], v[6], v[10], v[14]);
  dft4<SIGN>(v[3], v[7], v[11], v[15]);
  // now v[n0 + 4*k1]; twiddle by W16^(n0*k1)
  v[1 + 4] = twid32<SIGN, 2>(v[1 + 4]);
  v[2 + 4] = twid32<SIGN, 4>(v[2 + 4]);
  v[3 + 4] = twid32<SIGN, 6>(v[3 + 4]);
  v[1 + 8] = twid32<SIGN, 4>(v[1 + 8]);
  v[2 + 8] = SIGN < 0 ? cmake(v[2 + 8].y, -v[2 + 8].x) : cmake(-v[2 + 8].y, v[2 + 8].x); // W16^4 = (SIGN i)
  v[3 + 8] = twid32<SIGN, 12>(v[3 + 8]);
  v[1 + 12] = twid32<SIGN, 6>(v[1 + 12]);
  v[2 + 12] = twid32<SIGN, 12>(v[2 + 12]);
  v[3 + 12] = twid32<SIGN, 18>(v[3 + 12]);
  cf o[7];
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) {
    const cf a0 = v[4 * k1], a1 = v[4 * k1 + 1], a2 = v[4 * k1 + 2], a3 = v[4 * k1 + 3];
    o[k1] = cadd(cadd(a0, a2), cadd(a1, a3));
    if (k1 < 3) o[k1 + 4] = cadd_i<SIGN>(csub(a0, a2), csub(a1, a3));
  }
#pragma unroll
  for (int k = 0; k < 7; k++) v[k] = o[k];
}

// 4-point DFT from four values that stay untouched (separate single-instruction statements: the compiler is free to put
// the results elsewhere) -- the first step of a transform whose raw inputs are still wanted afterwards
template <int SIGN> B2_HD void dft4_nd(cf &o0, cf &o1, cf &o2, cf &o3, cf a0, cf a1, cf a2, cf a3)
{
  const cf t0 = cadd(a0, a2), t1 = csub(a0, a2);
  const cf t2 = cadd(a1, a3), d = csub(a1, a3);
  o0 = cadd(t0, t2);
  o1 = cadd_i<SIGN>(t1, d);
  o2 = csub(t0, t2);
  o3 = csub_i<SIGN>(t1, d);
}
// 16-point DFT of in[0..16) into out[0..16); `in` is not written
template <int SIGN> B2_HD void dft16_nd(cf *out, const cf *in)
{
#pragma unroll
  for (int n0 = 0; n0 < 4; n0++) dft4_nd<SIGN>(out[n0], out[n0 + 4], out[n0 + 8], out[n0 + 12], in[n0], in[n0 + 4], in[n0 + 8], in[n0 + 12]);
  dft16_tail<SIGN>(out);
}

struct Wave1kFft {
  static constexpr int F = 1024;
  static constexpr int L = 64;  // lanes
  static constexpr int E = 16;  // points per lane
  static constexpr int P = 17;  // row pitch of the exchange region (complex values)
  static constexpr int X_ELEMS = 64 * P;
  static constexpr int TW_ELEMS = 15 * 64; // stage-twiddle table, [q - 1][T]

  struct Tw {
    const cf *tab; // tab[(q - 1) * 64] = W_F^(T q) for this lane (the table pointer plus T)
    cf w64;        // W_64^(T & 31)
    cf w32;        // W_32^(T & 15)
  };

  // `tw` is the table tw[k] = exp(-2*pi*i*k/F), k in [0, F); `table` the workgroup's LDS copy
  template <class TW> B2_HD static void fill_table(int tid, int nthreads, const TW *tw, cf *table)
  {
    for (int e = tid; e < TW_ELEMS; e += nthreads) table[e] = tw[(((e >> 6) + 1) * (e & 63)) & (F - 1)];
  }
  template <class TW> B2_HD static void load_twiddles(int t, const TW *tw, const cf *table, Tw &w)
  {
    w.tab = table + t;
    w.w64 = tw[16 * (t & 31)];
    w.w32 = tw[32 * (t & 15)];
  }

  // v[k1] = in[T + 64*k1] on entry; NZ = 9: the inputs v[9..15] are zero and not read
  template <int SIGN, int NZ = 16> B2_HD static void s1(cf *v, const Tw &w)
  {
    static_assert(NZ == 9 || NZ == 16, "");
    if (NZ == 9) dft16_nz9<SIGN>(v);
    else dft16<SIGN>(v);
#pragma unroll
    for (int q = 1; q < 16; q++) v[q] = twid<SIGN>(v[q], w.tab[(q - 1) * 64]);
  }

#if defined(__HIPCC__)
  // Both swaps as asm with their own two wait states, like WaveFft::sw: the operands come out of asm statements the
  // compiler's hazard recogniser cannot see into.
  __device__ __forceinline__ static void sw32(cf *v)
  {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int p = 0; p < 8; p += 4)
      asm("s_nop 1\n\t"
          "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
          "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
          "v_permlane32_swap_b32 %8, %9\n\tv_permlane32_swap_b32 %10, %11\n\t"
          "v_permlane32_swap_b32 %12, %13\n\tv_permlane32_swap_b32 %14, %15"
          : "+v"(v[2 * p].x), "+v"(v[2 * p + 1].x), "+v"(v[2 * p].y), "+v"(v[2 * p + 1].y),
            "+v"(v[2 * p + 2].x), "+v"(v[2 * p + 3].x), "+v"(v[2 * p + 2].y), "+v"(v[2 * p + 3].y),
            "+v"(v[2 * p + 4].x), "+v"(v[2 * p + 5].x), "+v"(v[2 * p + 4].y), "+v"(v[2 * p + 5].y),
            "+v"(v[2 * p + 6].x), "+v"(v[2 * p + 7].x), "+v"(v[2 * p + 6].y), "+v"(v[2 * p + 7].y));
#endif
  }
  __device__ __forceinline__ static void sw16(cf *v)
  {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int p = 0; p < 8; p += 4)
      asm("s_nop 1\n\t"
          "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
          "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
          "v_permlane16_swap_b32 %8, %9\n\tv_permlane16_swap_b32 %10, %11\n\t"
          "v_permlane16_swap_b32 %12, %13\n\tv_permlane16_swap_b32 %14, %15"
          : "+v"(v[2 * p].x), "+v"(v[2 * p + 1].x), "+v"(v[2 * p].y), "+v"(v[2 * p + 1].y),
            "+v"(v[2 * p + 2].x), "+v"(v[2 * p + 3].x), "+v"(v[2 * p + 2].y), "+v"(v[2 * p + 3].y),
            "+v"(v[2 * p + 4].x), "+v"(v[2 * p + 5].x), "+v"(v[2 * p + 4].y), "+v"(v[2 * p + 5].y),
            "+v"(v[2 * p + 6].x), "+v"(v[2 * p + 7].x), "+v"(v[2 * p + 6].y), "+v"(v[2 * p + 7].y));
#endif
  }
#endif
  // what either swap does, on the register arrays of the lane with the swapped bit clear (lo) and set (hi)
  static inline void sw_host(cf *lo, cf *hi)
  {
    for (int p = 0; p < 8; p++) {
      const cf a_hi = hi[2 * p], b_lo = lo[2 * p + 1];
      lo[2 * p + 1] = a_hi;
      hi[2 * p] = b_lo;
    }
  }

  // B1: on the pairs SW1 left
  template <int SIGN> B2_HD static void b1(cf *v, const Tw &w)
  {
#pragma unroll
    for (int p = 0; p < 8; p++) {
      const cf u0 = cadd(v[2 * p], v[2 * p + 1]);
      const cf u1 = twid<SIGN>(csub(v[2 * p], v[2 * p + 1]), w.w64);
      v[2 * p] = u0;
      v[2 * p + 1] = u1;
    }
  }
  // B2 + the exchange writes: on the pairs SW2 left
  template <int SIGN> B2_HD static void b2(int t, cf *v, const Tw &w, cf *X)
  {
    const int l4 = t & 15, g = (t >> 4) & 1, h = t >> 5;
    cf *base = X + (h + 16 * g) * P + l4; // row (q = 2p + h) + 16 e (= g) + 32 b, column t1
#pragma unroll
    for (int p = 0; p < 8; p++) {
      const cf s0 = cadd(v[2 * p], v[2 * p + 1]);
      const cf s1 = twid<SIGN>(csub(v[2 * p], v[2 * p + 1]), w.w32);
      base[(2 * p) * P] = s0;      // b = 0
      base[(2 * p + 32) * P] = s1; // b = 1
    }
  }
  // leaves out[T + 64*a] in v[a]; OUT7: only a < 7 (the rest of v is undefined)
  template <int SIGN, bool OUT7 = false> B2_HD static void s3(int t, cf *v, const cf *X)
  {
#pragma unroll
    for (int t1 = 0; t1 < 16; t1++) v[t1] = X[t * P + t1];
    if (OUT7) dft16_out7<SIGN>(v);
    else dft16<SIGN>(v);
  }

  // S1 out of place: out = the 16-point DFT of `in` times the stage twiddles; `in` keeps its values
  template <int SIGN> B2_HD static void s1_nd(cf *out, const cf *in, const Tw &w)
  {
    dft16_nd<SIGN>(out, in);
#pragma unroll
    for (int q = 1; q < 16; q++) out[q] = twid<SIGN>(out[q], w.tab[(q - 1) * 64]);
  }

#if defined(__HIPCC__)
  // everything after S1
  template <int SIGN, bool OUT7 = false> __device__ __forceinline__ static void finish(int t, cf *v, const Tw &w, cf *X)
  {
    sw32(v);
    b1<SIGN>(v, w);
    sw16(v);
    b2<SIGN>(t, v, w, X);
    s3<SIGN, OUT7>(t, v, X);
  }
  template <int SIGN, int NZ = 16, bool OUT7 = false> __device__ __forceinline__ static void transform(int t, cf *v, const Tw &w, cf *X)
  {
    s1<SIGN, NZ>(v, w);
    sw32(v);
    b1<SIGN>(v, w);
    sw16(v);
    b2<SIGN>(t, v, w, X);
    s3<SIGN, OUT7>(t, v, X);
  }
#endif
};

} // namespace blah2
