// Wave-level 2048-point complex FFT for gfx950: ONE wave64, 32 points per lane in
// registers, ONE exchange through a wave-private LDS region, no workgroup barrier.
//
// Input index n = t + 64*k1 (lane t, register k1), output index m = R + 64*a (lane R,
// register a): the transform is self-sorting in this layout, so the inverse is the same
// code with SIGN = +1 (conjugated roots, unnormalised), and a cross spectrum is a
// register-wise product.
//
//   S1  lane t          : 32-point DFT over k1 -> q, times W_2048^(t*q)
//   SW  lanes t, t+32   : v_permlane32_swap on the register pairs (2p, 2p+1): afterwards lane
//                         (h, t1) = (t >> 5, t & 31) holds z_q[t1] and z_q[t1 + 32] for the 16
//                         values q = 2p + h
//   S2  lane (h, t1)    : u0 = z[t1] + z[t1+32], u1 = (z[t1] - z[t1+32]) * W_64^t1;
//                         write rows X[q][t1] = u0, X[q + 32][t1] = u1
//   S3  lane R          : read row R (32 values), 32-point DFT over t1 -> a
// because for the 64-point DFT across the lanes, t = t1 + 32*t0 and m2 = 2a + b give
//   W_64^(t*m2) = W_32^(t1*a) * W_64^(t1*b) * (-1)^(t0*b),
// and m = q + 32*m2 = (q + 32*b) + 64*a = R + 64*a.
//
// Against the workgroup transform of fft_wg.hpp (16 points per thread, two exchanges, three
// barriers per transform): half the LDS traffic, no barriers, the same number of butterflies.
//
// LDS: rows of 32 complex values at a pitch of 33 (S3 reads one row per lane: 32 lanes of a
// ds_read_b64 group hit 32 distinct bank pairs; S2 writes 16 consecutive t1 of one row per
// lane group).  64 * 33 * 8 B = 16.5 KB per wave.
//
// Stage twiddles W_2048^(t*q), q = 1..31, come from a table in LDS laid out [q][t] (one
// conflict-free ds_read_b64 per value, shared by the waves of a workgroup: 15.9 KB) -- 62
// registers would not fit beside x, y and the accumulator (3 x 64) at two waves per SIMD.
//
// Every stage is a pure per-lane function of (lane, registers, LDS) except SW, which the
// host emulation (tests/host/emulate_fft.cpp) performs on the two lanes' arrays.
#pragma once

#include "fft_wg.hpp"

namespace blah2 {

// u[k1] *= W_32^(N0*k1), k1 = 1..7 (compile-time exponents)
template <int SIGN, int N0, int K1 = 1> B2_HD void dft32_twiddles(cf *u)
{
  if constexpr (K1 < 8) {
    u[K1] = twid32<SIGN, N0 * K1>(u[K1]);
    dft32_twiddles<SIGN, N0, K1 + 1>(u);
  }
}

// 32-point DFT, in place, natural order out.  n = n0 + 4*n1, k = k1 + 8*k0.
// NZ = 24 / 28: the inputs v[NZ..31] are zero (zero padding of a segment window) and are not read.
template <int SIGN, int NZ = 32> B2_HD void dft32(cf *v)
{
  static_assert(NZ == 24 || NZ == 28 || NZ == 32, "");
  cf u[4][8];
#pragma unroll
  for (int n0 = 0; n0 < 4; n0++) {
#pragma unroll
    for (int n1 = 0; n1 < NZ / 4; n1++) u[n0][n1] = v[n0 + 4 * n1];
    dft8_z<SIGN, NZ / 4>(u[n0]); // -> u[n0][k1]
  }
  dft32_twiddles<SIGN, 1>(u[1]);
  dft32_twiddles<SIGN, 2>(u[2]);
  dft32_twiddles<SIGN, 3>(u[3]);
#pragma unroll
  for (int k1 = 0; k1 < 8; k1++) {
    dft4<SIGN>(u[0][k1], u[1][k1], u[2][k1], u[3][k1]); // -> u[k0][k1]
#pragma unroll
    for (int k0 = 0; k0 < 4; k0++) v[k1 + 8 * k0] = u[k0][k1];
  }
}

// 32-point DFT of which only the outputs v[0..6] are wanted (the inverse transform of a lag window of at
// most 7*64 lags): the DFT over n0 reduces to the k0 = 0 sums for k1 = 0..6.  v[7..31] are left undefined.
template <int SIGN> B2_HD void dft32_out7(cf *v)
{
  cf u[4][8];
#pragma unroll
  for (int n0 = 0; n0 < 4; n0++) {
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) u[n0][n1] = v[n0 + 4 * n1];
    dft8<SIGN>(u[n0]); // -> u[n0][k1]
  }
  dft32_twiddles<SIGN, 1>(u[1]);
  dft32_twiddles<SIGN, 2>(u[2]);
  dft32_twiddles<SIGN, 3>(u[3]);
#pragma unroll
  for (int k1 = 0; k1 < 7; k1++) v[k1] = cadd(cadd(u[0][k1], u[2][k1]), cadd(u[1][k1], u[3][k1]));
}

struct WaveFft {
  static constexpr int F = 2048;
  static constexpr int L = 64;  // lanes
  static constexpr int E = 32;  // points per lane
  static constexpr int P = 33;  // row pitch of the exchange region (complex values)
  static constexpr int X_ELEMS = 64 * P;

  static constexpr int TW_ELEMS = 31 * 64; // stage-twiddle table, [q - 1][t]

  struct Tw {
    const cf *tab; // tab[(q - 1) * 64] = W_F^(t*q) for this lane (the table pointer plus t)
    cf w64;        // W_64^(t & 31)
  };

  // `tw` is the table tw[k] = exp(-2*pi*i*k/F), k in [0, F); `table` the workgroup's LDS copy
  // (or any [q - 1][t] array), filled by fill_table
  template <class TW> B2_HD static void fill_table(int tid, int nthreads, const TW *tw, cf *table)
  {
    for (int e = tid; e < TW_ELEMS; e += nthreads) table[e] = tw[(((e >> 6) + 1) * (e & 63)) & (F - 1)];
  }
  template <class TW> B2_HD static void load_twiddles(int t, const TW *tw, const cf *table, Tw &w)
  {
    w.tab = table + t;
    w.w64 = tw[32 * (t & 31)];
  }

  // v[k1] = in[t + 64*k1] on entry
  template <int SIGN, int NZ = 32> B2_HD static void s1(cf *v, const Tw &w)
  {
    dft32<SIGN, NZ>(v);
#pragma unroll
    for (int q = 1; q < 32; q++) v[q] = twid<SIGN>(v[q], w.tab[(q - 1) * 64]);
  }

#if defined(__HIPCC__)
  // lanes t and t + 32 exchange: (v[2p], v[2p+1]) <- (z_q[t1], z_q[t1+32]), q = 2p + (t >> 5)
  __device__ __forceinline__ static void sw(cf *v)
  {
#if defined(__HIP_DEVICE_COMPILE__)
    // Written as asm with its own two wait states (once per group of eight swaps): the operands were
    // just produced by inline-asm packed instructions, which the compiler's hazard recogniser
    // (VALU write -> v_permlane read) cannot see into, so the builtin form gets no s_nop and reads
    // stale registers.
#pragma unroll
    for (int p = 0; p < 16; p += 4)
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
#endif
  // what sw() does, on the register arrays of lanes t1 (lo) and t1 + 32 (hi)
  static inline void sw_host(cf *lo, cf *hi)
  {
    for (int p = 0; p < 16; p++) {
      const cf a_lo = lo[2 * p], b_lo = lo[2 * p + 1], a_hi = hi[2 * p], b_hi = hi[2 * p + 1];
      lo[2 * p] = a_lo; lo[2 * p + 1] = a_hi;
      hi[2 * p] = b_lo; hi[2 * p + 1] = b_hi;
    }
  }

  template <int SIGN> B2_HD static void s2(int t, cf *v, const Tw &w, cf *X)
  {
    const int h = t >> 5, t1 = t & 31;
#pragma unroll
    for (int p = 0; p < 16; p++) {
      const cf u0 = cadd(v[2 * p], v[2 * p + 1]);
      const cf u1 = twid<SIGN>(csub(v[2 * p], v[2 * p + 1]), w.w64);
      const int q = 2 * p + h;
      X[q * P + t1] = u0;
      X[(q + 32) * P + t1] = u1;
    }
  }
  // leaves out[t + 64*a] in v[a]; OUT7: only a < 7 (the rest of v is undefined)
  template <int SIGN, bool OUT7 = false> B2_HD static void s3(int t, cf *v, const cf *X)
  {
#pragma unroll
    for (int t1 = 0; t1 < 32; t1++) v[t1] = X[t * P + t1];
    if (OUT7) dft32_out7<SIGN>(v);
    else dft32<SIGN>(v);
  }

#if defined(__HIPCC__)
  template <int SIGN, int NZ = 32, bool OUT7 = false> __device__ __forceinline__ static void transform(int t, cf *v, const Tw &w, cf *X)
  {
    s1<SIGN, NZ>(v, w);
    sw(v);
    s2<SIGN>(t, v, w, X);
    s3<SIGN, OUT7>(t, v, X);
  }
#endif
};

} // namespace blah2
