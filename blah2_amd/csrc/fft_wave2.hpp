// Two-wave 4096-point complex FFT for gfx950: 128 threads (two wave64s), 32 points per thread in
// registers, ONE exchange through LDS and two two-wave barriers per transform -- the one-wave
// 2048-point transform of fft_wave.hpp one size up.  (The workgroup transform it replaces at
// F = 4096, fft_wg.hpp with R3 = 16: 256 threads, 16 points per thread, two exchanges, three
// four-wave barriers.)
//
// Thread (wave w, lane L) carries the LOGICAL index
//     T = l4 + 16 w + 32 g + 64 h,     l4 = L & 15,  g = (L >> 4) & 1,  h = L >> 5,
// i.e. the two index bits the radix-2 steps pair up are lane bits (partner lanes L +- 32 and L +- 16),
// and the wave index is bit 4.  Input index n = T + 128 k1 (register k1), output index m = T + 128 a
// (register a): self-sorting in this layout, so the inverse is the same code with conjugated roots
// (unnormalised) and a cross spectrum is a register-wise product.
//
//   S1   thread T         : 32-point DFT over k1 -> q, times W_4096^(T q)
//   SW1  lanes L, L +- 32 : v_permlane32_swap on the register pairs (2p, 2p+1): lane h then holds z_q[t] and
//                           z_q[t + 64], t = T mod 64, for the 16 values q = 2p + h
//   B1                    : u0 = z[t] + z[t+64],  u1 = (z[t] - z[t+64]) W_128^t              (e = 0, 1)
//   SW2  lanes L, L +- 16 : v_permlane16_swap on the pairs (u0, u1): lane g then holds u_e[t1] and u_e[t1 + 32],
//                           t1 = l4 + 16 w, for e = g
//   B2                    : s0 = u[t1] + u[t1+32],  s1 = (u[t1] - u[t1+32]) W_64^t1         (b = 0, 1);
//                           write row (q, e, b), column t1 of the exchange region
//   S3   thread R         : read row R (32 values over t1), 32-point DFT over t1 -> a
// because for the 128-point DFT across the threads, t = t1 + 32 t0 + 64 t2 and m2 = 4a + 2b + e give
//   W_128^(t m2) = W_32^(t1 a) W_64^(t1 b) W_128^((t1 + 32 t0) e) (-1)^(t0 b) (-1)^(t2 e),
// and m = q + 32 m2 = (q + 32 e + 64 b) + 128 a = R + 128 a.
//
// LDS.  Exchange region: 128 rows of 32 complex values at a pitch of 33 (33.8 KB).  Row R = q + 32 e + 64 b sits
// in slot sigma(R) = R with bits 4 and 5 swapped: the 32 lanes of a half wave then read 32 consecutive slots
// (conflict-free at the odd pitch), and a writer's slot is a compile-time constant plus a per-lane constant, so
// every access is base + immediate.  Writes go out 16 consecutive columns of one row per 16-lane group.
// Stage twiddles W_4096^(T q), q = 1..31: a [q][T] table would be 31.7 KB per workgroup; they are the product
// of two small tables, W_4096^(l4 q) [q][16] and W_4096^(16 c q) [q][8] with c = w + 2 g + 4 h (5.9 KB, one
// more complex multiply per twiddle) -- which lets a workgroup be ONE pair of waves with its own barriers,
// four of them per CU (39.7 KB each).
//
// Every stage is a pure per-thread function of (thread, registers, LDS) except the two lane swaps, which the
// host emulation (tests/host/emulate_fft.cpp) performs on the two lanes' arrays.
#pragma once

#include "fft_wave.hpp"

#include <stddef.h>

namespace blah2 {

// 32-point DFT whose inputs v[16..31] are zero (not read): the zero-padded half of a reference segment
template <int SIGN> B2_HD void dft32_nz16(cf *v)
{
  cf u[4][8];
#pragma unroll
  for (int n0 = 0; n0 < 4; n0++) {
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++) u[n0][n1] = v[n0 + 4 * n1];
    // 8-point DFT of (u0, u1, u2, u3, 0, 0, 0, 0): both 4-point halves have their last two inputs zero
    dft4_z23<SIGN>(u[n0][0], u[n0][2], u[n0][4], u[n0][6]);
    dft4_z23<SIGN>(u[n0][1], u[n0][3], u[n0][5], u[n0][7]);
    const cf b1 = twid32<SIGN, 4>(u[n0][3]);
    const cf b3 = twid32<SIGN, 12>(u[n0][7]);
    dft8_tail<SIGN>(u[n0], b1, b3);
  }
  dft32_twiddles<SIGN, 1>(u[1]);
  dft32_twiddles<SIGN, 2>(u[2]);
  dft32_twiddles<SIGN, 3>(u[3]);
#pragma unroll
  for (int k1 = 0; k1 < 8; k1++) {
    dft4<SIGN>(u[0][k1], u[1][k1], u[2][k1], u[3][k1]);
#pragma unroll
    for (int k0 = 0; k0 < 4; k0++) v[k1 + 8 * k0] = u[k0][k1];
  }
}

struct Wave2Fft {
  static constexpr int F = 4096;
  static constexpr int NT = 128; // threads
  static constexpr int E = 32;   // points per thread
  static constexpr int P = 33;   // row pitch of the exchange region (complex values)
  static constexpr int X_ELEMS = 128 * P;
  static constexpr int TA_ELEMS = 31 * 16; // W_F^(l4 q),   [q - 1][l4]
  static constexpr int TB_ELEMS = 31 * 8;  // W_F^(16 c q), [q - 1][c]
  static constexpr int TW_ELEMS = TA_ELEMS + TB_ELEMS;
  static constexpr size_t LDS_BYTES = (size_t)(TW_ELEMS + X_ELEMS) * 8;

  // logical index of thread (wave, lane)
  B2_HD static int logical(int wave, int lane) { return (lane & 15) + 16 * wave + 32 * ((lane >> 4) & 1) + 64 * (lane >> 5); }

  struct Tw {
    const cf *ta; // ta[(q - 1) * 16] = W_F^(l4 q)    (the table pointer plus l4)
    const cf *tb; // tb[(q - 1) * 8]  = W_F^(16 c q)  (plus c)
    cf w128;      // W_128^(T mod 64)
    cf w64;       // W_64^(l4 + 16 w)
  };

  // `tw` is the table tw[k] = exp(-2*pi*i*k/F), k in [0, F); `table` the workgroup's LDS copy of the two factor tables
  template <class TW> B2_HD static void fill_table(int tid, int nthreads, const TW *tw, cf *table)
  {
    for (int e = tid; e < TA_ELEMS; e += nthreads) table[e] = tw[(((e >> 4) + 1) * (e & 15)) & (F - 1)];
    for (int e = tid; e < TB_ELEMS; e += nthreads) table[TA_ELEMS + e] = tw[(16 * ((e >> 3) + 1) * (e & 7)) & (F - 1)];
  }
  template <class TW> B2_HD static void load_twiddles(int wave, int lane, const TW *tw, const cf *table, Tw &w)
  {
    const int l4 = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
    w.ta = table + l4;
    w.tb = table + TA_ELEMS + (wave + 2 * g + 4 * h);
    w.w128 = tw[32 * (l4 + 16 * wave + 32 * g)];
    w.w64 = tw[64 * (l4 + 16 * wave)];
  }

  // v[k1] = in[T + 128*k1] on entry; NZ = 16: the inputs v[16..31] are zero and not read
  template <int SIGN, int NZ = 32> B2_HD static void s1(cf *v, const Tw &w)
  {
    static_assert(NZ == 16 || NZ == 32, "");
    if (NZ == 16) dft32_nz16<SIGN>(v);
    else dft32<SIGN>(v);
#pragma unroll
    for (int q = 1; q < 32; q++) v[q] = twid<SIGN>(v[q], cmul(w.ta[(q - 1) * 16], w.tb[(q - 1) * 8]));
  }

#if defined(__HIPCC__)
  // lanes L and L +- 16 exchange: (v[2p], v[2p+1]) <- (v[2p] of the lane with g = 0, ... of the lane with g = 1) for g = 0,
  // (v[2p+1] of g = 0, of g = 1) for g = 1 -- v_permlane16_swap swaps the odd 16-lane rows of its first operand with the
  // even rows of its second.  Asm with its own wait states, like WaveFft::sw (the operands come out of asm statements).
  __device__ __forceinline__ static void sw16(cf *v)
  {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int p = 0; p < 16; p += 4)
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

  // B1: on the pairs SW1 left; w128 = W_128^(T mod 64)
  template <int SIGN> B2_HD static void b1(cf *v, const Tw &w)
  {
#pragma unroll
    for (int p = 0; p < 16; p++) {
      const cf u0 = cadd(v[2 * p], v[2 * p + 1]);
      const cf u1 = twid<SIGN>(csub(v[2 * p], v[2 * p + 1]), w.w128);
      v[2 * p] = u0;
      v[2 * p + 1] = u1;
    }
  }
  // B2 + the exchange writes: on the pairs SW2 left
  template <int SIGN> B2_HD static void b2(int wave, int lane, cf *v, const Tw &w, cf *X)
  {
    const int l4 = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
    cf *base = X + (h + 16 * g) * P + (l4 + 16 * wave); // slot (h + 16 g) + the compile-time part below, column t1
#pragma unroll
    for (int p = 0; p < 16; p++) {
      const cf s0 = cadd(v[2 * p], v[2 * p + 1]);
      const cf s1 = twid<SIGN>(csub(v[2 * p], v[2 * p + 1]), w.w64);
      const int slot = ((2 * p) & 15) + 32 * (p >> 3);
      base[slot * P] = s0;        // b = 0
      base[(slot + 64) * P] = s1; // b = 1
    }
  }
  // leaves out[T + 128*a] in v[a]
  template <int SIGN> B2_HD static void s3(int wave, int lane, cf *v, const cf *X)
  {
    const cf *row = X + ((lane & 31) + 32 * wave + 64 * (lane >> 5)) * P;
#pragma unroll
    for (int t1 = 0; t1 < 32; t1++) v[t1] = row[t1];
    dft32<SIGN>(v);
  }

#if defined(__HIPCC__)
  // The whole transform.  Two workgroup barriers: the workgroup IS the pair of waves.  The second one keeps the next
  // transform's writes behind this one's reads.
  template <int SIGN, int NZ = 32> __device__ __forceinline__ static void transform(int wave, int lane, cf *v, const Tw &w, cf *X)
  {
    s1<SIGN, NZ>(v, w);
    WaveFft::sw(v);
    b1<SIGN>(v, w);
    sw16(v);
    b2<SIGN>(wave, lane, v, w, X);
    __syncthreads();
    s3<SIGN>(wave, lane, v, X);
    __syncthreads();
  }
#endif
};

} // namespace blah2
