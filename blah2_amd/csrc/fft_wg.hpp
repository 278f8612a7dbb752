// Workgroup-level power-of-two complex FFT for gfx950, F = 16 * 16 * R3 points
// (R3 = 4, 8, 16  ->  F = 1024, 2048, 4096), T = 16 * R3 threads, 16 points
// per thread held in registers.  Two LDS exchanges per transform.
//
// Forward (decimation in frequency), input index n = n' + T*k1 with
// n' = u + R3*k2 the thread id of stage 1, output index m = q + 16*r + 256*s:
//   S1  thread n'        : 16-point DFT over k1 -> q, times W_F^(n'*q), write A[q][n']
//   S2  thread (q, u)    : read A[q][u + R3*k2], 16-point DFT over k2 -> r, write B[q][u][r]
//   S3  thread pair(q,r) : read B[q][u][r], times W_T^(u*r), R3-point DFT over u -> s
// The spectrum is left in that permuted register layout; X and Y use the same
// layout, so the cross spectrum Y*conj(X) is an element-wise register product.
// The inverse is the exact mirror (decimation in time), starting from the
// permuted layout and ending in natural order z[n' + T*c]:
//   I   pair(q,r)        : inverse R3-point DFT over s -> a, times conj W_T^(r*a), write B[q][a][r]
//   II  thread (q, a)    : read B[q][a][r], inverse 16-point DFT over r -> b, write A[q][a + R3*b]
//   III thread n'        : read A[q][n'], times conj W_F^(n'*q), inverse 16-point DFT over q -> c
// so no bit-reversal pass exists anywhere.  Twiddles W_F^(n'*q) (15 values)
// and W_T^(r*u) (R3-1 values) live in registers for the lifetime of the
// workgroup and serve both directions.
//
// LDS layout (complex fp32).  hipcc pairs the 8-byte accesses of a thread into
// ds_read2_b64 / ds_write2_b64, which the LDS services in groups of 16
// consecutive lanes over 32 dword banks (MI355X_MICROARCH.md, LDS table), so a
// group is conflict-free iff its 16 complex indices are distinct mod 16:
//   A[q*PA + n'],        PA = 17*R3: S1 / III touch 16 consecutive n';
//                        S2 / II groups hold 16/R3 values of q x R3 values of u and
//                        q*PA = q*R3 (mod 16) spreads them.
//   B[q*PB + u*17 + r],  PB = 17*R3: S3 / I groups are 16 consecutive r; S2 / II
//                        groups see (q*R3 + u) mod 16, all distinct.
// Both buffers have the same size, 272*R3 elements.
//
// Every stage is a pure per-thread function of (thread id, registers, LDS), so
// the same code is compiled for the host by tests/host/emulate_fft.cpp, which
// runs the threads of a workgroup one after another between barriers to check
// the index algebra without a GPU.
#pragma once

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define B2_HD __host__ __device__ __forceinline__
#else
#define B2_HD inline
#endif

namespace blah2 {

// Complex fp32 as a plain {re, im} pair.
//
// Device code computes on the pair with gfx950's PACKED fp32 instructions (v_pk_add_f32,
// v_pk_mul_f32, v_pk_fma_f32: two lanes' worth of fp32 per instruction, VOP3P): a complex add is one
// instruction, a complex multiply two, and the swaps and sign flips of x(-i), conj() and
// multiply-accumulate are the instructions' op_sel / neg_lo / neg_hi operand modifiers, i.e. free.
// A 2048-point transform drops from 545 to ~290 VALU instructions per thread and the range kernel from
// 230 to 190 VGPRs.  It is NOT faster per flop: a packed instruction occupies a wave for twice as long
// as a scalar one (measured: the range kernel's time is unchanged, the register-starved correlation
// kernel gains 16 %), so what it buys is registers and instruction-cache footprint.
// The instructions are written as inline asm: left to the compiler (SLP vectoriser, or a float2
// vector type -- both tried in round 1) the packed forms came with a v_mov per repacked operand and
// register-tuple pressure that cost more than they saved; the library is still built with
// -fno-slp-vectorize so that the remaining scalar code stays scalar.  Host code (the emulation test)
// uses the scalar definitions; define B2_SCALAR_COMPLEX to get them on the device too.
struct alignas(8) cf {
  float x, y;
};

B2_HD cf cmake(float x, float y) { cf r; r.x = x; r.y = y; return r; }

#if defined(__HIP_DEVICE_COMPILE__) && !defined(B2_SCALAR_COMPLEX)
#define B2_PACKED_COMPLEX 1
typedef float b2_pk __attribute__((ext_vector_type(2)));
#define B2_V(a) __builtin_bit_cast(b2_pk, a)
#define B2_C(v) __builtin_bit_cast(cf, v)
// op_sel[i] / op_sel_hi[i]: which half of source i feeds the low / high result (default 0 / 1)
__device__ __forceinline__ cf cadd(cf a, cf b) { b2_pk r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(B2_V(a)), "v"(B2_V(b))); return B2_C(r); }
__device__ __forceinline__ cf csub(cf a, cf b) { b2_pk r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(B2_V(a)), "v"(B2_V(b))); return B2_C(r); }
// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ cf cadd_mi(cf a, cf b) { b2_pk r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(B2_V(a)), "v"(B2_V(b))); return B2_C(r); }
// a + (+i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ cf cadd_pi(cf a, cf b) { b2_pk r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(B2_V(a)), "v"(B2_V(b))); return B2_C(r); }
// A complex multiply is a dependent pair (v_pk_mul, v_pk_fma).  Both instructions sit in ONE asm
// statement: between two asm statements of which the second reads what the first wrote, the
// compiler's hazard recogniser -- which cannot see what the statements are -- inserts an s_nop (one in
// six instructions of a transform were such nops); the hardware interlocks plain VALU dependencies
// itself.  The result is early-clobber because the second instruction still reads both sources.
// a * b
__device__ __forceinline__ cf cmul(cf a, cf b)
{
  b2_pk r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"                                              // (a.x b.x, a.y b.x)
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"                // (-a.y b.y, a.x b.y) + that
      : "=&v"(r) : "v"(B2_V(a)), "v"(B2_V(b)));
  return B2_C(r);
}
// a * conj(b)
__device__ __forceinline__ cf cmulc(cf a, cf b)
{
  b2_pk r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"                // (a.y b.y, -a.x b.y) + that
      : "=&v"(r) : "v"(B2_V(a)), "v"(B2_V(b)));
  return B2_C(r);
}
// acc + a * conj(b)
__device__ __forceinline__ cf cmacc(cf acc, cf a, cf b)
{
  b2_pk r = B2_V(acc);
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\t"                                        // acc + (a.x b.x, a.y b.x)
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
      : "+v"(r) : "v"(B2_V(a)), "v"(B2_V(b)));
  return B2_C(r);
}
// a * w / a * conj(w) for a compile-time constant w: the pair lives in SGPRs
template <int SIGN> __device__ __forceinline__ cf twid_k(cf a, float wx, float wy)
{
  const b2_pk w = {wx, wy};
  b2_pk r;
  if (SIGN < 0)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=&v"(r) : "v"(B2_V(a)), "s"(w));
  else
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=&v"(r) : "v"(B2_V(a)), "s"(w));
  return B2_C(r);
}
// a + (SIGN i) b   (SIGN < 0: the forward transform's -i)
template <int SIGN> __device__ __forceinline__ cf cadd_i(cf a, cf b) { return SIGN < 0 ? cadd_mi(a, b) : cadd_pi(a, b); }
// a - (SIGN i) b
template <int SIGN> __device__ __forceinline__ cf csub_i(cf a, cf b) { return SIGN < 0 ? cadd_pi(a, b) : cadd_mi(a, b); }
#else
B2_HD cf cadd(cf a, cf b) { return cmake(a.x + b.x, a.y + b.y); }
B2_HD cf csub(cf a, cf b) { return cmake(a.x - b.x, a.y - b.y); }
// a * b
B2_HD cf cmul(cf a, cf b) { return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * conj(b)
B2_HD cf cmulc(cf a, cf b) { return cmake(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
// acc + a * conj(b)
B2_HD cf cmacc(cf acc, cf a, cf b)
{
  return cmake(acc.x + (a.x * b.x + a.y * b.y), acc.y + (a.y * b.x - a.x * b.y));
}
template <int SIGN> B2_HD cf twid_k(cf a, float wx, float wy) { return SIGN < 0 ? cmul(a, cmake(wx, wy)) : cmulc(a, cmake(wx, wy)); }
// a + (SIGN i) b,  a - (SIGN i) b
template <int SIGN> B2_HD cf cadd_i(cf a, cf b) { return SIGN < 0 ? cmake(a.x + b.y, a.y - b.x) : cmake(a.x - b.y, a.y + b.x); }
template <int SIGN> B2_HD cf csub_i(cf a, cf b) { return SIGN < 0 ? cmake(a.x - b.y, a.y + b.x) : cmake(a.x + b.y, a.y - b.x); }
#endif
// multiply by -i (SIGN < 0) or +i (SIGN > 0)
template <int SIGN> B2_HD cf mul_i(cf a)
{
  return SIGN < 0 ? cmake(a.y, -a.x) : cmake(-a.y, a.x);
}
// a * w for SIGN < 0, a * conj(w) for SIGN > 0 (w is always the forward root)
template <int SIGN> B2_HD cf twid(cf a, cf w) { return SIGN < 0 ? cmul(a, w) : cmulc(a, w); }

// W_32^j = exp(-2 pi i j / 32): (cos, -sin) = (w32_c(j), w32_s(j))
B2_HD constexpr float w32_c(int j)
{
  constexpr float c[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                          0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                          0.19509032201612826785f, 0.0f};
  j &= 31;
  return j <= 8 ? c[j] : j <= 16 ? -c[16 - j] : j <= 24 ? -c[j - 16] : c[32 - j];
}
B2_HD constexpr float w32_s(int j) { return w32_c(j + 8); } // -sin(2 pi j / 32) = cos(2 pi (j + 8) / 32)

// a * W_32^J (SIGN < 0) or a * conj(W_32^J) (SIGN > 0) for a compile-time J: the constant twiddles of
// the 8-, 16- and 32-point kernels.  Every such root is u * (c_k + i sigma s_k) with u in {1, i, -1, -i},
// sigma = +-1 and k in [0, 4], (c_k, s_k) = (cos, sin)(2 pi k / 32): the device form keeps ONE scalar
// register pair (c_k, s_k) per k and expresses u, sigma and the conjugation in the packed instructions'
// operand-select and negate modifiers -- five constant pairs for all transforms instead of one pair per
// distinct root (which overflowed the scalar registers of the 32-point kernel into v_readlane spills).
#if defined(B2_PACKED_COMPLEX)
// t = (a.x, a.y) * (SX w[CX]);  r = (-a.y, a.x) * (SY w[CY]) + t, with NX / NY = 1 for negative SX / SY
#define B2_TWK(CX, NX, CY, NLO, NHI)                                                                              \
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0," #CX "] op_sel_hi:[1," #CX "] neg_lo:[0," #NX "] neg_hi:[0," #NX "]\n\t" \
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1," #CY ",0] op_sel_hi:[0," #CY ",1] neg_lo:[0," #NLO ",0] neg_hi:[0," #NHI ",0]" \
      : "=&v"(r) : "v"(B2_V(a)), "s"(w))
template <int SIGN, int J> __device__ __forceinline__ cf twid32(cf a)
{
  constexpr int j = J & 31, quad = j >> 3, rem = j & 7;
  constexpr int k = rem <= 4 ? rem : 8 - rem;           // W^rem = W^k or (-i) conj(W^k)
  constexpr int e = (quad + (rem > 4 ? 1 : 0)) & 3;     // forward root = (-i)^e * (c_k - i s_k) or (-i)^e * (c_k + i s_k)
  constexpr int sig_f = rem > 4 ? +1 : -1;              // sign of the imaginary part of the base, forward
  // conjugation (SIGN > 0): u -> conj(u), sigma -> -sigma
  constexpr int sigma = SIGN < 0 ? sig_f : -sig_f;
  constexpr int u = SIGN < 0 ? (4 - e) & 3 : e;         // u = i^u:  (-i)^e = i^(4-e); conj -> i^e
  // m = i^u (c + i sigma s):  u=0: (c, sigma s)  u=1: (-sigma s, c)  u=2: (-c, -sigma s)  u=3: (sigma s, -c)
  constexpr int cx = (u & 1), cy = 1 - cx;
  constexpr int sx = u == 0 ? +1 : u == 1 ? -sigma : u == 2 ? -1 : sigma;
  constexpr int sy = u == 0 ? sigma : u == 1 ? +1 : u == 2 ? -sigma : -1;
  const b2_pk w = {w32_c(k), -w32_s(k)}; // (cos, sin), both >= 0
  b2_pk r;
  if constexpr (j == 0) return a;
  else if constexpr (cx == 0 && sx > 0 && sy > 0) B2_TWK(0, 0, 1, 1, 0);
  else if constexpr (cx == 0 && sx > 0 && sy < 0) B2_TWK(0, 0, 1, 0, 1);
  else if constexpr (cx == 0 && sx < 0 && sy > 0) B2_TWK(0, 1, 1, 1, 0);
  else if constexpr (cx == 0 && sx < 0 && sy < 0) B2_TWK(0, 1, 1, 0, 1);
  else if constexpr (cx == 1 && sx > 0 && sy > 0) B2_TWK(1, 0, 0, 1, 0);
  else if constexpr (cx == 1 && sx > 0 && sy < 0) B2_TWK(1, 0, 0, 0, 1);
  else if constexpr (cx == 1 && sx < 0 && sy > 0) B2_TWK(1, 1, 0, 1, 0);
  else B2_TWK(1, 1, 0, 0, 1);
  return B2_C(r);
}
#undef B2_TWK
#else
template <int SIGN, int J> B2_HD cf twid32(cf a) { return twid_k<SIGN>(a, w32_c(J), w32_s(J)); }
#endif

// 4-point DFT, in place, natural order out.  SIGN = -1 forward, +1 inverse.
#if defined(B2_PACKED_COMPLEX) && !defined(B2_SPLIT_BUTTERFLIES)
// The device forms are ONE asm statement per butterfly group (eight packed adds): a statement boundary
// between dependent packed instructions costs an s_nop from the compiler's hazard recogniser (see cmul).
// One-wave range kernel -4.6 %, workgroup range kernel -1 % (measured).  The grouped form holds two
// more registers per group while it runs; a translation unit whose kernels sit at the register cap
// (clutter.hip: the correlation kernels) defines B2_SPLIT_BUTTERFLIES to keep one statement per add.
#define B2_SUB " neg_lo:[0,1] neg_hi:[0,1]"
#define B2_AMI " op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" /* a + (-i) b */
#define B2_API " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" /* a + (+i) b */
// %0..%3 = a0..a3 (a0, a1 end up as scratch; a2, a3 receive outputs 2, 3), %4, %5 = outputs 0, 1
#define B2_DFT4_BODY(F0, F1, PI, MI)                                      \
  "v_pk_add_f32 %4, %0, %2" F0 "\n\t"  /* t0 */                           \
  "v_pk_add_f32 %0, %0, %2" F1 "\n\t"  /* t1 */                           \
  "v_pk_add_f32 %5, %1, %3\n\t"        /* t2 */                           \
  "v_pk_add_f32 %1, %1, %3" B2_SUB "\n\t" /* d */                         \
  "v_pk_add_f32 %2, %4, %5" B2_SUB "\n\t" /* out2 = t0 - t2 */            \
  "v_pk_add_f32 %4, %4, %5\n\t"        /* out0 = t0 + t2 */               \
  "v_pk_add_f32 %5, %0, %1" PI "\n\t"  /* out1 = t1 + (SIGN i) d */       \
  "v_pk_add_f32 %3, %0, %1" MI           /* out3 = t1 - (SIGN i) d */
template <int SIGN> __device__ __forceinline__ void dft4(cf &a0, cf &a1, cf &a2, cf &a3)
{
  b2_pk x0 = B2_V(a0), x1 = B2_V(a1), x2 = B2_V(a2), x3 = B2_V(a3), o0, o1;
  if (SIGN < 0) asm(B2_DFT4_BODY("", B2_SUB, B2_AMI, B2_API) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&v"(o0), "=&v"(o1));
  else asm(B2_DFT4_BODY("", B2_SUB, B2_API, B2_AMI) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&v"(o0), "=&v"(o1));
  a0 = B2_C(o0); a1 = B2_C(o1); a2 = B2_C(x2); a3 = B2_C(x3);
}
// the same with its third input still to be multiplied by (SIGN i)
template <int SIGN> __device__ __forceinline__ void dft4_rot2(cf &a0, cf &a1, cf &a2r, cf &a3)
{
  b2_pk x0 = B2_V(a0), x1 = B2_V(a1), x2 = B2_V(a2r), x3 = B2_V(a3), o0, o1;
  if (SIGN < 0) asm(B2_DFT4_BODY(B2_AMI, B2_API, B2_AMI, B2_API) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&v"(o0), "=&v"(o1));
  else asm(B2_DFT4_BODY(B2_API, B2_AMI, B2_API, B2_AMI) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&v"(o0), "=&v"(o1));
  a0 = B2_C(o0); a1 = B2_C(o1); a2r = B2_C(x2); a3 = B2_C(x3);
}
// last stage of the 8-point DFT: (e_k, o_k) -> (e_k + o_k, e_k - o_k), the third pair with o times (SIGN i);
// sums land in s0..s3, differences replace the o operands
template <int SIGN> __device__ __forceinline__ void dft8_tail(cf *v, cf b1, cf b3)
{
  b2_pk o0 = B2_V(v[1]), o1 = B2_V(b1), o2 = B2_V(v[5]), o3 = B2_V(b3), s0, s1, s2, s3;
#define B2_DFT8_TAIL(PI, MI)                                                   \
  "v_pk_add_f32 %4, %8, %0\n\tv_pk_add_f32 %0, %8, %0" B2_SUB "\n\t"          \
  "v_pk_add_f32 %5, %9, %1\n\tv_pk_add_f32 %1, %9, %1" B2_SUB "\n\t"          \
  "v_pk_add_f32 %6, %10, %2" PI "\n\tv_pk_add_f32 %2, %10, %2" MI "\n\t"      \
  "v_pk_add_f32 %7, %11, %3\n\tv_pk_add_f32 %3, %11, %3" B2_SUB
  if (SIGN < 0)
    asm(B2_DFT8_TAIL(B2_AMI, B2_API) : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
        : "v"(B2_V(v[0])), "v"(B2_V(v[2])), "v"(B2_V(v[4])), "v"(B2_V(v[6])));
  else
    asm(B2_DFT8_TAIL(B2_API, B2_AMI) : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
        : "v"(B2_V(v[0])), "v"(B2_V(v[2])), "v"(B2_V(v[4])), "v"(B2_V(v[6])));
#undef B2_DFT8_TAIL
  v[0] = B2_C(s0); v[4] = B2_C(o0);
  v[1] = B2_C(s1); v[5] = B2_C(o1);
  v[2] = B2_C(s2); v[6] = B2_C(o2);
  v[3] = B2_C(s3); v[7] = B2_C(o3);
}
#undef B2_DFT4_BODY
#else
template <int SIGN> B2_HD void dft4(cf &a0, cf &a1, cf &a2, cf &a3)
{
  const cf t0 = cadd(a0, a2), t1 = csub(a0, a2);
  const cf t2 = cadd(a1, a3), d = csub(a1, a3);
  a0 = cadd(t0, t2);
  a1 = cadd_i<SIGN>(t1, d); // t1 + (SIGN i)(a1 - a3)
  a2 = csub(t0, t2);
  a3 = csub_i<SIGN>(t1, d);
}
// the same with its third input still to be multiplied by (SIGN i)
template <int SIGN> B2_HD void dft4_rot2(cf &a0, cf &a1, cf &a2r, cf &a3)
{
  const cf t0 = cadd_i<SIGN>(a0, a2r), t1 = csub_i<SIGN>(a0, a2r);
  const cf t2 = cadd(a1, a3), d = csub(a1, a3);
  a0 = cadd(t0, t2);
  a1 = cadd_i<SIGN>(t1, d);
  a2r = csub(t0, t2);
  a3 = csub_i<SIGN>(t1, d);
}
template <int SIGN> B2_HD void dft8_tail(cf *v, cf b1, cf b3)
{
  const cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o2 = v[5];
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, b1); v[5] = csub(e1, b1);
  v[2] = cadd_i<SIGN>(e2, o2); v[6] = csub_i<SIGN>(e2, o2);
  v[3] = cadd(e3, b3); v[7] = csub(e3, b3);
}
#endif

#define B2_SQH 0.70710678118654752440f
#define B2_C16 0.92387953251128675613f
#define B2_S16 0.38268343236508977173f

// 4-point DFTs with trailing zero inputs (the zero-padded reference segments)
template <int SIGN> B2_HD void dft4_z3(cf &a0, cf &a1, cf &a2, cf &a3) // a3 = 0 on entry
{
  const cf t0 = cadd(a0, a2), t1 = csub(a0, a2), d = a1;
  a0 = cadd(t0, d);
  a1 = cadd_i<SIGN>(t1, d);
  a2 = csub(t0, d);
  a3 = csub_i<SIGN>(t1, d);
}
template <int SIGN> B2_HD void dft4_z23(cf &a0, cf &a1, cf &a2, cf &a3) // a2 = a3 = 0 on entry
{
  const cf e = a0, d = a1;
  a0 = cadd(e, d);
  a1 = cadd_i<SIGN>(e, d);
  a2 = csub(e, d);
  a3 = csub_i<SIGN>(e, d);
}
// 8-point DFT, in place, natural order out.
template <int SIGN> B2_HD void dft8(cf *v)
{
  // n = n0 + 2*n1 ; k = k1 + 4*k0
  dft4<SIGN>(v[0], v[2], v[4], v[6]); // n0 = 0 -> k1 at v[0],v[2],v[4],v[6]
  dft4<SIGN>(v[1], v[3], v[5], v[7]); // n0 = 1
  // twiddle W8^(k1) on the odd set; W8^2 = (SIGN i) is folded into the butterfly
  const cf b1 = twid32<SIGN, 4>(v[3]);
  const cf b3 = twid32<SIGN, 12>(v[7]);
  dft8_tail<SIGN>(v, b1, b3);
}

// steps after the first of the 16-point DFT: twiddles, DFT over n0, transpose
template <int SIGN> B2_HD void dft16_tail(cf *v)
{
  // now v[n0 + 4*k1]; twiddle by W16^(n0*k1) (W16^4 = (SIGN i) on v[10] is folded into step B)
  v[1 + 4] = twid32<SIGN, 2>(v[1 + 4]);       // W16^1
  v[2 + 4] = twid32<SIGN, 4>(v[2 + 4]);       // W16^2
  v[3 + 4] = twid32<SIGN, 6>(v[3 + 4]);       // W16^3
  v[1 + 8] = twid32<SIGN, 4>(v[1 + 8]);       // W16^2
  v[3 + 8] = twid32<SIGN, 12>(v[3 + 8]);     // W16^6
  v[1 + 12] = twid32<SIGN, 6>(v[1 + 12]);   // W16^3
  v[2 + 12] = twid32<SIGN, 12>(v[2 + 12]);  // W16^6
  v[3 + 12] = twid32<SIGN, 18>(v[3 + 12]);  // W16^9
  // Step B: DFT over n0 for each k1; result k = k1 + 4*k0 lands at v[4*k1 + k0]
  dft4<SIGN>(v[0], v[1], v[2], v[3]);
  dft4<SIGN>(v[4], v[5], v[6], v[7]);
  dft4_rot2<SIGN>(v[8], v[9], v[10], v[11]);
  dft4<SIGN>(v[12], v[13], v[14], v[15]);
  // transpose the 4x4 so that v[k] is natural: v[4*k1 + k0] -> v[k1 + 4*k0]
  cf t;
  t = v[1]; v[1] = v[4]; v[4] = t;
  t = v[2]; v[2] = v[8]; v[8] = t;
  t = v[3]; v[3] = v[12]; v[12] = t;
  t = v[6]; v[6] = v[9]; v[9] = t;
  t = v[7]; v[7] = v[13]; v[13] = t;
  t = v[11]; v[11] = v[14]; v[14] = t;
}


// 8-point DFT whose inputs v[K..7] are zero (not read), K = 6, 7 or 8
template <int SIGN, int K> B2_HD void dft8_z(cf *v)
{
  static_assert(K >= 6 && K <= 8, "");
  if (K <= 6) dft4_z3<SIGN>(v[0], v[2], v[4], v[6]);
  else dft4<SIGN>(v[0], v[2], v[4], v[6]);
  if (K <= 7) dft4_z3<SIGN>(v[1], v[3], v[5], v[7]);
  else dft4<SIGN>(v[1], v[3], v[5], v[7]);
  const cf b1 = twid32<SIGN, 4>(v[3]);
  const cf b3 = twid32<SIGN, 12>(v[7]);
  dft8_tail<SIGN>(v, b1, b3);
}

// 16-point DFT, in place, natural order out.
template <int SIGN> B2_HD void dft16(cf *v)
{
  // n = n0 + 4*n1 ; k = k1 + 4*k0.  Step A: DFT over n1 for each n0.
  dft4<SIGN>(v[0], v[4], v[8], v[12]);
  dft4<SIGN>(v[1], v[5], v[9], v[13]);
  dft4<SIGN>(v[2], v[6], v[10], v[14]);
  dft4<SIGN>(v[3], v[7], v[11], v[15]);
  dft16_tail<SIGN>(v);
}

// 16-point DFT whose inputs v[9..15] are zero (not read): 18 instead of 32 adds in the first step
template <int SIGN> B2_HD void dft16_nz9(cf *v)
{
  dft4_z3<SIGN>(v[0], v[4], v[8], v[12]);
  dft4_z23<SIGN>(v[1], v[5], v[9], v[13]);
  dft4_z23<SIGN>(v[2], v[6], v[10], v[14]);
  dft4_z23<SIGN>(v[3], v[7], v[11], v[15]);
  dft16_tail<SIGN>(v);
}

template <int R, int SIGN> B2_HD void dftR(cf *v)
{
  if (R == 16) dft16<SIGN>(v);
  else if (R == 8) dft8<SIGN>(v);
  else dft4<SIGN>(v[0], v[1], v[2], v[3]);
}

// One value out of an exchange buffer as ONE ds_read_b64.  Left to the compiler, neighbouring reads are paired into
// ds_read2_b64 / ds_read2st64_b64, which on gfx950 take 8 cycles of the CU's LDS per 64 lanes where two ds_read_b64 take 4.3
// (the doubled read path serves b64 and b128 only; tools/membench/ldsrate.hip).  A wavefront-scope relaxed atomic load is a
// plain ds_read_b64 that the merging passes leave alone.  The workgroup transforms gain 3-4 % from it; the one-wave
// transforms (fft_wave*.hpp) LOSE as much when their pairs are split -- they are bound by instruction issue -- and keep them.
B2_HD cf lds_ld(const cf *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long q = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  return cmake(__uint_as_float((unsigned)q), __uint_as_float((unsigned)(q >> 32)));
#else
  return *p;
#endif
}

template <int R3> struct WgFft {
  static constexpr int T = 16 * R3;   // threads per transform
  static constexpr int F = 256 * R3;  // transform length
  static constexpr int NP = 16 / R3;  // (q,r) pairs per thread in S3 / I
  static constexpr int PA = 17 * R3;
  static constexpr int SU = 17;
  static constexpr int PB = 17 * R3;
  static constexpr int A_ELEMS = 16 * PA;
  static constexpr int B_ELEMS = 16 * PB;
  static constexpr int NTW3 = R3 - 1;

  // tw1[q-1] = W_F^(t*q), q = 1..15 ; tw3[u-1] = W_T^((t%16)*u), u = 1..R3-1
  // `tw` is the table tw[k] = exp(-2*pi*i*k/F), k in [0, F).
  template <class TW> B2_HD static void load_twiddles(int t, const TW *tw, cf *tw1, cf *tw3)
  {
#pragma unroll
    for (int q = 1; q < 16; q++) tw1[q - 1] = tw[(t * q) & (F - 1)];
    const int r = t & 15;
#pragma unroll
    for (int u = 1; u < R3; u++) tw3[u - 1] = tw[(16 * r * u) & (F - 1)];
  }

  template <class TW> B2_HD static void load_tw1(int t, const TW *tw, cf *tw1)
  {
#pragma unroll
    for (int q = 1; q < 16; q++) tw1[q - 1] = tw[(t * q) & (F - 1)];
  }
  template <class TW> B2_HD static void load_tw3(int t, const TW *tw, cf *tw3)
  {
    const int r = t & 15;
#pragma unroll
    for (int u = 1; u < R3; u++) tw3[u - 1] = tw[(16 * r * u) & (F - 1)];
  }

  // ---- forward -----------------------------------------------------------
  // v[k1] = in[t + T*k1] on entry
  B2_HD static void fwd_s1(int t, cf *v, const cf *tw1, cf *A)
  {
    dft16<-1>(v);
#pragma unroll
    for (int q = 1; q < 16; q++) v[q] = cmul(v[q], tw1[q - 1]);
#pragma unroll
    for (int q = 0; q < 16; q++) A[q * PA + t] = v[q];
  }
  // the same for a zero-padded segment whose inputs v[9..15] are zero (and were not loaded)
  B2_HD static void fwd_s1_nz9(int t, cf *v, const cf *tw1, cf *A)
  {
    dft16_nz9<-1>(v);
#pragma unroll
    for (int q = 1; q < 16; q++) v[q] = cmul(v[q], tw1[q - 1]);
#pragma unroll
    for (int q = 0; q < 16; q++) A[q * PA + t] = v[q];
  }
  B2_HD static void fwd_s2_load(int t, cf *v, const cf *A)
  {
    const int q = t / R3, u = t % R3;
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = lds_ld(A + q * PA + u + R3 * k);
  }
  B2_HD static void fwd_s2_store(int t, const cf *v, cf *B)
  {
    const int q = t / R3, u = t % R3;
#pragma unroll
    for (int r = 0; r < 16; r++) B[q * PB + u * SU + r] = v[r];
  }
  B2_HD static void fwd_s2(int t, cf *v, const cf *A, cf *B)
  {
    fwd_s2_load(t, v, A);
    dft16<-1>(v);
    fwd_s2_store(t, v, B);
  }
  // leaves X[q + 16*r + 256*s] in v[j*R3 + s], (16*q + r) = t + T*j
  B2_HD static void fwd_s3(int t, cf *v, const cf *tw3, const cf *B)
  {
    const int r = t & 15;
#pragma unroll
    for (int j = 0; j < NP; j++) {
      const int q = (t >> 4) + R3 * j;
      cf *w = v + j * R3;
#pragma unroll
      for (int u = 0; u < R3; u++) w[u] = lds_ld(B + q * PB + u * SU + r);
#pragma unroll
      for (int u = 1; u < R3; u++) w[u] = cmul(w[u], tw3[u - 1]);
      dftR<R3, -1>(w);
    }
  }

  // ---- inverse (unnormalised: returns F * ifft) ----------------------------
  B2_HD static void inv_s1(int t, cf *v, const cf *tw3, cf *B)
  {
    const int r = t & 15;
#pragma unroll
    for (int j = 0; j < NP; j++) {
      const int q = (t >> 4) + R3 * j;
      cf *w = v + j * R3;
      dftR<R3, +1>(w);
#pragma unroll
      for (int a = 1; a < R3; a++) w[a] = cmulc(w[a], tw3[a - 1]);
#pragma unroll
      for (int a = 0; a < R3; a++) B[q * PB + a * SU + r] = w[a];
    }
  }
  B2_HD static void inv_s2_load(int t, cf *v, const cf *B)
  {
    const int q = t / R3, a = t % R3;
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds_ld(B + q * PB + a * SU + r);
  }
  B2_HD static void inv_s2_store(int t, const cf *v, cf *A)
  {
    const int q = t / R3, a = t % R3;
#pragma unroll
    for (int b = 0; b < 16; b++) A[q * PA + a + R3 * b] = v[b];
  }
  B2_HD static void inv_s2(int t, cf *v, const cf *B, cf *A)
  {
    inv_s2_load(t, v, B);
    dft16<+1>(v);
    inv_s2_store(t, v, A);
  }
  // leaves z[t + T*c] in v[c]
  B2_HD static void inv_s3(int t, cf *v, const cf *tw1, const cf *A)
  {
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = lds_ld(A + q * PA + t);
#pragma unroll
    for (int q = 1; q < 16; q++) v[q] = cmulc(v[q], tw1[q - 1]);
    dft16<+1>(v);
  }
};

} // namespace blah2
