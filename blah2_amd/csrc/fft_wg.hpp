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

// Complex fp32 as a plain {re, im} pair with scalar arithmetic.  The library is
// built with -fno-slp-vectorize: v_pk_*_f32 has the same lanes-per-cycle rate as
// the scalar v_*_f32 forms on gfx950 (157 TF either way), and letting the SLP
// vectoriser (or a float2 ext-vector type, which was tried) form packed ops
// costs one v_mov per repacked operand (17-30 % of the VALU stream) and 64-bit
// register-tuple constraints that pushed the range kernel into scratch spills.
struct alignas(8) cf {
  float x, y;
};

B2_HD cf cmake(float x, float y) { cf r; r.x = x; r.y = y; return r; }
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
// multiply by -i (SIGN < 0) or +i (SIGN > 0)
template <int SIGN> B2_HD cf mul_i(cf a)
{
  return SIGN < 0 ? cmake(a.y, -a.x) : cmake(-a.y, a.x);
}
// a * w for SIGN < 0, a * conj(w) for SIGN > 0 (w is always the forward root)
template <int SIGN> B2_HD cf twid(cf a, cf w) { return SIGN < 0 ? cmul(a, w) : cmulc(a, w); }

// 4-point DFT, in place, natural order out.  SIGN = -1 forward, +1 inverse.
template <int SIGN> B2_HD void dft4(cf &a0, cf &a1, cf &a2, cf &a3)
{
  const cf t0 = cadd(a0, a2), t1 = csub(a0, a2);
  const cf t2 = cadd(a1, a3), t3 = mul_i<SIGN>(csub(a1, a3));
  a0 = cadd(t0, t2);
  a1 = cadd(t1, t3);
  a2 = csub(t0, t2);
  a3 = csub(t1, t3);
}

#define B2_SQH 0.70710678118654752440f
#define B2_C16 0.92387953251128675613f
#define B2_S16 0.38268343236508977173f

// 8-point DFT, in place, natural order out.
template <int SIGN> B2_HD void dft8(cf *v)
{
  // n = n0 + 2*n1 ; k = k1 + 4*k0
  dft4<SIGN>(v[0], v[2], v[4], v[6]); // n0 = 0 -> k1 at v[0],v[2],v[4],v[6]
  dft4<SIGN>(v[1], v[3], v[5], v[7]); // n0 = 1
  // twiddle W8^(k1) on the odd set
  const cf w1 = cmake(B2_SQH, -B2_SQH);
  const cf w3 = cmake(-B2_SQH, -B2_SQH);
  const cf b1 = twid<SIGN>(v[3], w1);
  const cf b2 = mul_i<SIGN>(v[5]);
  const cf b3 = twid<SIGN>(v[7], w3);
  const cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, b1); v[5] = csub(e1, b1);
  v[2] = cadd(e2, b2); v[6] = csub(e2, b2);
  v[3] = cadd(e3, b3); v[7] = csub(e3, b3);
}

// 16-point DFT, in place, natural order out.
template <int SIGN> B2_HD void dft16(cf *v)
{
  // n = n0 + 4*n1 ; k = k1 + 4*k0.  Step A: DFT over n1 for each n0.
  dft4<SIGN>(v[0], v[4], v[8], v[12]);
  dft4<SIGN>(v[1], v[5], v[9], v[13]);
  dft4<SIGN>(v[2], v[6], v[10], v[14]);
  dft4<SIGN>(v[3], v[7], v[11], v[15]);
  // now v[n0 + 4*k1]; twiddle by W16^(n0*k1)
  const cf w1 = cmake(B2_C16, -B2_S16);  // W16^1
  const cf w2 = cmake(B2_SQH, -B2_SQH);  // W16^2
  const cf w3 = cmake(B2_S16, -B2_C16);  // W16^3
  const cf w6 = cmake(-B2_SQH, -B2_SQH); // W16^6
  const cf w9 = cmake(-B2_C16, B2_S16);  // W16^9
  v[1 + 4] = twid<SIGN>(v[1 + 4], w1);
  v[2 + 4] = twid<SIGN>(v[2 + 4], w2);
  v[3 + 4] = twid<SIGN>(v[3 + 4], w3);
  v[1 + 8] = twid<SIGN>(v[1 + 8], w2);
  v[2 + 8] = mul_i<SIGN>(v[2 + 8]); // W16^4
  v[3 + 8] = twid<SIGN>(v[3 + 8], w6);
  v[1 + 12] = twid<SIGN>(v[1 + 12], w3);
  v[2 + 12] = twid<SIGN>(v[2 + 12], w6);
  v[3 + 12] = twid<SIGN>(v[3 + 12], w9);
  // Step B: DFT over n0 for each k1; result k = k1 + 4*k0 lands at v[4*k1 + k0]
  dft4<SIGN>(v[0], v[1], v[2], v[3]);
  dft4<SIGN>(v[4], v[5], v[6], v[7]);
  dft4<SIGN>(v[8], v[9], v[10], v[11]);
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

template <int R, int SIGN> B2_HD void dftR(cf *v)
{
  if (R == 16) dft16<SIGN>(v);
  else if (R == 8) dft8<SIGN>(v);
  else dft4<SIGN>(v[0], v[1], v[2], v[3]);
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
  B2_HD static void fwd_s2_load(int t, cf *v, const cf *A)
  {
    const int q = t / R3, u = t % R3;
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = A[q * PA + u + R3 * k];
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
      for (int u = 0; u < R3; u++) w[u] = B[q * PB + u * SU + r];
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
    for (int r = 0; r < 16; r++) v[r] = B[q * PB + a * SU + r];
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
    for (int q = 0; q < 16; q++) v[q] = A[q * PA + t];
#pragma unroll
    for (int q = 1; q < 16; q++) v[q] = cmulc(v[q], tw1[q - 1]);
    dft16<+1>(v);
  }
};

} // namespace blah2
