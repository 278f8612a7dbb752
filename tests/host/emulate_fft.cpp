// Host emulation of the workgroup FFT and of the segmented range correlation.
// Compiles blah2_amd/csrc/fft_wg.hpp + range_core.hpp for the CPU and runs the
// threads of one workgroup one after another between barriers, so the index
// algebra, twiddle placement, LDS layouts and segment masks can be checked
// without a GPU.  Used by tests/test_host_emulation.py (not gpu).
//
//   emulate_fft fft                       -> prints max rel errors of fwd/inv for R3=4,8,16
//   emulate_fft range R3 nCorr nD dMin dMax nSeg segLen seed
//                                          -> prints max rel error vs the direct definition
#include "../../blah2_amd/csrc/range_core.hpp"
#include "../../blah2_amd/csrc/fft_wg8.hpp"
#include "../../blah2_amd/csrc/fft_wave.hpp"
#include "../../blah2_amd/csrc/fft_wave2.hpp"
#include "../../blah2_amd/csrc/fft_wave1k.hpp"

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace blah2;
using cd = std::complex<double>;

template <int R3> std::vector<cf> make_table()
{
  constexpr int F = 256 * R3;
  std::vector<cf> tw(F);
  for (int k = 0; k < F; k++) {
    double a = -2.0 * M_PI * k / F;
    tw[k] = cmake((float)std::cos(a), (float)std::sin(a));
  }
  return tw;
}

template <int R3> struct Wg {
  using W = WgFft<R3>;
  static constexpr int T = W::T;
  std::vector<cf> A, B, tw;
  std::vector<cf> tw1, tw3; // per thread
  Wg() : A(W::A_ELEMS), B(W::B_ELEMS), tw(make_table<R3>()), tw1(T * 15), tw3(T * 16)
  {
    for (int t = 0; t < T; t++) W::load_twiddles(t, tw.data(), &tw1[t * 15], &tw3[t * 16]);
  }
  // v: T x 16 registers
  void forward(std::vector<cf> &v)
  {
    for (int t = 0; t < T; t++) W::fwd_s1(t, &v[t * 16], &tw1[t * 15], A.data());
    for (int t = 0; t < T; t++) W::fwd_s2(t, &v[t * 16], A.data(), B.data());
    for (int t = 0; t < T; t++) W::fwd_s3(t, &v[t * 16], &tw3[t * 16], B.data());
  }
  void inverse(std::vector<cf> &v)
  {
    for (int t = 0; t < T; t++) W::inv_s1(t, &v[t * 16], &tw3[t * 16], B.data());
    for (int t = 0; t < T; t++) W::inv_s2(t, &v[t * 16], B.data(), A.data());
    for (int t = 0; t < T; t++) W::inv_s3(t, &v[t * 16], &tw1[t * 15], A.data());
  }
};

template <int R3> int test_fft()
{
  using W = WgFft<R3>;
  constexpr int T = W::T, F = W::F;
  std::mt19937 gen(1234 + R3);
  std::uniform_real_distribution<float> dist(-1.f, 1.f);
  std::vector<cf> in(F);
  for (auto &c : in) c = cmake(dist(gen), dist(gen));
  std::vector<cf> v(T * 16);
  for (int t = 0; t < T; t++)
    for (int k = 0; k < 16; k++) v[t * 16 + k] = in[t + T * k];
  Wg<R3> wg;
  wg.forward(v);
  // direct DFT in double
  std::vector<cd> X(F);
  for (int m = 0; m < F; m++) {
    cd acc = 0;
    for (int n = 0; n < F; n++) {
      double a = -2.0 * M_PI * (double)(((long)m * n) % F) / F;
      acc += cd(in[n].x, in[n].y) * cd(std::cos(a), std::sin(a));
    }
    X[m] = acc;
  }
  double peak = 0, err = 0;
  for (auto &c : X) peak = std::max(peak, std::abs(c));
  for (int t = 0; t < T; t++)
    for (int j = 0; j < W::NP; j++)
      for (int s = 0; s < R3; s++) {
        const int p = t + T * j, q = p / 16, r = p % 16;
        const int m = q + 16 * r + 256 * s;
        const cf g = v[t * 16 + j * R3 + s];
        err = std::max(err, std::abs(cd(g.x, g.y) - X[m]));
      }
  wg.inverse(v);
  double ierr = 0;
  for (int t = 0; t < T; t++)
    for (int c = 0; c < 16; c++) {
      const cf g = v[t * 16 + c];
      const cf e = in[t + T * c];
      ierr = std::max(ierr, (double)std::abs(cd(g.x / F - e.x, g.y / F - e.y)));
    }
  std::printf("R3=%d F=%d fwd_rel_err=%.3e inv_abs_err=%.3e\n", R3, F, err / peak, ierr);
  return (err / peak < 2e-6 && ierr < 2e-6) ? 0 : 1;
}

// 8-points-per-thread, four-stage transform (fft_wg8.hpp)
template <int R4> int test_fft8()
{
  using W = WgFft8<R4>;
  constexpr int T = W::T, F = W::F;
  std::mt19937 gen(4321 + R4);
  std::uniform_real_distribution<float> dist(-1.f, 1.f);
  std::vector<cf> in(F), tw(F);
  for (auto &c : in) c = cmake(dist(gen), dist(gen));
  for (int k = 0; k < F; k++) { double a = -2.0 * M_PI * k / F; tw[k] = cmake((float)std::cos(a), (float)std::sin(a)); }
  std::vector<cf> v(T * 8), tw1(T * 7), tw2(T * 7), tw3(T * 7), A(W::BUF_ELEMS), B(W::BUF_ELEMS);
  for (int t = 0; t < T; t++) {
    W::load_twiddles(t, tw.data(), &tw1[t * 7], &tw2[t * 7], &tw3[t * 7]);
    for (int k = 0; k < 8; k++) v[t * 8 + k] = in[t + T * k];
  }
  for (int t = 0; t < T; t++) W::fwd_s1(t, &v[t * 8], &tw1[t * 7], A.data());
  for (int t = 0; t < T; t++) { W::fwd_s2_load(t, &v[t * 8], A.data()); W::fwd_s2_store(t, &v[t * 8], &tw2[t * 7], B.data()); }
  for (int t = 0; t < T; t++) { W::fwd_s3_load(t, &v[t * 8], B.data()); W::fwd_s3_store(t, &v[t * 8], &tw3[t * 7], A.data()); }
  for (int t = 0; t < T; t++) W::fwd_s4(t, &v[t * 8], A.data());
  std::vector<cd> X(F);
  for (int m = 0; m < F; m++) {
    cd acc = 0;
    for (int n = 0; n < F; n++) {
      double a = -2.0 * M_PI * (double)(((long)m * n) % F) / F;
      acc += cd(in[n].x, in[n].y) * cd(std::cos(a), std::sin(a));
    }
    X[m] = acc;
  }
  double peak = 0, err = 0;
  for (auto &c : X) peak = std::max(peak, std::abs(c));
  for (int t = 0; t < T; t++)
    for (int j = 0; j < W::NP; j++)
      for (int q4 = 0; q4 < R4; q4++) {
        const int rho = t + T * j, q12 = rho % 64, q3 = rho / 64;
        const int m = (q12 / 8) + 8 * (q12 % 8) + 64 * q3 + 512 * q4;
        const cf g = v[t * 8 + j * R4 + q4];
        err = std::max(err, std::abs(cd(g.x, g.y) - X[m]));
      }
  for (int t = 0; t < T; t++) W::inv_s4(t, &v[t * 8], A.data());
  for (int t = 0; t < T; t++) { W::inv_s3_load(t, &v[t * 8], &tw3[t * 7], A.data()); }
  for (int t = 0; t < T; t++) { W::inv_s3_store(t, &v[t * 8], B.data()); }
  for (int t = 0; t < T; t++) { W::inv_s2_load(t, &v[t * 8], &tw2[t * 7], B.data()); }
  for (int t = 0; t < T; t++) { W::inv_s2_store(t, &v[t * 8], A.data()); }
  for (int t = 0; t < T; t++) W::inv_s1(t, &v[t * 8], &tw1[t * 7], A.data());
  double ierr = 0;
  for (int t = 0; t < T; t++)
    for (int c = 0; c < 8; c++) {
      const cf g = v[t * 8 + c];
      const cf e = in[t + T * c];
      ierr = std::max(ierr, (double)std::abs(cd(g.x / F - e.x, g.y / F - e.y)));
    }
  std::printf("E8 R4=%d F=%d fwd_rel_err=%.3e inv_abs_err=%.3e\n", R4, F, err / peak, ierr);
  return (err / peak < 2e-6 && ierr < 2e-6) ? 0 : 1;
}

// one-wave transform (fft_wave.hpp): 64 lanes x 32 points, the lane exchange emulated on lane pairs
template <int SIGN> void wave_transform(std::vector<cf> &v, const std::vector<cf> &tw, std::vector<cf> &X)
{
  using W = WaveFft;
  std::vector<W::Tw> w(64);
  std::vector<cf> table(W::TW_ELEMS);
  for (int t = 0; t < 64; t++) W::fill_table(t, 64, tw.data(), table.data());
  for (int t = 0; t < 64; t++) W::load_twiddles(t, tw.data(), table.data(), w[t]);
  for (int t = 0; t < 64; t++) W::s1<SIGN>(&v[t * 32], w[t]);
  for (int t1 = 0; t1 < 32; t1++) W::sw_host(&v[t1 * 32], &v[(t1 + 32) * 32]);
  for (int t = 0; t < 64; t++) W::s2<SIGN>(t, &v[t * 32], w[t], X.data());
  for (int t = 0; t < 64; t++) W::s3<SIGN>(t, &v[t * 32], X.data());
}

int test_fft_wave()
{
  using W = WaveFft;
  constexpr int F = W::F;
  std::mt19937 gen(777);
  std::uniform_real_distribution<float> dist(-1.f, 1.f);
  std::vector<cf> in(F), tw(F), X(W::X_ELEMS);
  for (auto &c : in) c = cmake(dist(gen), dist(gen));
  for (int k = 0; k < F; k++) { double a = -2.0 * M_PI * k / F; tw[k] = cmake((float)std::cos(a), (float)std::sin(a)); }
  // the 32-point kernel on its own
  double e32 = 0;
  for (int sign = -1; sign <= 1; sign += 2) {
    cf v[32];
    for (int k = 0; k < 32; k++) v[k] = in[k];
    if (sign < 0) dft32<-1>(v); else dft32<+1>(v);
    for (int m = 0; m < 32; m++) {
      cd acc = 0;
      for (int n = 0; n < 32; n++) acc += cd(in[n].x, in[n].y) * std::polar(1.0, sign * 2.0 * M_PI * ((m * n) % 32) / 32.0);
      e32 = std::max(e32, std::abs(cd(v[m].x, v[m].y) - acc));
    }
  }
  std::vector<cf> v(F);
  for (int t = 0; t < 64; t++)
    for (int k = 0; k < 32; k++) v[t * 32 + k] = in[t + 64 * k];
  wave_transform<-1>(v, tw, X);
  double peak = 0, err = 0;
  for (int m = 0; m < F; m++) {
    cd acc = 0;
    for (int n = 0; n < F; n++) {
      double a = -2.0 * M_PI * (double)(((long)m * n) % F) / F;
      acc += cd(in[n].x, in[n].y) * cd(std::cos(a), std::sin(a));
    }
    peak = std::max(peak, std::abs(acc));
    const cf g = v[(m % 64) * 32 + m / 64];
    err = std::max(err, std::abs(cd(g.x, g.y) - acc));
  }
  wave_transform<+1>(v, tw, X);
  double ierr = 0;
  for (int t = 0; t < 64; t++)
    for (int c = 0; c < 32; c++) {
      const cf g = v[t * 32 + c], e = in[t + 64 * c];
      ierr = std::max(ierr, (double)std::abs(cd(g.x / F - e.x, g.y / F - e.y)));
    }
  std::printf("WAVE F=%d dft32_abs_err=%.3e fwd_rel_err=%.3e inv_abs_err=%.3e\n", F, e32, err / peak, ierr);
  return (e32 < 2e-5 && err / peak < 2e-6 && ierr < 2e-6) ? 0 : 1;
}

// two-wave 4096-point transform (fft_wave2.hpp): 2 x 64 lanes x 32 points; both lane exchanges emulated on lane pairs
template <int SIGN, int NZ = 32> void wave2_transform(std::vector<cf> &v, const std::vector<cf> &tw, std::vector<cf> &X)
{
  using W = Wave2Fft;
  std::vector<W::Tw> w(128);
  std::vector<cf> table(W::TW_ELEMS);
  for (int t = 0; t < 128; t++) W::fill_table(t, 128, tw.data(), table.data());
  for (int t = 0; t < 128; t++) W::load_twiddles(t >> 6, t & 63, tw.data(), table.data(), w[t]);
  for (int t = 0; t < 128; t++) W::s1<SIGN, NZ>(&v[t * 32], w[t]);
  for (int wv = 0; wv < 2; wv++)
    for (int l = 0; l < 32; l++) WaveFft::sw_host(&v[(wv * 64 + l) * 32], &v[(wv * 64 + l + 32) * 32]);
  for (int t = 0; t < 128; t++) W::b1<SIGN>(&v[t * 32], w[t]);
  for (int wv = 0; wv < 2; wv++)
    for (int l = 0; l < 64; l++)
      if (!(l & 16)) WaveFft::sw_host(&v[(wv * 64 + l) * 32], &v[(wv * 64 + l + 16) * 32]); // v_permlane16_swap: same pattern on rows
  for (int t = 0; t < 128; t++) W::b2<SIGN>(t >> 6, t & 63, &v[t * 32], w[t], X.data());
  for (int t = 0; t < 128; t++) W::s3<SIGN>(t >> 6, t & 63, &v[t * 32], X.data());
}

int test_fft_wave2()
{
  using W = Wave2Fft;
  constexpr int F = W::F;
  std::mt19937 gen(4242);
  std::uniform_real_distribution<float> dist(-1.f, 1.f);
  std::vector<cf> in(F), tw(F), X(W::X_ELEMS);
  for (auto &c : in) c = cmake(dist(gen), dist(gen));
  for (int k = 0; k < F; k++) { double a = -2.0 * M_PI * k / F; tw[k] = cmake((float)std::cos(a), (float)std::sin(a)); }
  auto T_of = [](int t) { return W::logical(t >> 6, t & 63); };
  double worst = 0, worst_inv = 0, worst_nz = 0;
  for (int nz : {32, 16}) {
    std::vector<cf> src = in;
    if (nz == 16) for (int n = 2048; n < F; n++) src[n] = cmake(0.f, 0.f);
    std::vector<cf> v(F);
    for (int t = 0; t < 128; t++)
      for (int k = 0; k < 32; k++) v[t * 32 + k] = (nz == 16 && k >= 16) ? cmake(77.f, -55.f) /* never read */ : src[T_of(t) + 128 * k];
    if (nz == 16) wave2_transform<-1, 16>(v, tw, X); else wave2_transform<-1>(v, tw, X);
    // reference spectrum by fp64 FFT-free evaluation on a subset of bins would be slow: use the full O(N^2) sum in fp64
    std::vector<cd> ref(F);
    double peak = 0;
    for (int m = 0; m < F; m++) {
      cd acc = 0;
      for (int n = 0; n < (nz == 16 ? 2048 : F); n++) {
        const double a = -2.0 * M_PI * (double)(((long)m * n) % F) / F;
        acc += cd(src[n].x, src[n].y) * cd(std::cos(a), std::sin(a));
      }
      ref[m] = acc;
      peak = std::max(peak, std::abs(acc));
    }
    double err = 0;
    for (int t = 0; t < 128; t++)
      for (int a = 0; a < 32; a++) {
        const cf g = v[t * 32 + a];
        err = std::max(err, std::abs(cd(g.x, g.y) - ref[T_of(t) + 128 * a]));
      }
    if (nz == 32) {
      worst = err / peak;
      wave2_transform<+1>(v, tw, X);
      for (int t = 0; t < 128; t++)
        for (int c = 0; c < 32; c++) {
          const cf g = v[t * 32 + c], e = in[T_of(t) + 128 * c];
          worst_inv = std::max(worst_inv, (double)std::abs(cd(g.x / F - e.x, g.y / F - e.y)));
        }
    } else {
      worst_nz = err / peak;
    }
  }
  std::printf("WAVE2 F=%d fwd_rel_err=%.3e inv_abs_err=%.3e nz16_rel_err=%.3e\n", F, worst, worst_inv, worst_nz);
  return (worst < 2e-6 && worst_inv < 2e-6 && worst_nz < 2e-6) ? 0 : 1;
}

// one-wave 1024-point transform (fft_wave1k.hpp): 64 lanes x 16 points; both lane exchanges emulated on lane pairs
template <int SIGN, int NZ = 16, bool OUT7 = false> void wave1k_transform(std::vector<cf> &v, const std::vector<cf> &tw, std::vector<cf> &X)
{
  using W = Wave1kFft;
  std::vector<W::Tw> w(64);
  std::vector<cf> table(W::TW_ELEMS);
  for (int t = 0; t < 64; t++) W::fill_table(t, 64, tw.data(), table.data());
  for (int t = 0; t < 64; t++) W::load_twiddles(t, tw.data(), table.data(), w[t]);
  for (int t = 0; t < 64; t++) W::s1<SIGN, NZ>(&v[t * 16], w[t]);
  for (int l = 0; l < 32; l++) W::sw_host(&v[l * 16], &v[(l + 32) * 16]);
  for (int t = 0; t < 64; t++) W::b1<SIGN>(&v[t * 16], w[t]);
  for (int l = 0; l < 64; l++)
    if (!(l & 16)) W::sw_host(&v[l * 16], &v[(l + 16) * 16]);
  for (int t = 0; t < 64; t++) W::b2<SIGN>(t, &v[t * 16], w[t], X.data());
  for (int t = 0; t < 64; t++) W::s3<SIGN, OUT7>(t, &v[t * 16], X.data());
}

int test_fft_wave1k()
{
  using W = Wave1kFft;
  constexpr int F = W::F;
  std::mt19937 gen(1024);
  std::uniform_real_distribution<float> dist(-1.f, 1.f);
  std::vector<cf> in(F), tw(F), X(W::X_ELEMS);
  for (auto &c : in) c = cmake(dist(gen), dist(gen));
  for (int k = 0; k < F; k++) { double a = -2.0 * M_PI * k / F; tw[k] = cmake((float)std::cos(a), (float)std::sin(a)); }
  double worst = 0, worst_inv = 0, worst_nz = 0, worst_o7 = 0;
  for (int nz : {16, 9}) {
    std::vector<cf> src = in;
    if (nz == 9) for (int n = 9 * 64; n < F; n++) src[n] = cmake(0.f, 0.f);
    std::vector<cf> v(F);
    for (int t = 0; t < 64; t++)
      for (int k = 0; k < 16; k++) v[t * 16 + k] = (k >= nz) ? cmake(77.f, -55.f) /* never read */ : src[t + 64 * k];
    if (nz == 9) wave1k_transform<-1, 9>(v, tw, X); else wave1k_transform<-1>(v, tw, X);
    std::vector<cd> ref(F);
    double peak = 0;
    for (int m = 0; m < F; m++) {
      cd acc = 0;
      for (int n = 0; n < F; n++) {
        const double a = -2.0 * M_PI * (double)(((long)m * n) % F) / F;
        acc += cd(src[n].x, src[n].y) * cd(std::cos(a), std::sin(a));
      }
      ref[m] = acc;
      peak = std::max(peak, std::abs(acc));
    }
    double err = 0;
    for (int t = 0; t < 64; t++)
      for (int a = 0; a < 16; a++) {
        const cf g = v[t * 16 + a];
        err = std::max(err, std::abs(cd(g.x, g.y) - ref[t + 64 * a]));
      }
    if (nz == 16) {
      worst = err / peak;
      std::vector<cf> v7 = v;
      wave1k_transform<+1>(v, tw, X);
      wave1k_transform<+1, 16, true>(v7, tw, X);
      for (int t = 0; t < 64; t++)
        for (int c = 0; c < 16; c++) {
          const cf g = v[t * 16 + c], e = in[t + 64 * c];
          worst_inv = std::max(worst_inv, (double)std::abs(cd(g.x / F - e.x, g.y / F - e.y)));
          if (c < 7) worst_o7 = std::max(worst_o7, (double)std::abs(cd(v7[t * 16 + c].x / F - e.x, v7[t * 16 + c].y / F - e.y)));
        }
    } else {
      worst_nz = err / peak;
    }
  }
  std::printf("WAVE1K F=%d fwd_rel_err=%.3e inv_abs_err=%.3e nz9_rel_err=%.3e out7_abs_err=%.3e\n", F, worst, worst_inv, worst_nz, worst_o7);
  return (worst < 2e-6 && worst_inv < 2e-6 && worst_nz < 2e-6 && worst_o7 < 2e-6) ? 0 : 1;
}

// the pruned 16-point DFT of the zero-padded reference segments against the full one
int test_dft16_nz9()
{
  std::mt19937 gen(99);
  std::uniform_real_distribution<float> dist(-1.f, 1.f);
  double err = 0;
  for (int rep = 0; rep < 4; rep++) {
    cf a[16], b[16];
    for (int k = 0; k < 16; k++) a[k] = b[k] = k < 9 ? cmake(dist(gen), dist(gen)) : cmake(0.f, 0.f);
    for (int k = 9; k < 16; k++) b[k] = cmake(123.f, -7.f); // never read
    if (rep & 1) { dft16<+1>(a); dft16_nz9<+1>(b); } else { dft16<-1>(a); dft16_nz9<-1>(b); }
    for (int k = 0; k < 16; k++) err = std::max(err, (double)std::hypot(a[k].x - b[k].x, a[k].y - b[k].y));
  }
  // dft32 with trailing zero inputs / with only the first seven outputs
  for (int rep = 0; rep < 6; rep++) {
    const int nz = rep < 2 ? 24 : (rep < 4 ? 28 : 32);
    cf a[32], b[32];
    for (int k = 0; k < 32; k++) a[k] = b[k] = k < nz ? cmake(dist(gen), dist(gen)) : cmake(0.f, 0.f);
    for (int k = nz; k < 32; k++) b[k] = cmake(55.f, 66.f); // never read
    dft32<-1>(a);
    if (nz == 24) dft32<-1, 24>(b); else if (nz == 28) dft32<-1, 28>(b); else dft32_out7<-1>(b);
    for (int k = 0; k < (nz == 32 ? 7 : 32); k++) err = std::max(err, (double)std::hypot(a[k].x - b[k].x, a[k].y - b[k].y));
  }
  std::printf("NZ9 dft16 / dft32 pruned_vs_full_abs_err=%.3e\n", err);
  return err < 1e-6 ? 0 : 1;
}

template <int R3>
int test_range(int nCorr, int nD, int dMin, int dMax, int nSeg, int segLen, unsigned seed)
{
  using W = WgFft<R3>;
  constexpr int T = W::T, F = W::F;
  RangePlan p;
  p.nCorr = nCorr; p.nDoppler = nD; p.nDelay = dMax - dMin + 1; p.delayMin = dMin;
  p.nSeg = nSeg; p.segLen = segLen; p.scale = 1.0f / F;
  p.colOff = 0; p.nTilesOut = (p.nDelay + 15) >> 4;
  if (segLen + p.nDelay - 1 > F || nSeg * segLen < nCorr) { std::printf("bad plan\n"); return 2; }
  const long n = (long)nCorr * nD;
  std::mt19937 gen(seed);
  std::normal_distribution<float> dist(0.f, 300.f);
  std::vector<cf> x(n), y(n);
  for (long i = 0; i < n; i++) { x[i] = cmake(std::round(dist(gen)), std::round(dist(gen))); }
  for (long i = 0; i < n; i++) {
    y[i] = cmake(std::round(0.5f * x[i].x + dist(gen) * 0.1f), std::round(0.5f * x[i].y + dist(gen) * 0.1f));
  }
  InC32 in{x.data(), y.data()};
  const int nTiles = (p.nDelay + 15) / 16;
  std::vector<cf> out((size_t)nTiles * nD * 16);
  Wg<R3> wg;
  std::vector<cf> v(T * 16), yv(T * 16), acc(T * 16);
  for (int pulse = 0; pulse < nD; pulse++) {
    const int64_t base = (int64_t)pulse * nCorr;
    std::fill(acc.begin(), acc.end(), cmake(0, 0));
    for (int s = 0; s < nSeg; s++) {
      // same barrier structure as range_kernel: x through P, y through Q
      std::vector<cf> &P = wg.A, &Q = wg.B;
      for (int t = 0; t < T; t++) {
        load_seg_x<R3>(in, p, base, s, t, &v[t * 16]);
        load_seg_y<R3>(in, p, base, s, t, &yv[t * 16]);
        mask_seg_x<R3>(p, s, t, &v[t * 16]);
        mask_seg_y<R3>(p, s, t, &yv[t * 16]);
        W::fwd_s1(t, &v[t * 16], &wg.tw1[t * 15], P.data());
        W::fwd_s1(t, &yv[t * 16], &wg.tw1[t * 15], Q.data());
      }
      for (int t = 0; t < T; t++) {
        W::fwd_s2_load(t, &v[t * 16], P.data());
        W::fwd_s2_load(t, &yv[t * 16], Q.data());
        dft16<-1>(&v[t * 16]);
        dft16<-1>(&yv[t * 16]);
      }
      for (int t = 0; t < T; t++) {
        W::fwd_s2_store(t, &v[t * 16], P.data());
        W::fwd_s2_store(t, &yv[t * 16], Q.data());
      }
      for (int t = 0; t < T; t++) {
        W::fwd_s3(t, &v[t * 16], &wg.tw3[t * 16], P.data());
        W::fwd_s3(t, &yv[t * 16], &wg.tw3[t * 16], Q.data());
        for (int e = 0; e < 16; e++) acc[t * 16 + e] = cmacc(acc[t * 16 + e], yv[t * 16 + e], v[t * 16 + e]);
      }
    }
    v = acc;
    wg.inverse(v);
    for (int t = 0; t < T; t++) store_lags<R3>(out.data(), p, 0, pulse, t, &v[t * 16]);
  }
  // direct definition in double
  double peak = 0, err = 0;
  for (int pulse = 0; pulse < nD; pulse++)
    for (int j = 0; j < p.nDelay; j++) {
      const int d = dMin + j;
      cd accd = 0;
      for (int k = 0; k < nCorr; k++) {
        const int ky = k + d;
        if (ky < 0 || ky >= nCorr) continue;
        const cf a = y[(long)pulse * nCorr + ky], b = x[(long)pulse * nCorr + k];
        accd += cd(a.x, a.y) * std::conj(cd(b.x, b.y));
      }
      const cf g = out[rmap_index(nD, nTiles, 0, pulse, j)];
      peak = std::max(peak, std::abs(accd));
      err = std::max(err, std::abs(cd(g.x, g.y) - accd));
    }
  std::printf("range R3=%d nCorr=%d nD=%d lags=[%d,%d] nSeg=%d segLen=%d rel_err=%.3e\n", R3, nCorr,
              nD, dMin, dMax, nSeg, segLen, err / peak);
  return err / peak < 1e-5 ? 0 : 1;
}

int main(int argc, char **argv)
{
  if (argc >= 2 && !std::strcmp(argv[1], "fft"))
    return test_fft<4>() | test_fft<8>() | test_fft<16>() | test_fft8<2>() | test_fft8<4>() | test_fft8<8>() | test_fft_wave() | test_fft_wave2() | test_fft_wave1k() | test_dft16_nz9();
  if (argc >= 10 && !std::strcmp(argv[1], "range")) {
    const int R3 = std::atoi(argv[2]);
    const int a[7] = {std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]), std::atoi(argv[6]),
                      std::atoi(argv[7]), std::atoi(argv[8]), std::atoi(argv[9])};
    if (R3 == 4) return test_range<4>(a[0], a[1], a[2], a[3], a[4], a[5], a[6]);
    if (R3 == 8) return test_range<8>(a[0], a[1], a[2], a[3], a[4], a[5], a[6]);
    if (R3 == 16) return test_range<16>(a[0], a[1], a[2], a[3], a[4], a[5], a[6]);
  }
  std::fprintf(stderr, "usage: emulate_fft fft | range R3 nCorr nD dMin dMax nSeg segLen seed\n");
  return 2;
}
