// TEST INFRASTRUCTURE ONLY -- part of oracle/ (see oracle/README.md).
//
// fp64 complex DFT behind the fftw3.h shim.  Mixed-radix Stockham autosort
// (radices 4,2,3,5 specialised, 7/11/13 generic) and Bluestein's chirp-z for
// lengths with a prime factor above 13 (e.g. WienerHopf's nSamples+nBins+1,
// /root/reference/src/process/clutter/WienerHopf.cpp:39-44).
//
// Twiddles come from one master table W[k] = exp(-2*pi*i*k/N) evaluated with
// sincos on the exactly reduced angle, so every factor used in a pass is
// correctly rounded to < 1 ulp; error growth is the usual O(eps*log N).

#include "fftw3.h"

#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

namespace {

using cd = std::complex<double>;

std::vector<int> factorize(int n)
{
  std::vector<int> f;
  while (n % 4 == 0) { f.push_back(4); n /= 4; }
  while (n % 2 == 0) { f.push_back(2); n /= 2; }
  for (int p = 3; (long)p * p <= n; p += 2)
    while (n % p == 0) { f.push_back(p); n /= p; }
  if (n > 1) f.push_back(n);
  return f;
}

// exp(-2*pi*i*k/n) with the angle folded into the first octant first
cd unit_root(long k, long n)
{
  k %= n;
  if (k < 0) k += n;
  // use symmetry so that the argument handed to sin/cos is in [0, pi/4]
  long k8 = 8 * k;
  double c, s;
  auto cs = [&](long num) { // angle = 2*pi*num/(8n) ... num in [0, n]
    double a = (2.0 * M_PI * (double)num) / (8.0 * (double)n);
    c = std::cos(a); s = std::sin(a);
  };
  if (k8 < n)            { cs(k8);           return cd(c, -s); }
  else if (k8 < 2 * n)   { cs(2 * n - k8);   return cd(s, -c); }
  else if (k8 < 3 * n)   { cs(k8 - 2 * n);   return cd(-s, -c); }
  else if (k8 < 4 * n)   { cs(4 * n - k8);   return cd(-c, -s); }
  else if (k8 < 5 * n)   { cs(k8 - 4 * n);   return cd(-c, s); }
  else if (k8 < 6 * n)   { cs(6 * n - k8);   return cd(-s, c); }
  else if (k8 < 7 * n)   { cs(k8 - 6 * n);   return cd(s, c); }
  else                   { cs(8 * n - k8);   return cd(c, s); }
}

struct Stockham {
  int n = 0;
  std::vector<int> radices;
  std::vector<cd> W;      // forward roots exp(-2 pi i k/n)
  std::vector<cd> work;

  bool init(int n_)
  {
    n = n_;
    radices = factorize(n);
    for (int r : radices) if (r > 13) return false;
    W.resize(n);
    for (int k = 0; k < n; k++) W[k] = unit_root(k, n);
    work.resize(n);
    return true;
  }

  // forward transform of x (length n) in place; y is scratch
  void forward(cd *x)
  {
    cd *src = x, *dst = work.data();
    int len = n, s = 1;
    for (int r : radices) {
      pass(r, len, s, src, dst);
      len /= r; s *= r;
      std::swap(src, dst);
    }
    if (src != x) std::memcpy(x, src, sizeof(cd) * n);
  }

  // one Stockham pass: len = current sub-length, s = stride
  void pass(int r, int len, int s, const cd *x, cd *y)
  {
    const int m = len / r;
    const int tw = n / len; // W_len^p = W[p*tw]
    switch (r) {
    case 2:
      for (int p = 0; p < m; p++) {
        const cd w = W[(long)p * tw];
        const cd *a = x + (long)s * p, *b = x + (long)s * (p + m);
        cd *o0 = y + (long)s * (2 * p), *o1 = o0 + s;
        for (int q = 0; q < s; q++) {
          cd u = a[q], v = b[q];
          o0[q] = u + v;
          o1[q] = (u - v) * w;
        }
      }
      break;
    case 4:
      for (int p = 0; p < m; p++) {
        const cd w1 = W[(long)p * tw], w2 = W[(long)(2 * p) * tw % n], w3 = W[(long)(3L * p) * tw % n];
        const cd *a = x + (long)s * p, *b = a + (long)s * m, *c = b + (long)s * m, *d = c + (long)s * m;
        cd *o0 = y + (long)s * (4 * p), *o1 = o0 + s, *o2 = o1 + s, *o3 = o2 + s;
        for (int q = 0; q < s; q++) {
          cd t0 = a[q] + c[q], t1 = a[q] - c[q];
          cd t2 = b[q] + d[q], t3 = b[q] - d[q];
          cd jt3(t3.imag(), -t3.real()); // -i * t3
          o0[q] = t0 + t2;
          o1[q] = (t1 + jt3) * w1;
          o2[q] = (t0 - t2) * w2;
          o3[q] = (t1 - jt3) * w3;
        }
      }
      break;
    case 3: {
      const double c3 = -0.5, s3 = -0.86602540378443864676; // exp(-2 pi i/3)
      for (int p = 0; p < m; p++) {
        const cd w1 = W[(long)p * tw], w2 = W[(long)(2 * p) * tw % n];
        const cd *a = x + (long)s * p, *b = a + (long)s * m, *c = b + (long)s * m;
        cd *o0 = y + (long)s * (3 * p), *o1 = o0 + s, *o2 = o1 + s;
        for (int q = 0; q < s; q++) {
          cd t1 = b[q] + c[q], t2 = b[q] - c[q];
          cd u = a[q] + c3 * t1;
          cd v(-s3 * t2.imag(), s3 * t2.real()); // i*s3*t2
          o0[q] = a[q] + t1;
          o1[q] = (u + v) * w1;
          o2[q] = (u - v) * w2;
        }
      }
      break;
    }
    default:
      generic(r, len, s, x, y);
    }
  }

  void generic(int r, int len, int s, const cd *x, cd *y)
  {
    const int m = len / r;
    const int tw = n / len;
    const int rs = n / r; // W_r^j = W[j*rs]
    cd in[16], out[16];
    for (int p = 0; p < m; p++) {
      for (int q = 0; q < s; q++) {
        for (int k = 0; k < r; k++) in[k] = x[q + (long)s * (p + (long)k * m)];
        for (int j = 0; j < r; j++) {
          cd acc = in[0];
          for (int k = 1; k < r; k++) acc += in[k] * W[((long)j * k % r) * rs];
          out[j] = acc;
        }
        y[q + (long)s * ((long)r * p)] = out[0];
        for (int j = 1; j < r; j++)
          y[q + (long)s * ((long)r * p + j)] = out[j] * W[((long)j * p % len) * tw];
      }
    }
  }
};

struct Bluestein {
  int n = 0, m = 0;
  Stockham fft;          // length m (power of two)
  std::vector<cd> chirp; // exp(-i*pi*k^2/n), k in [0,n)
  std::vector<cd> B;     // forward FFT of the conjugate-chirp kernel
  std::vector<cd> buf;

  void init(int n_)
  {
    n = n_;
    m = 1;
    while (m < 2 * n - 1) m <<= 1;
    fft.init(m);
    chirp.resize(n);
    for (long k = 0; k < n; k++) chirp[k] = unit_root((k * k) % (2L * n), 2L * n);
    B.assign(m, cd(0, 0));
    B[0] = std::conj(chirp[0]);
    for (int k = 1; k < n; k++) B[k] = B[m - k] = std::conj(chirp[k]);
    fft.forward(B.data());
    buf.resize(m);
  }

  void forward(cd *x)
  {
    for (int k = 0; k < n; k++) buf[k] = x[k] * chirp[k];
    for (int k = n; k < m; k++) buf[k] = cd(0, 0);
    fft.forward(buf.data());
    for (int k = 0; k < m; k++) buf[k] = std::conj(buf[k] * B[k]);
    // inverse via conj(FFT(conj(.)))/m
    fft.forward(buf.data());
    const double inv = 1.0 / m;
    for (int k = 0; k < n; k++) x[k] = std::conj(buf[k]) * inv * chirp[k];
  }
};

} // namespace

struct oracle_fft_plan_s {
  int n;
  int sign;
  cd *in;
  cd *out;
  bool use_bluestein;
  Stockham st;
  Bluestein bl;
};

extern "C" {

fftw_plan fftw_plan_dft_1d(int n, fftw_complex *in, fftw_complex *out, int sign, unsigned)
{
  auto *p = new oracle_fft_plan_s;
  p->n = n;
  p->sign = sign;
  p->in = reinterpret_cast<cd *>(in);
  p->out = reinterpret_cast<cd *>(out);
  p->use_bluestein = !p->st.init(n);
  if (p->use_bluestein) p->bl.init(n);
  return p;
}

void fftw_execute(const fftw_plan p)
{
  const int n = p->n;
  if (p->out != p->in) std::memcpy(p->out, p->in, sizeof(cd) * n);
  cd *x = p->out;
  // backward = conj(forward(conj(x)))
  if (p->sign > 0) for (int k = 0; k < n; k++) x[k] = std::conj(x[k]);
  if (p->use_bluestein) p->bl.forward(x); else p->st.forward(x);
  if (p->sign > 0) for (int k = 0; k < n; k++) x[k] = std::conj(x[k]);
}

void fftw_destroy_plan(fftw_plan p) { delete p; }
int fftw_init_threads(void) { return 1; }
void fftw_plan_with_nthreads(int) {}
void fftw_cleanup_threads(void) {}

} // extern "C"
