// The drop-in classes against the COMPILED REFERENCE's values (tests/golden/*.npz, written by tests/golden/make_golden.py
// from /root/reference's own sources), through the C++ class surface a blah2 maintainer links -- IqData::push_back,
// SpectrumAnalyser / WienerHopf / Ambiguity::process(IqData*, IqData*), Map::set_metrics, CfarDetector1D, Centroid,
// Interpolate -- with the gates of the Python parity tests (tests/test_ambiguity_gpu.py, test_clutter_gpu.py,
// test_cfar_gpu.py).  Modelled on the reference's "Process_File" (TestAmbiguity.cpp:147-178), whose capture is not
// shipped; the call sequence is blah2.cpp:254-287.
//
//     test_golden fixture.bin [fixture.bin ...]      (flat little-endian dumps of the .npz, tests/test_host_cpp_gpu.py)
//
// Every fixture runs through four host paths that must all give the reference's values:
//   per-cpi   the FIFO detached from its pinned shadow: one narrow + upload per channel and CPI
//   eager     three consecutive CPIs of ONE pair of FIFOs, the shadow attached from the second on (samples narrowed and
//             uploaded while they are pushed); the middle CPI carries other data, the first and third are checked
//   wrapped   extra samples pushed in front, so that the CPI's front sits somewhere inside the ring and wraps
//   peek      y read through the FIFO between the filter and the map (the device-only front is brought home)
// Exit code 0 = all checks passed.
#include "data/IqData.h"
#include "data/Map.h"
#include "process/ambiguity/Ambiguity.h"
#include "process/clutter/WienerHopf.h"
#include "process/detection/Centroid.h"
#include "process/detection/CfarDetector1D.h"
#include "process/detection/Interpolate.h"
#include "process/spectrum/SpectrumAnalyser.h"

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <vector>

typedef std::complex<double> cd;

static int failures = 0;
static std::string ctx;
#define CHECK(cond)                                                                                              \
  do {                                                                                                           \
    if (!(cond)) { std::printf("CHECK FAILED [%s] %s:%d: %s\n", ctx.c_str(), __FILE__, __LINE__, #cond); failures++; } \
  } while (0)
#define CHECK_LE(val, lim)                                                                                       \
  do {                                                                                                           \
    const double v_ = (val), l_ = (lim);                                                                         \
    if (!(v_ <= l_)) { std::printf("CHECK FAILED [%s] %s:%d: %s = %.3e > %.3e\n", ctx.c_str(), __FILE__, __LINE__, #val, v_, l_); failures++; } \
  } while (0)

// gates of tests/test_ambiguity_gpu.py:19-21, test_clutter_gpu.py:15, test_cfar_gpu.py
static const double PEAK_TOL = 1e-5, CELL_TOL = 1e-4, DB_TOL = 1e-3, Y_TOL = 1e-4, SNR_TOL = 1e-3, SPEC_TOL = 1e-9;

struct DetList { std::vector<double> delay, doppler, snr; };

struct Fixture {
  std::string name;
  int64_t fs, n, dmin, dmax, fmin, fmax, rh;
  int64_t nD, nDelay, nCorr, nfft;
  double cpi, dopplerMiddle;
  std::vector<int16_t> iq;            // [n][4]: I1 Q1 I2 Q2
  std::vector<double> delayAxis, dopplerAxis;
  std::vector<cd> map;                // [nD][nDelay]
  double metrics[2];
  double det[7];                      // pfa, nGuard, nTrain, minDelay, minDoppler, nCentroid, centroid resolution
  DetList cfar, centroid, interp;
  int64_t cdmin, cdmax, clutterOk;
  std::vector<cd> clutterY, chainMap;
  double chainMetrics[2];
  DetList chainCfar;
  std::vector<cd> spectrum;
  int64_t spectrumNFreq;
};

struct Reader {
  FILE *f;
  bool ok = true;
  template <class T> void get(T *p, size_t cnt) { if (std::fread(p, sizeof(T), cnt, f) != cnt) ok = false; }
  int64_t i64() { int64_t v = 0; get(&v, 1); return v; }
  double f64() { double v = 0; get(&v, 1); return v; }
  template <class T> void vec(std::vector<T> &v) { const int64_t c = i64(); if (!ok || c < 0 || c > (int64_t)1 << 31) { ok = false; return; } v.resize((size_t)c); get(v.data(), (size_t)c); }
  void dets(DetList &d) { vec(d.delay); vec(d.doppler); vec(d.snr); }
};

static bool load(const char *path, Fixture &g)
{
  Reader r{std::fopen(path, "rb")};
  if (!r.f) return false;
  char magic[8];
  r.get(magic, 8);
  if (std::memcmp(magic, "B2GOLD01", 8) != 0) { std::fclose(r.f); return false; }
  g.fs = r.i64(); g.n = r.i64(); g.dmin = r.i64(); g.dmax = r.i64(); g.fmin = r.i64(); g.fmax = r.i64(); g.rh = r.i64();
  g.nD = r.i64(); g.nDelay = r.i64(); g.nCorr = r.i64(); g.nfft = r.i64();
  g.cpi = r.f64(); g.dopplerMiddle = r.f64();
  r.vec(g.iq); r.vec(g.delayAxis); r.vec(g.dopplerAxis); r.vec(g.map);
  r.get(g.metrics, 2); r.get(g.det, 7);
  r.dets(g.cfar); r.dets(g.centroid); r.dets(g.interp);
  g.cdmin = r.i64(); g.cdmax = r.i64(); g.clutterOk = r.i64();
  r.vec(g.clutterY); r.vec(g.chainMap); r.get(g.chainMetrics, 2); r.dets(g.chainCfar);
  r.vec(g.spectrum); g.spectrumNFreq = r.i64();
  std::fclose(r.f);
  return r.ok && (int64_t)g.iq.size() == 4 * g.n && (int64_t)g.map.size() == g.nD * g.nDelay;
}

static void push_cpi(const Fixture &g, IqData &x, IqData &y)
{
  for (int64_t i = 0; i < g.n; i++) { // blah2.cpp:254-258: sample by sample
    x.push_back({(double)g.iq[4 * i], (double)g.iq[4 * i + 1]});
    y.push_back({(double)g.iq[4 * i + 2], (double)g.iq[4 * i + 3]});
  }
}
static void push_noise(IqData &x, IqData &y, uint32_t count, unsigned seed)
{
  std::mt19937 gen(seed);
  std::uniform_int_distribution<int> u(-500, 500);
  for (uint32_t i = 0; i < count; i++) { x.push_back({(double)u(gen), (double)u(gen)}); y.push_back({(double)u(gen), (double)u(gen)}); }
}

static void check_map(const Map<cd> *m, const std::vector<cd> &ref, const Fixture &g)
{
  CHECK((int64_t)m->data.size() == g.nD && (int64_t)m->data[0].size() == g.nDelay);
  if ((int64_t)m->data.size() != g.nD || (int64_t)m->data[0].size() != g.nDelay) return;
  CHECK(m->delay.size() == g.delayAxis.size() && m->doppler.size() == g.dopplerAxis.size());
  for (size_t j = 0; j < g.delayAxis.size() && j < m->delay.size(); j++) CHECK((double)m->delay[j] == g.delayAxis[j]);
  for (size_t i = 0; i < g.dopplerAxis.size() && i < m->doppler.size(); i++) CHECK(m->doppler[i] == g.dopplerAxis[i]);
  double peak = 0, mean = 0, emax = 0;
  for (const cd &v : ref) { peak = std::max(peak, std::abs(v)); mean += std::abs(v); }
  mean /= (double)ref.size();
  double relmax = 0;
  for (int64_t i = 0; i < g.nD; i++)
    for (int64_t j = 0; j < g.nDelay; j++) {
      const cd r = ref[(size_t)(i * g.nDelay + j)];
      const double e = std::abs(m->data[i][j] - r);
      emax = std::max(emax, e);
      if (std::abs(r) > mean) relmax = std::max(relmax, e / std::abs(r));
    }
  CHECK_LE(emax / peak, PEAK_TOL);
  CHECK_LE(relmax, CELL_TOL);
}

static void check_dets(Detection *d, const DetList &ref, double snrTol, double posTol)
{
  const auto a = d->get_delay(), b = d->get_doppler(), c = d->get_snr();
  CHECK(a.size() == ref.delay.size());
  if (a.size() != ref.delay.size()) return;
  for (size_t i = 0; i < a.size(); i++) {
    CHECK_LE(std::fabs(a[i] - ref.delay[i]), posTol);
    CHECK_LE(std::fabs(b[i] - ref.doppler[i]), posTol);
    CHECK_LE(std::fabs(c[i] - ref.snr[i]), snrTol);
  }
}

// the detector chain of blah2.cpp:285-287 on `map`, against the three reference lists
static void check_detector_chain(const Fixture &g, Map<cd> *map, bool all_stages)
{
  CfarDetector1D cfar(g.det[0], (int8_t)g.det[1], (int8_t)g.det[2], (int8_t)g.det[3], g.det[4]);
  auto d0 = cfar.process(map);
  check_dets(d0.get(), all_stages ? g.cfar : g.chainCfar, SNR_TOL, 0.0); // cells: exact; the fixtures' targets are far from the threshold
  if (!all_stages) return;
  Centroid centroid((uint16_t)g.det[5], (uint16_t)g.det[5], g.det[6]); // blah2.cpp:183
  auto d1 = centroid.process(d0.get());
  check_dets(d1.get(), g.centroid, SNR_TOL, 0.0);
  Interpolate interpolate(true, true);
  auto d2 = interpolate.process(d1.get(), map);
  // interpolated positions are ratios of fp32-map cell differences: 1e-3 bins / Hz (measured ~1e-5)
  check_dets(d2.get(), g.interp, 2e-3, 1e-3);
}

// one CPI that is already in the FIFOs, through blah2.cpp:264-287
static void run_sequence(const Fixture &g, IqData &x, IqData &y, SpectrumAnalyser &spec, WienerHopf *filter, Ambiguity &amb,
                         bool with_filter, bool peek)
{
  // :264 the spectrum of the reference channel (x is not consumed)
  spec.process(&x);
  {
    const auto &s = x.get_spectrum();
    CHECK(s.size() == g.spectrum.size());
    CHECK((int64_t)x.get_frequency().size() == g.spectrumNFreq);
    double smax = 0, emax = 0;
    for (const cd &v : g.spectrum) smax = std::max(smax, std::abs(v));
    for (size_t k = 0; k < s.size() && k < g.spectrum.size(); k++) emax = std::max(emax, std::abs(s[k] - g.spectrum[k]));
    CHECK_LE(emax / smax, SPEC_TOL);
  }
  if (with_filter) {
    // :270 y <- y - w * x; the filtered channel stays on the device as the front of y
    const bool ok = filter->process(&x, &y);
    CHECK(ok == (g.clutterOk != 0));
    CHECK((int64_t)y.get_length() == g.n && (int64_t)x.get_length() == g.n);
    if (peek) {
      const auto yd = y.get_data(); // brings the device-only front home
      CHECK((int64_t)yd.size() == g.n);
      double ymax = 0, emax = 0;
      for (const cd &v : g.clutterY) ymax = std::max(ymax, std::abs(v));
      for (size_t i = 0; i < yd.size() && i < g.clutterY.size(); i++) emax = std::max(emax, std::abs(yd[i] - g.clutterY[i]));
      CHECK_LE(emax / ymax, Y_TOL);
    }
  }
  // :278-279
  Map<cd> *map = amb.process(&x, &y);
  map->set_metrics();
  CHECK((int64_t)x.get_length() == g.n - g.nCorr * g.nD && x.get_length() == y.get_length()); // consumed by pop_front
  if (!with_filter) {
    check_map(map, g.map, g);
    CHECK_LE(std::fabs(map->noisePower - g.metrics[0]), DB_TOL);
    CHECK_LE(std::fabs(map->maxPower - g.metrics[1]), DB_TOL);
    check_detector_chain(g, map, true);
  } else if (g.clutterOk) {
    // after cancellation the reference level is the uncancelled map's (tests/test_full_chain_gpu.py): a tap error shows up
    // coherently at zero Doppler; cell-wise on the cells within 20 dB of the cancelled map's peak
    double peak0 = 0, peak = 0, emax = 0, relmax = 0;
    for (const cd &v : g.map) peak0 = std::max(peak0, std::abs(v));
    for (const cd &v : g.chainMap) peak = std::max(peak, std::abs(v));
    for (int64_t i = 0; i < g.nD; i++)
      for (int64_t j = 0; j < g.nDelay; j++) {
        const cd r = g.chainMap[(size_t)(i * g.nDelay + j)];
        const double e = std::abs(map->data[i][j] - r);
        emax = std::max(emax, e);
        if (std::abs(r) > 0.1 * peak) relmax = std::max(relmax, e / std::abs(r));
      }
    CHECK_LE(emax / peak0, 1e-4);
    CHECK_LE(relmax, 1e-3);
    CHECK_LE(std::fabs(map->noisePower - g.chainMetrics[0]), DB_TOL);
    CHECK_LE(std::fabs(map->maxPower - g.chainMetrics[1]), DB_TOL);
    check_detector_chain(g, map, false);
  }
}

static void run_fixture(const Fixture &g)
{
  const uint32_t n = (uint32_t)g.n;
  // the clutter filter's own geometry is the fixture's (it may differ from the map's delay window)
  for (int with_filter = 0; with_filter < 2; with_filter++) {
    const char *fl = with_filter ? "+filter" : "";
    { // per-CPI path
      ctx = g.name + " per-cpi" + fl;
      IqData x{n}, y{n};
      x.detach_shadow(); y.detach_shadow();
      SpectrumAnalyser spec(n, 2000.0); // blah2.cpp:198
      WienerHopf filter((int32_t)g.cdmin, (int32_t)g.cdmax, n);
      Ambiguity amb((int32_t)g.dmin, (int32_t)g.dmax, (int32_t)g.fmin, (int32_t)g.fmax, (uint32_t)g.fs, n, g.rh != 0);
      CHECK(amb.get_n_doppler_bins() == g.nD && amb.get_n_delay_bins() == g.nDelay && amb.get_n_corr() == g.nCorr && amb.get_nfft() == g.nfft);
      CHECK(amb.get_cpi() == g.cpi && amb.get_doppler_middle() == g.dopplerMiddle);
      push_cpi(g, x, y);
      run_sequence(g, x, y, spec, &filter, amb, with_filter != 0, with_filter != 0);
    }
    { // three CPIs of one pair of FIFOs: eager uploads from the second on; first and third carry the fixture
      IqData x{n}, y{n};
      SpectrumAnalyser spec(n, 2000.0);
      WienerHopf filter((int32_t)g.cdmin, (int32_t)g.cdmax, n);
      Ambiguity amb((int32_t)g.dmin, (int32_t)g.dmax, (int32_t)g.fmin, (int32_t)g.fmax, (uint32_t)g.fs, n, g.rh != 0);
      for (int c = 0; c < 3; c++) {
        ctx = g.name + " eager cpi " + std::to_string(c) + fl;
        if (c == 1) {
          push_noise(x, y, n, 77); // other data: what it leaves behind must not be seen by the third CPI
          spec.process(&x);
          if (with_filter) (void)filter.process(&x, &y);
          (void)amb.process(&x, &y);
          continue;
        }
        push_cpi(g, x, y); // evicts what the previous CPI left (IqData.cpp:37-47)
        run_sequence(g, x, y, spec, &filter, amb, with_filter != 0, false);
      }
    }
    { // the front somewhere inside the ring, wrapping around its end; with a reader of y in between when filtering
      ctx = g.name + " wrapped" + fl;
      IqData x{n}, y{n};
      SpectrumAnalyser spec(n, 2000.0);
      WienerHopf filter((int32_t)g.cdmin, (int32_t)g.cdmax, n);
      Ambiguity amb((int32_t)g.dmin, (int32_t)g.dmax, (int32_t)g.fmin, (int32_t)g.fmax, (uint32_t)g.fs, n, g.rh != 0);
      push_noise(x, y, n, 5);           // a first CPI: attaches the eager path
      spec.process(&x);
      (void)amb.process(&x, &y);
      push_noise(x, y, n / 3 + 17, 6);  // moves the ring's head to an odd place
      push_cpi(g, x, y);
      run_sequence(g, x, y, spec, &filter, amb, with_filter != 0, with_filter != 0);
    }
  }
}

int main(int argc, char **argv)
{
  if (argc < 2) { std::printf("usage: test_golden fixture.bin [...]\n"); return 2; }
  for (int a = 1; a < argc; a++) {
    Fixture g;
    if (!load(argv[a], g)) { std::printf("cannot read %s\n", argv[a]); return 2; }
    const std::string p(argv[a]);
    const size_t s = p.find_last_of('/');
    g.name = p.substr(s == std::string::npos ? 0 : s + 1);
    const int before = failures;
    run_fixture(g);
    std::printf("%s: %s\n", g.name.c_str(), failures == before ? "ok" : "FAILED");
  }
  std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
  return failures ? 1 : 0;
}
