// C++ test of the host classes, modelled on the reference's
// test/unit/process/ambiguity/TestAmbiguity.cpp (Catch2 is not available in
// this image, so plain checks).  Exercises the exact call sequence of
// blah2.cpp:268-287 through the drop-in classes.  Run on a GPU box:
//     test_ambiguity [--sequence]        (the values of the compiled reference: test_golden.cpp)
// Exit code 0 = all checks passed.
#include "data/IqData.h"
#include "data/Map.h"
#include "process/ambiguity/Ambiguity.h"
#include "process/clutter/WienerHopf.h"
#include "process/spectrum/SpectrumAnalyser.h"
#include "process/detection/Centroid.h"
#include "process/detection/CfarDetector1D.h"
#include "process/detection/Interpolate.h"
#include "process/meta/HammingNumber.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <random>
#include <string>
#include <stdexcept>

static int failures = 0;
#define CHECK(cond)                                                     \
  do {                                                                  \
    if (!(cond)) { std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
  } while (0)

static void random_iq(IqData &iq, unsigned seed)
{
  std::mt19937 gen(seed);
  std::uniform_real_distribution<> dist(-100.0, 100.0);
  for (uint32_t i = 0; i < iq.get_n(); ++i) iq.push_back({dist(gen), dist(gen)});
}

static void sequence_at_cfg2();

int main(int argc, char **argv)
{
  if (argc > 1 && std::string(argv[1]) == "--sequence") { // bench.py: only the timed sequence of blah2.cpp:264-287
    sequence_at_cfg2();
    std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
  }
  // TestHammingNumber.cpp:15-17
  CHECK(next_hamming(104) == 108);
  CHECK(next_hamming(3322) == 3375);
  CHECK(next_hamming(19043) == 19200);

  const int32_t delayMin = -10, delayMax = 300, dopplerMin = -300, dopplerMax = 300;
  const uint32_t fs = 2000000;
  const float tCpi = 0.5;
  const uint32_t nSamples = tCpi * fs;

  for (int rh = 0; rh < 2; rh++) {
    // TestAmbiguity.cpp "Constructor" / "Constructor_Round"
    Ambiguity ambiguity(delayMin, delayMax, dopplerMin, dopplerMax, fs, nSamples, rh != 0);
    CHECK(std::fabs(ambiguity.get_cpi() - tCpi) <= 0.02);
    CHECK(ambiguity.get_doppler_middle() == 0);
    CHECK(ambiguity.get_n_corr() == 3322);
    CHECK(ambiguity.get_n_delay_bins() == delayMax + std::abs(delayMin) + 1);
    CHECK(ambiguity.get_n_doppler_bins() == 301);
    CHECK(ambiguity.get_nfft() == (rh ? 6750u : 6643u));

    // "Process_Simple"
    IqData x{nSamples}, y{nSamples};
    random_iq(x, 1);
    random_iq(y, 2);
    const auto t0 = std::chrono::steady_clock::now();
    auto map = ambiguity.process(&x, &y);
    map->set_metrics();
    const auto t1 = std::chrono::steady_clock::now();
    std::printf("Ambiguity::process(IqData*, IqData*) + set_metrics, %u samples: %.2f ms wall (FIFO pops, c64->c32, H2D, kernels, D2H, Map fill)\n",
                nSamples, std::chrono::duration<double, std::milli>(t1 - t0).count());
    CHECK(map->maxPower > 0.0);
    CHECK(map->noisePower > 0.0);
    CHECK(x.get_length() == nSamples - 3322u * 301u); // process() consumes by pop_front
    CHECK(ambiguity.get_n_samples() == 3322u * 301u);

    // underflow behaves like IqData::pop_front
    bool threw = false;
    try { ambiguity.process(&x, &y); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
  }

  // the t2 loop of blah2.cpp:268-287 with an injected target
  {
    const uint32_t n = 200000, fsl = 1000000;
    IqData x{n}, y{n};
    std::mt19937 gen(7);
    std::normal_distribution<> g(0.0, 300.0);
    std::vector<std::complex<double>> xs(n);
    for (auto &v : xs) v = {std::round(g(gen)), std::round(g(gen))};
    for (uint32_t i = 0; i < n; i++) {
      std::complex<double> t = 0.8 * xs[i];
      if (i >= 37) t += 0.05 * xs[i - 37] * std::exp(std::complex<double>(0, 2 * M_PI * (-63.0) * i / fsl));
      t += std::complex<double>(0.1 * g(gen), 0.1 * g(gen));
      x.push_back(xs[i]);
      y.push_back({std::round(t.real()), std::round(t.imag())});
    }
    // blah2.cpp:264: spectrum of the reference channel, x is not consumed; a few
    // bins against the defining sum  X[(k*D + nfft/2 + 1) mod nfft]
    {
      SpectrumAnalyser spectrumAnalyser(n, 2000);
      spectrumAnalyser.process(&x);
      CHECK(x.get_length() == n);
      const std::string sj = x.to_json(42);
      CHECK(sj.find("\"frequency\":[],\"spectrum\":[") != std::string::npos); // the axis the reference emits is empty
      const uint32_t D = 100, nS = 2000, N = 200000;
      std::vector<std::complex<double>> got = x.get_spectrum();
      CHECK(got.size() == nS);
      for (uint32_t k : {0u, 1u, 999u, 1000u, 1999u}) {
        const uint64_t bin = ((uint64_t)k * D + N / 2 + 1) % N;
        std::complex<double> acc = 0;
        for (uint32_t i = 0; i < N; i++) acc += xs[i] * std::exp(std::complex<double>(0, -2 * M_PI * (double)((bin * i) % N) / N));
        CHECK(got.size() == nS && std::abs(got[k] - acc) <= 1e-9 * std::sqrt((double)N) * 300.0 * 1e3);
      }
    }
    WienerHopf filter(-10, 100, n);
    Ambiguity ambiguity(-10, 100, -100, 100, fsl, n, true);
    CfarDetector1D cfar(1e-5, 2, 6, 5, 15.0);
    CHECK(filter.process(&x, &y));
    auto map = ambiguity.process(&x, &y);
    map->set_metrics();
    auto det1 = cfar.process(map);
    Centroid centroid(6, 6, 1.0 / 0.2);
    Interpolate interpolate(true, true);
    auto det2 = centroid.process(det1.get());        // blah2.cpp:286
    auto det = interpolate.process(det2.get(), map); // blah2.cpp:287
    CHECK(det2->get_nDetections() <= det1->get_nDetections());
    bool found = false;
    auto dl = det->get_delay();
    auto dp = det->get_doppler();
    for (size_t i = 0; i < dl.size(); i++)
      if (std::fabs(dl[i] - 37) < 1.0 && std::fabs(dp[i] + 63.0) < 5.1) found = true;
    CHECK(found);
    const std::string js = map->delay_bin_to_km(map->to_json(1234567890123ull), fsl);
    CHECK(js.find("{\"timestamp\":1234567890123,\"nRows\":41,\"nCols\":111,\"noisePower\":") == 0);
    CHECK(js.find("\"delay\":[-2.99,-2.69") != std::string::npos); // -10*c/fs/1000 = -2.9979 -> truncated
    std::printf("chain: %zu detections, noisePower %.3f maxPower %.3f, json %zu bytes\n", dl.size(),
                map->noisePower, map->maxPower, js.size());

    // Map::set_metrics is not sticky (reference Map.cpp:187-206 always reduces `data`): change the cells
    // and the values follow; CfarDetector1D then runs on the caller's cells (uploaded), not on the
    // engine's stale device copy.
    const double noise0 = map->noisePower, max0 = map->maxPower;
    const size_t nDet0 = det1->get_nDetections();
    map->set_metrics();
    CHECK(map->noisePower == noise0 && map->maxPower == max0); // untouched: adopted again
    for (auto &row : map->data)
      for (auto &v : row) v *= 10.0; // +10 dB on every cell
    map->set_metrics();
    CHECK(std::fabs(map->noisePower - (noise0 + 10.0)) < 1e-3);
    CHECK(std::fabs(map->maxPower - max0) < 1e-3);
    auto det10 = cfar.process(map); // scale-free detector: same cells fire, snr unchanged (both shift by 10 dB)
    CHECK(det10->get_nDetections() == nDet0);
    CHECK(det10->get_delay() == det1->get_delay() && det10->get_doppler() == det1->get_doppler());
    // a map the caller builds from scratch: one strong cell on a flat floor
    Map<std::complex<double>> own(5, 40);
    for (int j = 0; j < 40; j++) own.delay.push_back(j - 2);
    for (int i = 0; i < 5; i++) own.doppler.push_back(20.0 * (i - 2));
    own.data[4][20] = {1000.0, 0.0};
    own.set_metrics();
    CHECK(std::fabs(own.noisePower - 30.0 / 200.0) < 1e-12 && std::fabs(own.maxPower - (30.0 - 0.15)) < 1e-12);
    auto detOwn = CfarDetector1D(1e-3, 1, 4, 0, 15.0).process(&own);
    CHECK(detOwn->get_nDetections() == 1 && detOwn->get_delay()[0] == 18.0 && detOwn->get_doppler()[0] == 40.0);
  }
  // The device-resident chain (util/DeviceContext.h): (i) WienerHopf leaves the filtered channel in HBM as the front of y;
  // reading y through the FIFO first (get_data) brings it to the host and must not change what Ambiguity computes
  // (the device values are fp32, their host copies exact); (ii) the sequence of blah2.cpp:264-287 at BASELINE configs[1]
  // (2 MS/s, 1 s CPI), timed per CPI.
  {
    const uint32_t n = 200000, fsl = 1000000;
    std::mt19937 gen(11);
    std::normal_distribution<> g(0.0, 300.0);
    std::vector<std::complex<double>> xs(n), ys(n);
    for (uint32_t i = 0; i < n; i++) {
      xs[i] = {std::round(g(gen)), std::round(g(gen))};
      std::complex<double> t = 0.8 * xs[i] + std::complex<double>(0.1 * g(gen), 0.1 * g(gen));
      if (i >= 21) t += 0.05 * xs[i - 21] * std::exp(std::complex<double>(0, 2 * M_PI * 40.0 * i / fsl));
      ys[i] = {std::round(t.real()), std::round(t.imag())};
    }
    std::vector<std::vector<std::complex<double>>> maps[2];
    double noise[2] = {0, 0};
    for (int pass = 0; pass < 2; pass++) {
      IqData x{n}, y{n};
      for (uint32_t i = 0; i < n; i++) { x.push_back(xs[i]); y.push_back(ys[i]); }
      WienerHopf filter(-10, 100, n);
      Ambiguity ambiguity(-10, 100, -100, 100, fsl, n, true);
      CHECK(filter.process(&x, &y));
      CHECK(y.get_length() == n && y.device_front_count() == n);
      if (pass == 1) {
        const auto yd = y.get_data(); // materialises
        CHECK(yd.size() == n && y.device_front_count() == 0);
        double e_in = 0, e_out = 0;
        for (uint32_t i = 0; i < n; i++) { e_in += std::norm(ys[i]); e_out += std::norm(yd[i]); }
        CHECK(e_out < 0.1 * e_in); // the direct path is cancelled
        CHECK(x.get_data()[12345] == xs[12345]); // x untouched
      }
      auto map = ambiguity.process(&x, &y);
      map->set_metrics();
      maps[pass] = map->data;
      noise[pass] = map->noisePower;
      CHECK(x.get_length() == n - ambiguity.get_n_samples() && y.get_length() == x.get_length());
    }
    CHECK(maps[0] == maps[1] && noise[0] == noise[1]);
  }
  sequence_at_cfg2();
  std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
  return failures ? 1 : 0;
}

static void sequence_at_cfg2()
{
  {
    const uint32_t fs2 = 2000000, n2 = 2000000;
    // CPI 0 attaches the eager path.  One device-to-host copy in the first few CPIs after its uploads began takes 7 ms, once
    // per process (the runtime growing a pool, by the look of it): the stage times are MEDIANS over the timed CPIs, every
    // CPI's total is printed
    const int nCpi = 10, nWarm = 2;
    SpectrumAnalyser spectrumAnalyser(n2, 2000);
    WienerHopf filter(-10, 400, n2);
    Ambiguity ambiguity(-10, 400, -256, 256, fs2, n2, true);
    CfarDetector1D cfar(1e-5, 2, 6, 5, 15.0);
    IqData x{n2}, y{n2};
    std::mt19937 gen(5);
    std::uniform_int_distribution<int> u(-300, 300);
    std::vector<std::complex<double>> xs(n2);
    std::vector<double> seqs, parts[5];
    size_t nDet = 0;
    std::vector<std::complex<double>> ysv(n2);
    double t_cut = 0;
    for (int c = 0; c < nCpi; c++) {
      for (auto &v : xs) v = {(double)u(gen), (double)u(gen)};
      for (uint32_t i = 0; i < n2; i++) {
        std::complex<double> t = 0.8 * xs[i] + std::complex<double>(u(gen) * 0.1, u(gen) * 0.1);
        if (i >= 37) t += 0.05 * xs[i - 37] * std::exp(std::complex<double>(0, 2 * M_PI * (-63.0) * i / fs2));
        ysv[i] = {std::round(t.real()), std::round(t.imag())};
      }
      // blah2.cpp:254-258: the cut, sample by sample.  Timed apart from the sequence: from the second CPI on push_back also
      // narrows into the pinned shadow and sends every 256 k samples on their way (util/DeviceContext.h)
      const auto tc0 = std::chrono::steady_clock::now();
      for (uint32_t i = 0; i < n2; i++) {
        x.push_back(xs[i]);
        y.push_back(ysv[i]);
      }
      if (c >= nWarm) t_cut += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count();
      auto tic = std::chrono::steady_clock::now();
      auto lap = [&](int k) {
        const auto now = std::chrono::steady_clock::now();
        if (c >= nWarm) parts[k].push_back(std::chrono::duration<double, std::milli>(now - tic).count());
        tic = now;
      };
      const auto t0 = tic;
      spectrumAnalyser.process(&x); lap(0);           // :264
      CHECK(filter.process(&x, &y)); lap(1);          // :270
      auto map = ambiguity.process(&x, &y); lap(2);   // :278
      map->set_metrics(); lap(3);                     // :279
      auto det = cfar.process(map); lap(4);           // :285
      if (c >= nWarm) seqs.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      nDet = det->get_nDetections();
      bool found = false;
      for (size_t i = 0; i < nDet; i++)
        if (det->get_delay()[i] == 37.0 && std::fabs(det->get_doppler()[i] + 63.0) < 1.01) found = true;
      CHECK(found);
      CHECK(x.get_length() == n2 - ambiguity.get_n_samples());
    }
    auto median = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::printf("sequence blah2.cpp:264-287 at 2 MS/s x 1 s (Spectrum, WienerHopf 410 taps, Ambiguity 513 x 411, set_metrics, CFAR): "
                "%.2f ms/CPI  [spectrum %.2f, filter %.2f, ambiguity %.2f, set_metrics %.2f, cfar %.2f]; %zu detections\n",
                median(seqs), median(parts[0]), median(parts[1]), median(parts[2]), median(parts[3]), median(parts[4]), nDet);
    std::printf("  (medians of %zu CPIs; each CPI:", seqs.size());
    for (double v : seqs) std::printf(" %.2f", v);
    std::printf(" ms)\n");
    std::printf("cut blah2.cpp:254-258 (2 x %u push_back, narrowing + eager upload inside): %.2f ms/CPI\n", n2, t_cut / (nCpi - nWarm));
  }
  // the eager path (samples narrowed and uploaded as they are pushed) against the per-CPI path, also with a front that
  // wraps around the end of the ring and with a reader of y between the filter and the map
  {
    const uint32_t n = 200000, fsl = 1000000;
    std::mt19937 gen(11);
    std::uniform_int_distribution<int> u(-300, 300);
    auto run = [&](bool eager, uint32_t extra, bool peek) {
      IqData x{n}, y{n};
      WienerHopf filter(-10, 100, n);
      Ambiguity ambiguity(-10, 100, -100, 100, fsl, n, true);
      std::mt19937 g2(23);
      std::vector<double> cells;
      for (int c = 0; c < 3; c++) {
        if (!eager) { x.detach_shadow(); y.detach_shadow(); } // keeps this FIFO on the per-CPI path
        const uint32_t m = n + (c == 2 ? extra : 0);       // `extra` more than the FIFO holds: the oldest are evicted, the front moves
        for (uint32_t i = 0; i < m; i++) {
          const std::complex<double> a{(double)u(g2), (double)u(g2)};
          x.push_back(a);
          y.push_back({std::round(0.7 * a.real() + 0.1 * u(g2)), std::round(0.7 * a.imag() + 0.1 * u(g2))});
        }
        CHECK(filter.process(&x, &y));
        if (peek && c == 2) CHECK(y.get_data().size() == n);
        auto map = ambiguity.process(&x, &y);
        if (c == 2)
          for (const auto &row : map->data)
            for (const auto &v : row) { cells.push_back(v.real()); cells.push_back(v.imag()); }
      }
      return cells;
    };
    const auto ref = run(false, 0, false);
    CHECK(run(true, 0, false) == ref);
    CHECK(run(true, 0, true) == ref);
    const auto refw = run(false, 5000, false);
    CHECK(refw != ref);
    CHECK(run(true, 5000, false) == refw);
    CHECK(run(true, 5000, true) == refw);
  }
}
