// TEST INFRASTRUCTURE ONLY -- part of oracle/ (see oracle/README.md).
//
// extern "C" driver around the *reference's own classes*, compiled from
// /root/reference/src (Ambiguity.cpp, HammingNumber.cpp, WienerHopf.cpp,
// CfarDetector1D.cpp, Centroid.cpp, Interpolate.cpp) so that Python tests
// and bench.py's cpu_baseline leg can run the reference through ctypes.
// The call order mirrors the t2 loop of /root/reference/src/blah2.cpp:263-289.

#include "process/ambiguity/Ambiguity.h"
#include "process/clutter/WienerHopf.h"
#include "process/spectrum/SpectrumAnalyser.h"

extern std::vector<std::complex<double>> g_ref_last_spectrum;
extern std::vector<double> g_ref_last_frequency;
#include "process/detection/Centroid.h"
#include "process/detection/CfarDetector1D.h"
#include "process/detection/Interpolate.h"
#include "process/meta/HammingNumber.h"

#include <chrono>
#include <complex>
#include <cstring>
#include <memory>

namespace {
struct RefAmb {
  std::unique_ptr<Ambiguity> amb;
  Map<std::complex<double>> *map = nullptr; // owned by amb
  std::unique_ptr<Detection> det;           // last detector-chain output
  uint32_t nSamples = 0;
};

void fill(IqData &q, const double *iq, uint32_t n)
{
  for (uint32_t i = 0; i < n; i++) q.push_back({iq[2 * i], iq[2 * i + 1]});
}

int64_t copy_det(Detection *d, double *delay, double *doppler, double *snr, int64_t cap)
{
  auto a = d->get_delay();
  auto b = d->get_doppler();
  auto c = d->get_snr();
  const int64_t n = (int64_t)a.size();
  for (int64_t i = 0; i < n && i < cap; i++) { delay[i] = a[i]; doppler[i] = b[i]; snr[i] = c[i]; }
  return n;
}
} // namespace

extern "C" {

uint32_t ref_next_hamming(uint32_t v) { return next_hamming(v); }

void *ref_amb_create(int32_t dMin, int32_t dMax, int32_t fMin, int32_t fMax, uint32_t fs,
                     uint32_t n, int roundHamming)
{
  auto *h = new RefAmb;
  h->amb = std::make_unique<Ambiguity>(dMin, dMax, fMin, fMax, fs, n, roundHamming != 0);
  h->nSamples = n;
  return h;
}

void ref_amb_destroy(void *p) { delete static_cast<RefAmb *>(p); }

// out[0..6] = nDopplerBins, nDelayBins, nCorr, nfft, cpi, dopplerMiddle, nSamples(after ctor)
void ref_amb_dims(void *p, double *out)
{
  auto *h = static_cast<RefAmb *>(p);
  out[0] = h->amb->get_n_doppler_bins();
  out[1] = h->amb->get_n_delay_bins();
  out[2] = h->amb->get_n_corr();
  out[3] = h->amb->get_nfft();
  out[4] = h->amb->get_cpi();
  out[5] = h->amb->get_doppler_middle();
  out[6] = h->amb->get_n_samples();
}

// x, y: interleaved (re, im) doubles, n complex samples each.
// map_out: nDoppler*nDelay interleaved complex doubles, row-major [doppler][delay].
// metrics[0] = noisePower, metrics[1] = maxPower (Map::set_metrics, blah2.cpp:279)
// leftover[0..1] = samples left in x / y after process() (it consumes by pop_front)
// Returns wall seconds spent inside Ambiguity::process + set_metrics only.
double ref_amb_process(void *p, const double *x, const double *y, uint32_t n, double *map_out,
                       double *delay_axis, double *doppler_axis, double *metrics,
                       uint32_t *leftover)
{
  auto *h = static_cast<RefAmb *>(p);
  IqData qx(n), qy(n);
  fill(qx, x, n);
  fill(qy, y, n);
  const auto t0 = std::chrono::steady_clock::now();
  h->map = h->amb->process(&qx, &qy);
  h->map->set_metrics();
  const auto t1 = std::chrono::steady_clock::now();
  const uint32_t nR = h->map->get_nRows(), nC = h->map->get_nCols();
  if (map_out)
    for (uint32_t i = 0; i < nR; i++)
      std::memcpy(map_out + 2 * (size_t)i * nC, h->map->data[i].data(), sizeof(double) * 2 * nC);
  if (delay_axis) for (uint32_t j = 0; j < nC; j++) delay_axis[j] = h->map->delay[j];
  if (doppler_axis) for (uint32_t i = 0; i < nR; i++) doppler_axis[i] = h->map->doppler[i];
  if (metrics) { metrics[0] = h->map->noisePower; metrics[1] = h->map->maxPower; }
  if (leftover) { leftover[0] = qx.get_length(); leftover[1] = qy.get_length(); }
  return std::chrono::duration<double>(t1 - t0).count();
}

// Detector chain on the map left by the last ref_amb_process
// (blah2.cpp:285-287).  stage: 0 = CFAR only, 1 = +Centroid, 2 = +Interpolate.
int64_t ref_detect(void *p, double pfa, int nGuard, int nTrain, int minDelay, double minDoppler,
                   int nCentroid, double centroidRes, int stage, double *delay, double *doppler,
                   double *snr, int64_t cap)
{
  auto *h = static_cast<RefAmb *>(p);
  if (!h->map) return -1;
  CfarDetector1D cfar(pfa, (int8_t)nGuard, (int8_t)nTrain, (int8_t)minDelay, minDoppler);
  std::unique_ptr<Detection> d = cfar.process(h->map);
  if (stage >= 1) {
    // blah2.cpp:183: Centroid(nCentroid, nCentroid, 1/tCpi) -- the *configured* CPI
    Centroid cen((uint16_t)nCentroid, (uint16_t)nCentroid, centroidRes);
    d = cen.process(d.get());
  }
  if (stage >= 2) {
    Interpolate itp(true, true);
    d = itp.process(d.get(), h->map);
  }
  h->det = std::move(d);
  return copy_det(h->det.get(), delay, doppler, snr, cap);
}

// WienerHopf::process (blah2.cpp:166,270).  y_out gets the filtered
// surveillance channel.  Returns 1 on success, 0 when the reference reports
// failure (CPI skipped), and the wall seconds in *secs.
int ref_wiener_process(int32_t dMin, int32_t dMax, uint32_t n, const double *x, const double *y,
                       double *y_out, double *secs)
{
  WienerHopf wh(dMin, dMax, n);
  IqData qx(n), qy(n);
  fill(qx, x, n);
  fill(qy, y, n);
  const auto t0 = std::chrono::steady_clock::now();
  const bool ok = wh.process(&qx, &qy);
  const auto t1 = std::chrono::steady_clock::now();
  if (secs) *secs = std::chrono::duration<double>(t1 - t0).count();
  if (ok && y_out) {
    auto d = qy.get_data();
    for (uint32_t i = 0; i < n && i < d.size(); i++) { y_out[2 * i] = d[i].real(); y_out[2 * i + 1] = d[i].imag(); }
  }
  return ok ? 1 : 0;
}

// SpectrumAnalyser(n, bandwidth).process(x): writes up to cap spectrum values
// (re, im) and returns their count; *n_frequency = length of the frequency axis
// the reference produced.
int ref_spectrum_process(uint32_t n, double bandwidth, const double *x, double *spectrum_out, uint32_t cap,
                         uint32_t *n_frequency)
{
  SpectrumAnalyser sa(n, bandwidth);
  IqData qx(n);
  fill(qx, x, n);
  sa.process(&qx);
  const uint32_t m = (uint32_t)g_ref_last_spectrum.size();
  for (uint32_t i = 0; i < m && i < cap; i++) {
    spectrum_out[2 * i] = g_ref_last_spectrum[i].real();
    spectrum_out[2 * i + 1] = g_ref_last_spectrum[i].imag();
  }
  if (n_frequency) *n_frequency = (uint32_t)g_ref_last_frequency.size();
  return (int)m;
}

} // extern "C"
