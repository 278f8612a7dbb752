#include "process/ambiguity/Ambiguity.h"

#include "blah2hip.h"
#include "util/DeviceContext.h"

#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>

int Ambiguity::default_device()
{
  const char *e = std::getenv("BLAH2HIP_DEVICE");
  return e ? std::atoi(e) : 0;
}

Ambiguity::Ambiguity(int32_t delayMin, int32_t delayMax, int32_t dopplerMin, int32_t dopplerMax,
                     uint32_t _fs, uint32_t n, bool roundHamming)
    : fs(_fs), nSamples(n)
{
  // the reference constructor cannot fail; here a missing GPU or an
  // unsupported geometry is a hard error (there is no CPU fallback)
  if (blah2hip_amb_create(delayMin, delayMax, dopplerMin, dopplerMax, _fs, n, roundHamming ? 1 : 0,
                          default_device(), 1, &engine) != BLAH2HIP_OK)
    throw std::runtime_error(std::string("Ambiguity: ") + blah2hip_last_error());
  blah2hip_amb_dims_t d;
  blah2hip_amb_get_dims(engine, &d);
  nDelayBins = (uint16_t)d.n_delay_bins;
  nDopplerBins = (uint16_t)d.n_doppler_bins;
  nCorr = (uint16_t)d.n_corr;
  dopplerMiddle = d.doppler_middle;
  cpi = d.cpi;
  nfft = d.nfft;

  map = std::make_unique<Map<Complex>>(nDopplerBins, nDelayBins);
  std::vector<int32_t> delay(nDelayBins);
  std::vector<double> doppler(nDopplerBins);
  blah2hip_amb_get_axes(engine, delay.data(), doppler.data());
  map->delay.assign(delay.begin(), delay.end());
  map->doppler.assign(doppler.begin(), doppler.end());

  try { // a constructor that throws runs no destructor: give back what has been acquired
    DeviceContext &dc = DeviceContext::get();
    mapF = (float *)dc.alloc_pinned(2 * (size_t)nDopplerBins * nDelayBins * sizeof(float));
    metF = (double *)dc.alloc_pinned(2 * sizeof(double));
  } catch (...) {
    if (mapF) DeviceContext::get().free_pinned(mapF);
    blah2hip_amb_destroy(engine);
    throw;
  }
}

Ambiguity::~Ambiguity()
{
  DeviceContext &dc = DeviceContext::get();
  dc.free_pinned(mapF);
  dc.free_pinned(metF);
  blah2hip_amb_destroy(engine);
}

Map<std::complex<double>> *Ambiguity::process(IqData *x, IqData *y)
{
  const uint32_t used = (uint32_t)nCorr * nDopplerBins;
  // Ambiguity.cpp:95-102 rotates EVERY sample of x about the Doppler centre
  // before the range loop; the GPU applies the same rotation to the samples it
  // consumes, the ones left in the FIFO are rotated here so that x ends up in
  // the state the reference leaves it in.
  const uint32_t total = x->get_length();
  // Ambiguity.cpp:105-112 pops nCorr samples of each channel per pulse.  The samples are consumed where they are
  // resident -- uploaded once per CPI and shared with SpectrumAnalyser / WienerHopf, or, for y behind the clutter
  // filter, already in HBM -- and the FIFOs drop them without reading (util/DeviceContext.h).
  nSamples = used;
  if (x->get_length() < used || y->get_length() < used) { // the reference's pops empty the FIFOs, then throw
    x->drop_front(used);
    y->drop_front(used);
  }
  DeviceContext &dc = DeviceContext::get();
  const void *dx = dc.resident(x, used);
  const void *dy = dc.resident(y, used);
  if (blah2hip_amb_process_dev(engine, BLAH2HIP_FMT_C32, dx, dy, 1, used, nullptr, nullptr, dc.stream()) != BLAH2HIP_OK)
    throw std::runtime_error(std::string("Ambiguity::process: ") + blah2hip_last_error());
  const void *dMap = nullptr;
  const double *dMet = nullptr;
  blah2hip_amb_result_ptrs(engine, &dMap, &dMet);
  dc.d2h(mapF, dMap, 2 * (size_t)nDopplerBins * nDelayBins * sizeof(float));
  dc.d2h(metF, dMet, 2 * sizeof(double));
  x->drop_front(used);
  dc.consumed(x, used);
  y->drop_front(used);
  dc.consumed(y, used);
  if (dopplerMiddle != 0 && total > used) {
    const std::complex<double> j(0, 1);
    for (uint32_t i = used; i < total; i++)
      x->push_back(x->pop_front() * std::exp(1.0 * j * 2.0 * M_PI * dopplerMiddle * ((double)i / fs)));
  }
  dc.sync();
  const double metrics[2] = {metF[0], metF[1]};
  const float *mf = mapF;
  const uint32_t nDel = nDelayBins;
  Map<Complex> *mp = map.get();
  dc.parallel_for(nDopplerBins, 32, [mf, nDel, mp](size_t r0, size_t r1) {
    for (size_t i = r0; i < r1; i++) {
      const float *row = mf + 2 * i * nDel;
      std::vector<Complex> &dst = mp->data[i];
      for (uint32_t k = 0; k < nDel; k++) dst[k] = Complex(row[2 * k], row[2 * k + 1]);
    }
  });
  map->bind_engine(engine, 0, metrics[0], metrics[1]);
  return map.get();
}

double Ambiguity::get_doppler_middle() const { return dopplerMiddle; }
uint16_t Ambiguity::get_n_delay_bins() const { return nDelayBins; }
uint16_t Ambiguity::get_n_doppler_bins() const { return nDopplerBins; }
uint16_t Ambiguity::get_n_corr() const { return nCorr; }
double Ambiguity::get_cpi() const { return cpi; }
uint32_t Ambiguity::get_nfft() const { return nfft; }
uint32_t Ambiguity::get_n_samples() const { return nSamples; }
