#include "process/ambiguity/Ambiguity.h"

#include "blah2hip.h"

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

  const size_t used = (size_t)nCorr * nDopplerBins;
  bufX.resize(2 * used);
  bufY.resize(2 * used);
  mapF.resize(2 * (size_t)nDopplerBins * nDelayBins);
}

Ambiguity::~Ambiguity() { blah2hip_amb_destroy(engine); }

Map<std::complex<double>> *Ambiguity::process(IqData *x, IqData *y)
{
  const uint32_t used = (uint32_t)nCorr * nDopplerBins;
  // Ambiguity.cpp:95-102 rotates EVERY sample of x about the Doppler centre
  // before the range loop; the GPU applies the same rotation to the samples it
  // consumes, the ones left in the FIFO are rotated here so that x ends up in
  // the state the reference leaves it in.
  const uint32_t total = x->get_length();
  // Ambiguity.cpp:105-112: pops nCorr samples of each channel per pulse
  nSamples = used;
  x->pop_front_block(bufX.data(), used); // throws "Attempting to pop from an empty deque"
  y->pop_front_block(bufY.data(), used);
  if (dopplerMiddle != 0 && total > used) {
    const std::complex<double> j(0, 1);
    for (uint32_t i = used; i < total; i++)
      x->push_back(x->pop_front() * std::exp(1.0 * j * 2.0 * M_PI * dopplerMiddle * ((double)i / fs)));
  }
  double metrics[2] = {0, 0};
  if (blah2hip_amb_process_c64(engine, bufX.data(), bufY.data(), used, mapF.data(), metrics) != BLAH2HIP_OK)
    throw std::runtime_error(std::string("Ambiguity::process: ") + blah2hip_last_error());
  for (uint32_t i = 0; i < nDopplerBins; i++) {
    const float *row = mapF.data() + 2 * (size_t)i * nDelayBins;
    std::vector<Complex> &dst = map->data[i];
    for (uint32_t k = 0; k < nDelayBins; k++) dst[k] = Complex(row[2 * k], row[2 * k + 1]);
  }
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
