#include "process/spectrum/SpectrumAnalyser.h"

#include "blah2hip.h"
#include "process/ambiguity/Ambiguity.h"
#include "util/DeviceContext.h"

#include <stdexcept>
#include <string>

SpectrumAnalyser::SpectrumAnalyser(uint32_t _n, double _bandwidth) : n(_n), bandwidth(_bandwidth)
{
  if (blah2hip_spectrum_create(n, bandwidth, Ambiguity::default_device(), 1, &engine) != BLAH2HIP_OK)
    throw std::runtime_error(std::string("SpectrumAnalyser: ") + blah2hip_last_error());
  blah2hip_spectrum_get_dims(engine, &decimation, &nSpectrum, &nfft);
  try { // a constructor that throws runs no destructor: give back what has been acquired
    DeviceContext &dc = DeviceContext::get();
    dSpec = (double *)dc.alloc_device(2 * (size_t)nSpectrum * sizeof(double));
    hSpec = (double *)dc.alloc_pinned(2 * (size_t)nSpectrum * sizeof(double));
  } catch (...) {
    if (dSpec) DeviceContext::get().free_device(dSpec);
    blah2hip_spectrum_destroy(engine);
    throw;
  }
}

SpectrumAnalyser::~SpectrumAnalyser()
{
  DeviceContext &dc = DeviceContext::get();
  dc.free_device(dSpec);
  dc.free_pinned(hSpec);
  blah2hip_spectrum_destroy(engine);
}

void SpectrumAnalyser::process(IqData *x)
{
  // SpectrumAnalyser.cpp:33-37 reads the first nfft entries of a copy of the FIFO: here the channel is (made) resident on
  // the device once per CPI and shared with WienerHopf and Ambiguity (util/DeviceContext.h)
  if (x->get_length() < nfft) throw std::runtime_error("SpectrumAnalyser::process: fewer samples than nfft in the buffer");
  DeviceContext &dc = DeviceContext::get();
  const void *dx = dc.resident(x, (uint32_t)nfft);
  if (blah2hip_spectrum_process_dev(engine, BLAH2HIP_FMT_C32, dx, 1, nfft, dSpec, dc.stream()) != BLAH2HIP_OK)
    throw std::runtime_error(std::string("SpectrumAnalyser::process: ") + blah2hip_last_error());
  dc.d2h(hSpec, dSpec, 2 * (size_t)nSpectrum * sizeof(double));
  dc.sync();
  const double *bufS = hSpec;
  std::vector<std::complex<double>> spectrum(nSpectrum);
  for (uint32_t k = 0; k < nSpectrum; k++) spectrum[k] = {bufS[2 * k], bufS[2 * k + 1]};
  x->update_spectrum(spectrum);

  // SpectrumAnalyser.cpp:57-68.  The loop counter is a uint32_t, so the start value
  // -nSpectrum/2 is (2^32 - nSpectrum)/2 and the body never runs: the frequency axis
  // the reference publishes is empty.  Same arithmetic here, on purpose.
  std::vector<double> frequency;
  double offset = 0;
  if (decimation % 2 == 0) offset = bandwidth / 2;
  for (uint32_t i = -nSpectrum / 2; i < nSpectrum / 2; i++)
    frequency.push_back(((i * bandwidth) + offset + 204640000) / 1000);
  x->update_frequency(frequency);
}
