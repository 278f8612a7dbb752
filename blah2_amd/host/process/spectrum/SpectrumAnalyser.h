// SpectrumAnalyser: decimated spectrum of the reference channel for the web
// front-end (reference surface: src/process/spectrum/SpectrumAnalyser.h:53-62,
// constructed blah2.cpp:199, called blah2.cpp:264).  process() reads the first
// nfft samples of x without consuming them and stores nSpectrum complex bins and
// the frequency axis in x (IqData::update_spectrum / update_frequency).
#ifndef BLAH2HIP_HOST_SPECTRUMANALYSER_H
#define BLAH2HIP_HOST_SPECTRUMANALYSER_H

#include "data/IqData.h"

#include <stdint.h>
#include <vector>

struct blah2hip_spectrum_s;

class SpectrumAnalyser
{
public:
  SpectrumAnalyser(uint32_t n, double bandwidth);
  ~SpectrumAnalyser();
  SpectrumAnalyser(const SpectrumAnalyser &) = delete;
  SpectrumAnalyser &operator=(const SpectrumAnalyser &) = delete;

  void process(IqData *x);

private:
  blah2hip_spectrum_s *engine = nullptr;
  uint32_t n;
  double bandwidth;
  uint32_t decimation = 0;
  uint32_t nSpectrum = 0;
  uint64_t nfft = 0;
  double *dSpec = nullptr; // device: nSpectrum complex fp64
  double *hSpec = nullptr; // pinned host copy
};

#endif
