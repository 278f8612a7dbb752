#include "process/detection/CfarDetector1D.h"

#include "blah2hip.h"

#include <stdexcept>
#include <string>
#include <vector>

CfarDetector1D::CfarDetector1D(double _pfa, int8_t _nGuard, int8_t _nTrain, int8_t _minDelay, double _minDoppler)
    : pfa(_pfa), nGuard(_nGuard), nTrain(_nTrain), minDelay(_minDelay), minDoppler(_minDoppler) {}

CfarDetector1D::~CfarDetector1D() {}

std::unique_ptr<Detection> CfarDetector1D::process(Map<std::complex<double>> *x)
{
  blah2hip_amb_s *engine = x->get_engine();
  if (!engine)
    throw std::runtime_error("CfarDetector1D::process: the map does not come from the GPU Ambiguity engine");
  const uint32_t cap = x->get_nRows() * x->get_nCols();
  std::vector<double> delay(cap), doppler(cap), snr(cap);
  uint32_t n = 0;
  if (blah2hip_cfar1d_process(engine, x->get_engine_cpi(), pfa, nGuard, nTrain, minDelay, minDoppler,
                              delay.data(), doppler.data(), snr.data(), cap, &n) != BLAH2HIP_OK)
    throw std::runtime_error(std::string("CfarDetector1D::process: ") + blah2hip_last_error());
  delay.resize(n);
  doppler.resize(n);
  snr.resize(n);
  return std::make_unique<Detection>(delay, doppler, snr);
}
