// CfarDetector1D: 1-D cell-averaging CFAR along delay, per Doppler row
// (reference surface: src/process/detection/CfarDetector1D.h:46-55).  Runs on
// the GPU copy of the map that Ambiguity::process left on the device.
#ifndef BLAH2HIP_HOST_CFARDETECTOR1D_H
#define BLAH2HIP_HOST_CFARDETECTOR1D_H

#include "data/Detection.h"
#include "data/Map.h"

#include <complex>
#include <memory>
#include <stdint.h>

class CfarDetector1D
{
public:
  CfarDetector1D(double pfa, int8_t nGuard, int8_t nTrain, int8_t minDelay, double minDoppler);
  ~CfarDetector1D();
  std::unique_ptr<Detection> process(Map<std::complex<double>> *x);

private:
  double pfa;
  int8_t nGuard, nTrain, minDelay;
  double minDoppler;
};

#endif
