// Interpolate: quadratic sub-bin refinement of detections in delay and Doppler
// (reference surface: src/process/detection/Interpolate.h).
#ifndef BLAH2HIP_HOST_INTERPOLATE_H
#define BLAH2HIP_HOST_INTERPOLATE_H

#include "data/Detection.h"
#include "data/Map.h"

#include <complex>
#include <memory>

class Interpolate
{
public:
  Interpolate(bool doDelay, bool doDoppler);
  ~Interpolate();
  std::unique_ptr<Detection> process(Detection *x, Map<std::complex<double>> *y);

private:
  bool doDelay, doDoppler;
};

#endif
