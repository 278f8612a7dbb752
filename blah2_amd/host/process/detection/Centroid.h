// Centroid: non-maximum suppression over the CFAR detection list
// (reference surface: src/process/detection/Centroid.h).
#ifndef BLAH2HIP_HOST_CENTROID_H
#define BLAH2HIP_HOST_CENTROID_H

#include "data/Detection.h"

#include <memory>
#include <stdint.h>

class Centroid
{
public:
  Centroid(uint16_t nDelay, uint16_t nDoppler, double resolutionDoppler);
  ~Centroid();
  std::unique_ptr<Detection> process(Detection *x);

private:
  uint16_t nDelay, nDoppler;
  double resolutionDoppler;
};

#endif
