#include "process/detection/Centroid.h"

#include "blah2hip.h"

#include <stdexcept>
#include <vector>

Centroid::Centroid(uint16_t _nDelay, uint16_t _nDoppler, double _resolutionDoppler)
    : nDelay(_nDelay), nDoppler(_nDoppler), resolutionDoppler(_resolutionDoppler) {}
Centroid::~Centroid() {}

std::unique_ptr<Detection> Centroid::process(Detection *x)
{
  const std::vector<double> d = x->get_delay(), f = x->get_doppler(), s = x->get_snr();
  const uint32_t n = (uint32_t)s.size();
  std::vector<double> od(n), of(n), os(n);
  uint32_t k = 0;
  if (blah2hip_centroid(d.data(), f.data(), s.data(), n, nDelay, nDoppler, resolutionDoppler, od.data(),
                        of.data(), os.data(), &k) != BLAH2HIP_OK)
    throw std::runtime_error(blah2hip_last_error());
  od.resize(k); of.resize(k); os.resize(k);
  return std::make_unique<Detection>(od, of, os);
}
