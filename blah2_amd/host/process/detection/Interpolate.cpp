#include "process/detection/Interpolate.h"

#include "blah2hip.h"

#include <stdexcept>
#include <vector>

Interpolate::Interpolate(bool _doDelay, bool _doDoppler) : doDelay(_doDelay), doDoppler(_doDoppler) {}
Interpolate::~Interpolate() {}

std::unique_ptr<Detection> Interpolate::process(Detection *x, Map<std::complex<double>> *y)
{
  const std::vector<double> d = x->get_delay(), f = x->get_doppler(), s = x->get_snr();
  const uint32_t n = (uint32_t)s.size();
  const uint32_t nR = y->get_nRows(), nC = y->get_nCols();
  // the C ABI takes the map as it lives on the device: complex fp32, row-major
  std::vector<float> m(2 * (size_t)nR * nC);
  for (uint32_t i = 0; i < nR; i++)
    for (uint32_t j = 0; j < nC; j++) {
      m[2 * ((size_t)i * nC + j)] = (float)y->data[i][j].real();
      m[2 * ((size_t)i * nC + j) + 1] = (float)y->data[i][j].imag();
    }
  std::vector<int32_t> dax(y->delay.begin(), y->delay.end());
  std::vector<double> fax(y->doppler.begin(), y->doppler.end());
  std::vector<double> od(n), of(n), os(n);
  uint32_t k = 0;
  if (blah2hip_interpolate(d.data(), f.data(), s.data(), n, m.data(), nR, nC, dax.data(), fax.data(),
                           y->noisePower, doDelay, doDoppler, od.data(), of.data(), os.data(), &k) != BLAH2HIP_OK)
    throw std::runtime_error(blah2hip_last_error());
  od.resize(k); of.resize(k); os.resize(k);
  return std::make_unique<Detection>(od, of, os);
}
