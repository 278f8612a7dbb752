#include "process/detection/CfarDetector1D.h"

#include "blah2hip.h"

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

CfarDetector1D::CfarDetector1D(double _pfa, int8_t _nGuard, int8_t _nTrain, int8_t _minDelay, double _minDoppler)
    : pfa(_pfa), nGuard(_nGuard), nTrain(_nTrain), minDelay(_minDelay), minDoppler(_minDoppler) {}

CfarDetector1D::~CfarDetector1D() {}

std::unique_ptr<Detection> CfarDetector1D::process(Map<std::complex<double>> *x)
{
  const uint32_t nRows = x->get_nRows(), nCols = x->get_nCols();
  const uint32_t cap = nRows * nCols;
  std::vector<double> delay(cap), doppler(cap), snr(cap);
  uint32_t n = 0;
  int rc;
  if (blah2hip_amb_s *engine = x->get_engine()) {
    // the engine still holds this very map on the device: no upload
    rc = blah2hip_cfar1d_process(engine, x->get_engine_cpi(), pfa, nGuard, nTrain, minDelay, minDoppler,
                                 delay.data(), doppler.data(), snr.data(), cap, &n);
  } else {
    // any other Map (built or modified by the caller): upload its cells and run the same GPU kernel
    std::vector<float> cells(2 * (size_t)cap);
    for (uint32_t i = 0; i < nRows; i++)
      for (uint32_t j = 0; j < nCols; j++) {
        cells[2 * ((size_t)i * nCols + j)] = (float)x->data[i][j].real();
        cells[2 * ((size_t)i * nCols + j) + 1] = (float)x->data[i][j].imag();
      }
    std::vector<int32_t> dax(x->delay.begin(), x->delay.end());
    std::vector<double> fax(x->doppler.begin(), x->doppler.end());
    if (dax.size() != nCols || fax.size() != nRows)
      throw std::runtime_error("CfarDetector1D::process: Map::delay / Map::doppler do not match the map size");
    const char *e = std::getenv("BLAH2HIP_DEVICE");
    rc = blah2hip_cfar1d_map(cells.data(), nRows, nCols, dax.data(), fax.data(), x->noisePower, pfa, nGuard, nTrain,
                             minDelay, minDoppler, e ? std::atoi(e) : 0, delay.data(), doppler.data(), snr.data(), cap, &n);
  }
  if (rc != BLAH2HIP_OK)
    throw std::runtime_error(std::string("CfarDetector1D::process: ") + blah2hip_last_error());
  delay.resize(n);
  doppler.resize(n);
  snr.resize(n);
  return std::make_unique<Detection>(delay, doppler, snr);
}
