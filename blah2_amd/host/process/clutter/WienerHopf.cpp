#include "process/clutter/WienerHopf.h"

#include "blah2hip.h"
#include "process/ambiguity/Ambiguity.h"

#include <iostream>
#include <stdexcept>
#include <string>

WienerHopf::WienerHopf(int32_t delayMin, int32_t delayMax, uint32_t _nSamples) : nSamples(_nSamples)
{
  if (blah2hip_clutter_create(delayMin, delayMax, _nSamples, Ambiguity::default_device(), 1, &engine) != BLAH2HIP_OK)
    throw std::runtime_error(std::string("WienerHopf: ") + blah2hip_last_error());
  bufX.resize(2 * (size_t)nSamples);
  bufY.resize(2 * (size_t)nSamples);
  bufOut.resize(2 * (size_t)nSamples);
}

WienerHopf::~WienerHopf() { blah2hip_clutter_destroy(engine); }

bool WienerHopf::process(IqData *x, IqData *y)
{
  // the reference copies both deques (WienerHopf.cpp:61-62) and indexes the
  // first nSamples entries; x is left untouched
  const std::deque<std::complex<double>> xd = x->get_data(), yd = y->get_data();
  if (xd.size() < nSamples || yd.size() < nSamples)
    throw std::runtime_error("WienerHopf::process: fewer samples than nSamples in the buffers");
  for (uint32_t i = 0; i < nSamples; i++) {
    bufX[2 * i] = xd[i].real(); bufX[2 * i + 1] = xd[i].imag();
    bufY[2 * i] = yd[i].real(); bufY[2 * i + 1] = yd[i].imag();
  }
  int ok = 0;
  if (blah2hip_clutter_process_c64(engine, bufX.data(), bufY.data(), nSamples, bufOut.data(), &ok) != BLAH2HIP_OK)
    throw std::runtime_error(std::string("WienerHopf::process: ") + blah2hip_last_error());
  if (!ok) {
    std::cerr << "Chol decomposition failed, skip clutter filter" << std::endl; // WienerHopf.cpp:114
    return false;
  }
  // WienerHopf.cpp:156-160
  y->clear();
  for (uint32_t i = 0; i < nSamples; i++) y->push_back({bufOut[2 * i], bufOut[2 * i + 1]});
  return true;
}
