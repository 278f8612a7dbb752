#include "process/clutter/WienerHopf.h"

#include "blah2hip.h"
#include "process/ambiguity/Ambiguity.h"
#include "util/DeviceContext.h"

#include <iostream>
#include <stdexcept>
#include <string>

WienerHopf::WienerHopf(int32_t delayMin, int32_t delayMax, uint32_t _nSamples) : nSamples(_nSamples)
{
  if (blah2hip_clutter_create(delayMin, delayMax, _nSamples, Ambiguity::default_device(), 1, &engine) != BLAH2HIP_OK)
    throw std::runtime_error(std::string("WienerHopf: ") + blah2hip_last_error());
  try { // a constructor that throws runs no destructor: give back what has been acquired
    DeviceContext &dc = DeviceContext::get();
    dOk = (int32_t *)dc.alloc_device(sizeof(int32_t));
    hOk = (int32_t *)dc.alloc_pinned(sizeof(int32_t));
  } catch (...) {
    if (dOk) DeviceContext::get().free_device(dOk);
    blah2hip_clutter_destroy(engine);
    throw;
  }
}

WienerHopf::~WienerHopf()
{
  DeviceContext &dc = DeviceContext::get();
  dc.free_device(dOk);
  dc.free_pinned(hOk);
  blah2hip_clutter_destroy(engine);
}

bool WienerHopf::process(IqData *x, IqData *y)
{
  // the reference copies both deques (WienerHopf.cpp:61-62) and indexes the first nSamples entries; x is left untouched.
  // Both channels are (made) resident on the device -- uploaded once per CPI, shared with SpectrumAnalyser and Ambiguity.
  if (x->get_length() < nSamples || y->get_length() < nSamples)
    throw std::runtime_error("WienerHopf::process: fewer samples than nSamples in the buffers");
  DeviceContext &dc = DeviceContext::get();
  const void *dx = dc.resident(x, nSamples);
  const void *dy = dc.resident(y, nSamples);
  void *dyf = dc.front_buffer(y, nSamples);
  if (blah2hip_clutter_process_dev_fmt(engine, BLAH2HIP_FMT_C32, dx, dy, 1, nSamples, dyf, nSamples, dOk, dc.stream()) != BLAH2HIP_OK)
    throw std::runtime_error(std::string("WienerHopf::process: ") + blah2hip_last_error());
  dc.d2h(hOk, dOk, sizeof(int32_t));
  dc.sync();
  if (!*hOk) {
    std::cerr << "Chol decomposition failed, skip clutter filter" << std::endl; // WienerHopf.cpp:114
    return false;
  }
  // WienerHopf.cpp:156-160 clears y and pushes the nSamples filtered values: y then holds exactly those.  They stay in
  // HBM as the new front of y (Ambiguity consumes them there); the host copy is written only if somebody reads y first.
  y->keep_front(nSamples);
  dc.adopt_front(y, nSamples);
  return true;
}
