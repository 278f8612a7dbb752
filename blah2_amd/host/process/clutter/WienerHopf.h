// WienerHopf: least-squares FIR clutter canceller (reference surface:
// src/process/clutter/WienerHopf.h:68-78).  process() reads x, replaces the
// contents of y with the filtered surveillance channel and returns true; when
// the normal equations are not positive definite it returns false and leaves
// y alone (the caller then skips the CPI, blah2.cpp:270-273).
#ifndef BLAH2HIP_HOST_WIENERHOPF_H
#define BLAH2HIP_HOST_WIENERHOPF_H

#include "data/IqData.h"

#include <stdint.h>
#include <vector>

struct blah2hip_clutter_s;

class WienerHopf
{
public:
  WienerHopf(int32_t delayMin, int32_t delayMax, uint32_t nSamples);
  ~WienerHopf();
  WienerHopf(const WienerHopf &) = delete;
  WienerHopf &operator=(const WienerHopf &) = delete;

  bool process(IqData *x, IqData *y);

private:
  blah2hip_clutter_s *engine = nullptr;
  uint32_t nSamples;
  int32_t *dOk = nullptr, *hOk = nullptr; // device flag and its pinned host copy
};

#endif
