// WienerHopf clutter filter entry points (include/blah2hip.h).
// PLACEHOLDER for the first GPU bring-up of the ambiguity chain: the real
// implementation (segmented FFT correlation -> Toeplitz solve -> overlap-save
// FIR) replaces this file; until then every call reports UNSUPPORTED loudly.
#include "blah2hip.h"

#include <string>

extern "C" {
int blah2hip_clutter_create(int32_t, int32_t, uint32_t, int, uint32_t, blah2hip_clutter_t *out)
{
  if (out) *out = nullptr;
  return BLAH2HIP_ERR_UNSUPPORTED;
}
int blah2hip_clutter_destroy(blah2hip_clutter_t) { return BLAH2HIP_OK; }
int blah2hip_clutter_process_c64(blah2hip_clutter_t, const double *, const double *, uint32_t, double *, int *) { return BLAH2HIP_ERR_UNSUPPORTED; }
int blah2hip_clutter_process_c32(blah2hip_clutter_t, const float *, const float *, uint32_t, float *, int *) { return BLAH2HIP_ERR_UNSUPPORTED; }
int blah2hip_clutter_process_dev(blah2hip_clutter_t, const void *, const void *, uint32_t, uint64_t, void *, int32_t *, void *) { return BLAH2HIP_ERR_UNSUPPORTED; }
}
