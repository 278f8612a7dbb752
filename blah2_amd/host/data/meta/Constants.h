// Physical constants of the host-side data classes.
//
// Only the speed of light is needed on this path: Map::delay_bin_to_km and
// Detection::delay_bin_to_km turn a delay bin into a bistatic range in km as
// bin * c / fs / 1000 (reference: src/data/Map.cpp:170, src/data/Detection.cpp:66).
// The reference keeps c in an unsigned 32-bit integer (src/data/meta/Constants.h:13),
// so `Constants::c / (double)fs` is evaluated from that integer value; the same type
// is used here so that the conversion rounds identically.
#ifndef BLAH2HIP_HOST_CONSTANTS_H
#define BLAH2HIP_HOST_CONSTANTS_H

#include <stdint.h>

namespace Constants
{
const uint32_t c = 299792458; // m/s, exact by definition of the metre
}

#endif
