// Physical constants used by the delay-bin -> km conversion
// (reference: src/data/meta/Constants.h:13, c stored as uint32_t).
#ifndef BLAH2HIP_HOST_CONSTANTS_H
#define BLAH2HIP_HOST_CONSTANTS_H
#include <stdint.h>
namespace Constants
{
const uint32_t c = 299792458;
}
#endif
