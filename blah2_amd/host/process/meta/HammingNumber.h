// next_hamming(): FFT-length helper of the reference
// (src/process/meta/HammingNumber.h / .cpp:38-48).  The reference walks a
// generator of 5-smooth numbers; only the result is part of the contract
// (TestHammingNumber.cpp:15-17), so it is computed directly.
#ifndef BLAH2HIP_HOST_HAMMINGNUMBER_H
#define BLAH2HIP_HOST_HAMMINGNUMBER_H
#include <stdint.h>
/// smallest integer of the form 2^a 3^b 5^c that is STRICTLY greater than value
uint32_t next_hamming(uint32_t value);
#endif
