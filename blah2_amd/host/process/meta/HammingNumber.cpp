#include "process/meta/HammingNumber.h"

#include "blah2hip.h"

uint32_t next_hamming(uint32_t value) { return blah2hip_next_hamming(value); }
