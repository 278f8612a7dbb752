/* blah2host.h -- C ABI over the HOST-side product classes of blah2_amd/host (the C++ classes with
 * the reference's own surface: Map, Detection).  It exists so that non-C++ callers -- the Python
 * replay driver and the tests -- serialise through the SAME C++ code that drops into blah2.cpp,
 * not through a re-implementation.  Nothing here touches the GPU.
 *
 *   Map<T>::to_json + delay_bin_to_km      src/data/Map.cpp:115-185        (called blah2.cpp:304-305)
 *   Detection::to_json + delay_bin_to_km   src/data/Detection.cpp:47-106   (called blah2.cpp:315-316)
 *   rapidjson Writer::Double with SetMaxDecimalPlaces(2)  (Map.cpp:157-160)  -> blah2host_format_double
 *
 * Strings are written into caller-owned buffers; *len receives the length needed (without the
 * terminating NUL); the return value is 0 on success, -6 (capacity) when `cap` is too small
 * (nothing usable is written then), -1 on a bad argument, -2 when the host classes threw (out of memory):
 * no exception crosses this boundary.
 */
#ifndef BLAH2HOST_H
#define BLAH2HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* map: complex fp32 cells [n_doppler][n_delay] as blah2hip_amb_* deliver them; delay (bins) and
 * doppler (Hz) are Map::delay / Map::doppler.  fs > 0: the document's "delay" array is rewritten
 * in km like blah2.cpp:305 does (Map::delay_bin_to_km); fs = 0: bins. */
int blah2host_map_json(const float *map, uint32_t n_doppler, uint32_t n_delay, const int32_t *delay,
                       const double *doppler, double noise_power, double max_power, uint64_t timestamp,
                       uint32_t fs, char *out, size_t cap, size_t *len);
int blah2host_detection_json(const double *delay, const double *doppler, const double *snr, uint32_t count,
                             uint64_t timestamp, uint32_t fs, char *out, size_t cap, size_t *len);
/* one double as the reference's JSON writer prints it (max_decimals = 2 in every reference document) */
int blah2host_format_double(double v, int max_decimals, char *out, size_t cap, size_t *len);

#ifdef __cplusplus
}
#endif
#endif /* BLAH2HOST_H */
