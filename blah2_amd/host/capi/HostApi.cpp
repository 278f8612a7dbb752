// C ABI of include/blah2host.h: thin marshalling onto the host classes.
#include "blah2host.h"

#include "data/Detection.h"
#include "data/Map.h"
#include "util/JsonOut.h"

#include <complex>
#include <cstring>
#include <string>
#include <vector>

namespace {

int emit(const std::string &s, char *out, size_t cap, size_t *len)
{
  if (len) *len = s.size();
  if (!out || cap < s.size() + 1) return -6;
  std::memcpy(out, s.data(), s.size());
  out[s.size()] = '\0';
  return 0;
}

} // namespace

extern "C" {

int blah2host_map_json(const float *map, uint32_t n_doppler, uint32_t n_delay, const int32_t *delay,
                       const double *doppler, double noise_power, double max_power, uint64_t timestamp,
                       uint32_t fs, char *out, size_t cap, size_t *len)
{
  if (!map || !delay || !doppler || n_doppler == 0 || n_delay == 0) return -1;
  try {
  Map<std::complex<double>> m(n_doppler, n_delay);
  for (uint32_t i = 0; i < n_doppler; i++) {
    const float *row = map + 2 * (size_t)i * n_delay;
    for (uint32_t j = 0; j < n_delay; j++) m.data[i][j] = std::complex<double>(row[2 * j], row[2 * j + 1]);
  }
  m.delay.assign(delay, delay + n_delay);
  m.doppler.assign(doppler, doppler + n_doppler);
  m.noisePower = noise_power; // what Map::set_metrics left there (blah2.cpp:279)
  m.maxPower = max_power;
  std::string json = m.to_json(timestamp);
  if (fs) json = m.delay_bin_to_km(json, fs);
  return emit(json, out, cap, len);
  } catch (...) { return -2; } // nothing may unwind through the C ABI (std::bad_alloc, ...)
}

int blah2host_detection_json(const double *delay, const double *doppler, const double *snr, uint32_t count,
                             uint64_t timestamp, uint32_t fs, char *out, size_t cap, size_t *len)
{
  if (count && (!delay || !doppler || !snr)) return -1;
  try {
  Detection d(std::vector<double>(delay, delay + count), std::vector<double>(doppler, doppler + count),
              std::vector<double>(snr, snr + count));
  std::string json = d.to_json(timestamp);
  if (fs) json = d.delay_bin_to_km(json, fs);
  return emit(json, out, cap, len);
  } catch (...) { return -2; }
}

int blah2host_format_double(double v, int max_decimals, char *out, size_t cap, size_t *len)
{
  try {
  std::string s;
  blah2json::write_double(s, v, max_decimals);
  return emit(s, out, cap, len);
  } catch (...) { return -2; }
}

} // extern "C"
