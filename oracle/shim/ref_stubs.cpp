// TEST INFRASTRUCTURE ONLY -- part of oracle/ (see oracle/README.md).
//
// The reference's IqData.cpp / Map.cpp / Detection.cpp cannot be compiled in
// this image because they #include rapidjson (absent) for their to_json/save
// members.  Their *headers* are rapidjson-free and are used as they are from
// /root/reference/src; this file supplies the numeric (non-JSON) members the
// hot path touches, restated from:
//   IqData     /root/reference/src/data/IqData.cpp:11-81
//   Map<T>     /root/reference/src/data/Map.cpp:13-113 and :187-206
//   Detection  /root/reference/src/data/Detection.cpp:13-45
// The JSON members are deliberately left undefined (nothing in oracle/_ref
// calls them).

#include "data/Detection.h"
#include "data/IqData.h"
#include "data/Map.h"

#include <cmath>
#include <stdexcept>

// ---------------------------------------------------------------- IqData --
// bounded FIFO: push_back evicts the oldest sample when full (IqData.cpp:42-53)
IqData::IqData(uint32_t cap) : n(cap), data(new std::deque<std::complex<double>>) {}
uint32_t IqData::get_n() { return n; }
uint32_t IqData::get_length() { return static_cast<uint32_t>(data->size()); }
void IqData::lock() { mutex_lock.lock(); }
void IqData::unlock() { mutex_lock.unlock(); }
std::deque<std::complex<double>> IqData::get_data() { return *data; }

void IqData::push_back(std::complex<double> s)
{
  if (data->size() >= n) data->pop_front();
  data->push_back(s);
}

std::complex<double> IqData::pop_front()
{
  // IqData.cpp:57-59: underflow is a runtime_error
  if (data->empty()) throw std::runtime_error("Attempting to pop from an empty deque");
  const std::complex<double> s = data->front();
  data->pop_front();
  return s;
}

void IqData::clear() { data->clear(); }

// IqData.cpp:83-91.  The members are private and only reachable through to_json
// (rapidjson), so the last values handed over are also kept where ref_capi.cpp can
// read them back.
std::vector<std::complex<double>> g_ref_last_spectrum;
std::vector<double> g_ref_last_frequency;
void IqData::update_spectrum(std::vector<std::complex<double>> s) { spectrum = s; g_ref_last_spectrum = s; }
void IqData::update_frequency(std::vector<double> f) { frequency = f; g_ref_last_frequency = f; }

// ---------------------------------------------------------------- Map<T> --
// rows = Doppler, cols = delay; cells start at 1 (Map.cpp:18)
template <class T>
Map<T>::Map(uint32_t r, uint32_t c)
    : nRows(r), nCols(c), data(r, std::vector<T>(c, T(1))), noisePower(0), maxPower(0) {}

template <class T> void Map<T>::set_row(uint32_t i, std::vector<T> row)
{
  for (uint32_t j = 0; j < nCols; j++) data[i][j] = row[j];
}
template <class T> void Map<T>::set_col(uint32_t i, std::vector<T> col)
{
  for (uint32_t j = 0; j < nRows; j++) data[j][i] = col[j];
}
template <class T> uint32_t Map<T>::get_nRows() { return nRows; }
template <class T> uint32_t Map<T>::get_nCols() { return nCols; }
template <class T> std::vector<T> Map<T>::get_row(uint32_t r) { return data[r]; }
template <class T> std::vector<T> Map<T>::get_col(uint32_t c)
{
  std::vector<T> out;
  out.reserve(nRows);
  for (uint32_t i = 0; i < nRows; i++) out.push_back(data[i][c]);
  return out;
}

// exact-equality search, 0 on miss (Map.cpp:102-113)
template <class T> uint32_t Map<T>::doppler_hz_to_bin(double hz)
{
  for (size_t i = 0; i < doppler.size(); i++)
    if (doppler[i] == hz) return static_cast<uint32_t>(i);
  return 0;
}

// Map.cpp:187-206: v = 10*log10|z| (amplitude, not power); noise = mean(v);
// the running max starts at 0; maxPower = max - noise.  Accumulation order
// is row-major, as in the reference, so the fp64 sum is bit-identical.
template <class T> void Map<T>::set_metrics()
{
  double sum = 0.0, peak = 0.0;
  for (uint32_t i = 0; i < nRows; i++)
    for (uint32_t j = 0; j < nCols; j++) {
      const double v = 10 * std::log10(std::abs(data[i][j]));
      sum = sum + v;
      peak = (peak < v) ? v : peak;
    }
  const double mean = sum / (nRows * nCols);
  noisePower = mean;
  maxPower = peak - mean;
}

template class Map<std::complex<double>>;
template class Map<double>;

// ------------------------------------------------------------- Detection --
Detection::Detection(std::vector<double> d, std::vector<double> f, std::vector<double> s)
    : delay(std::move(d)), doppler(std::move(f)), snr(std::move(s)) {}
Detection::Detection(double d, double f, double s) : delay{d}, doppler{f}, snr{s} {}
std::vector<double> Detection::get_delay() { return delay; }
std::vector<double> Detection::get_doppler() { return doppler; }
std::vector<double> Detection::get_snr() { return snr; }
size_t Detection::get_nDetections() { return delay.size(); }
