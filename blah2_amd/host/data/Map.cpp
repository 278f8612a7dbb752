#include "data/Map.h"

#include "data/meta/Constants.h"
#include "util/JsonOut.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>

// every cell starts at 1 like the reference (Map.cpp:18), so 10*log10|z| is 0
template <class T>
Map<T>::Map(uint32_t rows, uint32_t cols)
    : nRows(rows), nCols(cols), data(rows, std::vector<T>(cols, T(1))), noisePower(0.0), maxPower(0.0) {}

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
template <class T> std::vector<T> Map<T>::get_row(uint32_t row) { return data[row]; }

template <class T> std::vector<T> Map<T>::get_col(uint32_t col)
{
  std::vector<T> out(nRows);
  for (uint32_t i = 0; i < nRows; i++) out[i] = data[i][col];
  return out;
}

template <class T> Map<double> *Map<T>::get_map_db()
{
  Map<double> *db = new Map<double>(nRows, nCols);
  for (uint32_t i = 0; i < nRows; i++)
    for (uint32_t j = 0; j < nCols; j++) db->data[i][j] = 10.0 * std::log10(std::abs(data[i][j]));
  return db;
}

template <class T> void Map<T>::print()
{
  for (const auto &row : data) {
    for (const auto &v : row) std::cout << v << " ";
    std::cout << std::endl;
  }
}

// exact-equality search that returns 0 on a miss (reference Map.cpp:102-113)
template <class T> uint32_t Map<T>::doppler_hz_to_bin(double hz)
{
  for (size_t i = 0; i < doppler.size(); i++)
    if (doppler[i] == hz) return (uint32_t)i;
  return 0;
}

// A fingerprint of the cells (every cell's bit pattern): tells whether `data` is still what the engine delivered.
// Four independent multiply-xor chains per row (the single FNV chain it replaces was latency-bound: one dependent
// multiply per 8 bytes, 0.34 ms per pass over a 513 x 411 map), folded with the row index so that swapped rows differ.
template <class T> uint64_t Map<T>::fingerprint() const
{
  static_assert(sizeof(T) % sizeof(uint64_t) == 0, "cells are whole 64-bit words");
  uint64_t h = 1469598103934665603ull;
  uint64_t r = 0;
  for (const auto &row : data) {
    const unsigned char *p = reinterpret_cast<const unsigned char *>(row.data());
    const size_t nw = row.size() * sizeof(T) / sizeof(uint64_t);
    uint64_t a0 = 0x9e3779b97f4a7c15ull, a1 = 0xc2b2ae3d27d4eb4full, a2 = 0x165667b19e3779f9ull, a3 = 0x27d4eb2f165667c5ull;
    size_t i = 0;
    for (; i + 4 <= nw; i += 4) {
      uint64_t w[4];
      std::memcpy(w, p + i * sizeof(uint64_t), sizeof w); // no aliasing of complex<double> through uint64_t*
      a0 = (a0 ^ w[0]) * 0x100000001b3ull;
      a1 = (a1 ^ w[1]) * 0x100000001b3ull;
      a2 = (a2 ^ w[2]) * 0x100000001b3ull;
      a3 = (a3 ^ w[3]) * 0x100000001b3ull;
    }
    for (; i < nw; i++) {
      uint64_t w;
      std::memcpy(&w, p + i * sizeof(uint64_t), sizeof w);
      a0 = (a0 ^ w) * 0x100000001b3ull;
    }
    const uint64_t rowh = (a0 ^ (a1 >> 7)) + (a2 ^ (a3 << 9)) + (a1 * 31) + a3;
    h = (h ^ (rowh + ++r)) * 1099511628211ull;
  }
  return h;
}

// Map::set_metrics (reference Map.cpp:187-206): noisePower = mean(10 log10|z|),
// maxPower = max(0, max 10 log10|z|) - noisePower.  For a map that came out of
// the GPU engine AND whose cells are untouched since, the reduction was fused into
// the Doppler kernel and the values are adopted from there; any other map
// (built by the caller, or modified after Ambiguity::process) is reduced here in
// fp64 exactly like the reference.
template <class T> void Map<T>::set_metrics()
{
  if (engineMetricsValid && fingerprint() == engineFingerprint) {
    noisePower = engineNoise;
    maxPower = engineMax;
    return;
  }
  engineMetricsValid = false;
  double sum = 0.0, peak = 0.0;
  for (uint32_t i = 0; i < nRows; i++)
    for (uint32_t j = 0; j < nCols; j++) {
      const double v = 10.0 * std::log10(std::abs(data[i][j]));
      sum += v;
      if (peak < v) peak = v;
    }
  noisePower = sum / ((double)nRows * nCols);
  maxPower = peak - noisePower;
}

// Field order of the reference document (Map.cpp:148-155):
// timestamp,nRows,nCols,noisePower,maxPower,delay[],doppler[],data[][]
template <class T> std::string Map<T>::to_json(uint64_t timestamp)
{
  blah2json::Writer w(2);
  w.begin_object();
  w.key("timestamp"); w.value(timestamp);
  w.key("nRows"); w.value(nRows);
  w.key("nCols"); w.value(nCols);
  w.key("noisePower"); w.value(noisePower);
  w.key("maxPower"); w.value(maxPower);
  w.key("delay"); w.begin_array();
  for (int d : delay) w.value(d);
  w.end_array();
  w.key("doppler"); w.begin_array();
  for (uint32_t i = 0; i < nRows; i++) w.value(doppler[i]);
  w.end_array();
  w.key("data"); w.begin_array();
  for (const auto &row : data) {
    w.begin_array();
    for (const auto &v : row) w.value(10.0 * std::log10(std::abs(v)) - noisePower);
    w.end_array();
  }
  w.end_array();
  w.end_object();
  return w.str();
}

// The reference re-parses the document and rewrites "delay" in km
// (Map.cpp:165-185): delay*c/fs/1000 with 2 decimals.
template <class T> std::string Map<T>::delay_bin_to_km(std::string json, uint32_t fs)
{
  std::string arr = "[";
  for (size_t i = 0; i < delay.size(); i++) {
    if (i) arr.push_back(',');
    blah2json::write_double(arr, 1.0 * delay[i] * (Constants::c / (double)fs) / 1000, 2);
  }
  arr.push_back(']');
  return blah2json::replace_array(json, "delay", arr);
}

// append one JSON object to a file that holds a JSON array (Map.cpp:208-262)
template <class T> bool Map<T>::save(std::string json, std::string path)
{
  FILE *fp = std::fopen(path.c_str(), "rb+");
  if (!fp) {
    fp = std::fopen(path.c_str(), "wb+");
    if (!fp) return false;
    std::fputs("[]", fp);
    std::fflush(fp);
  }
  std::fseek(fp, 0, SEEK_SET);
  if (std::fgetc(fp) != '[') { std::fclose(fp); return false; }
  const bool empty = std::fgetc(fp) == ']';
  std::fseek(fp, -1, SEEK_END);
  if (std::fgetc(fp) != ']') { std::fclose(fp); return false; }
  std::fseek(fp, -1, SEEK_END);
  if (!empty) std::fputc(',', fp);
  std::fwrite(json.data(), 1, json.size(), fp);
  std::fputc(']', fp);
  std::fclose(fp);
  return true;
}

template class Map<std::complex<double>>;
template class Map<double>;
