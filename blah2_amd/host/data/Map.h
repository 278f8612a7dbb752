// Map<T>: the delay-Doppler map product (rows = Doppler bins, columns = delay
// bins) with the public members and methods of the reference's
// src/data/Map.h:19-112, so detection / interpolation / JSON code written
// against blah2 keeps working.  A map returned by the GPU Ambiguity class also
// remembers which engine handle holds its device copy, which lets
// CfarDetector1D run on the GPU without re-uploading it.
#ifndef BLAH2HIP_HOST_MAP_H
#define BLAH2HIP_HOST_MAP_H

#include <complex>
#include <deque>
#include <stdint.h>
#include <string>
#include <vector>

struct blah2hip_amb_s;

template <typename T> class Map
{
private:
  uint32_t nRows;
  uint32_t nCols;

  // device-side provenance (not part of the reference surface)
  blah2hip_amb_s *engine = nullptr;
  uint32_t engineCpi = 0;
  double engineNoise = 0.0, engineMax = 0.0;
  bool engineMetricsValid = false;
  uint64_t engineFingerprint = 0; // of `data` as the engine delivered it
  uint64_t fingerprint() const;

public:
  std::vector<std::vector<T>> data;
  std::deque<int> delay;
  std::deque<double> doppler;
  double noisePower;
  double maxPower;

  Map(uint32_t nRows, uint32_t nCols);

  void set_row(uint32_t i, std::vector<T> row);
  void set_col(uint32_t i, std::vector<T> col);
  void set_metrics();
  uint32_t get_nRows();
  uint32_t get_nCols();
  std::vector<T> get_row(uint32_t row);
  std::vector<T> get_col(uint32_t col);
  Map<double> *get_map_db();
  void print();
  uint32_t doppler_hz_to_bin(double dopplerHz);
  std::string to_json(uint64_t timestamp);
  std::string delay_bin_to_km(std::string json, uint32_t fs);
  bool save(std::string json, std::string path);

  // ---- extensions used by the GPU classes --------------------------------
  // called by Ambiguity::process after it has filled `data`
  void bind_engine(blah2hip_amb_s *h, uint32_t cpi, double noise, double peak)
  {
    engine = h; engineCpi = cpi; engineNoise = noise; engineMax = peak; engineMetricsValid = true;
    engineFingerprint = fingerprint();
  }
  // the engine whose device copy still equals `data` (nullptr once the caller has changed the cells)
  blah2hip_amb_s *get_engine() const { return (engine && fingerprint() == engineFingerprint) ? engine : nullptr; }
  uint32_t get_engine_cpi() const { return engineCpi; }
};

#endif
