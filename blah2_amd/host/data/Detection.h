// Detection: parallel delay / Doppler / SNR vectors (reference surface:
// src/data/Detection.h:13-70).
#ifndef BLAH2HIP_HOST_DETECTION_H
#define BLAH2HIP_HOST_DETECTION_H

#include <complex>
#include <stdint.h>
#include <string>
#include <vector>

class Detection
{
public:
  Detection(std::vector<double> delay, std::vector<double> doppler, std::vector<double> snr);
  Detection(double delay, double doppler, double snr);
  std::vector<double> get_delay();
  std::vector<double> get_doppler();
  std::vector<double> get_snr();
  size_t get_nDetections();
  std::string to_json(uint64_t timestamp);
  std::string delay_bin_to_km(std::string json, uint32_t fs);
  bool save(std::string json, std::string path);

private:
  std::vector<double> delay;
  std::vector<double> doppler;
  std::vector<double> snr;
};

#endif
