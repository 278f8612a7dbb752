#include "data/Detection.h"

#include "data/meta/Constants.h"
#include "util/JsonOut.h"

#include <cstdio>

Detection::Detection(std::vector<double> d, std::vector<double> f, std::vector<double> s)
    : delay(std::move(d)), doppler(std::move(f)), snr(std::move(s)) {}
Detection::Detection(double d, double f, double s) : delay{d}, doppler{f}, snr{s} {}

std::vector<double> Detection::get_delay() { return delay; }
std::vector<double> Detection::get_doppler() { return doppler; }
std::vector<double> Detection::get_snr() { return snr; }
size_t Detection::get_nDetections() { return delay.size(); }

// reference Detection.cpp:47-85: timestamp, delay[], doppler[], snr[], 2 decimals
std::string Detection::to_json(uint64_t timestamp)
{
  blah2json::Writer w(2);
  w.begin_object();
  w.key("timestamp"); w.value(timestamp);
  const std::vector<double> *arrs[3] = {&delay, &doppler, &snr};
  const char *names[3] = {"delay", "doppler", "snr"};
  for (int k = 0; k < 3; k++) {
    w.key(names[k]);
    w.begin_array();
    for (double v : *arrs[k]) w.value(v);
    w.end_array();
  }
  w.end_object();
  return w.str();
}

// reference Detection.cpp:87-106
std::string Detection::delay_bin_to_km(std::string json, uint32_t fs)
{
  std::string arr = "[";
  for (size_t i = 0; i < delay.size(); i++) {
    if (i) arr.push_back(',');
    blah2json::write_double(arr, 1.0 * delay[i] * (Constants::c / (double)fs) / 1000, 2);
  }
  arr.push_back(']');
  return blah2json::replace_array(json, "delay", arr);
}

bool Detection::save(std::string json, std::string path)
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
