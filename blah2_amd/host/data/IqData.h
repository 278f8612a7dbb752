// IqData: bounded FIFO of complex<double> samples shared between the capture
// and processing threads.  Same public surface as the reference's
// src/data/IqData.h:16-100 so blah2.cpp compiles against it unchanged.
#ifndef BLAH2HIP_HOST_IQDATA_H
#define BLAH2HIP_HOST_IQDATA_H

#include <complex>
#include <deque>
#include <mutex>
#include <stdint.h>
#include <string>
#include <vector>

class IqData
{
public:
  explicit IqData(uint32_t n);
  ~IqData();
  IqData(const IqData &) = delete;
  IqData &operator=(const IqData &) = delete;

  uint32_t get_n();
  uint32_t get_length();
  void lock();
  void unlock();
  std::deque<std::complex<double>> get_data();
  void push_back(std::complex<double> sample);
  std::complex<double> pop_front();
  void print();
  void clear();
  void update_spectrum(std::vector<std::complex<double>> spectrum);
  void update_frequency(std::vector<double> frequency);
  std::string to_json(uint64_t timestamp);

  // extension used by the GPU classes: move up to `count` front samples into
  // a contiguous interleaved (re,im) buffer; throws like pop_front on underflow.
  void pop_front_block(double *dst, uint32_t count);
  // extension: read back what update_spectrum / update_frequency stored (the
  // reference only exposes them through to_json)
  const std::vector<std::complex<double>> &get_spectrum() const { return spectrum; }
  const std::vector<double> &get_frequency() const { return frequency; }

private:
  uint32_t n;
  std::mutex mutex_lock;
  std::deque<std::complex<double>> *data;
  double min = 0, max = 0, mean = 0;
  std::vector<std::complex<double>> spectrum;
  std::vector<double> frequency;
};

#endif
