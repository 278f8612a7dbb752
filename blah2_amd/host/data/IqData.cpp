#include "data/IqData.h"

#include "util/JsonOut.h"

#include <cmath>
#include <iostream>
#include <stdexcept>

IqData::IqData(uint32_t capacity) : n(capacity), data(new std::deque<std::complex<double>>) {}
IqData::~IqData() { delete data; }

uint32_t IqData::get_n() { return n; }
uint32_t IqData::get_length() { return static_cast<uint32_t>(data->size()); }
void IqData::lock() { mutex_lock.lock(); }
void IqData::unlock() { mutex_lock.unlock(); }
std::deque<std::complex<double>> IqData::get_data() { return *data; }

// reference IqData.cpp:42-53: when full, the oldest sample is evicted
void IqData::push_back(std::complex<double> sample)
{
  if (data->size() >= n) data->pop_front();
  data->push_back(sample);
}

// reference IqData.cpp:55-63
std::complex<double> IqData::pop_front()
{
  if (data->empty()) throw std::runtime_error("Attempting to pop from an empty deque");
  const std::complex<double> s = data->front();
  data->pop_front();
  return s;
}

void IqData::pop_front_block(double *dst, uint32_t count)
{
  // same observable behaviour as `count` calls of pop_front() (the FIFO is emptied and the
  // same exception is thrown on underflow), without the per-sample bookkeeping
  const size_t have = data->size();
  const size_t take = have < count ? have : (size_t)count;
  auto it = data->begin();
  for (size_t i = 0; i < take; i++, ++it) {
    dst[2 * i] = it->real();
    dst[2 * i + 1] = it->imag();
  }
  data->erase(data->begin(), data->begin() + (std::ptrdiff_t)take);
  if (take < count) throw std::runtime_error("Attempting to pop from an empty deque");
}

void IqData::print()
{
  std::cout << data->size() << std::endl;
  while (!data->empty()) {
    std::cout << data->front() << std::endl;
    data->pop_front();
  }
}

void IqData::clear() { data->clear(); }
void IqData::update_spectrum(std::vector<std::complex<double>> s) { spectrum = std::move(s); }
void IqData::update_frequency(std::vector<double> f) { frequency = std::move(f); }

// reference IqData.cpp:92-125: timestamp,min,max,mean,frequency[],spectrum[] (dB)
std::string IqData::to_json(uint64_t timestamp)
{
  blah2json::Writer w(2);
  w.begin_object();
  w.key("timestamp"); w.value(timestamp);
  w.key("min"); w.value(min);
  w.key("max"); w.value(max);
  w.key("mean"); w.value(mean);
  w.key("frequency"); w.begin_array();
  for (double f : frequency) w.value(f);
  w.end_array();
  w.key("spectrum"); w.begin_array();
  for (const auto &s : spectrum) w.value(10 * std::log10(std::abs(s)));
  w.end_array();
  w.end_object();
  return w.str();
}
