#include "data/IqData.h"

#include "util/JsonOut.h"

#include <algorithm>
#include <cmath>
#include <iostream>
#include <stdexcept>

void (*IqData::destroyed_hook)(IqData *) = nullptr;

IqData::IqData(uint32_t capacity) : n(capacity) {}
IqData::~IqData()
{
  if (destroyed_hook) destroyed_hook(this);
}

uint32_t IqData::get_n() { return n; }
uint32_t IqData::get_length() { return static_cast<uint32_t>(count); }
void IqData::lock() { mutex_lock.lock(); }
void IqData::unlock() { mutex_lock.unlock(); }

void IqData::spans(size_t first, size_t cnt, std::pair<const std::complex<double> *, size_t> out[2]) const
{
  out[0] = {nullptr, 0};
  out[1] = {nullptr, 0};
  if (!cnt) return;
  const size_t cap = ring.size(), start = (head + first) % cap;
  const size_t a = std::min(cnt, cap - start);
  out[0] = {ring.data() + start, a};
  if (a < cnt) out[1] = {ring.data(), cnt - a};
}

// the storage doubles until it holds n samples (an IqData of a whole capture buffer does not claim it up front)
void IqData::grow()
{
  const size_t cap = ring.size();
  size_t want = cap ? 2 * cap : 1024;
  if (n && want > n) want = n;
  if (want <= cap) want = cap + 1; // n == 0 (or n < 1024): still room for the sample being pushed
  std::vector<std::complex<double>> bigger(want);
  std::pair<const std::complex<double> *, size_t> sp[2];
  spans(0, count, sp);
  size_t o = 0;
  for (const auto &s : sp) { std::copy(s.first, s.first + s.second, bigger.begin() + (std::ptrdiff_t)o); o += s.second; }
  ring.swap(bigger);
  head = 0;
}

void IqData::materialise()
{
  if (!devCount) return;
  const uint32_t cnt = std::min<uint32_t>(devCount, (uint32_t)count);
  std::pair<const std::complex<double> *, size_t> sp[2];
  spans(0, cnt, sp);
  uint32_t first = 0;
  for (const auto &s : sp)
    if (s.second) {
      devSrc->read(devSkip + first, (uint32_t)s.second, const_cast<std::complex<double> *>(s.first));
      first += (uint32_t)s.second;
    }
  devCount = 0;
  devSkip = 0;
  devSrc = nullptr;
  sh.mirrored = 0; // the ring now holds samples its shadow (and the device's copy of it) never saw
}

bool IqData::attach_shadow(float *buf, size_t chunk, void (*hook)(IqData *, void *), void *user)
{
  if (!n || !buf || !hook || !chunk) return false;
  if (ring.size() != n) { // fix the storage at its capacity, front sample at position 0
    std::vector<std::complex<double>> full(n);
    std::pair<const std::complex<double> *, size_t> sp[2];
    spans(0, count, sp);
    size_t o = 0;
    for (const auto &s : sp) { std::copy(s.first, s.first + s.second, full.begin() + (std::ptrdiff_t)o); o += s.second; }
    ring.swap(full);
    head = 0;
  }
  sh = Shadow();
  sh.buf = buf;
  sh.chunk = chunk;
  sh.hook = hook;
  sh.user = user;
  sh.mirrored = 0; // samples already in the FIFO have no shadow: the eager path starts once they have left
  sh.pendStart = (head + count) % ring.size();
  sh.pend = 0;
  return true;
}

void IqData::detach_shadow() { sh = Shadow(); }

void IqData::shadow_take_pending(size_t &start, size_t &cnt)
{
  start = sh.pendStart;
  cnt = sh.pend;
  if (!ring.empty()) sh.pendStart = (sh.pendStart + sh.pend) % ring.size();
  sh.pend = 0;
}

void IqData::set_device_front(uint32_t cnt, IqDeviceFront *src)
{
  devCount = std::min<uint32_t>(cnt, (uint32_t)count);
  devSkip = 0;
  devSrc = devCount ? src : nullptr;
  gen++;
}

// reference IqData.cpp:33-36 returns a copy of the deque
std::deque<std::complex<double>> IqData::get_data()
{
  materialise();
  std::deque<std::complex<double>> out;
  std::pair<const std::complex<double> *, size_t> sp[2];
  spans(0, count, sp);
  for (const auto &s : sp) out.insert(out.end(), s.first, s.first + s.second);
  return out;
}

// reference IqData.cpp:42-53: when full, the oldest sample is evicted
void IqData::push_back(std::complex<double> sample)
{
  if (count >= n && count > 0) { // evict the front sample
    if (devCount) { devCount--; devSkip++; } // a device-only one simply ceases to exist
    head = (head + 1) % ring.size();
    count--;
    if (sh.mirrored > count) sh.mirrored = count;
  }
  if (count == ring.size()) grow();
  size_t pos = head + count;
  if (pos >= ring.size()) pos -= ring.size();
  ring[pos] = sample;
  count++;
  gen++;
  if (sh.buf) {
    sh.buf[2 * pos] = (float)sample.real();
    sh.buf[2 * pos + 1] = (float)sample.imag();
    if (sh.mirrored < count) sh.mirrored++;
    if (++sh.pend >= sh.chunk) sh.hook(this, sh.user);
  }
}

// reference IqData.cpp:55-63
std::complex<double> IqData::pop_front()
{
  if (!count) throw std::runtime_error("Attempting to pop from an empty deque");
  materialise();
  const std::complex<double> s = ring[head];
  head = (head + 1) % ring.size();
  count--;
  if (sh.mirrored > count) sh.mirrored = count;
  gen++;
  return s;
}

void IqData::pop_front_block(double *dst, uint32_t cnt)
{
  // same observable behaviour as `cnt` calls of pop_front() (the FIFO is emptied and the
  // same exception is thrown on underflow), without the per-sample bookkeeping
  materialise();
  const size_t take = std::min<size_t>(count, cnt);
  std::pair<const std::complex<double> *, size_t> sp[2];
  spans(0, take, sp);
  size_t o = 0;
  for (const auto &s : sp)
    for (size_t i = 0; i < s.second; i++, o++) {
      dst[2 * o] = s.first[i].real();
      dst[2 * o + 1] = s.first[i].imag();
    }
  if (take) head = (head + take) % ring.size();
  count -= take;
  if (sh.mirrored > count) sh.mirrored = count;
  gen++;
  if (take < cnt) throw std::runtime_error("Attempting to pop from an empty deque");
}

void IqData::drop_front(uint32_t cnt)
{
  const size_t take = std::min<size_t>(count, cnt);
  if (devCount) { // consumed where they live: nothing to bring back
    const uint32_t d = (uint32_t)std::min<size_t>(devCount, take);
    devCount -= d;
    devSkip += d;
    if (!devCount) { devSrc = nullptr; devSkip = 0; }
  }
  if (take) head = (head + take) % ring.size();
  count -= take;
  if (sh.mirrored > count) sh.mirrored = count;
  gen++;
  if (take < cnt) throw std::runtime_error("Attempting to pop from an empty deque");
}

void IqData::keep_front(uint32_t cnt)
{
  if (cnt < count) { // the most recent samples go: so do their shadows, and what was pushed but not yet reported
    const size_t gone = count - cnt;
    sh.mirrored = sh.mirrored > gone ? sh.mirrored - gone : 0;
    sh.pend = sh.pend > gone ? sh.pend - gone : 0;
    count = cnt;
    if (!ring.empty()) sh.pendStart = (head + count + ring.size() - sh.pend) % ring.size(); // the unreported stretch ends at the new back
  }
  if (devCount > count) devCount = (uint32_t)count;
  gen++;
}

void IqData::copy_front_c32(uint32_t first, uint32_t cnt, float *dst) const
{
  std::pair<const std::complex<double> *, size_t> sp[2];
  spans(first, cnt, sp);
  for (const auto &s : sp) {
    const double *src = reinterpret_cast<const double *>(s.first); // complex<double> is (re, im) by the standard
    for (size_t i = 0; i < 2 * s.second; i++) dst[i] = (float)src[i];
    dst += 2 * s.second;
  }
}

void IqData::print()
{
  materialise();
  std::cout << count << std::endl;
  while (count) {
    std::cout << ring[head] << std::endl;
    head = (head + 1) % ring.size();
    count--;
  }
  sh.mirrored = 0;
  gen++;
}

void IqData::clear()
{
  head = 0;
  count = 0;
  sh.pendStart = 0;
  sh.pend = 0;
  sh.mirrored = 0;
  devCount = 0;
  devSkip = 0;
  devSrc = nullptr;
  gen++;
}
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
