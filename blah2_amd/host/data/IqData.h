// IqData: bounded FIFO of complex<double> samples shared between the capture
// and processing threads.  Same public surface as the reference's
// src/data/IqData.h:16-100 so blah2.cpp compiles against it unchanged.
//
// Inside it is a ring buffer, not the reference's std::deque (only get_data() hands a deque out, as a copy, like
// the reference's): push_back / pop_front are O(1) without node allocations, and the processing classes can take the
// front samples as (at most two) contiguous spans -- blah2.cpp:264-287 runs Spectrum, WienerHopf and Ambiguity on the
// same x, y, and the GPU classes narrow and upload each channel ONCE per CPI (util/DeviceContext.h).  The front of the
// FIFO may also live on the device only: WienerHopf replaces y's samples (WienerHopf.cpp:156-160) by the filtered
// channel it has just computed in HBM, Ambiguity consumes them there, and the host copy is written only if somebody
// reads those samples through this class first.
#ifndef BLAH2HIP_HOST_IQDATA_H
#define BLAH2HIP_HOST_IQDATA_H

#include <complex>
#include <deque>
#include <mutex>
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>

// where the samples at the front of an IqData live while they are on the device only
struct IqDeviceFront {
  virtual ~IqDeviceFront() {}
  // writes samples [first, first + count) of the device-resident front into dst
  virtual void read(uint32_t first, uint32_t count, std::complex<double> *dst) = 0;
};

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

  // ---- extensions used by the GPU classes ---------------------------------------------------
  // move up to `count` front samples into a contiguous interleaved (re,im) buffer; throws like pop_front on underflow
  void pop_front_block(double *dst, uint32_t count);
  // what `count` calls of pop_front() leave behind, without reading the samples (O(1)); on underflow the FIFO is
  // emptied and the same exception is thrown
  void drop_front(uint32_t count);
  // drops everything behind the first `count` samples (WienerHopf.cpp:156-160 clears y and refills it with nSamples)
  void keep_front(uint32_t count);
  // samples [first, first + count) as interleaved fp32 (re,im), the conversion blah2hip's fp32 kernels start from.
  // The caller guarantees first + count <= get_length() and that the range is not device-only.
  void copy_front_c32(uint32_t first, uint32_t count, float *dst) const;
  // a counter that changes with every change of the samples (the device cache's validity check)
  uint64_t generation() const { return gen; }
  // the first `count` samples now live on the device (`src` reads them back on demand; it must outlive that state,
  // DeviceContext owns it); the host copies of those samples are stale until materialised
  void set_device_front(uint32_t count, IqDeviceFront *src);
  uint32_t device_front_count() const { return devCount; }
  // read back what update_spectrum / update_frequency stored (the reference only exposes them through to_json)
  const std::vector<std::complex<double>> &get_spectrum() const { return spectrum; }
  const std::vector<double> &get_frequency() const { return frequency; }
  // called when an IqData dies, so that a cache keyed by its address can forget it
  static void (*destroyed_hook)(IqData *);

  // ---- fp32 shadow of the ring (util/DeviceContext.h) ------------------------------------------
  // From attach_shadow() on, push_back also narrows each sample into buf[2 * position], position = its index in the
  // ring (whose storage is fixed at its capacity n by the call), and calls hook(this, user) every `chunk` pushes: the
  // device context uploads that stretch while the caller is still pushing (blah2.cpp:254-258 fills x and y sample by
  // sample right before it processes them), so that a CPI is resident by the time the first class asks for it.
  // False (nothing attached) when the capacity is unbounded (n == 0).  `buf` holds 2 * n floats and stays the caller's.
  bool attach_shadow(float *buf, size_t chunk, void (*hook)(IqData *, void *), void *user);
  void detach_shadow();
  // every sample in the FIFO has its shadow: those pushed since attach_shadow() have, so this holds once the older ones
  // have left (x and y turn over once per CPI); device-only samples written back into the ring have none either
  bool shadow_valid() const { return sh.buf && sh.mirrored == count; }
  // the stretch of ring positions pushed since the last call: [start, start + cnt) modulo the capacity
  void shadow_take_pending(size_t &start, size_t &cnt);
  size_t ring_capacity() const { return ring.size(); }
  size_t head_pos() const { return head; }

private:
  uint32_t n;
  std::mutex mutex_lock;
  std::vector<std::complex<double>> ring; // storage, grows up to n
  size_t head = 0, count = 0;
  uint64_t gen = 0;
  uint32_t devCount = 0;          // front samples whose truth is on the device
  uint32_t devSkip = 0;           // samples of that device view already consumed or evicted in front of them
  IqDeviceFront *devSrc = nullptr;
  struct Shadow {
    float *buf = nullptr;
    size_t mirrored = 0; // the most recent `mirrored` samples of the FIFO have their shadow
    size_t pendStart = 0, pend = 0, chunk = 0;
    void (*hook)(IqData *, void *) = nullptr;
    void *user = nullptr;
  } sh;
  double min = 0, max = 0, mean = 0;
  std::vector<std::complex<double>> spectrum;
  std::vector<double> frequency;

  void grow();
  void materialise(); // device-only front samples -> host
  // the (at most two) contiguous spans of samples [first, first + cnt)
  void spans(size_t first, size_t cnt, std::pair<const std::complex<double> *, size_t> out[2]) const;
};

#endif
