// DeviceContext: the per-process device state the drop-in classes share.
//
// blah2.cpp:264-287 calls SpectrumAnalyser, WienerHopf, Ambiguity and CfarDetector1D one after the other on the SAME
// x, y of a CPI.  Each class used to drain the FIFO, narrow to fp32 and upload on its own (and WienerHopf's filtered
// channel went device -> host FIFO -> device between two GPU stages): ~6 ms per 1 M-sample CPI against 36 us of kernels.
// Here a channel is narrowed (by a few threads, straight out of IqData's ring) into pinned staging and uploaded ONCE
// per CPI; every class finds it resident as long as IqData::generation() has not moved; WienerHopf's output stays in
// HBM as the new front of y (IqData::set_device_front) and is consumed there by Ambiguity.
// From the second CPI of a FIFO on even that happens ahead of time: IqData::push_back narrows every sample into a
// pinned shadow of its ring as it arrives (blah2.cpp:254-258 moves a CPI into x and y sample by sample right before
// :264-287 runs) and every 256 k samples that stretch is sent to the same positions of a device ring, so the first
// class finds the CPI resident and the sequence starts with kernels, not with 32 MB of narrowing and 32 MB of PCIe.
// (The uploads are enqueued from inside push_back, on the context's stream: like the reference, which holds a FIFO's lock
// around both, a FIFO's pushes and the classes that process it must not run concurrently.  FIFOs the classes never see --
// the capture buffers -- are never attached.)
// Built on the blah2hip_ctx_* part of the C ABI: no HIP headers on this side.
#ifndef BLAH2HIP_HOST_DEVICECONTEXT_H
#define BLAH2HIP_HOST_DEVICECONTEXT_H

#include "data/IqData.h"

#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <stdint.h>
#include <thread>
#include <vector>

struct blah2hip_ctx_s;

class DeviceContext
{
public:
  static DeviceContext &get(); // created on first use for device Ambiguity::default_device()

  void *stream() const;
  void sync();
  // the first `count` samples of q as a complex fp32 plane on the device (uploaded now, or still resident)
  const void *resident(IqData *q, uint32_t count);
  // a device buffer of `count` complex fp32 samples that will become q's front (WienerHopf's output)
  void *front_buffer(IqData *q, uint32_t count);
  // q's first `count` samples are now what front_buffer(q, count) holds (written by work enqueued on stream())
  void adopt_front(IqData *q, uint32_t count);
  // q dropped its first `count` samples (IqData::drop_front): the resident view follows
  void consumed(IqData *q, uint32_t count);
  // asynchronous copies on stream() (+ sync() before the host reads)
  void d2h(void *hptr, const void *dptr, size_t bytes);
  // device / pinned host memory for a class's own results (freed with the process)
  void *alloc_device(size_t bytes);
  void *alloc_pinned(size_t bytes);
  void free_device(void *p);
  void free_pinned(void *p);
  // splits [0, n) over the worker threads
  void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)> &fn);

  DeviceContext(const DeviceContext &) = delete;
  DeviceContext &operator=(const DeviceContext &) = delete;

private:
  DeviceContext();
  ~DeviceContext();
  static void forget(IqData *q);

  struct Mirror : IqDeviceFront {
    DeviceContext *ctx = nullptr;
    float *dev = nullptr;    // complex fp32 plane
    size_t cap = 0;          // samples
    float *devFront = nullptr; // second plane: a filter's output for this channel
    size_t capFront = 0;
    // eager path: a pinned fp32 shadow of the FIFO's ring, filled by IqData::push_back, and its device copy at the same
    // positions, uploaded in stretches while the caller is still pushing
    float *shadow = nullptr;
    float *devRing = nullptr;
    size_t ringCap = 0;      // samples; 0 = not attached
    bool tried = false;      // attach attempted (capacity unbounded or too large: stays on the per-CPI path)
    const float *view = nullptr; // what resident() hands out: dev, devRing or devFront, + offset
    uint32_t viewCount = 0;
    uint64_t gen = ~0ull;    // IqData::generation() the view belongs to
    void read(uint32_t first, uint32_t count, std::complex<double> *dst) override;
  };
  blah2hip_ctx_s *ctx = nullptr;
  // One context per process: its staging buffers, worker job and mirrors are shared state.  Every public entry point (and
  // the eager-upload hook IqData::push_back calls) holds `api`, so two processing chains on two threads -- or an IqData
  // destroyed on another thread -- take turns instead of clobbering each other; within a chain nothing changes.
  std::recursive_mutex api;
  std::mutex mu;
  std::map<IqData *, Mirror *> mirrors;
  // two pinned staging buffers, used in turn: while one channel's upload drains, the next channel is narrowed into the other
  float *pinned[2] = {nullptr, nullptr};
  size_t pinnedSamples[2] = {0, 0};
  int pinnedNext = 0;
  int uploadsSinceSync = 0;
  Mirror &mirror_of(IqData *q);
  float *staging(size_t samples);
  void try_attach(IqData *q, Mirror &m);
  void flush_pending(IqData *q, Mirror &m);
  static void on_chunk(IqData *q, void *mirror);

  // worker threads
  std::vector<std::thread> workers;
  std::mutex pmu;
  std::condition_variable pcv, pdone;
  const std::function<void(size_t, size_t)> *job = nullptr;
  size_t jobN = 0, jobGrain = 0, jobNext = 0, jobPending = 0;
  uint64_t jobId = 0;
  bool stopping = false;
  void worker();
};

#endif
