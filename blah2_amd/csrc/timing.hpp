// Per-kernel HIP-event timing shared by the ambiguity and clutter handles: when
// enabled, every launch is bracketed by an event pair recorded on the launch
// stream; collect() synchronises, sums per slot and recycles the events.
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

namespace blah2 {

template <int NSLOT> struct KernelTimer {
  struct Pair { hipEvent_t a, b; };
  bool enabled = false;
  std::vector<Pair> ev[NSLOT];
  std::vector<Pair> pool;

  hipError_t tic(int k, hipStream_t st)
  {
    if (!enabled) return hipSuccess;
    Pair p;
    if (!pool.empty()) {
      p = pool.back();
      pool.pop_back();
    } else {
      hipError_t e = hipEventCreate(&p.a);
      if (e != hipSuccess) return e;
      e = hipEventCreate(&p.b);
      if (e != hipSuccess) { (void)hipEventDestroy(p.a); return e; }
    }
    ev[k].push_back(p);
    return hipEventRecord(p.a, st);
  }
  hipError_t toc(int k, hipStream_t st)
  {
    if (!enabled) return hipSuccess;
    return hipEventRecord(ev[k].back().b, st);
  }
  // ms_total[k], launches[k] since the last collect; the caller has synchronised the device
  hipError_t collect(double *ms_total, uint32_t *launches)
  {
    for (int k = 0; k < NSLOT; k++) {
      double tot = 0.0;
      for (auto &p : ev[k]) {
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, p.a, p.b);
        if (e != hipSuccess) return e;
        tot += ms;
        pool.push_back(p);
      }
      ms_total[k] = tot;
      launches[k] = (uint32_t)ev[k].size();
      ev[k].clear();
    }
    return hipSuccess;
  }
  void destroy()
  {
    for (auto &v : ev) {
      for (auto &p : v) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
      v.clear();
    }
    for (auto &p : pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    pool.clear();
  }
};

} // namespace blah2
