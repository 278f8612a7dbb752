// Trace builds only (-DRANGEW_TRACE / -DDOPW_TRACE / -DC2T_TRACE, tools/build_trace.sh): the per-phase s_memtime ticks a
// kernel's waves collect are added into a device global and printed by the kernel itself (every 8th launch: the totals so
// far, as fractions).  Nothing of it appears in a kernel signature, an argument struct or in host code; without the
// macros this header is empty.
#pragma once
#if defined(RANGEW_TRACE) || defined(DOPW_TRACE) || defined(C2T_TRACE)
#include <hip/hip_runtime.h>
namespace blah2 {
__device__ unsigned long long trace_buckets[16];
__device__ unsigned int trace_launches;
// one lane per wave: its buckets; the first wave of the launch also counts the launch and prints every 8th
template <int N> __device__ __forceinline__ void trace_finish(const char *tag, const uint64_t (&tr)[N], bool first_wave_of_grid)
{
  for (int k = 0; k < N; k++) atomicAdd(&trace_buckets[k], (unsigned long long)tr[k]);
  if (first_wave_of_grid && (atomicAdd(&trace_launches, 1u) & 7u) == 7u) {
    double tot = 0;
    for (int k = 0; k < N; k++) tot += (double)trace_buckets[k];
    printf("[%s trace] buckets:", tag);
    for (int k = 0; k < N; k++) printf(" %.3f", (double)trace_buckets[k] / tot);
    printf(" of %.3e ticks (waves that have finished so far, %u launches)\n", tot, trace_launches);
  }
}
} // namespace blah2
#endif
