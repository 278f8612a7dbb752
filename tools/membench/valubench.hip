// How does VALU throughput of radix-16 butterfly code depend on waves/SIMD?
// Occupancy is throttled with a dummy dynamic-LDS allocation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../blah2_amd/csrc/fft_wg.hpp"
using namespace blah2;

__global__ __launch_bounds__(64) void k(cf *out, const cf *tw, int iters)
{
  extern __shared__ char smem[];
  cf v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = cmake((float)(threadIdx.x + i), (float)(blockIdx.x - i));
  cf w[15];
#pragma unroll
  for (int i = 0; i < 15; i++) w[i] = tw[(threadIdx.x * (i + 1)) & 1023];
  for (int it = 0; it < iters; it++) {
    dft16<-1>(v);
#pragma unroll
    for (int i = 1; i < 16; i++) v[i] = cmul(v[i], w[i - 1]);
  }
  cf s = v[0];
#pragma unroll
  for (int i = 1; i < 16; i++) s = cadd(s, v[i]);
  if (s.x == 1234.5f) out[blockIdx.x * 64 + threadIdx.x] = s;
  if (iters < 0) smem[threadIdx.x] = 1;
}

int main()
{
  cf *out, *tw;
  hipMalloc(&out, 1 << 24); hipMalloc(&tw, 1024 * 8); hipMemset(tw, 0, 8192);
  const int iters = 2000;
  // per iteration per thread: dft16 (~174 fp instr) + 15 cmul (60) ~ 234 VALU, ~ 144+60+90 = flops ~ 300
  for (int wavesPerSimd : {1, 2, 3, 4, 6, 8}) {
    const int perCU = wavesPerSimd * 4;
    const size_t lds = (160 * 1024) / perCU - 256; // one 64-thread block per slot
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = 256 * perCU;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<grid, 64, lds>>>(out, tw, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<<<grid, 64, lds>>>(out, tw, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waveIters = (double)grid * iters;
    std::printf("waves/SIMD %d: %.3f ms, %.1f ns per wave-iteration per SIMD-slot, relative throughput %.2f Gwave-iter/s\n",
                wavesPerSimd, ms, ms * 1e6 / iters, waveIters / (ms * 1e-3) / 1e9);
  }
  return 0;
}
