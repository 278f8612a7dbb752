// The library's range kernel launched bare (no Doppler/metrics kernels around it, no
// Python), same buffers as rangepat.hip.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../include -I../../blah2_amd/csrc rangekern.hip -o rangekern
#include "kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using namespace blah2;

__global__ void fill_random(unsigned *p, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = __float_as_uint((float)((int)(h % 1201u) - 600)); // integer-valued like int16 IQ, zero mean
  }
}

template <class K> void run(const char *name, K kern, const RangeArgs &a, InC32 in, int nCpi, int grid, size_t lds)
{
  hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(128), lds, 0, a, in);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(128), lds, 0, a, in);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::printf("%-44s grid %4d: %.2f us/CPI  (%s)\n", name, grid, ms * 1e3 / 5 / nCpi, hipGetErrorString(hipGetLastError()));
}

int main()
{
  const int nCpi = 128, nD = 513, nCorr = 3898, nDelay = 411;
  const size_t n = (size_t)nCpi * 2000000;
  cf *x, *y, *out, *tw;
  hipMalloc(&x, n * 8); hipMalloc(&y, n * 8);
  hipMalloc(&out, (size_t)nCpi * 26 * nD * 16 * 8); hipMalloc(&tw, 2048 * 8);
  fill_random<<<4096, 256>>>((unsigned *)x, n * 2); fill_random<<<4096, 256>>>((unsigned *)y, n * 2);
  std::vector<cf> h(2048);
  for (int k = 0; k < 2048; k++) { h[k].x = (float)std::cos(-2 * M_PI * k / 2048); h[k].y = (float)std::sin(-2 * M_PI * k / 2048); }
  hipMemcpy(tw, h.data(), 2048 * 8, hipMemcpyHostToDevice);
  hipDeviceSynchronize();
  RangeArgs a;
  a.plan = RangePlan{nCorr, nD, nDelay, -10, 3, 1300, 1.f / 2048};
  a.tw = tw; a.out = out; a.cpiStride = 2000000; a.nPulses = nCpi * nD;
  InC32 in{x, y};
  const size_t lds = (size_t)(WgFft<8>::A_ELEMS + WgFft<8>::B_ELEMS) * sizeof(cf);
  // the shipped kernel at several grid sizes (1024 = LDS-limited residency, 4 workgroups per CU).
  // The ablated variants this tool used to time (arithmetic / LDS / loads switched off one by one;
  // results in DESIGN.md section 4) were template parameters of the product kernel in round 1 and
  // were removed from it; rangepat.hip still times the bare load/store pattern.
  for (int pass = 0; pass < 2; pass++)
    for (int grid : {512, 768, 1024, 2048})
      run("range_kernel<8, InC32>", range_kernel<8, InC32>, a, in, nCpi, grid, lds);
  return 0;
}
