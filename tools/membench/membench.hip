// Streaming-read microbenchmark: what does the load path of the range kernel
// cost by itself?  8 vs 16 bytes per lane, aligned vs pulse-like misaligned
// bases, at the occupancy of the range kernel (few waves per CU, 16 loads in
// flight per thread).  Build: hipcc --offload-arch=gfx950 -O3 membench.hip -o membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct f2 { float x, y; };
struct f4 { float x, y, z, w; };

// each workgroup (T threads) reads `chunk` consecutive elements starting at base + wg*stride, 16 per thread
template <class V, int T> __global__ __launch_bounds__(T) void rd(const V *in, float *out, long stride, int nChunks, int reps, int skew)
{
  float acc = 0.f;
  for (int c = blockIdx.x; c < nChunks; c += gridDim.x) {
    const V *p = in + (long)c * stride + skew;
    for (int r = 0; r < reps; r++) {
      V v[16];
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = p[(long)r * 16 * T + threadIdx.x + T * k];
#pragma unroll
      for (int k = 0; k < 16; k++) acc += v[k].x + v[k].y;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <class V, int T> double run(const void *buf, float *out, size_t bytes, int wgPerCU, int skewElems, long strideElems, int reps)
{
  const long perChunk = (long)reps * 16 * T;
  const int nChunks = (int)(bytes / sizeof(V) / (strideElems ? strideElems : perChunk)) - 1;
  const long stride = strideElems ? strideElems : perChunk;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int grid = 256 * wgPerCU;
  rd<V, T><<<grid, T>>>((const V *)buf, out, stride, nChunks, reps, skewElems);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; i++) rd<V, T><<<grid, T>>>((const V *)buf, out, stride, nChunks, reps, skewElems);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  const double moved = 5.0 * (double)nChunks * perChunk * sizeof(V);
  return moved / (ms * 1e-3) / 1e12;
}

int main()
{
  const size_t bytes = 1ull << 30;
  void *buf; float *out;
  hipMalloc(&buf, bytes + (1 << 20)); hipMalloc(&out, 64);
  hipMemset(buf, 1, bytes + (1 << 20));
  std::printf("TB/s (1 GiB working set, 16 loads in flight per thread)\n");
  for (int wg : {2, 4, 8, 16}) {
    std::printf("64-thread WGs x %2d per CU : 8B aligned %.2f | 8B skew+1 %.2f | 8B pulse-stride(3898) %.2f | 16B aligned %.2f | 16B skew 8B %.2f\n", wg,
                run<f2, 64>(buf, out, bytes, wg, 0, 0, 4), run<f2, 64>(buf, out, bytes, wg, 1, 0, 4),
                run<f2, 64>(buf, out, bytes, wg, 0, 3898, 3), run<f4, 64>(buf, out, bytes, wg, 0, 0, 4),
                run<f4, 64>((const char *)buf + 8, out, bytes, wg, 0, 0, 4));
  }
  for (int wg : {1, 2, 4}) {
    std::printf("128-thread WGs x %2d per CU: 8B aligned %.2f | 8B pulse-stride(3898) %.2f | 16B aligned %.2f | 16B skew 8B %.2f\n", wg,
                run<f2, 128>(buf, out, bytes, wg, 0, 0, 2), run<f2, 128>(buf, out, bytes, wg, 0, 3898, 1),
                run<f4, 128>(buf, out, bytes, wg, 0, 0, 2), run<f4, 128>((const char *)buf + 8, out, bytes, wg, 0, 0, 2));
  }
  return 0;
}
