// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, in
// the access patterns this engine's kernels use (MI355X_MICROARCH.md, HBM section: only the
// 16 B/lane streaming read is calibrated there, "calibrate on a known byte count in your own
// access pattern").  One kernel per pattern, each moving exactly BYTES bytes of a 1 GiB buffer:
//   cal_read8 / cal_read16        streaming reads, 8 / 16 bytes per lane
//   cal_write8                    streaming writes, 8 bytes per lane (whole 128-byte lines)
//   cal_write_seg<S>              S-byte row segments (S = 32, 64, 128) at a pitch of 3288 bytes
//                                 (= 411 complex cells: the final map's row pitch at cfg 2), i.e. the
//                                 store pattern of the Doppler tile kernels (4, 8, 16 columns)
// Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./pmccal   (and WRITE_SIZE in its own pass)
// Build: hipcc --offload-arch=gfx950 -O3 pmccal.hip -o pmccal
#include <hip/hip_runtime.h>
#include <cstdio>

struct f2 { float x, y; };
struct f4 { float x, y, z, w; };
constexpr size_t BYTES = 1ull << 30;

template <class V> __device__ void rd(const V *in, float *out)
{
  const size_t n = BYTES / sizeof(V);
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i].x;
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void cal_read8(const f2 *in, float *out) { rd(in, out); }
__global__ void cal_read16(const f4 *in, float *out) { rd(in, out); }
// the first 64 bytes of every 128-byte line, 8 bytes per lane (the 8-column Doppler tile read)
__global__ void cal_read_half(const f2 *in, float *out)
{
  const size_t n = BYTES / sizeof(f2) / 2; // elements read
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[(i >> 3) * 16 + (i & 7)].x;
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void cal_write8(f2 *o)
{
  const size_t n = BYTES / sizeof(f2);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = f2{1.f, 2.f};
}
// rows of S bytes at pitch 3288 B; lanes of a wave cover 512/S consecutive rows, 8 bytes per lane
template <int S> __global__ void cal_write_seg(char *o)
{
  constexpr int PITCH = 3288, LPR = S / 8; // lanes per row
  const size_t rows = BYTES / PITCH;       // rows in the buffer; S*rows bytes are written
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < rows * LPR; t += nthreads) {
    const size_t row = t / LPR, c = t % LPR;
    *reinterpret_cast<f2 *>(o + row * PITCH + c * 8) = f2{1.f, 2.f};
  }
}

int main()
{
  void *buf; float *out;
  hipMalloc(&buf, BYTES + 4096); hipMalloc(&out, 64);
  hipMemset(buf, 1, BYTES + 4096);
  hipDeviceSynchronize();
  const int grid = 256 * 8, T = 256;
  for (int rep = 0; rep < 3; rep++) {
    cal_read8<<<grid, T>>>((const f2 *)buf, out);
    cal_read16<<<grid, T>>>((const f4 *)buf, out);
    cal_read_half<<<grid, T>>>((const f2 *)buf, out);
    cal_write8<<<grid, T>>>((f2 *)buf);
    cal_write_seg<32><<<grid, T>>>((char *)buf);
    cal_write_seg<64><<<grid, T>>>((char *)buf);
    cal_write_seg<128><<<grid, T>>>((char *)buf);
  }
  hipDeviceSynchronize();
  const double rows = (double)(BYTES / 3288);
  std::printf("known bytes per dispatch: read_half 536870912 (of 1 GiB of lines touched) read8 %.0f read16 %.0f write8 %.0f seg32 %.0f seg64 %.0f seg128 %.0f (%s)\n", (double)BYTES,
              (double)BYTES, (double)BYTES, 32 * rows, 64 * rows, 128 * rows, hipGetErrorString(hipGetLastError()));
  return 0;
}
