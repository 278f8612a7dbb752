// Throughput per CU of the cross-lane moves the stream CFAR kernel could use, with the CU full (16 waves of 64):
// ds_bpermute_b32, ds_read_b64 / ds_read2_b64 / ds_write_b64 at lane-linear addresses, v_mov_b32_dpp wave_shl:1,
// and v_add_f64 beside ds_bpermute_b32 (do the two pipes overlap).
//   hipcc --offload-arch=gfx950 -O3 tools/membench/ldsrate.hip -o tools/membench/ldsrate && gpurun -- tools/membench/ldsrate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 512
#define ILP 8

__global__ __launch_bounds__(1024) void k_bperm(int *out, unsigned long long *cyc)
{
  int x[ILP];
  const int lane = threadIdx.x & 63;
  const int addr = ((lane + 3) & 63) << 2;
#pragma unroll
  for (int i = 0; i < ILP; i++) x[i] = threadIdx.x * 7 + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++)
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = __builtin_amdgcn_ds_bpermute(addr, x[i]);
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  int s = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// MODE 0: ds_read_b64, 1: ds_read2_b64 (two offsets), 2: ds_write_b64, 3: ds_read2st64_b64, 4: ds_write2_b64, 5: ds_read_b128, 6: ds_write_b128
template <int MODE> __global__ __launch_bounds__(1024) void k_lds(double *out, unsigned long long *cyc)
{
  __shared__ double buf[16][64 * 2 + 32 + 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *p = &buf[wave][(MODE >= 5 ? 2 * lane : lane) + 8];
  for (int i = lane; i < 64 * 2 + 32 + 64; i += 64) buf[wave][i] = i;
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; i++) x[i] = threadIdx.x + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      if (MODE == 0) { double v; asm volatile("ds_read_b64 %0, %1 offset:24" : "=v"(v) : "v"((unsigned)(size_t)p)); x[i] = v; }
      if (MODE == 1) { double __attribute__((ext_vector_type(2))) v; asm volatile("ds_read2_b64 %0, %1 offset0:3 offset1:5" : "=v"(v) : "v"((unsigned)(size_t)p)); x[i] = v.x; }
      if (MODE == 2) { asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(size_t)p), "v"(x[i])); }
      if (MODE == 3) { double __attribute__((ext_vector_type(2))) v; asm volatile("ds_read2st64_b64 %0, %1 offset0:0 offset1:1" : "=v"(v) : "v"((unsigned)(size_t)p)); x[i] = v.x; }
      if (MODE == 4) { asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:80" ::"v"((unsigned)(size_t)p), "v"(x[i]), "v"(x[(i + 1) % ILP])); }
      if (MODE == 5) { double __attribute__((ext_vector_type(2))) v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)p)); x[i] = v.x; }
      if (MODE == 6) { double __attribute__((ext_vector_type(2))) v = {x[i], x[(i + 1) % ILP]}; asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)p), "v"(v)); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ __launch_bounds__(1024) void k_dpp(int *out, unsigned long long *cyc)
{
  int x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; i++) x[i] = threadIdx.x * 7 + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++)
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = __builtin_amdgcn_update_dpp(0, x[i], 0x130, 0xf, 0xf, false); // wave_shl:1
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  int s = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// NADD fp64 additions per bpermute, independent of it
template <int NADD> __global__ __launch_bounds__(1024) void k_mix(double *out, unsigned long long *cyc)
{
  int x[ILP];
  double y[ILP];
  const int lane = threadIdx.x & 63;
  const int addr = ((lane + 3) & 63) << 2;
#pragma unroll
  for (int i = 0; i < ILP; i++) { x[i] = threadIdx.x * 7 + i; y[i] = threadIdx.x + i; }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++)
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      x[i] = __builtin_amdgcn_ds_bpermute(addr, x[i]);
#pragma unroll
      for (int k = 0; k < NADD; k++) y[(i + k) % ILP] += 1.5;
    }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) s += x[i] + y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
  void *out;
  unsigned long long *cyc, h;
  hipMalloc(&out, 256 * 1024 * 8 * 4);
  hipMalloc(&cyc, 8);
  const double ops = (double)REP * ILP * 16; // wave-level instructions per CU
  auto report = [&](const char *name, double per) {
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %8.2f cycles per wave instruction and CU (%llu cycles, 16 waves)\n", name, (double)h / ops / per, h);
  };
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_bperm, dim3(256), dim3(1024), 0, 0, (int *)out, cyc); report("ds_bpermute_b32", 1);
    hipLaunchKernelGGL(k_lds<0>, dim3(256), dim3(1024), 0, 0, (double *)out, cyc); report("ds_read_b64", 1);
    hipLaunchKernelGGL(k_lds<1>, dim3(256), dim3(1024), 0, 0, (double *)out, cyc); report("ds_read2_b64", 1);
    hipLaunchKernelGGL(k_lds<2>, dim3(256), dim3(1024), 0, 0, (double *)out, cyc); report("ds_write_b64", 1);
    hipLaunchKernelGGL(k_lds<3>, dim3(256), dim3(1024), 0, 0, (double *)out, cyc); report("ds_read2st64_b64", 1);
    hipLaunchKernelGGL(k_lds<4>, dim3(256), dim3(1024), 0, 0, (double *)out, cyc); report("ds_write2_b64", 1);
    hipLaunchKernelGGL(k_lds<5>, dim3(256), dim3(1024), 0, 0, (double *)out, cyc); report("ds_read_b128", 1);
    hipLaunchKernelGGL(k_lds<6>, dim3(256), dim3(1024), 0, 0, (double *)out, cyc); report("ds_write_b128", 1);
    hipLaunchKernelGGL(k_dpp, dim3(256), dim3(1024), 0, 0, (int *)out, cyc); report("v_mov_b32_dpp wave_shl:1", 1);
    hipLaunchKernelGGL(k_mix<1>, dim3(256), dim3(1024), 0, 0, (double *)out, cyc); report("bpermute + 1 v_add_f64", 1);
    hipLaunchKernelGGL(k_mix<2>, dim3(256), dim3(1024), 0, 0, (double *)out, cyc); report("bpermute + 2 v_add_f64", 1);
  }
  return 0;
}
