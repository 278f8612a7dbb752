// Latencies of the dependent links of the look-ahead solve's coefficient chain, one wave: cycles per iteration of
//   x = fma(x, a, b)                              (1 link)
//   s = readlane(x, k); x = fma(x, s, b)          (broadcast of a lane + its use)
//   x = rcp(x)                                    (the reciprocal estimate)
//   x = wave_shr(x); x = fma(x, a, b)             (the shift + its use)
//   exec juggling + ds_write (put_coef)           (per call)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 512
__device__ __forceinline__ double rl(double v, int l)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double shr1(double v)
{
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true));
}
typedef double d2_t __attribute__((ext_vector_type(2)));
__global__ void k(double *out, double a, double b, unsigned long long *cyc)
{
  __shared__ double lds[64];
  double x = 1.0 + threadIdx.x * 1e-3;
  unsigned long long t0, t1;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) x = __builtin_fma(x, a, b);
  t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) cyc[0] = t1 - t0;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) { const double s = rl(x, r & 63); x = __builtin_fma(x, s, b); }
  t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) cyc[1] = t1 - t0;
  x = 0.5 + 1e-3 * threadIdx.x;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) x = __builtin_amdgcn_rcp(x);
  t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) cyc[2] = t1 - t0;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) { x = shr1(x); x = __builtin_fma(x, a, b); }
  t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) cyc[3] = t1 - t0;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
    const d2_t e = {x, x + 1.0};
    const uint32_t ad = (uint32_t)(uintptr_t)(lds + (r & 31));
    asm volatile("s_mov_b64 exec, 1\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:16\n\ts_mov_b64 exec, -1\n\ts_nop 4" :: "v"(ad), "v"(e) : "memory");
    x = __builtin_fma(x, a, b);
  }
  t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) cyc[4] = t1 - t0;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) { const double s = rl(x, r & 63); const double m = s * a; x = __builtin_fma(x, m, b); }
  t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) cyc[5] = t1 - t0;
  // readlane alone in a dependent chain through an integer
  int v = threadIdx.x;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) { v = __builtin_amdgcn_readlane(v, r & 63) + threadIdx.x; }
  t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) cyc[6] = t1 - t0;
  out[threadIdx.x] = x + lds[threadIdx.x & 31] + v;
}
int main()
{
  double *o; unsigned long long *c, h[8];
  hipMalloc(&o, 4096); hipMalloc(&c, 64);
  for (int i = 0; i < 3; i++) k<<<1, 64>>>(o, 1.0000001, 1e-9, c);
  hipDeviceSynchronize();
  hipMemcpy(h, c, 56, hipMemcpyDeviceToHost);
  const char *nm[7] = {"fma", "readlane(2) + fma", "rcp", "wave_shr(2) + fma", "put_coef + fma", "readlane(2) + mul + fma", "readlane + v_add (int)"};
  for (int i = 0; i < 7; i++) printf("%-28s %.1f cycles per iteration\n", nm[i], (double)h[i] / REP);
  return 0;
}
