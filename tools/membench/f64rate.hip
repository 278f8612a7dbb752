// Issue rate / latency of the instructions the look-ahead Toeplitz solve is made of, from ONE wave (and from 2 / 4 waves of
// one SIMD): v_fma_f64 independent and dependent, v_readlane_b32 -> SGPR -> VALU use, v_mov_b32_dpp wave_shr:1.
//   hipcc --offload-arch=gfx950 -O3 tools/membench/f64rate.hip -o tools/membench/f64rate && gpurun -- tools/membench/f64rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256

template <int ILP> __global__ void fma_indep(double *out, double a, double b, unsigned long long *cyc)
{
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; i++) x[i] = threadIdx.x + i;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int i = 0; i < ILP; i++) x[i] = __builtin_fma(x[i], a, b);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void f32_indep(float *out, float a, float b, unsigned long long *cyc)
{
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = __builtin_fmaf(x[i], a, b);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// readlane -> scalar operand of an fma, 4 independent chains
__global__ void readlane_use(double *out, unsigned long long *cyc)
{
  double x[4];
  int v = threadIdx.x * 3 + 1;
#pragma unroll
  for (int i = 0; i < 4; i++) x[i] = threadIdx.x + i;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int lo = __builtin_amdgcn_readlane(v, (r + k) & 63), hi = __builtin_amdgcn_readlane(v + 7, (r + k + 1) & 63);
      const double s = __hiloint2double(hi & 0x3fffffff, lo);
#pragma unroll
      for (int i = 0; i < 4; i++) x[i] = __builtin_fma(x[i], s, 1.0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void dpp_shift(int *out, unsigned long long *cyc)
{
  int x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = __builtin_amdgcn_update_dpp(0, x[i], 0x138, 0xf, 0xf, true) + 1;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// accuracy of v_rcp_f64 and of one / two Newton steps on it
__global__ void rcp_acc(double *err)
{
  double worst0 = 0, worst1 = 0, worst2 = 0;
  unsigned long long st = 88172645463325252ull + threadIdx.x * 7919ull;
  for (int it = 0; it < 20000; it++) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    const double d = 1e-12 + (double)(st >> 11) * (1.0 / 9007199254740992.0); // (0, 1]
    double r = __builtin_amdgcn_rcp(d);
    const double ex = 1.0 / d;
    worst0 = fmax(worst0, fabs(r - ex) / ex);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    worst1 = fmax(worst1, fabs(r - ex) / ex);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    worst2 = fmax(worst2, fabs(r - ex) / ex);
  }
  err[threadIdx.x * 3 + 0] = worst0; err[threadIdx.x * 3 + 1] = worst1; err[threadIdx.x * 3 + 2] = worst2;
}

int main()
{
  {
    double *e; hipMalloc(&e, 64 * 3 * 8);
    rcp_acc<<<1, 64>>>(e);
    double h[192]; hipMemcpy(h, e, sizeof(h), hipMemcpyDeviceToHost);
    double w[3] = {0, 0, 0};
    for (int i = 0; i < 64; i++) for (int k = 0; k < 3; k++) w[k] = h[3 * i + k] > w[k] ? h[3 * i + k] : w[k];
    printf("v_rcp_f64 worst relative error over 1.28 M values in (0, 1]: raw %.3e, one Newton step %.3e, two %.3e\n", w[0], w[1], w[2]);
  }
  double *o; unsigned long long *c, h;
  hipMalloc(&o, 1 << 20); hipMalloc(&c, 8);
  auto run = [&](const char *name, auto launch, int instr) {
    for (int w : {1, 2, 4, 8, 16}) { // waves per workgroup (one workgroup): 4 per CU = 1 per SIMD, 8 = 2 per SIMD
      for (int it = 0; it < 3; it++) launch(w);
      hipDeviceSynchronize();
      hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
      printf("%-28s %2d waves in the workgroup: %.2f cycles per instruction of a wave\n", name, w, (double)h / (REP * instr));
    }
  };
  run("v_fma_f64 dependent (ILP 1)", [&](int w) { fma_indep<1><<<1, 64 * w>>>(o, 1.0000001, 1e-9, c); }, 4);
  run("v_fma_f64 ILP 2", [&](int w) { fma_indep<2><<<1, 64 * w>>>(o, 1.0000001, 1e-9, c); }, 8);
  run("v_fma_f64 ILP 4", [&](int w) { fma_indep<4><<<1, 64 * w>>>(o, 1.0000001, 1e-9, c); }, 16);
  run("v_fma_f64 ILP 8", [&](int w) { fma_indep<8><<<1, 64 * w>>>(o, 1.0000001, 1e-9, c); }, 32);
  run("v_fma_f32 ILP 8", [&](int w) { f32_indep<<<1, 64 * w>>>((float *)o, 1.0000001f, 1e-9f, c); }, 32);
  run("2 readlane + 4 fma_f64", [&](int w) { readlane_use<<<1, 64 * w>>>(o, c); }, 24);
  run("dpp wave_shr + add, ILP 8", [&](int w) { dpp_shift<<<1, 64 * w>>>((int *)o, c); }, 64);
  return 0;
}
