// VALU issue rate on gfx950: wave64 instructions per cycle per SIMD for independent streams of
// v_fma_f32 / v_add_f32 / v_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 at 1, 2, 4, 8 waves per SIMD.
// (What bounds the FFT kernels: DESIGN.md section 4.)  Build: hipcc --offload-arch=gfx950 -O3 issuerate.hip -o issuerate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int MODE> __global__ __launch_bounds__(64) void k(float *out, int iters, float seed)
{
  float a[8];
  v2f p[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; p[i] = {a[i], a[i] + 1.f}; }
  const float c = seed * 0.5f, d = seed + 0.25f;
  const v2f pc = {c, d};
  for (int it = 0; it < iters; it++) {
    // 64 instructions per iteration, 8 independent chains
    if (MODE == 0) { REP8(for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));) }
    if (MODE == 1) { REP8(for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));) }
    if (MODE == 2) { REP8(for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));) }
    if (MODE == 3) { REP8(for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));) }
    if (MODE == 4) { REP8(for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));) }
    if (MODE == 5) { REP8(for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %2" : "+v"(a[i]) : "v"(c), "v"(d));) } // 128 per iteration, dependent pairs
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
  if (s == 1234.5f) out[0] = s;
}
template <int MODE> void run(const char *name, int perIter)
{
  float *out; hipMalloc(&out, 64);
  const int iters = 4000;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  for (int wps : {1, 2, 4, 8}) {
    const int grid = prop.multiProcessorCount * 4 * wps;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<grid, 64>>>(out, iters, 1.0f); hipDeviceSynchronize();
    hipEventRecord(a); k<MODE><<<grid, 64>>>(out, iters, 1.0f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)grid * iters * perIter;                       // wave-instructions
    const double simds = prop.multiProcessorCount * 4.0;
    const double clk = prop.clockRate * 1e3;                                   // Hz (nominal)
    std::printf("%-14s waves/SIMD %d: %.3f ms  %.3f wave-instr/cycle/SIMD at the nominal %.2f GHz  (%.1f G wave-instr/s/SIMD)\n", name, wps, ms,
                instr / (ms * 1e-3) / simds / clk, clk / 1e9, instr / (ms * 1e-3) / simds / 1e9);
  }
}
int main()
{
  run<0>("v_fma_f32", 64); run<1>("v_add_f32", 64); run<2>("v_mul_f32", 64); run<3>("v_pk_fma_f32", 64); run<4>("v_pk_add_f32", 64); run<5>("add;sub dep", 128);
  return 0;
}
