// Does a 16-byte buffer load / LDS-DMA (buffer_load_dwordx4 ... lds) range-check per dword?  A window of complex fp32
// samples starts and ends on 8-byte boundaries, so a lane's 16 bytes can straddle either end of the descriptor's range.
// Build: hipcc --offload-arch=gfx950 -O3 gldsprobe.hip -o gldsprobe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *src, float *out, int bytes)
{
  __shared__ __attribute__((aligned(16))) float buf[64 * 4 * 2];
  __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, bytes, 0x00020000);
  const int t = threadIdx.x;
  for (int i = t; i < 512; i += 64) buf[i] = -7.f;
  __syncthreads();
  // lane t reads 16 bytes at byte offset 16 t - 8: lane 0 straddles the start (offset -8), the lane at the end straddles the end
  const int off = 16 * t - 8;
  const u4 r = __builtin_amdgcn_raw_buffer_load_b128(d, off, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (lds_void *)buf, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int c = 0; c < 4; c++) {
    out[t * 8 + c] = __uint_as_float(r[c]);
    out[t * 8 + 4 + c] = buf[t * 4 + c];
  }
}
int main()
{
  const int n = 30; // floats: 120 bytes; lane 7 reads bytes 104..119 (in), lane 8 reads 120..135 (out), choose n so a lane straddles: bytes 16t-8 .. 16t+7
  float h[64], *d, *o, ho[512];
  for (int i = 0; i < 64; i++) h[i] = 100.f + i;
  hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, n * 4 - 8 + 8); // 120 bytes: lane 8 covers 120..135 fully out; lane 7: 104..119 fully in.  Use 116 for a straddle:
  hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
  printf("range 120 bytes\n");
  for (int t : {0, 1, 7, 8}) { printf("lane %d reg:", t); for (int c = 0; c < 4; c++) printf(" %g", ho[t * 8 + c]); printf("   lds:"); for (int c = 0; c < 4; c++) printf(" %g", ho[t * 8 + 4 + c]); printf("\n"); }
  k<<<1, 64>>>(d, o, 116); // lane 7 covers bytes 104..119: dwords 26, 27, 28 in range (< 116), dword 29 out
  hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
  printf("range 116 bytes (lane 7: three dwords in, one out)\n");
  for (int t : {0, 6, 7, 8}) { printf("lane %d reg:", t); for (int c = 0; c < 4; c++) printf(" %g", ho[t * 8 + c]); printf("   lds:"); for (int c = 0; c < 4; c++) printf(" %g", ho[t * 8 + 4 + c]); printf("\n"); }
  return 0;
}
