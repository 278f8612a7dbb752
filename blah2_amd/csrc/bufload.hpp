// Raw buffer loads for the correlation kernels (gfx950 only; device code).
#pragma once

#include <hip/hip_runtime.h>

#include "range_core.hpp"

namespace blah2 {

// ---- raw buffer loads --------------------------------------------------------
// A buffer descriptor carries (base, num_records); a load whose byte offset is
// >= num_records -- or negative, i.e. huge as an unsigned -- returns 0 without
// touching memory.  That is exactly the zero padding of the segment windows:
//   x': descriptor = [pulse + s*segLen, + min(segLen, nCorr - s*segLen)) samples
//   y': descriptor = the whole pulse, offset = (s*segLen + delayMin + m) samples
// so the 32 loads of a segment need no clamp, no select and no per-load address
// arithmetic: one VGPR offset per channel, the k-dependent part in the
// instruction's immediate (12 bits) and, in steps of 4 KiB, in a handful of VGPR
// adds (y) or in soffset (x).  Measured on gfx950 (tools/membench/bufprobe.hip):
// the range check sees voffset + immediate (32-bit wrap) + soffset, and a negative
// voffset stays out of range whatever soffset is -- which is why y, whose offset
// can be negative, keeps everything in voffset + immediate.
// Written as inline asm: the b64 builtin of this clang loads a single dword, and
// pairs of b32 builtins do not merge once LICM has hoisted `offset + 4`.  The
// compiler does not count these loads in its s_waitcnt bookkeeping; bufwait<N>()
// is the explicit wait and ties the destination registers to it.
typedef int b2_v4i __attribute__((ext_vector_type(4)));
typedef float b2_v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ b2_v4i make_rsrc(const void *base, int bytes)
{
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  b2_v4i d;
  d.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  d.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) & 0xffff; // stride 0: raw buffer
  d.z = __builtin_amdgcn_readfirstlane(bytes);
  d.w = 0x00020000;
  return d;
}

// one channel's sample format: byte stride between samples, the raw register type and its conversion
struct ChanC32 {
  static constexpr int STRIDE = 8;
  using raw = b2_v2f;
  template <int IMM> static __device__ __forceinline__ void ld(raw &r, b2_v4i d, int voff, int soff)
  {
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:%4" : "=v"(r) : "v"(voff), "s"(d), "s"(soff), "n"(IMM) : "memory");
  }
  static __device__ __forceinline__ cf cvt(raw r) { return cmake(r.x, r.y); }
};
struct ChanI16 { // one tuner's (I, Q) int16 pair inside the 8-byte I1 Q1 I2 Q2 word
  static constexpr int STRIDE = 8;
  using raw = uint32_t;
  template <int IMM> static __device__ __forceinline__ void ld(raw &r, b2_v4i d, int voff, int soff)
  {
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(r) : "v"(voff), "s"(d), "s"(soff), "n"(IMM) : "memory");
  }
  static __device__ __forceinline__ cf cvt(raw r) { return cmake((float)(int16_t)(r & 0xffffu), (float)(int16_t)(r >> 16)); }
};
struct ChanF16 {
  static constexpr int STRIDE = 4;
  using raw = uint32_t;
  template <int IMM> static __device__ __forceinline__ void ld(raw &r, b2_v4i d, int voff, int soff)
  {
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(r) : "v"(voff), "s"(d), "s"(soff), "n"(IMM) : "memory");
  }
  static __device__ __forceinline__ cf cvt(raw r)
  {
    return cmake((float)__builtin_bit_cast(_Float16, (uint16_t)(r & 0xffffu)), (float)__builtin_bit_cast(_Float16, (uint16_t)(r >> 16)));
  }
};

// the two channels of an input format: X = reference, Y = surveillance
template <class In> struct BufLoad;
template <> struct BufLoad<InC32> {
  using X = ChanC32;
  using Y = ChanC32;
  static __device__ __forceinline__ const void *xp(const InC32 &in, int64_t i) { return in.x + i; }
  static __device__ __forceinline__ const void *yp(const InC32 &in, int64_t i) { return in.y + i; }
};
template <> struct BufLoad<InI16> {
  using X = ChanI16;
  using Y = ChanI16;
  static __device__ __forceinline__ const void *xp(const InI16 &in, int64_t i) { return in.iq + 4 * i; }
  static __device__ __forceinline__ const void *yp(const InI16 &in, int64_t i) { return in.iq + 4 * i + 2; }
};
template <> struct BufLoad<InF16> {
  using X = ChanF16;
  using Y = ChanF16;
  static __device__ __forceinline__ const void *xp(const InF16 &in, int64_t i) { return in.x + 2 * i; }
  static __device__ __forceinline__ const void *yp(const InF16 &in, int64_t i) { return in.y + 2 * i; }
};
template <> struct BufLoad<InI16C32> {
  using X = ChanI16;
  using Y = ChanC32;
  static __device__ __forceinline__ const void *xp(const InI16C32 &in, int64_t i) { return in.iq + 4 * i; }
  static __device__ __forceinline__ const void *yp(const InI16C32 &in, int64_t i) { return in.y + i; }
};

// wait until at most N of the loads issued so far are outstanding; r[0..E) are
// tied to the wait so that nothing reads them earlier
template <int N, int E, class R> __device__ __forceinline__ void bufwait(R *r)
{
  static_assert(E == 4 || E == 8 || E == 9 || E == 16, "");
  if constexpr (E == 4)
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N) : "memory");
  else if constexpr (E == 9)
    asm volatile("s_waitcnt vmcnt(%9)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8])
                 : "n"(N)
                 : "memory");
  else if constexpr (E == 16)
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                   "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "n"(N)
                 : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "n"(N)
                 : "memory");
}

// k-th load of a channel, byte offset voff + k*STEP: the multiple of 4096 goes to
// soffset (SOFF: x, offsets never negative) or must already be in vbase[k*STEP/4096] (y)
template <class Chan, int STEP, int E, bool SOFF, int K = 0>
__device__ __forceinline__ void bufload_chan(typename Chan::raw *r, b2_v4i d, const int *vbase)
{
  if constexpr (K < E) {
    constexpr int OFF = K * STEP;
    if constexpr (SOFF) Chan::template ld<(OFF & 4095)>(r[K], d, vbase[0], OFF & ~4095);
    else Chan::template ld<(OFF & 4095)>(r[K], d, vbase[OFF >> 12], 0);
    bufload_chan<Chan, STEP, E, SOFF, K + 1>(r, d, vbase);
  }
}

// ---- the same through the compiler's builtins ---------------------------------------------------
// For kernels that keep loads in flight across other work (the prefetching range kernel) or mix loads and stores
// freely: the compiler counts these itself, so the waits between issue and use are its own.
typedef unsigned b2_v2u __attribute__((ext_vector_type(2)));
template <class Chan> struct RawBuiltin;
template <> struct RawBuiltin<ChanC32> {
  using raw = b2_v2u;
  static __device__ __forceinline__ raw ld(__amdgpu_buffer_rsrc_t d, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b64(d, voff, soff, 0); }
  static __device__ __forceinline__ cf cvt(raw r) { return cmake(__uint_as_float(r.x), __uint_as_float(r.y)); } // not __builtin_bit_cast on a vector element: this clang reads element 0 for both
};
template <> struct RawBuiltin<ChanI16> {
  using raw = unsigned;
  static __device__ __forceinline__ raw ld(__amdgpu_buffer_rsrc_t d, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b32(d, voff, soff, 0); }
  static __device__ __forceinline__ cf cvt(raw r) { return ChanI16::cvt(r); }
};
template <> struct RawBuiltin<ChanF16> {
  using raw = unsigned;
  static __device__ __forceinline__ raw ld(__amdgpu_buffer_rsrc_t d, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b32(d, voff, soff, 0); }
  static __device__ __forceinline__ cf cvt(raw r) { return ChanF16::cvt(r); }
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_b(const void *base, int bytes)
{
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(u), (short)0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// complex fp32 store; out-of-range offsets (negative included) are dropped by the range check
__device__ __forceinline__ void bufstore_c32(__amdgpu_buffer_rsrc_t d, int voff, cf v, int soff = 0)
{
  b2_v2u r;
  r.x = __float_as_uint(v.x);
  r.y = __float_as_uint(v.y);
  __builtin_amdgcn_raw_buffer_store_b64(r, d, voff, soff, 0);
}

} // namespace blah2
