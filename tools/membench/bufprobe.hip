// What does a raw buffer load return on gfx950 for offsets around num_records,
// with the offset split between voffset / soffset / the immediate, and for
// "negative" (wrapped) voffsets?  Prints one line per case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const unsigned *buf, unsigned *out, int nrec_bytes, const int *voffs, const int *soffs, int ncase)
{
  // base points 1024 dwords into the allocation so that small negative offsets stay inside it
  __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void *)(buf + 1024), (short)0, nrec_bytes, 0x00020000);
  for (int c = 0; c < ncase; c++) {
    const int vo = voffs[c], so = __builtin_amdgcn_readfirstlane(soffs[c]);
    out[c * 4 + 0] = __builtin_amdgcn_raw_buffer_load_b32(d, vo, so, 0);
    out[c * 4 + 1] = __builtin_amdgcn_raw_buffer_load_b32(d, vo + 4, so, 0); // merged into dwordx2 or not
    out[c * 4 + 2] = 0;
    out[c * 4 + 3] = 0;
  }
}

// same with the constant part in the instruction's immediate offset field
__global__ void probe_imm(const unsigned *buf, unsigned *out, int nrec_bytes, const int *voffs, int ncase)
{
  __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void *)(buf + 1024), (short)0, nrec_bytes, 0x00020000);
  for (int c = 0; c < ncase; c++) {
    const int vo = voffs[c];
    out[c] = __builtin_amdgcn_raw_buffer_load_b32(d, vo + 1024, 0, 0); // compiler may fold +1024 into the immediate
  }
}

int main()
{
  const int N = 1 << 16;
  std::vector<unsigned> h(N);
  for (int i = 0; i < N; i++) h[i] = 0xA0000000u + i; // dword i of the allocation
  unsigned *d, *o;
  int *dv, *ds;
  hipMalloc(&d, N * 4); hipMalloc(&o, 4096 * 4); hipMalloc(&dv, 256 * 4); hipMalloc(&ds, 256 * 4);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  const int nrec = 4000; // bytes: dwords 0..999 of the view (allocation dwords 1024..2023) are in range
  struct C { int vo, so; const char *what; };
  std::vector<C> cs = {
    {0, 0, "first"}, {3992, 0, "last pair in range"}, {3996, 0, "second dword OOB"}, {4000, 0, "first OOB"},
    {0, 3992, "soffset only, in range"}, {0, 4000, "soffset only, OOB"}, {0, 8192, "soffset > num_records"},
    {1000, 2992, "split, in range (sum 3992)"}, {1000, 3000, "split, OOB (sum 4000)"}, {2000, 8192, "split, soffset > num_records"},
    {-80, 0, "negative voffset"}, {-80, 1024, "negative voffset + soffset 1024 (true offset 944)"},
    {-80, 80, "negative voffset + soffset 80 (true offset 0)"}, {-4, 0, "voffset -4"},
  };
  std::vector<int> vo, so;
  for (auto &c : cs) { vo.push_back(c.vo); so.push_back(c.so); }
  hipMemcpy(dv, vo.data(), vo.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(ds, so.data(), so.size() * 4, hipMemcpyHostToDevice);
  hipMemset(o, 0xFF, 4096 * 4);
  probe<<<1, 1>>>(d, o, nrec, dv, ds, (int)cs.size());
  std::vector<unsigned> r(4096);
  hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
  std::printf("num_records = %d bytes; in-range data reads 0xA0000400 + offset/4\n", nrec);
  for (size_t c = 0; c < cs.size(); c++)
    std::printf("voffset %6d soffset %6d : %08x %08x   %s\n", cs[c].vo, cs[c].so, r[c * 4], r[c * 4 + 1], cs[c].what);
  hipMemset(o, 0xFF, 4096 * 4);
  probe_imm<<<1, 1>>>(d, o, nrec, dv, (int)cs.size());
  hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
  for (size_t c = 0; c < cs.size(); c++)
    std::printf("voffset %6d + imm 1024 : %08x\n", cs[c].vo, r[c]);
  std::printf("hip status: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
