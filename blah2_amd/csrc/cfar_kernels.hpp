// Detector kernels of the blah2 engine (gfx950 / MI355X only).
//
//   cfar1d_kernel                  CfarDetector1D::process            CfarDetector1D.cpp:23-100
//   cfar2d_tile_kernel             2-D CA-CFAR (BASELINE configs[2]; extension): one read of the map
//   cfar2d_stream_kernel           the same in one pass without LDS arrays: a wave per strip of columns walks down the rows
//   sat_rows / sat_cols / cfar2d   the same detector through a summed-area table (windows beyond the tile kernel's halo)
//
// All paths are relative to /root/reference/src.
#pragma once

#include <hip/hip_runtime.h>

#include "blah2hip.h"
#include "fft_wg.hpp"
#include "trace.hpp"

namespace blah2 {

// --------------------------------------------------------------------------
// CfarDetector1D::process (CfarDetector1D.cpp:23-100): cell-averaging CFAR
// along delay for each Doppler row with |doppler| >= minDoppler.  One
// workgroup per row; |z|^2 of the row is staged in LDS as fp64 and the window
// sum runs in the reference's index order (leading cells need k > 0, trailing
// k >= 0, :61,:68).  alpha[n] = n*(pfa^(-1/n)-1) is tabulated on the host with
// the same libm pow the reference calls (:76).  Hits are appended through a
// per-CPI atomic counter; the host API sorts them into row-major order.
struct CfarArgs {
  const cf *map;         // [nCpi][nD][nDelay]
  const double *metrics; // [nCpi][2]
  const double *doppler; // [nD] Hz
  const double *alpha;   // [2*nTrain+1]
  const int32_t *delayAxis; // Map::delay, [nDelay]; nullptr = delayMin + j (the engine's own axis)
  blah2hip_hit_t *hits;  // [nCpi][cap]
  uint32_t *count;       // [nCpi]
  int32_t nD, nDelay, delayMin;
  int32_t nGuard, nTrain, minDelay;
  double minDoppler;
  uint32_t cap;
};

// STAGED = false: rows longer than the LDS holds as fp64 (more than 19 k delay bins; blah2hip_cfar1d_map on a host-built
// map): the same window sums with |z|^2 formed straight from the L2-resident row.
template <bool STAGED>
__global__ void cfar1d_kernel(CfarArgs a)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sq = reinterpret_cast<double *>(smem);
  const int row = blockIdx.x, cpi = blockIdx.y;
  if (fabs(a.doppler[row]) < a.minDoppler) return; // :40
  const cf *z = a.map + ((size_t)cpi * a.nD + row) * a.nDelay;
  auto sqv = [&](int k) -> double {
    if (STAGED) return sq[k];
    const cf c = z[k];
    return (double)c.x * (double)c.x + (double)c.y * (double)c.y;
  };
  if (STAGED) {
    for (int j = threadIdx.x; j < a.nDelay; j += blockDim.x) {
      const cf c = z[j];
      sq[j] = (double)c.x * (double)c.x + (double)c.y * (double)c.y; // |z*z| (:47)
    }
    __syncthreads();
  }
  const double noisePower = a.metrics[2 * cpi];
  for (int j = threadIdx.x; j < a.nDelay; j += blockDim.x) {
    if ((a.delayAxis ? a.delayAxis[j] : j + a.delayMin) < a.minDelay) continue; // :53  x->delay[j] < minDelay
    int n = 0;
    double tot = 0.0;
    for (int k = j - a.nGuard - a.nTrain; k < j - a.nGuard; k++)
      if (k > 0 && k < a.nDelay) { tot += sqv(k); n++; }
    for (int k = j + a.nGuard + 1; k < j + a.nGuard + a.nTrain + 1; k++)
      if (k >= 0 && k < a.nDelay) { tot += sqv(k); n++; }
    if (n == 0) continue; // alpha = 0*inf = NaN in the reference: never exceeds
    const double thr = a.alpha[n] * (tot / n);
    const double sj = sqv(j);
    if (sj > thr) {
      const uint32_t slot = atomicAdd(&a.count[cpi], 1u);
      if (slot < a.cap) {
        blah2hip_hit_t h;
        h.row = row;
        h.col = j;
        h.snr = 5.0 * log10(sj) - noisePower; // 10 log10|z| - noisePower (:48)
        a.hits[(size_t)cpi * a.cap + slot] = h;
      }
    }
  }
}

// --------------------------------------------------------------------------
// 2-D cell-averaging CFAR (BASELINE.json configs[2]; the reference only has the
// 1-D detector).  Definition: SURVEY.md section 8g / oracle cfar2d(): training
// cells = the (2(nGd+nTd)+1) x (2(nGf+nTf)+1) rectangle minus the guard box,
// in-bounds only, delay column 0 never trains (CfarDetector1D.cpp:61), statistic
// |z|^2, alpha = N (pfa^(-1/N) - 1).  With nGf = nTf = 0 it is the 1-D detector.
// Window sums come from an fp64 summed-area table built by two scan kernels.
struct Cfar2dArgs {
  const cf *map;         // [nCpi][nD][nDelay]
  const double *metrics; // [nCpi][2]
  const double *doppler; // [nD]
  const double *alpha;   // [maxN + 1]
  double *sat;           // [nCpi][nD + 1][nDelay + 1], row 0 and column 0 stay zero
  blah2hip_hit_t *hits;
  uint32_t *count;
  int32_t nD, nDelay, delayMin;
  int32_t ngD, ntD, ngF, ntF, minDelay;
  double minDoppler;
  uint32_t cap;
};

// row-wise inclusive prefix of |z|^2 (column 0 zeroed) into sat[i+1][1..]: the row is
// walked in coalesced chunks of 256 cells; inside a chunk a wave scans with shuffles,
// the four wave totals and the running carry are combined through LDS.
__global__ __launch_bounds__(256) void sat_rows_kernel(Cfar2dArgs a)
{
  __shared__ double wtot[2][4];
  const int row = blockIdx.x, cpi = blockIdx.y, t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;
  const cf *z = a.map + ((size_t)cpi * a.nD + row) * a.nDelay;
  double *out = a.sat + ((size_t)cpi * (a.nD + 1) + row + 1) * (a.nDelay + 1) + 1;
  double carry = 0.0;
  int buf = 0;
  for (int j0 = 0; j0 < a.nDelay; j0 += 256, buf ^= 1) {
    const int j = j0 + t;
    double val = 0.0;
    if (j < a.nDelay && j != 0) {
      const cf c = z[j];
      val = (double)c.x * (double)c.x + (double)c.y * (double)c.y;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double n = __shfl_up(val, off);
      if (lane >= off) val += n;
    }
    if (lane == 63) wtot[buf][wv] = val;
    __syncthreads(); // double-buffered totals: one barrier per chunk is enough
    double pre = carry;
#pragma unroll
    for (int w = 0; w < 4; w++) pre += (w < wv) ? wtot[buf][w] : 0.0;
    if (j < a.nDelay) out[j] = val + pre;
    carry += (wtot[buf][0] + wtot[buf][1]) + (wtot[buf][2] + wtot[buf][3]);
  }
}

// column-wise running sum of the row prefixes -> summed-area table.  One thread
// per column; rows are taken 16 at a time so that 16 independent loads are in
// flight before the (serial) running sum consumes them.
__global__ __launch_bounds__(64) void sat_cols_kernel(Cfar2dArgs a)
{
  const int j = blockIdx.x * 64 + threadIdx.x, cpi = blockIdx.y;
  if (j >= a.nDelay) return;
  const size_t W = (size_t)a.nDelay + 1;
  double *col = a.sat + (size_t)cpi * (a.nD + 1) * W + (j + 1);
  double run = 0.0;
  int i = 1;
  for (; i + 15 <= a.nD; i += 16) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = col[(size_t)(i + k) * W];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      run += v[k];
      col[(size_t)(i + k) * W] = run;
    }
  }
  for (; i <= a.nD; i++) {
    run += col[(size_t)i * W];
    col[(size_t)i * W] = run;
  }
}

__global__ __launch_bounds__(256) void cfar2d_kernel(Cfar2dArgs a)
{
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, cpi = blockIdx.z;
  if (j >= a.nDelay) return;
  if (fabs(a.doppler[i]) < a.minDoppler) return;
  if (j + a.delayMin < a.minDelay) return;
  const int nD = a.nD, nC = a.nDelay, W = nC + 1;
  const double *S = a.sat + (size_t)cpi * (nD + 1) * W;
  auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
  const int R0 = clampi(i - a.ngF - a.ntF, 0, nD), R1 = clampi(i + a.ngF + a.ntF + 1, 0, nD);
  const int G0 = clampi(i - a.ngF, 0, nD), G1 = clampi(i + a.ngF + 1, 0, nD);
  const int C0 = clampi(j - a.ngD - a.ntD, 0, nC), C1 = clampi(j + a.ngD + a.ntD + 1, 0, nC);
  const int H0 = clampi(j - a.ngD, 0, nC), H1 = clampi(j + a.ngD + 1, 0, nC);
  auto box = [&](int r0, int r1, int c0, int c1) {
    return S[(size_t)r1 * W + c1] - S[(size_t)r0 * W + c1] - S[(size_t)r1 * W + c0] + S[(size_t)r0 * W + c0];
  };
  auto cols = [](int c0, int c1) { return max(c1, 1) - max(c0, 1); }; // column 0 never trains
  const double tot = box(R0, R1, C0, C1) - box(G0, G1, H0, H1);
  const int n = (R1 - R0) * cols(C0, C1) - (G1 - G0) * cols(H0, H1);
  if (n <= 0) return;
  const cf c = a.map[((size_t)cpi * nD + i) * nC + j];
  const double sq = (double)c.x * (double)c.x + (double)c.y * (double)c.y;
  if (sq > a.alpha[n] * (tot / n)) {
    const uint32_t slot = atomicAdd(&a.count[cpi], 1u);
    if (slot < a.cap) {
      blah2hip_hit_t h;
      h.row = i;
      h.col = j;
      h.snr = 5.0 * log10(sq) - a.metrics[2 * cpi];
      a.hits[(size_t)cpi * a.cap + slot] = h;
    }
  }
}


// --------------------------------------------------------------------------
// The same detector in ONE pass over the map (the default for windows that fit the tile's halo budget;
// the summed-area-table kernels above moved 6.3 x the map and took twice the Doppler stage at cfg 3).
//
// A workgroup owns a tile of (64 - 2 hR) output rows x 64 output columns, hR = nGf + nTf Doppler rows and
// hC = nGd + nTd delay columns of halo around it.  |z|^2 of the tile + halo goes to LDS as fp64 (cells
// outside the map as 0 -- "in-bounds only" -- through raw buffer loads whose range check is that zero), and
// the window sum is taken separably and ADDITIVELY (no running differences and no prefix table: a |z|^2 that
// leaves a sliding sum leaves its rounding error behind, 1e-16 x peak^2 against a noise-level window):
//   rows   A[r][j] = sum over the training columns of row r  (nTd left of the guard + nTd right of it)
//          B[r][j] = A[r][j] + the 2 nGd + 1 guard columns    (the whole 2 hC + 1 window)
//   cols   tot[i][j] = sum_{nTf rows above the guard} B + sum_{2 nGf + 1 guard rows} A + sum_{nTf rows below} B
// Each sum runs over EIGHT neighbouring outputs at once out of a rotating eight-value register window
// (B2_STREAM8: W + 7 LDS reads for 8 x W additions), one continuous stream across the window; which
// accumulators a step feeds depends on the window position only, i.e. it is wave-uniform: a bit of a per-block
// scalar mask.  Lanes run across the other axis: lane <-> row while walking along rows (odd row pitch:
// conflict-free), lane <-> column while walking down columns.  The number of training cells follows from the
// clipped window like in cfar2d_kernel, and a cell that must not be tested (minDelay, minDoppler, outside the
// map) gets n = 0, whose threshold factor alpha[0] is NaN: it never exceeds.
// What bounds it: VALU instruction issue.  At two waves per SIMD every vector instruction of a thread costs
// 8 cycles of its tile, so the kernel is written for instruction COUNT (PMC of the first version: 822 vector and
// 960 scalar instructions per wave and tile, VALU 17 % busy, the rest branch and scalar latencies): raw buffer
// loads instead of clamped addresses, scalar masks instead of per-step compares, the tests without division
// (sq n > alpha tot) and without global loads (threshold table in LDS, Doppler axis through scalar loads).
// Workgroups are PERSISTENT: the next tile's cells are requested into registers once this tile's are
// in LDS and land during the two summation phases; XCD x walks a contiguous eighth of the tile
// sequence, so the halo a tile shares with its neighbours is served by that XCD's L2.
#ifndef C2T_ABLATE
#define C2T_ABLATE 0 /* tools/ only: 1 = no tile loads after the first, 2 = no row sums, 4 = no column sums / tests */
#endif
constexpr int C2T_ROWS = 64;                // tile rows incl. the halo (= lanes of the row phase)
constexpr int C2T_COLS = 64;                // output columns (= lanes of the column phase)
constexpr int C2T_WAVES = 8;                // row phase: eight output columns per wave; column phase: eight output rows per wave
constexpr int C2T_ABP = C2T_COLS + 1;       // row pitch of A and of B
constexpr int C2T_AB_ROWS = C2T_ROWS + 12;  // the batched loads of a run's last block read up to twelve rows beyond the last row used (row 75 at most)
constexpr int C2T_MAX_HR = 24, C2T_MAX_HC = 40;
__host__ __device__ inline int c2t_spitch(int hC) { return (C2T_COLS + 2 * hC) | 1; }
// S | cut0[64] | dopL[64] | alphaL[alphaN, even] | AB
__host__ __device__ inline int c2t_ab_offset(int hC, int alphaN) { return (C2T_ROWS * c2t_spitch(hC) + 2 * C2T_ROWS + alphaN + 1) & ~1; }
__host__ inline size_t c2t_lds_bytes(int hC, int alphaN)
{
  return ((size_t)c2t_ab_offset(hC, alphaN) + 2 * (size_t)C2T_AB_ROWS * C2T_ABP) * sizeof(double);
}

struct Cfar2dTileArgs {
  Cfar2dArgs d;
  int32_t nCpi, tilesX, tilesY, rowsOut; // rowsOut = 64 - 2 hR output rows per tile
  int32_t alphaLds;                      // entries of the threshold table staged in LDS (the whole table, or 0: read from L2)
};

// One run of N window positions through a rotating register window: at step p the eight values LD(p + m), m < 8,
// are handed to STEP, which adds them to the eight outputs' accumulators (N + 7 loads for 8 N additions).  Eight
// steps share ONE batch of eight loads issued up front -- one LDS round trip per eight steps; with a load per step
// the sums were a chain of LDS latencies -- and the steps of a block are straight-line code: the remainder block
// leaves through ONE forward branch after its last step.  (A first form selected the accumulator set per step
// from scalar masks: two taken branches per step, 960 scalar instructions per wave and tile.)  Positions up to
// N + 12 are read (the caller's arrays are padded for it); values beyond N + 6 are never used.
// A macro, not a function taking lambdas: the closures' by-reference accumulators were left in scratch.
#define B2_RUN8(TYPE, N, LD, STEP)                                                                  \
  do {                                                                                             \
    const int n_ = (N);                                                                            \
    if (n_ > 0) {                                                                                  \
      __label__ done_;                                                                             \
      TYPE o0 = LD(0), o1 = LD(1), o2 = LD(2), o3 = LD(3), o4 = LD(4), o5 = LD(5), o6 = LD(6);      \
      int p = 0;                                                                                   \
      _Pragma("unroll 1") for (; p + 8 <= n_; p += 8) {                                            \
        const TYPE n0 = LD(p + 7), n1 = LD(p + 8), n2 = LD(p + 9), n3 = LD(p + 10), n4 = LD(p + 11), n5 = LD(p + 12), \
                   n6 = LD(p + 13), n7 = LD(p + 14);                                               \
        STEP(o0, o1, o2, o3, o4, o5, o6, n0); STEP(o1, o2, o3, o4, o5, o6, n0, n1);                 \
        STEP(o2, o3, o4, o5, o6, n0, n1, n2); STEP(o3, o4, o5, o6, n0, n1, n2, n3);                 \
        STEP(o4, o5, o6, n0, n1, n2, n3, n4); STEP(o5, o6, n0, n1, n2, n3, n4, n5);                 \
        STEP(o6, n0, n1, n2, n3, n4, n5, n6); STEP(n0, n1, n2, n3, n4, n5, n6, n7);                 \
        o0 = n1; o1 = n2; o2 = n3; o3 = n4; o4 = n5; o5 = n6; o6 = n7;                              \
      }                                                                                            \
      const int r_ = n_ - p;                                                                       \
      if (r_ > 0) {                                                                                \
        const TYPE n0 = LD(p + 7), n1 = LD(p + 8), n2 = LD(p + 9), n3 = LD(p + 10), n4 = LD(p + 11), n5 = LD(p + 12); \
        STEP(o0, o1, o2, o3, o4, o5, o6, n0); if (r_ == 1) goto done_;                              \
        STEP(o1, o2, o3, o4, o5, o6, n0, n1); if (r_ == 2) goto done_;                              \
        STEP(o2, o3, o4, o5, o6, n0, n1, n2); if (r_ == 3) goto done_;                              \
        STEP(o3, o4, o5, o6, n0, n1, n2, n3); if (r_ == 4) goto done_;                              \
        STEP(o4, o5, o6, n0, n1, n2, n3, n4); if (r_ == 5) goto done_;                              \
        STEP(o5, o6, n0, n1, n2, n3, n4, n5); if (r_ == 6) goto done_;                              \
        { const TYPE n6 = LD(p + 13); STEP(o6, n0, n1, n2, n3, n4, n5, n6); }                       \
      }                                                                                            \
    done_:;                                                                                        \
    }                                                                                              \
  } while (0)

typedef unsigned c2t_v2u __attribute__((ext_vector_type(2)));

// NL: column loads per lane and row of the tile fill, ceil((64 + 2 hC) / 64); ALPHA_LDS: the threshold table is
// staged in LDS (a template parameter: as a run-time choice the compiler selected between the two POINTERS and
// issued a flat load, whose s_waitcnt vmcnt(0) waited for the whole tile prefetch in the first test)
template <int NL, bool ALPHA_LDS>
__global__ __launch_bounds__(64 * C2T_WAVES) void cfar2d_tile_kernel(Cfar2dTileArgs ta)
{
  const Cfar2dArgs &a = ta.d;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int hR = a.ngF + a.ntF, hC = a.ngD + a.ntD;
  const int SP = c2t_spitch(hC), SC = C2T_COLS + 2 * hC;
  double *S = reinterpret_cast<double *>(smem);
  double *cut0 = S + C2T_ROWS * SP; // |z|^2 of delay column 0 while it is zeroed in S (it never trains), per tile row
  // per tile row: rows of the clipped window and of the clipped guard box of the cell it holds, both 0 when that cell is not
  // tested here (halo rows, rows outside the map, |doppler| < minDoppler).  Worked out by wave 0 from the Doppler axis it
  // requests WITH the tile (any vector-memory wait in the test phase would wait for the whole prefetch: loads return in order)
  int2 *rowsL = reinterpret_cast<int2 *>(cut0 + C2T_ROWS);
  double *alphaL = cut0 + 2 * C2T_ROWS; // the threshold table (when it fits)
  // 16-byte aligned behind them (an even number of doubles from the 16-byte aligned base; no integer round trip of the
  // pointer: that would lose the LDS address space and turn every access into a flat one)
  double *A = S + c2t_ab_offset(hC, ta.alphaLds); // training-column sums of each tile row
  double *B = A + C2T_AB_ROWS * C2T_ABP;          // whole-window sums
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nD = a.nD, nC = a.nDelay;
  constexpr int RPW = C2T_ROWS / C2T_WAVES; // rows of the fill per wave
  if (ALPHA_LDS)
    for (int e = threadIdx.x; e < ta.alphaLds; e += 64 * C2T_WAVES) alphaL[e] = a.alpha[e]; // visible after the first barrier

  // XCD-aware tile walk (gridDim.x is a multiple of 8)
  const int tilesPerCpi = ta.tilesX * ta.tilesY;
  const int nAll = tilesPerCpi * ta.nCpi;
  const int chunk = (nAll + 7) >> 3;
  const int base = (blockIdx.x & 7) * chunk;
  const int cnt = min(chunk, nAll - base);
  const int step = gridDim.x >> 3;

  // tile fill: one raw buffer descriptor per map row (num_records = the row, or 0 for a row outside the map), byte
  // offset 8 (j0 - hC + lane) + 512 l -- negative or beyond the row reads as 0 without touching memory
  c2t_v2u nt[RPW][NL];
  double ndop;
  const bool last_l = lane + 64 * (NL - 1) < SC; // the last load of a row covers SC - 64 (NL - 1) columns
  auto tile_load = [&](int t) {
    const int cpi = t / tilesPerCpi, rem = t - cpi * tilesPerCpi;
    const int ty = rem / ta.tilesX, tx = rem - ty * ta.tilesX;
    const int iBase = ty * ta.rowsOut - hR + wave * RPW;
    const int voff = (tx * C2T_COLS - hC + lane) * 8;
    ndop = a.doppler[min(max(ty * ta.rowsOut - hR + lane, 0), nD - 1)];
#pragma unroll
    for (int k = 0; k < RPW; k++) {
      const int i = iBase + k;
      const bool rok = i >= 0 && i < nD;
      const cf *row = a.map + ((size_t)cpi * nD + (rok ? i : 0)) * nC;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)row, (short)0, rok ? nC * 8 : 0, 0x00020000);
#pragma unroll
      for (int l = 0; l < NL - 1; l++) nt[k][l] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 512 * l, 0, 0);
      if (last_l) nt[k][NL - 1] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 512 * (NL - 1), 0, 0);
    }
  };

#ifdef C2T_TRACE
  uint64_t tr[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime();
#define C2_T(k) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); tr[k] += now_ - t0_; t0_ = now_; }
#else
#define C2_T(k)
#endif
  int r = blockIdx.x >> 3;
  if (r < cnt) tile_load(base + r);
  for (; r < cnt; r += step) {
    C2_T(0)
    const int t = base + r;
    const int cpi = t / tilesPerCpi, rem = t - cpi * tilesPerCpi;
    const int ty = rem / ta.tilesX, tx = rem - ty * ta.tilesX;
    const int i0 = ty * ta.rowsOut, j0 = tx * C2T_COLS;
    const int c0 = hC - j0;                  // tile column of delay column 0
    const bool hasCol0 = c0 >= 0 && c0 < SC; // (the tiles of the first tile column, and their right neighbours when hC > 0)
    // phase 0: |z|^2 of tile + halo -> S (fp64)
    {
      if (wave == 0) {
        const int i = i0 - hR + lane; // the map row this tile row holds
        const bool live = lane >= hR && lane < hR + ta.rowsOut && i < nD && !(fabs(ndop) < a.minDoppler); // CfarDetector1D.cpp:40
        auto clampr = [nD](int v_) { return v_ < 0 ? 0 : (v_ > nD ? nD : v_); };
        rowsL[lane] = live ? make_int2(clampr(i + hR + 1) - clampr(i - hR), clampr(i + a.ngF + 1) - clampr(i - a.ngF)) : make_int2(0, 0);
      }
      double *srow = S + (wave * RPW) * SP + lane;
#pragma unroll
      for (int k = 0; k < RPW; k++) {
#pragma unroll
        for (int l = 0; l < NL; l++) {
          const double x = (double)__uint_as_float(nt[k][l].x), y = (double)__uint_as_float(nt[k][l].y);
          if (l < NL - 1 || last_l) srow[k * SP + 64 * l] = x * x + y * y;
        }
      }
      // delay column 0 never trains (CfarDetector1D.cpp:61): 0 during the row sums, its own value set aside
      if (hasCol0 && lane < RPW) {
        double *p0 = S + (wave * RPW + lane) * SP + c0;
        cut0[wave * RPW + lane] = *p0;
        *p0 = 0.0;
      }
    }
    C2_T(1)
    __syncthreads();
    C2_T(2)
#if !(C2T_ABLATE & 1)
    if (r + step < cnt) tile_load(base + r + step);
#endif
    C2_T(3)

    // phase 1: row sums.  lane <-> row, wave <-> eight output columns; window position p of output m is
    // S column 8 wave + m + p: training for p < nTd and p > nTd + 2 nGd, guard in between
#if !(C2T_ABLATE & 2)
    {
      const double *v = S + lane * SP + wave * 8;
      double t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0;
      double u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0, u5 = 0, u6 = 0, u7 = 0;
#define B2_T1(A0, A1, A2, A3, A4, A5, A6, A7) t0 += A0; t1 += A1; t2 += A2; t3 += A3; t4 += A4; t5 += A5; t6 += A6; t7 += A7
#define B2_U1(A0, A1, A2, A3, A4, A5, A6, A7) u0 += A0; u1 += A1; u2 += A2; u3 += A3; u4 += A4; u5 += A5; u6 += A6; u7 += A7
#define B2_LD1(P) v[P]
      B2_RUN8(double, a.ntD, B2_LD1, B2_T1);           // training columns left of the guard
      v += a.ntD;
      B2_RUN8(double, 2 * a.ngD + 1, B2_LD1, B2_U1);   // guard columns (and the cell itself)
      v += 2 * a.ngD + 1;
      B2_RUN8(double, a.ntD, B2_LD1, B2_T1);           // training columns right of it
#undef B2_LD1
#undef B2_T1
#undef B2_U1
      C2_T(9)
      double *pa = A + lane * C2T_ABP + wave * 8, *pb = B + lane * C2T_ABP + wave * 8;
      pa[0] = t0; pa[1] = t1; pa[2] = t2; pa[3] = t3; pa[4] = t4; pa[5] = t5; pa[6] = t6; pa[7] = t7;
      pb[0] = t0 + u0; pb[1] = t1 + u1; pb[2] = t2 + u2; pb[3] = t3 + u3;
      pb[4] = t4 + u4; pb[5] = t5 + u5; pb[6] = t6 + u6; pb[7] = t7 + u7;
    }
#endif
    C2_T(4)
    __syncthreads();
    C2_T(5)

    // phase 2: column sums, threshold, hits.  lane <-> column, wave <-> eight output rows; window position p of
    // output m is tile row 8 wave + m + p: B above and below the guard rows, A inside them
#if !(C2T_ABLATE & 4)
    if (8 * wave < ta.rowsOut) {
      const int j = j0 + lane, o = 8 * wave;
      // the test cells of this wave's rows: column 0 gets its value back (the row sums are done; only this wave reads these rows)
      if (hasCol0 && lane < 8 && o + lane + hR < C2T_ROWS) S[(o + lane + hR) * SP + c0] = cut0[o + lane + hR];
      double q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0, q7 = 0;
#define B2_Q(A0, A1, A2, A3, A4, A5, A6, A7) q0 += A0; q1 += A1; q2 += A2; q3 += A3; q4 += A4; q5 += A5; q6 += A6; q7 += A7
#define B2_LD2(P) v[(P) * C2T_ABP]
      const double *v = B + o * C2T_ABP + lane;
      B2_RUN8(double, a.ntF, B2_LD2, B2_Q);            // rows above the guard: the whole window of each
      v = A + (o + a.ntF) * C2T_ABP + lane;
      B2_RUN8(double, 2 * a.ngF + 1, B2_LD2, B2_Q);    // guard rows: their training columns only
      v = B + (o + a.ntF + 2 * a.ngF + 1) * C2T_ABP + lane;
      B2_RUN8(double, a.ntF, B2_LD2, B2_Q);            // rows below
#undef B2_LD2
#undef B2_Q
      C2_T(8)
      // The eight tests: every LDS operand is requested first, the eight comparisons are branch-free, and the (rare)
      // hits are appended afterwards.  n = 0 for a cell that is not tested (minDelay, minDoppler, outside the map):
      // alpha[0] is NaN, so it never exceeds (like a cell without training cells, CfarDetector1D.cpp:76).
      // sq > alpha (tot / n) is evaluated as sq n > alpha tot: no fp64 division.
      auto clampi = [](int v_, int lo, int hi) { return v_ < lo ? lo : (v_ > hi ? hi : v_); };
      auto cols = [](int c0_, int c1_) { return max(c1_, 1) - max(c0_, 1); }; // column 0 never trains
      const bool jok = j < nC && j + a.delayMin >= a.minDelay; // CfarDetector1D.cpp:53
      const int nColsAll = jok ? cols(clampi(j - hC, 0, nC), clampi(j + hC + 1, 0, nC)) : 0;
      const int nColsGuard = jok ? cols(clampi(j - a.ngD, 0, nC), clampi(j + a.ngD + 1, 0, nC)) : 0;
      const double tot[8] = {q0, q1, q2, q3, q4, q5, q6, q7};
      const double *cut = S + (o + hR) * SP + lane + hC;
      double sq[8], al[8];
      int nn[8];
      int2 rw[8];
#pragma unroll
      for (int m = 0; m < 8; m++) {
        sq[m] = cut[m * SP];
        rw[m] = rowsL[min(o + m + hR, C2T_ROWS - 1)];
      }
#pragma unroll
      for (int m = 0; m < 8; m++) {
        nn[m] = o + m < ta.rowsOut ? rw[m].x * nColsAll - rw[m].y * nColsGuard : 0; // >= 0: the guard box lies inside the window
        al[m] = ALPHA_LDS ? alphaL[nn[m]] : a.alpha[nn[m]];
      }
      uint32_t hitMask = 0;
#pragma unroll
      for (int m = 0; m < 8; m++) hitMask |= (sq[m] * (double)nn[m] > al[m] * tot[m]) ? (1u << m) : 0u;
      if (hitMask) {
        const double noise = a.metrics[2 * cpi];
        for (int m = 0; m < 8; m++) {
          if (!((hitMask >> m) & 1u)) continue;
          const uint32_t slot = atomicAdd(&a.count[cpi], 1u);
          if (slot < a.cap) {
            blah2hip_hit_t h;
            h.row = i0 + o + m;
            h.col = j;
            h.snr = 5.0 * log10(sq[m]) - noise;
            a.hits[(size_t)cpi * a.cap + slot] = h;
          }
        }
      }
    }
#endif
    C2_T(6)
    __syncthreads(); // S and AB are rewritten by the next tile
    C2_T(7)
  }
#ifdef C2T_TRACE // buckets: loop + decode, fill, barrier 1, issue, row stores, barrier 2, tests, barrier 3, column sums, row sums
  if (lane == 0) trace_finish("c2t", tr, blockIdx.x == 0 && threadIdx.x == 0);
#endif
}

// --------------------------------------------------------------------------
// The one-pass detector as a STREAM, without barriers (round 4; the default for the window shapes instantiated in
// C2S_SHAPES, the tile kernel above takes the others).  A WAVE owns a strip of 64 - 2 hC output columns (lane <-> column,
// hC halo lanes on either side) and walks down a segment of Doppler rows, one 512-byte row piece per step, U row pieces
// requested ahead.  Per row and lane, everything in fp64 and ADDITIVE (no running differences) like in the tile kernel:
//   along the row (across lanes): the lane's |z|^2 and its sum with the right neighbour's go to the wave's own piece of
//   LDS; the nTd training columns either side of the guard and the 2 nGd + 1 guard columns come back as pair sums
//   (and one cell where a count is odd) through ds_read_b64 at immediate offsets:
//     A(r, j) = the training columns of row r          B(r, j) = A + the guard columns = the whole window of row r
//   (2 writes + 10 reads and 10 additions for the 17-column window; see the comment in the kernel for why not
//   ds_bpermute and not ds_read2_b64)
//   down the column (in registers, a ring per quantity whose slots are compile-time after unrolling U rows):
//     TB(r) = B(r) + ... + B(r - nTf + 1),  TA(r) = A(r) + ... + A(r - 2 nGf)
//     tot(i = r - hR) = TB(r) + TA(r - nTf) + TB(r - nTf - 2 nGf - 1)
//   (the block of nTf rows is summed once and used twice): 6 additions per cell for the 9 rows of the 17 x 9 window.
// The tile kernel makes 26 additions per cell out of LDS between three workgroup barriers per tile.
// The cell under test is the lane's own |z|^2 of hR rows ago; n, alpha[n] and the test are the tile kernel's
// (sq n > alpha tot; n = 0 -> alpha = NaN -> never).  n depends on the row only through the clipped row counts,
// which are wave-uniform: alpha[n] is fetched (from LDS) when they change (the first and last hR rows of the map).
// Four waves (four neighbouring strips) share a workgroup only for the halo columns' sake (one L1); workgroups
// are laid out so that an XCD walks a contiguous range of strips and segments (neighbours' halos in its L2).
// What bounds it: the map's HBM stream and the LDS / VALU work, each about half of the time,
// not fully overlapped at four waves per SIMD.
#ifndef C2S_ABLATE
#define C2S_ABLATE 0 /* tools/ only: 1 = every row piece is the segment's first (cache hits), 2 = no sums along the row */
#endif
#ifndef C2S_V
#define C2S_V 2 /* rows whose shifts along the row are in flight together (tools/ builds other values for comparison) */
#endif
// ring length: a whole number of blocks of C2S_V rows that holds a block and the nTf + 2 nGf + 1 rows behind it
constexpr int c2s_ring(int ntf, int ngf) { return (C2S_V + ntf + 2 * ngf + 1 + C2S_V - 1) / C2S_V * C2S_V; }
constexpr int C2S_PITCH = 64 + 32;
// what the lanes of this wave wrote to LDS is visible to its other lanes (no instruction: DS operations of a wave execute in
// order; the fences keep the compiler from moving, merging or forwarding the accesses across this point)
// one ds_read_b64 (2 cycles of the CU's LDS on gfx950; as plain loads the compiler pairs them into ds_read2_b64: 8 cycles)
#define C2S_LD(P) __hip_atomic_load((P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)
#define C2S_LANES_SEE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

struct Cfar2dStreamArgs {
  Cfar2dArgs d;
  int32_t nCpi, strips, segs, rowsPerSeg, nTasks;
  int32_t deadLo, deadHi; // the rows [deadLo, deadHi) are not tested: |doppler| < minDoppler there and nowhere else (CfarDetector1D.cpp:40)
};

template <int NTD, int NGD, int NTF, int NGF>
__global__ __launch_bounds__(256) void cfar2d_stream_kernel(Cfar2dStreamArgs ta)
{
  const Cfar2dArgs &a = ta.d;
  constexpr int HC = NTD + NGD, HR = NTF + NGF, OUTW = 64 - 2 * HC;
  constexpr int GW = 2 * NGD + 1, GH = 2 * NGF + 1;
  constexpr int V = C2S_V, U = c2s_ring(NTF, NGF); // U: ring length = rows per round of the unrolled loop = row pieces requested ahead
  static_assert(OUTW >= 16 && HC + 1 <= 16, "window too wide for a one-wave strip");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the threshold table in LDS: read when the clipped row counts change, and an LDS read does not wait for the row pieces in flight
  constexpr int NALPHA = (2 * HC + 1) * (2 * HR + 1) + 1;
  __shared__ double alphaL[NALPHA];
  for (int e = threadIdx.x; e < NALPHA; e += 256) alphaL[e] = a.alpha[e];
  __syncthreads();
  const int chunk = gridDim.x >> 3; // gridDim.x is a multiple of 8: XCD x walks workgroups [x chunk, (x + 1) chunk)
  const int task = (((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3)) * 4 + wave;
  if (task >= ta.nTasks) return;
  __shared__ double rowL[4][C2S_V][2][C2S_PITCH]; // per wave and row in flight: |z|^2 and the pair sums, a pad of 16 on either side
  double *lds1 = &rowL[wave][0][0][16 + lane];
  const int perCpi = ta.segs * ta.strips;
  const int cpi = task / perCpi, rem = task - cpi * perCpi;
  const int seg = rem / ta.strips, strip = rem - seg * ta.strips;
  const int nD = a.nD, nC = a.nDelay;
  const int i0 = seg * ta.rowsPerSeg, i1 = min(i0 + ta.rowsPerSeg, nD);
  const int j = strip * OUTW - HC + lane;
  const int voff = j * 8;
  // per-lane constants: the clipped column counts of the cell this lane tests (0: it tests none)
  auto clampi = [](int v_, int lo, int hi) { return v_ < lo ? lo : (v_ > hi ? hi : v_); };
  auto cols = [](int c0_, int c1_) { return max(c1_, 1) - max(c0_, 1); }; // column 0 never trains (CfarDetector1D.cpp:61)
  const bool jok = lane >= HC && lane < 64 - HC && j < nC && j + a.delayMin >= a.minDelay; // CfarDetector1D.cpp:53
  const int nColsAll = jok ? cols(clampi(j - HC, 0, nC), clampi(j + HC + 1, 0, nC)) : 0;
  const int nColsGuard = jok ? cols(clampi(j - NGD, 0, nC), clampi(j + NGD + 1, 0, nC)) : 0;
  const bool isCol0 = j == 0;
  const cf *mapc = a.map + (size_t)cpi * nD * nC;

  const int rStart = i0 - HR, rLast = i1 - 1 + HR; // rows streamed; row r completes output row r - HR
  c2t_v2u pf[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int r = rStart + u;
    const bool rok = r >= 0 && r < nD && r <= rLast;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(mapc + (size_t)(rok ? r : 0) * nC), (short)0, rok ? nC * 8 : 0, 0x00020000);
    pf[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
  }
  double Bh[U], Ah[U], TBh[U], TAh[U], sqh[U];
#pragma unroll
  for (int u = 0; u < U; u++) Bh[u] = Ah[u] = TBh[u] = TAh[u] = sqh[u] = 0.0;
  int keyA = -1, keyG = -1;
  double nf = 0.0, al = 0.0;

  // The only vector-memory operations of the loop are the row pieces: any other load would have to be waited for with
  // everything requested before it (loads return in order), i.e. with the whole ring.
  for (int rb = rStart; rb <= rLast; rb += U) {
    uint32_t hitRows = 0;
#pragma unroll
    for (int ub = 0; ub < U; ub += V) {
      // ---- along the row, V rows side by side.  The lane's |z|^2 and its sum with the right neighbour's go to the wave's own
      // piece of LDS (ds_write_b64), the neighbours' come back through ds_read_b64 at immediate offsets: on gfx950 a 64-lane
      // ds_read_b64 takes 2 cycles of the CU's LDS, a ds_bpermute_b32 six (tools/membench/ldsrate.hip) -- the first version
      // of this kernel shifted registers through 14 bpermutes per row and was bound by exactly those.  Lanes of one wave
      // only: no barrier, DS operations of a wave execute in order.  Lanes outside [hC, 64 - hC) read the pads (anything):
      // they test nothing.
      double s1[V];
#pragma unroll
      for (int v = 0; v < V; v++) {
        const int u = ub + v;
        const double x = (double)__uint_as_float(pf[u].x), y = (double)__uint_as_float(pf[u].y);
        const double sq = x * x + y * y;
        {
          const int rn = (C2S_ABLATE & 1) ? max(rStart, 0) : rb + u + U;
          const bool rok = rn >= 0 && rn < nD && rn <= rLast;
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(mapc + (size_t)(rok ? rn : 0) * nC), (short)0, rok ? nC * 8 : 0, 0x00020000);
          pf[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
        }
        sqh[u] = sq;
        s1[v] = isCol0 ? 0.0 : sq;
        lds1[v * 2 * C2S_PITCH] = s1[v];
      }
#if C2S_ABLATE & 2
#pragma unroll
      for (int v = 0; v < V; v++) { Bh[ub + v] = s1[v] * 3.0; Ah[ub + v] = s1[v]; }
#else
      C2S_LANES_SEE();
#pragma unroll
      for (int v = 0; v < V; v++) lds1[(v * 2 + 1) * C2S_PITCH] = s1[v] + C2S_LD(lds1 + v * 2 * C2S_PITCH + 1);
      C2S_LANES_SEE();
#pragma unroll
      for (int v = 0; v < V; v++) {
        const int u = ub + v;
        const double *p1 = lds1 + v * 2 * C2S_PITCH, *p2 = p1 + C2S_PITCH;
        double A = 0.0, G = C2S_LD(p1 - NGD);
#pragma unroll
        for (int m = 0; m < NGD; m++) G += C2S_LD(p2 - NGD + 1 + 2 * m); // the 2 nGd + 1 guard columns: one cell and nGd pairs
        if (NTD > 0) {
          double L = 0.0, R = 0.0;                                       // nTd columns left and right of them: pairs (and one cell if nTd is odd)
#pragma unroll
          for (int m = 0; m < NTD / 2; m++) {
            const double l = C2S_LD(p2 - HC + 2 * m), r = C2S_LD(p2 + NGD + 1 + 2 * m);
            L = m ? L + l : l;
            R = m ? R + r : r;
          }
          if (NTD & 1) {
            const double l = C2S_LD(p1 - NGD - 1), r = C2S_LD(p1 + HC);
            L = NTD > 1 ? L + l : l;
            R = NTD > 1 ? R + r : r;
          }
          A = L + R;
        }
        Bh[u] = A + G; Ah[u] = A;
      }
#endif
      // ---- down the column, and the tests
#pragma unroll
      for (int u = ub; u < ub + V; u++) {
        double TB = 0.0, TA = Ah[u];
        if (NTF > 0) {
          TB = Bh[u];
#pragma unroll
          for (int k = 1; k < NTF; k++) TB += Bh[(u - k + 8 * U) % U];
        }
#pragma unroll
        for (int k = 1; k < GH; k++) TA += Ah[(u - k + 8 * U) % U];
        TBh[u] = TB; TAh[u] = TA;
        double tot = TAh[(u - NTF + 8 * U) % U];
        if (NTF > 0) tot = (TB + tot) + TBh[(u - NTF - GH + 8 * U) % U];
        const double cut = sqh[(u - HR + 8 * U) % U];
        // the test of output row i (wave-uniform: is the row tested at all, and with which row counts)
        const int i = rb + u - HR;
        if (i >= i0 && i < i1 && (i < ta.deadLo || i >= ta.deadHi)) {
          const int rA = min(i + HR + 1, nD) - max(i - HR, 0), rG = min(i + NGF + 1, nD) - max(i - NGF, 0);
          if (rA != keyA || rG != keyG) {
            keyA = rA; keyG = rG;
            const int nn = rA * nColsAll - rG * nColsGuard; // >= 0: the guard box lies inside the window
            nf = (double)nn;
            al = alphaL[nn];
          }
          hitRows |= (cut * nf > al * tot) ? (1u << u) : 0u;
        }
      }
    }
    // the (rare) hits of these U rows, in one place: the cell is read again, its |z|^2 is the same arithmetic
    if (hitRows) {
      for (int u = 0; u < U; u++) {
        if (!((hitRows >> u) & 1u)) continue;
        const int i = rb + u - HR;
        const cf c = mapc[(size_t)i * nC + j];
        const double cut = (double)c.x * (double)c.x + (double)c.y * (double)c.y;
        const uint32_t slot = atomicAdd(&a.count[cpi], 1u);
        if (slot < a.cap) {
          blah2hip_hit_t h;
          h.row = i;
          h.col = j;
          h.snr = 5.0 * log10(cut) - a.metrics[2 * cpi];
          a.hits[(size_t)cpi * a.cap + slot] = h;
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) HERE: left pending, the store would turn every wait of the loop into vmcnt(0)
    }
  }
}

// the window shapes (nTd, nGd, nTf, nGf) the stream kernel is instantiated for; one line per shape
#define C2S_SHAPES(X) \
  X(6, 2, 3, 1) /* config.yml:36-40 along delay + Doppler guard 1 / train 3: the bench's 17 x 9 window */ \
  X(6, 2, 0, 0) X(8, 2, 4, 1) X(4, 1, 2, 1) X(3, 1, 2, 1) X(1, 0, 0, 0) X(0, 0, 1, 0) X(2, 0, 1, 0) X(5, 2, 6, 2)

} // namespace blah2
