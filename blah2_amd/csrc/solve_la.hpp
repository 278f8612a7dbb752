// Hermitian Toeplitz solve of the clutter filter's normal equations on SEVERAL workgroups per CPI
// (WienerHopf.cpp:85-122: A = toeplitz(r), chol(A), two triangular solves) -- included by clutter.hip.
//
// clutter_solve_kernel (the round-2 form) runs the Schur/Levinson recursion one order at a time in ONE workgroup:
// nBins dependent orders with a workgroup barrier each (1.7 ms at 2047 taps whatever the batch).  Here the same
// recursion is cut along both axes -- orders in blocks of 32, indices in slices -- with nothing but 32 coefficient
// pairs per block crossing between waves, so that a CPI spreads over as many CUs as the launch leaves free.
//
// State per index j, role decided by the order boundary m (an index turns "lower" when its order has passed):
//     lower  j <= m :  U = F[j]   V = B[j-1]   Z =  x[j]      forward / backward predictor (B one index up), solution
//     upper  j >  m :  U = A[j]   V = C[j-1]   Z = -g[j]      their residuals T F, T B and the right-hand side's
// One order is the SAME element-wise map on every index, then a shift by one index:
//     W = V - conj(ef) U,   U' = U - ef V,   Z' = Z + dt W,   V'[j] = W[j-1]          ef = U[m+1] / s,   dt = -Z[m+1] / s'
//     index m+1 (it turns lower):  U' = -ef,  Z' = dt                                 s' = s (1 - |ef|^2)
// so the values at j after k orders depend on the values at [j-k, j] before them, and on the k pairs (ef, dt).  The
// matrix is positive definite iff r[0] > 0 and every 1 - |ef|^2 > 0: the condition under which the reference's chol()
// succeeds (WienerHopf.cpp:111); otherwise ok = 0 and the taps are zero.  fp64 throughout.
// (Derivation and a wave-level NumPy model of exactly this index algebra: tools/proto/toeplitz_front_bulk.py.)
//
//   * The FRONT (waves 0 and 1 of a CPI's first workgroup) produces the coefficients from the 32-index sub-windows
//     W_b = [32b+1, 32b+32].  The CHAIN wave runs the 32 orders of block b on W_b, reading (ef, dt) off the leading lane
//     -- the one dependent chain of the whole solve -- and publishes them in chunks of 8 orders (LDS for its companion,
//     granules in L2 for the bulk).  Its COMPANION wave supplies every next triangle: stale values creep up one lane
//     per order from lane 0 of a 64-lane set, so after 32 orders on [W_b | W_b+1] the upper half is still valid.  Per
//     block b it (i) applies c_(b-1) to D = [W_b | W_b+1] valid through block b-2, whose upper half is the feed-in the
//     bulk wave that owns W_b+1 published after ITS block b-2 -- two blocks of slack, so the bulk may run behind
//     without stalling the chain -- and (ii) follows the chain wave's chunks of block b on C = [W_b | W_b+1] valid
//     through b-1 (lower half: its own last hand-over, upper half: D's) and hands C.hi = W_b+1, valid through b, over
//     through LDS.  (Rounds of measurement behind this split: a wave issues in order, an fp64 instruction of a lone
//     wave costs 5-6 cycles, so the chain wave is bound by its instruction count -- 300 cycles per order with the
//     catch-up orders in the same wave, against a dependent chain of ~150.)
//   * BULK waves own S = 64 E - 32 consecutive indices (E per lane, lane-major: the shift is a register rename plus ONE
//     DPP wave shift) and a halo of 32 below them.  Per block: wait for the coefficient chunks, 32 orders, publish the
//     top 32 indices as the halo of the wave above, publish the front's feed-in if it lies here, refresh the own halo.
//
// Exchange inside the launch (MI355X_MICROARCH.md "inter-workgroup visibility"; per-XCD L2s are not coherent, a CU's L1
// is never refreshed): every shared word is an 8-byte agent-scope relaxed atomic (sc1: L2-served, write-through).
// Coefficient chunks are 64 data-tagged granules {epoch, 32 bits} -- the data is the flag, one sweep instruction per
// chunk; halos and feed-ins are sc1 payload stores, a drained store queue, then one flag word.  No slot is ever
// reused within a launch, so no flow control; the tag is a per-handle launch epoch (bumped by the reduce kernel that
// always precedes), so nothing is cleared between launches.  Every spin is bounded: on a timeout the fault word is set
// and ok stays 0.
#pragma once

namespace sla {

constexpr int KB = 32;                 // orders per block = halo width = sub-window width
constexpr unsigned SPIN_LIMIT = 1u << 20; // polls of a bounded wait (about a second); Args::spinLimit, a test can lower it

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;

struct Args {
  const dcx *rb;     // [nCpi][2][n]: r then b
  cf *w;             // [nCpi][n]
  int32_t *ok;       // [nCpi]
  u64 *mail;         // [nCpi][mailStride]
  const uint32_t *epoch;
  uint32_t *fault;   // set when a bounded spin ran out (sticky, per handle: a diagnostic)
  uint32_t spinLimit; // polls before a wait gives up
  int32_t n, NB, nbulk, G, nCpi;
  int64_t mailStride;                                   // u64 words per CPI
  int64_t offHalo, offFeed, offHaloFlag, offFeedFlag, offStatus; // coefficient granules at 0
};

// ---- shared words ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void st64(u64 *p, u64 v) { __hip_atomic_store((gu64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld64(const u64 *p) { return __hip_atomic_load((gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_d(u64 *p, double v) { st64(p, (u64)__double_as_longlong(v)); }
__device__ __forceinline__ double ld_d(const u64 *p) { return __longlong_as_double((long long)ld64(p)); }
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// What a wave does when a bounded wait runs out: the handle's sticky fault word (a diagnostic) and the CPI's own fault word
// in its mailbox (offStatus + 1 = this launch's tag).  The launch that follows on the stream (clutter_solve_kernel, gated
// on that word) then solves the CPI again in ONE workgroup, which waits for nobody: a wait that ran out -- workgroups of
// this launch that a busy chip dispatched late -- costs time, never a result, and is never mistaken for "not positive
// definite" (WienerHopf.cpp:111-115).
struct FaultRef {
  uint32_t *sticky;
  u64 *cpiWord;
  uint32_t tag;
  unsigned limit;
};
__device__ __forceinline__ FaultRef fault_ref(const Args &a, int cpi, uint32_t tag)
{
  return FaultRef{a.fault, a.mail + (size_t)cpi * a.mailStride + a.offStatus + 1, tag, a.spinLimit};
}

__device__ __forceinline__ bool spin_fail(unsigned &spins, const FaultRef &f)
{
  __builtin_amdgcn_s_sleep(1);
  if (++spins > f.limit) {
    __hip_atomic_store((gu32 *)f.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    st64(f.cpiWord, (u64)f.tag);
    return true;
  }
  return false;
}

// one relaxed poll of ONE word until it carries this launch's tag
__device__ __forceinline__ bool wait_flag(const u64 *flag, uint32_t tag, const FaultRef &fault)
{
  for (unsigned spins = 0;;) {
    const u64 v = ld64(flag);
    if (__builtin_amdgcn_readfirstlane((uint32_t)v) == tag) break;
    if (spin_fail(spins, fault)) return false;
  }
  asm volatile("" ::: "memory");
  return true;
}

// a chunk of 8 orders = 64 granules, lane L <-> field (L & 7) of order (L >> 3)
__device__ __forceinline__ bool sweep_chunk(const u64 *g, int lane, uint32_t tag, const FaultRef &fault, uint32_t &val)
{
  for (unsigned spins = 0;;) {
    const u64 x = ld64(g + lane);
    val = (uint32_t)x;
    if (__all((uint32_t)(x >> 32) == tag)) return true;
    if (spin_fail(spins, fault)) return false;
  }
}

__device__ __forceinline__ double rl_d(double v, int l)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// value of lane L-1 (lane 0: zero): DPP wave_shr:1
__device__ __forceinline__ double shr1(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rot32(double v, int lane) // value of lane (L + 32) & 63
{
  const int a = ((lane + 32) & 63) << 2;
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(a, __double2loint(v)));
}

// ---- the state of E indices per lane (position p = lane E + e) and one order on it ------------------------------------
template <int E> struct St {
  double Ur[E], Ui[E], Vr[E], Vi[E], Zr[E], Zi[E];
};

// (U, V, Z) of index j before order 0
__device__ __forceinline__ void init_index(const dcx *r, const dcx *b, int n, dcx x0, int j, double &ur, double &ui, double &vr,
                                           double &vi, double &zr, double &zi)
{
  ur = ui = vr = vi = zr = zi = 0.0;
  if (j < 0 || j >= n) return;
  if (j == 0) { ur = 1.0; zr = x0.x; zi = x0.y; return; }
  const dcx rj = r[j], rm = r[j - 1], bj = b[j];
  ur = rj.x; ui = rj.y; vr = rm.x; vi = rm.y;
  // Z = -(b[j] - r[j] x0)
  zr = __builtin_fma(rj.x, x0.x, __builtin_fma(-rj.y, x0.y, -bj.x));
  zi = __builtin_fma(rj.x, x0.y, __builtin_fma(rj.y, x0.x, -bj.y));
}

template <int E> __device__ __forceinline__ void step(St<E> &s, double efr, double efi, double dtr, double dti)
{
  double wr[E], wi[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    wr[e] = __builtin_fma(-efi, s.Ui[e], __builtin_fma(-efr, s.Ur[e], s.Vr[e])); // V - conj(ef) U
    wi[e] = __builtin_fma(efi, s.Ur[e], __builtin_fma(-efr, s.Ui[e], s.Vi[e]));
    const double ur = __builtin_fma(efi, s.Vi[e], __builtin_fma(-efr, s.Vr[e], s.Ur[e])); // U - ef V
    const double ui = __builtin_fma(-efi, s.Vr[e], __builtin_fma(-efr, s.Vi[e], s.Ui[e]));
    s.Ur[e] = ur; s.Ui[e] = ui;
    s.Zr[e] = __builtin_fma(-dti, wi[e], __builtin_fma(dtr, wr[e], s.Zr[e])); // Z + dt W
    s.Zi[e] = __builtin_fma(dti, wr[e], __builtin_fma(dtr, wi[e], s.Zi[e]));
  }
#pragma unroll
  for (int e = E - 1; e >= 1; e--) { s.Vr[e] = wr[e - 1]; s.Vi[e] = wi[e - 1]; }
  s.Vr[0] = shr1(wr[E - 1]);
  s.Vi[0] = shr1(wi[E - 1]);
}

// 8 orders whose coefficients (efr, efi, dtr, dti per order) sit in the wave's own LDS copy; the loads of an order are
// issued an order ahead (a readlane -> SGPR -> VALU path measured 11 cycles per instruction against 5.5 for VGPR operands,
// tools/membench/f64rate.hip).  FIX: the index that turns lower at order t sits at position p0 + t of this wave
struct __attribute__((aligned(32))) Coef { double efr, efi, dtr, dti; };
template <int E, bool FIX> __device__ __forceinline__ void apply_chunk(St<E> &s, const Coef *cl, int lane, int p0)
{
  // narrow slices have little to overlap an LDS round trip with: all 8 records in registers first; wide ones (E = 12
  // sits at the register cap and has 144 independent multiply-adds per order) two at a time
  constexpr int G = E <= 3 ? 8 : (E <= 6 ? 4 : 1);
#pragma unroll 1
  for (int t0 = 0; t0 < 8; t0 += G) {
    Coef c[G];
#pragma unroll
    for (int g = 0; g < G; g++) c[g] = cl[t0 + g];
#pragma unroll
    for (int g = 0; g < G; g++) {
      step<E>(s, c[g].efr, c[g].efi, c[g].dtr, c[g].dti);
      if (FIX) {
        const int pp = p0 + t0 + g, pl = pp / E, pe = pp - pl * E;
        const bool me = lane == pl;
#pragma unroll
        for (int e = 0; e < E; e++)
          if (pe == e) {
            s.Ur[e] = me ? -c[g].efr : s.Ur[e]; s.Ui[e] = me ? -c[g].efi : s.Ui[e];
            s.Zr[e] = me ? c[g].dtr : s.Zr[e];  s.Zi[e] = me ? c[g].dti : s.Zi[e];
          }
      }
    }
  }
}

#define SLA_SB __builtin_amdgcn_sched_barrier(0)
// one record from ONE lane of a fully active wave (uniform values), without a branch around the store
typedef double d2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void put_coef(Coef *p, double efr, double efi, double dtr, double dti)
{
  const d2_t e = {efr, efi}, d = {dtr, dti};
  const uint32_t a = (uint32_t)(uintptr_t)p; // LDS offset
  asm volatile("s_mov_b64 exec, 1\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\ts_mov_b64 exec, -1\n\ts_nop 4"
               :: "v"(a), "v"(e), "v"(d) : "memory");
}

#ifdef SLA_TRACE
#define SLA_T0 unsigned long long t0_ = wall_clock64()
#define SLA_LAP(acc) do { const unsigned long long t1_ = wall_clock64(); acc += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define SLA_T0 do { } while (0)
#define SLA_LAP(acc) do { } while (0)
#endif

// ---- the front: chain wave and companion wave ---------------------------------------------------------------------------
// What the two exchange through LDS.  The chain wave writes a chunk's 8 records (the same 64 dwords it publishes as
// granules) and then its chunk count, every lane into a slot of its own; the companion writes a hand-over's 32 x 6
// doubles and then its count.  The LDS serves a wave's requests in order, so a count that has arrived has its data behind
// it.  Records are double-buffered by block parity: the companion reads c_(b-1) while the chain wave writes c_b, and it
// has finished with c_(b-1) before it hands over the window the chain wave needs to start block b+1.
struct PairLds {
  Coef rec[2][KB];
  double hand[6][KB];
  unsigned seq[64];      // chunks the chain wave has published
  unsigned handseq[64];  // hand-overs the companion has made
  unsigned bad;          // one of the two has given up (not positive definite, or a bounded wait ran out)
};

__device__ __forceinline__ bool lds_wait_count(const unsigned *p, unsigned want, const unsigned *bad, const FaultRef &fault)
{
  for (unsigned spins = 0;;) {
    const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((int)(__builtin_amdgcn_readfirstlane(v) - want) >= 0) {
      asm volatile("" ::: "memory"); // what the count announces is read after it
      return true;
    }
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) return false;
    if (spin_fail(spins, fault)) return false;
  }
}

__device__ __forceinline__ void front_chain(const Args &a, int cpi, uint32_t tag, int lane, PairLds *pl)
{
  const FaultRef fr = fault_ref(a, cpi, tag);
  const int n = a.n, NB = a.NB;
  const dcx *r = a.rb + (size_t)cpi * 2 * n, *b = r + n;
  u64 *mail = a.mail + (size_t)cpi * a.mailStride;
  u64 *coef = mail;
  __builtin_amdgcn_s_setprio(3);
  if (lane == 0) a.ok[cpi] = 0; // stays 0 if a bounded wait runs out (the gated launch behind this one then decides)
  const double r0 = r[0].x;
  bool bad = !(r0 > 0.0) || !isfinite(r0);
  double inv_s = bad ? 0.0 : 1.0 / r0;
  const dcx b0 = b[0];
  const dcx x0 = {b0.x * inv_s, b0.y * inv_s};
  St<1> A; // lanes 0..31: W_blk; the upper lanes ride along unused
  init_index(r, b, n, x0, 1 + lane, A.Ur[0], A.Ui[0], A.Vr[0], A.Vi[0], A.Zr[0], A.Zi[0]);
  const bool lo = lane < 32;
  const int f8 = lane & 7, g8 = lane >> 3;
  int blk = 0, cdone = 0; // chunks published so far
#ifdef SLA_TRACE
  unsigned long long tr1 = 0, trw = 0;
  const unsigned long long wc0 = wall_clock64(), cy0 = __builtin_readcyclecounter();
#endif
  for (; blk < NB && !bad; blk++) {
    Coef *cl = pl->rec[blk & 1];
    uint32_t *clw = reinterpret_cast<uint32_t *>(cl);
    SLA_T0;
    // phase 1: the 32 orders of this block on A, coefficients off the leading lane.  Orders beyond the last (the tail of
    // the last block) are identities.  This loop is the one dependent chain of the whole solve -- 1/s -> ef -> D -> 1/D
    // (estimate + two Newton steps) -> 1/s', ten dependent fp64 operations of 12 cycles each -- and a wave issues in
    // order, so the statement order below IS the schedule (SLA_SB pins it): everything that is not on the chain sits
    // in its shadows, and Z's update by order i (it needs dt_i, the last thing the chain yields) runs inside order
    // i + 1 together with dt_i and the record of order i.
    const int live = min(KB, n - 1 - KB * blk);
#pragma unroll 1
    for (int c = 0; c < 4 && !bad; c++) {
      const int nl = min(8, max(0, live - 8 * c));
      if (nl == 8) {
        // Measured links (tools/membench/f64chain.hip): a dependent v_fma_f64 12 cycles, v_rcp_f64 48, a lane broadcast
        // (v_readlane -> SGPR -> VALU) 60, the wave shift (DPP) 40.  So: U's next leading value is broadcast as soon as U
        // is final, mid-order, and has arrived when 1/s' has; Z's runs one order behind (dt, update, broadcast of order
        // i - 1 inside order i); the coefficients are kept in lanes of eight registers (one select each per order, no
        // store inside the chain) and leave as granules after the chunk.
        const int i0 = 8 * c;
        double efr, efi;
        {
          const double ar = rl_d(A.Ur[0], i0), ai = rl_d(A.Ui[0], i0);
          efr = ar * inv_s; efi = ai * inv_s;
        }
        double zlr = rl_d(A.Zr[0], i0), zli = rl_d(A.Zi[0], i0);
        double pefr = 0.0, pefi = 0.0, pwr = 0.0, pwi = 0.0;
        uint32_t ck[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 8; t++) {
          const int i = i0 + t;
          double D = __builtin_fma(-efr, efr, 1.0);
          SLA_SB;
          const double t1 = __builtin_fma(-efr, A.Vr[0], A.Ur[0]), t2 = __builtin_fma(-efr, A.Vi[0], A.Ui[0]);
          SLA_SB;
          D = __builtin_fma(-efi, efi, D);
          SLA_SB;
          double wr = __builtin_fma(-efr, A.Ur[0], A.Vr[0]), wi = __builtin_fma(-efr, A.Ui[0], A.Vi[0]);
          SLA_SB;
          double rc = __builtin_amdgcn_rcp(D);
          SLA_SB;
          wr = __builtin_fma(-efi, A.Ui[0], wr); wi = __builtin_fma(efi, A.Ur[0], wi);
          SLA_SB;
          A.Ur[0] = __builtin_fma(efi, A.Vi[0], t1); A.Ui[0] = __builtin_fma(-efi, A.Vr[0], t2);
          SLA_SB;
          double dtr = 0.0, dti = 0.0;
          if (t > 0) { dtr = -zlr * inv_s; dti = -zli * inv_s; } // of order i - 1: 1/s after it is this order's 1/s
          SLA_SB;
          A.Vr[0] = shr1(wr); A.Vi[0] = shr1(wi);
          SLA_SB;
          const double nar = rl_d(A.Ur[0], i + 1), nai = rl_d(A.Ui[0], i + 1); // the next order's leading U
          SLA_SB;
          double er = __builtin_fma(-D, rc, 1.0);
          SLA_SB;
          if (t > 0) { A.Zr[0] = __builtin_fma(dtr, pwr, A.Zr[0]); A.Zi[0] = __builtin_fma(dtr, pwi, A.Zi[0]); }
          SLA_SB;
          rc = __builtin_fma(rc, er, rc);
          SLA_SB;
          if (t > 0) { A.Zr[0] = __builtin_fma(-dti, pwi, A.Zr[0]); A.Zi[0] = __builtin_fma(dti, pwr, A.Zi[0]); }
          SLA_SB;
          er = __builtin_fma(-D, rc, 1.0);
          SLA_SB;
          if (t > 0) { // order i - 1 into the lanes 8 (t - 1) ... 8 (t - 1) + 7
            const bool here = g8 == t - 1;
            ck[0] = here ? (uint32_t)__double2loint(pefr) : ck[0]; ck[1] = here ? (uint32_t)__double2hiint(pefr) : ck[1];
            ck[2] = here ? (uint32_t)__double2loint(pefi) : ck[2]; ck[3] = here ? (uint32_t)__double2hiint(pefi) : ck[3];
            ck[4] = here ? (uint32_t)__double2loint(dtr) : ck[4];  ck[5] = here ? (uint32_t)__double2hiint(dtr) : ck[5];
            ck[6] = here ? (uint32_t)__double2loint(dti) : ck[6];  ck[7] = here ? (uint32_t)__double2hiint(dti) : ck[7];
          }
          SLA_SB;
          rc = __builtin_fma(rc, er, rc);
          SLA_SB;
          if (t > 0) { zlr = rl_d(A.Zr[0], i); zli = rl_d(A.Zi[0], i); } // Z is through order i - 1 now
          bad = bad || !(D > 0.0) || !isfinite(D);
          SLA_SB;
          inv_s *= rc;
          SLA_SB;
          pefr = efr; pefi = efi; pwr = wr; pwi = wi;
          efr = nar * inv_s; efi = nai * inv_s;
          SLA_SB;
        }
        {
          const double dtr = -zlr * inv_s, dti = -zli * inv_s;
          A.Zr[0] = __builtin_fma(-dti, pwi, __builtin_fma(dtr, pwr, A.Zr[0]));
          A.Zi[0] = __builtin_fma(dti, pwr, __builtin_fma(dtr, pwi, A.Zi[0]));
          const bool here = g8 == 7;
          ck[0] = here ? (uint32_t)__double2loint(pefr) : ck[0]; ck[1] = here ? (uint32_t)__double2hiint(pefr) : ck[1];
          ck[2] = here ? (uint32_t)__double2loint(pefi) : ck[2]; ck[3] = here ? (uint32_t)__double2hiint(pefi) : ck[3];
          ck[4] = here ? (uint32_t)__double2loint(dtr) : ck[4];  ck[5] = here ? (uint32_t)__double2hiint(dtr) : ck[5];
          ck[6] = here ? (uint32_t)__double2loint(dti) : ck[6];  ck[7] = here ? (uint32_t)__double2hiint(dti) : ck[7];
        }
        // lane L <-> dword L of the chunk's 8 records: field L & 7 of order L >> 3
        const uint32_t w01 = (f8 & 1) ? ck[1] : ck[0], w23 = (f8 & 1) ? ck[3] : ck[2];
        const uint32_t w45 = (f8 & 1) ? ck[5] : ck[4], w67 = (f8 & 1) ? ck[7] : ck[6];
        const uint32_t w03 = (f8 & 2) ? w23 : w01, w47 = (f8 & 2) ? w67 : w45;
        clw[64 * c + lane] = (f8 & 4) ? w47 : w03;
      } else {
        // the last block's tail, order by order
#pragma unroll 1
        for (int t = 0; t < nl; t++) {
          const int i = 8 * c + t;
          const double ar = rl_d(A.Ur[0], i), ai = rl_d(A.Ui[0], i);
          const double zr = rl_d(A.Zr[0], i), zi = rl_d(A.Zi[0], i);
          const double efr = ar * inv_s, efi = ai * inv_s;
          const double D = __builtin_fma(-efi, efi, __builtin_fma(-efr, efr, 1.0));
          bad = bad || !(D > 0.0) || !isfinite(D);
          inv_s *= fast_rcp(D);
          const double dtr = -zr * inv_s, dti = -zi * inv_s;
          step<1>(A, efr, efi, dtr, dti);
          put_coef(cl + i, efr, efi, dtr, dti);
        }
        for (int t = nl; t < 8; t++) {
          step<1>(A, 0.0, 0.0, 0.0, 0.0);
          put_coef(cl + 8 * c + t, 0.0, 0.0, 0.0, 0.0);
        }
      }
      bad = __any(bad);
      if (bad) break;
      // the chunk as 64 granules: lane L <-> dword L of the 8 records
      st64(coef + ((size_t)(KB * blk + 8 * c)) * 8 + lane, ((u64)tag << 32) | clw[64 * c + lane]);
      cdone++;
      asm volatile("" ::: "memory");
      __hip_atomic_store(&pl->seq[lane], (unsigned)cdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // behind the records: a wave's LDS requests are served in order
    }
    if (bad) break;
    SLA_LAP(tr1);
    if (blk + 1 >= NB) break;
    // the next triangle: W_(blk+1), valid through this block, from the companion
    if (!lds_wait_count(&pl->handseq[0], (unsigned)blk + 1u, &pl->bad, fr)) return;
    {
      const int l = lane & 31;
      const double u0 = pl->hand[0][l], u1 = pl->hand[1][l], v0 = pl->hand[2][l], v1 = pl->hand[3][l], z0 = pl->hand[4][l], z1 = pl->hand[5][l];
      A.Ur[0] = lo ? u0 : 0.0; A.Ui[0] = lo ? u1 : 0.0; A.Vr[0] = lo ? v0 : 0.0;
      A.Vi[0] = lo ? v1 : 0.0; A.Zr[0] = lo ? z0 : 0.0; A.Zi[0] = lo ? z1 : 0.0;
    }
    SLA_LAP(trw);
  }
#ifdef SLA_TRACE
  if (lane == 0 && cpi == 0)
    printf("sla trace chain: n %d NB %d  orders %.2f us/block  hand-over wait %.2f  cycle counter %.0f MHz\n", n, NB,
           0.01 * tr1 / NB, 0.01 * trw / NB, 100.0 * (double)(__builtin_readcyclecounter() - cy0) / (double)(wall_clock64() - wc0));
#endif
  if (bad) {
    __hip_atomic_store(&pl->bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // not positive definite: say so BEFORE the remaining coefficients go out as identities, so that a bulk wave that
    // has consumed its last chunk finds the status set
    st64(mail + a.offStatus, (u64)tag);
    drain();
    for (int c = cdone; c < 4 * NB; c++) st64(coef + (size_t)c * 64 + lane, (u64)tag << 32);
    if (lane == 0) a.ok[cpi] = 0;
    return;
  }
  if (lane == 0) a.ok[cpi] = 1;
}

__device__ __forceinline__ void front_companion(const Args &a, int cpi, uint32_t tag, int lane, PairLds *pl)
{
  const FaultRef fr = fault_ref(a, cpi, tag);
  const int n = a.n, NB = a.NB;
  const dcx *r = a.rb + (size_t)cpi * 2 * n, *b = r + n;
  u64 *mail = a.mail + (size_t)cpi * a.mailStride;
  __builtin_amdgcn_s_setprio(2);
  const double r0 = r[0].x;
  const double inv0 = (r0 > 0.0 && isfinite(r0)) ? 1.0 / r0 : 0.0;
  const dcx b0 = b[0];
  const dcx x0 = {b0.x * inv0, b0.y * inv0};
  St<1> C, D;
  init_index(r, b, n, x0, 1 + lane, C.Ur[0], C.Ui[0], C.Vr[0], C.Vi[0], C.Zr[0], C.Zi[0]);         // [W_0 | W_1]
  init_index(r, b, n, x0, 1 + KB + lane, D.Ur[0], D.Ui[0], D.Vr[0], D.Vi[0], D.Zr[0], D.Zi[0]);    // [W_1 | W_2]
  const bool lo = lane < 32;
#ifdef SLA_TRACE
  unsigned long long trf = 0, tr2 = 0, trc = 0;
#endif
  for (int blk = 0; blk + 1 < NB; blk++) {
    SLA_T0;
    if (blk >= 1) {
      // (i) c_(blk-1) on D = [W_blk | W_(blk+1)] valid through blk-2; its upper half from the bulk (blk = 1: initial values)
      if (blk >= 2) {
        D.Ur[0] = rot32(D.Ur[0], lane); D.Ui[0] = rot32(D.Ui[0], lane); D.Vr[0] = rot32(D.Vr[0], lane);
        D.Vi[0] = rot32(D.Vi[0], lane); D.Zr[0] = rot32(D.Zr[0], lane); D.Zi[0] = rot32(D.Zi[0], lane);
        const int wb = blk + 1;
        if (!wait_flag(mail + a.offFeedFlag + wb, tag, fr)) { __hip_atomic_store(&pl->bad, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return; }
        if (!lo) {
          const u64 *p = mail + a.offFeed + ((size_t)wb * KB + (lane - 32)) * 6;
          D.Ur[0] = ld_d(p + 0); D.Ui[0] = ld_d(p + 1); D.Vr[0] = ld_d(p + 2);
          D.Vi[0] = ld_d(p + 3); D.Zr[0] = ld_d(p + 4); D.Zi[0] = ld_d(p + 5);
        }
      }
      SLA_LAP(trf);
      if (!lds_wait_count(&pl->seq[0], 4u * (unsigned)blk, &pl->bad, fr)) return;
      const Coef *cp = pl->rec[(blk - 1) & 1];
#pragma unroll 1
      for (int c = 0; c < 4; c++) apply_chunk<1, false>(D, cp + 8 * c, lane, 0);
      // C = [last hand-over (C.hi) | D.hi]
      double t;
      t = rot32(C.Ur[0], lane); C.Ur[0] = lo ? t : D.Ur[0];
      t = rot32(C.Ui[0], lane); C.Ui[0] = lo ? t : D.Ui[0];
      t = rot32(C.Vr[0], lane); C.Vr[0] = lo ? t : D.Vr[0];
      t = rot32(C.Vi[0], lane); C.Vi[0] = lo ? t : D.Vi[0];
      t = rot32(C.Zr[0], lane); C.Zr[0] = lo ? t : D.Zr[0];
      t = rot32(C.Zi[0], lane); C.Zi[0] = lo ? t : D.Zi[0];
      SLA_LAP(tr2);
    }
    // (ii) the chain wave's chunks of this block on C as they come
    const Coef *cq = pl->rec[blk & 1];
#pragma unroll 1
    for (int c = 0; c < 4; c++) {
      if (!lds_wait_count(&pl->seq[0], 4u * (unsigned)blk + (unsigned)c + 1u, &pl->bad, fr)) return;
      apply_chunk<1, false>(C, cq + 8 * c, lane, 0);
    }
    // hand C.hi = W_(blk+1), valid through this block, over
    if (!lo) {
      const int l = lane - 32;
      pl->hand[0][l] = C.Ur[0]; pl->hand[1][l] = C.Ui[0]; pl->hand[2][l] = C.Vr[0];
      pl->hand[3][l] = C.Vi[0]; pl->hand[4][l] = C.Zr[0]; pl->hand[5][l] = C.Zi[0];
    }
    asm volatile("" ::: "memory");
    __hip_atomic_store(&pl->handseq[lane], (unsigned)blk + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    SLA_LAP(trc);
  }
#ifdef SLA_TRACE
  if (lane == 0 && cpi == 0)
    printf("sla trace companion: feed wait %.2f us/block  catch-up orders %.2f  following + hand-over %.2f\n", 0.01 * trf / NB, 0.01 * tr2 / NB, 0.01 * trc / NB);
#endif
}

// ---- a bulk wave ----------------------------------------------------------------------------------------------------
template <int E> __device__ __forceinline__ void bulk(const Args &a, int cpi, uint32_t tag, int lane, int q, Coef *cl)
{
  const FaultRef fr = fault_ref(a, cpi, tag);
  constexpr int S = 64 * E - KB, P = 64 * E;
  const int n = a.n, NB = a.NB;
  const dcx *r = a.rb + (size_t)cpi * 2 * n, *b = r + n;
  u64 *mail = a.mail + (size_t)cpi * a.mailStride;
  const u64 *coef = mail;
  const int j0 = q * S + 1 - KB; // index of position 0
  const double r0 = r[0].x;
  const double inv0 = (r0 > 0.0 && isfinite(r0)) ? 1.0 / r0 : 0.0;
  const dcx b0 = b[0];
  const dcx x0 = {b0.x * inv0, b0.y * inv0};
  St<E> s;
#pragma unroll
  for (int e = 0; e < E; e++)
    init_index(r, b, n, x0, j0 + lane * E + e, s.Ur[e], s.Ui[e], s.Vr[e], s.Vi[e], s.Zr[e], s.Zi[e]);
#ifdef SLA_TRACE
  unsigned long long trs = 0, tra = 0, trp = 0, trh = 0;
#endif
  uint32_t *clw = reinterpret_cast<uint32_t *>(cl);
  u64 pre = ld64(coef + lane);
  for (int blk = 0; blk < NB; blk++) {
    const int p0 = KB * (blk + 1) - q * S; // position of W_blk's first index here (a multiple of 32)
    const bool fix = p0 >= 0 && p0 < P;
    SLA_T0;
#pragma unroll 1
    for (int c = 0; c < 4; c++) {
      // this chunk's granules were requested before the previous chunk's orders; re-read until every tag is this launch's
      u64 x = pre;
      for (unsigned spins = 0; !__all((uint32_t)(x >> 32) == tag);) {
        if (spin_fail(spins, fr)) return;
        x = ld64(coef + ((size_t)(KB * blk + 8 * c)) * 8 + lane);
      }
      if (4 * blk + c + 1 < 4 * NB) pre = ld64(coef + ((size_t)(KB * blk + 8 * c + 8)) * 8 + lane);
      clw[64 * c + lane] = (uint32_t)x;
      SLA_LAP(trs);
      if (fix) apply_chunk<E, true>(s, cl + 8 * c, lane, p0 + 8 * c);
      else apply_chunk<E, false>(s, cl + 8 * c, lane, 0);
      SLA_LAP(tra);
    }
    if (blk + 1 >= NB) break;
    // the top 32 positions are the halo of the wave above
    if (q + 1 < a.nbulk) {
      u64 *h = mail + a.offHalo + ((size_t)q * NB + blk) * (KB * 6);
#pragma unroll
      for (int e = 0; e < E; e++) {
        const int el = lane * E + e - (P - KB);
        if (el >= 0) {
          u64 *p = h + (size_t)el * 6;
          st_d(p + 0, s.Ur[e]); st_d(p + 1, s.Ui[e]); st_d(p + 2, s.Vr[e]);
          st_d(p + 3, s.Vi[e]); st_d(p + 4, s.Zr[e]); st_d(p + 5, s.Zi[e]);
        }
      }
      drain();
      if (lane == 0) st64(mail + a.offHaloFlag + (size_t)q * NB + blk, (u64)tag);
    }
    // the front's feed-in W_{blk+3}, if this wave owns it
    {
      const int wb = blk + 3, pf = KB * (wb + 1) - q * S;
      if (wb <= NB - 1 && pf >= KB && pf + KB <= P) {
        u64 *fd = mail + a.offFeed + (size_t)wb * (KB * 6);
#pragma unroll
        for (int e = 0; e < E; e++) {
          const int el = lane * E + e - pf;
          if (el >= 0 && el < KB) {
            u64 *p = fd + (size_t)el * 6;
            st_d(p + 0, s.Ur[e]); st_d(p + 1, s.Ui[e]); st_d(p + 2, s.Vr[e]);
            st_d(p + 3, s.Vi[e]); st_d(p + 4, s.Zr[e]); st_d(p + 5, s.Zi[e]);
          }
        }
        drain();
        if (lane == 0) st64(mail + a.offFeedFlag + wb, (u64)tag);
      }
    }
    SLA_LAP(trp);
    // own halo from the wave below (wave 0: the positions below index 0 are exact zeros, nothing creeps in)
    if (q > 0) {
      if (!wait_flag(mail + a.offHaloFlag + (size_t)(q - 1) * NB + blk, tag, fr)) return;
      const u64 *h = mail + a.offHalo + ((size_t)(q - 1) * NB + blk) * (KB * 6);
#pragma unroll
      for (int e = 0; e < E; e++) {
        const int el = lane * E + e;
        if (el < KB) {
          const u64 *p = h + (size_t)el * 6;
          s.Ur[e] = ld_d(p + 0); s.Ui[e] = ld_d(p + 1); s.Vr[e] = ld_d(p + 2);
          s.Vi[e] = ld_d(p + 3); s.Zr[e] = ld_d(p + 4); s.Zi[e] = ld_d(p + 5);
        }
      }
    }
    SLA_LAP(trh);
  }
#ifdef SLA_TRACE
  if (lane == 0 && cpi == 0 && (q == 0 || q == a.nbulk - 1))
    printf("sla trace bulk %d (E %d): coefficient wait %.2f us/block  apply %.2f  publish %.2f  halo %.2f\n", q, E,
           0.01 * trs / NB, 0.01 * tra / NB, 0.01 * trp / NB, 0.01 * trh / NB);
#endif
  // taps: Z of the own positions (wave 0 also holds index 0, at the top of its halo)
  const bool fail = (uint32_t)ld64(mail + a.offStatus) == tag;
  cf *w = a.w + (size_t)cpi * n;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int p = lane * E + e, j = j0 + p;
    if (j >= 0 && j < n && (p >= KB || (q == 0 && p == KB - 1)))
      w[j] = fail ? cmake(0.f, 0.f) : cmake((float)s.Zr[e], (float)s.Zi[e]);
  }
}

// grid: ceil(nCpi / 8) * 8 * G workgroups of NW waves.  The G workgroups of a CPI take block ids congruent mod 8 (same XCD
// while the dispatcher keeps its round-robin: their exchange is then served by one L2 -- speed only).  Roles in a CPI's
// first workgroup: wave 0 the chain wave, wave 1 its companion, then bulk waves (with 8 waves two of them share the front
// waves' SIMDs: the eight-wave form is the packed one, for batches in which the bulk waves set the pace and the chain
// wave, at the higher priority, mostly waits); every other workgroup is NW bulk waves.
template <int E, int NW> __global__ __launch_bounds__(64 * NW) void clutter_solve_la_kernel(Args a)
{
  const int G = a.G;
  const int chunk = blockIdx.x / (8 * G), within = blockIdx.x - chunk * 8 * G;
  const int member = within >> 3, cpi = chunk * 8 + (within & 7);
  if (cpi >= a.nCpi) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t tag = *a.epoch;
  __shared__ Coef lds[NW][KB]; // a bulk wave's own copy of the block's coefficients (1 KB each; no exchange through it)
  __shared__ PairLds pl;       // the front's two waves (a CPI's first workgroup)
  if (threadIdx.x < 64) { pl.seq[threadIdx.x] = 0; pl.handseq[threadIdx.x] = 0; }
  if (threadIdx.x == 0) pl.bad = 0;
  __syncthreads();
  int q;
  if (member == 0) {
    if (w == 0) { front_chain(a, cpi, tag, lane, &pl); return; }
    if (w == 1) { front_companion(a, cpi, tag, lane, &pl); return; }
    q = w - 2;
  } else {
    q = NW - 2 + (member - 1) * NW + w;
  }
  if (q < a.nbulk) bulk<E>(a, cpi, tag, lane, q, lds[w]);
}

// ---- host side: the plan of a launch --------------------------------------------------------------------------------
struct Plan {
  int E = 0, NW = 4, G = 0, nbulk = 0, NB = 0;
  int64_t stride = 0, offHalo = 0, offFeed = 0, offHaloFlag = 0, offFeedFlag = 0, offStatus = 0;
};
inline Plan make_plan(int n, int E, int NW)
{
  Plan p;
  p.E = E; p.NW = NW;
  const int S = 64 * E - KB;
  p.nbulk = std::max(1, (n - 1 + S - 1) / S);
  p.NB = std::max(1, (n - 1 + KB - 1) / KB);
  const int first = NW - 2; // bulk waves beside the front's two in a CPI's first workgroup
  p.G = 1 + (std::max(0, p.nbulk - first) + NW - 1) / NW;
  int64_t o = (int64_t)p.NB * KB * 8;
  p.offHalo = o; o += (int64_t)p.nbulk * p.NB * KB * 6;
  p.offFeed = o; o += (int64_t)p.NB * KB * 6;
  p.offHaloFlag = o; o += (int64_t)p.nbulk * p.NB;
  p.offFeedFlag = o; o += p.NB;
  p.offStatus = o; o += 8;
  p.stride = (o + 15) & ~(int64_t)15;
  return p;
}
constexpr int kE[4] = {2, 3, 6, 12};
// narrowest slices (lowest latency per block) whose workgroups all get a CU of their own, four waves per workgroup before
// eight; the widest slices otherwise.  `capacity(E, NW)` = workgroups of that instantiation the chip holds at once
// (occupancy x CUs): a plan whose G > 1 workgroups per CPI wait for each other is only used when the whole grid fits --
// also when a slice width is forced -- else the one-workgroup-per-CPI plan (E = 12, eight waves: G = 1 up to 4417 taps,
// nothing to wait for outside the workgroup).
template <class Cap> inline Plan choose_plan(int n, int nCpi, int numCU, int forceE, Cap capacity)
{
  const int64_t groups = (nCpi + 7) / 8 * 8;
  for (int E : kE) {
    if (forceE && E != forceE) continue;
    for (int NW : {4, 8}) {
      const Plan p = make_plan(n, E, NW);
      if ((int64_t)p.G * groups <= std::min<int64_t>(numCU, capacity(E, NW))) return p;
    }
  }
  const Plan p = make_plan(n, forceE ? forceE : 12, 8);
  if (p.G > 1 && (int64_t)p.G * groups > capacity(p.E, p.NW)) return make_plan(n, 12, 8);
  return p;
}

} // namespace sla
