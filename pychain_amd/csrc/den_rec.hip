// den_rec.hip - the two-barrier alpha / beta recursion (den_recursion_kernel): the state vector is normalised in a pass of
// its own every frame.  The form of round 1; it remains for the shapes the lazy-normalisation recursion does not take (more
// than four row groups per wave, rows of more than 9216 pdfs, arcs streamed from L2 because a wave's share exceeds its
// registers, per-sequence plans that do not all fit) and as the second opinion of the tests (option den_lazy = 0).
// Replaces chain-computation.cc:92-207,232-330 (pytorch_binding/src); see den_kernels.hip for the decomposition.
#include <hip/hip_runtime.h>
#include <cstring>
#include <type_traits>
#include <stdint.h>

#include "common.h"
#include "den_kernels.h"
#include "device_utils.h"
#include "plan_format.h"

namespace pychain_hip {

namespace {

#include "den_common.inc.h"
// ------------------------------------------------------------------------------------
// launch 1: alpha and beta recursions
// ------------------------------------------------------------------------------------
// DB: the nnet-output row is double-buffered in LDS (two 16 KiB regions, D <= 4096), so the row of
// the next frame is exp'd and stored by each wave right after ITS arc work - while slower waves
// still gather - instead of by all waves at once between the two barriers.
template <int VEC, int XCH, int R, bool DB>
__global__ __launch_bounds__(kNT) void den_recursion_kernel(const DenArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool fwd = blockIdx.x < (unsigned)a.B;
  const int b = fwd ? blockIdx.x : blockIdx.x - a.B;
  const int L = __builtin_amdgcn_readfirstlane(seq_len(a.lengths, b, a.T));
  const int nsteps = fwd ? L : L - 1;
  const int j_begin = 0, j_end = nsteps;             // (the whole sequence in one launch)
#ifdef PYCHAIN_EXP_ONLY_DIR                          // timing experiment: 0 = alpha workgroups only, 1 = beta only
  if ((int)fwd == PYCHAIN_EXP_ONLY_DIR) return;
#endif
  const int Hp = a.Hp, H = a.H, D = a.D, Dp = (D + 3) & ~3;
  const char* plan = a.plans + (size_t)b * a.plan_stride;
  const PlanHeader* hd = reinterpret_cast<const PlanHeader*>(plan);
  const TilePlan tp = fwd ? hd->alpha : hd->beta;
  const WaveEntry we = reinterpret_cast<const WaveEntry*>(plan + tp.off_wave_tab)[wave];
  const GroupEntry* gtab = reinterpret_cast<const GroupEntry*>(plan + tp.off_group_tab);
  const uint2* slots = reinterpret_cast<const uint2*>(plan + tp.off_slots);

  // LDS: cur = normalised state vector of the previous frame (gather operand U), xr = exp'd
  // nnet-output row (operand V), raw = this frame's un-normalised sums, lk = leaky probs.
  float* xr = reinterpret_cast<float*>(smem_raw);                      // DB: xr = buffer 0, xr + kXOff/4 = buffer 1
  float* cur = xr + (DB ? 2 * (kXOff / 4) : Dp);
  float* raw = cur + Hp;
  float* lk = raw + Hp;
  float* red = lk + Hp;              // [2][16]

  GroupRegs groups;
  groups.load<R>(we, gtab, lane);
  const uint2* wave_slots = slots + (size_t)__builtin_amdgcn_readfirstlane(we.slot_row_begin) * 64 + lane;
  ArcRegs<R> arcs;
  arcs.load(groups.nslots, wave_slots, lds_addr(cur), lds_addr(xr));
  const uint2* tail_slots = wave_slots + (size_t)R * 64;

  const float* leaky_g = reinterpret_cast<const float*>(plan + (fwd ? hd->off_leaky_a : hd->off_leaky_b));
  const float* start_g = reinterpret_cast<const float*>(plan + (fwd ? hd->off_init_a : hd->off_final_b));
  const float* xseq = a.x + (size_t)b * a.T * D;
  float* store = fwd ? a.alpha_store + (size_t)b * a.T * Hp : a.beta_store + (size_t)b * (a.T + 1) * Hp;
  const float coef = a.coef;
  const XBuf xbuf = make_xbuf(xseq, (size_t)a.T * D * sizeof(float));
  const XBuf sbuf = make_xbuf(store, (size_t)(a.T + 1) * Hp * sizeof(float));   // < 2 GiB: checked at launch

  float* totv = (fwd ? a.tot_a : a.tot_b) + (size_t)b * (a.T + 2);   // per-frame totals for den_finish_kernel (DenArgs::tot_a)
  int bad = (fwd && seq_len_bad(a.lengths, b, a.T)) ? 1 : 0;   // bit 0 not ok, bit 1 a NaN network output (den_lazy.inc.h)
  float tot, wtot;
  XRow<kNT, VEC, XCH> xq;
  if (tid < 32) red[tid] = 0.f;
  {
    // ---- frame 0 (alpha) / frame L (beta): chain-computation.cc:92-95,97-110,178-194 / :232-245,313-330
    float p0 = 0.f, p1 = 0.f;
    for (int i = tid; i < Hp; i += kNT) {
      const float l = leaky_g[i], s = start_g[i];
      lk[i] = l; raw[i] = s;
      p0 += s; p1 += s * l;
    }
    p0 = wave_sum(p0); p1 = wave_sum(p1);
    {
      const int t0 = fwd ? 0 : L - 1;                 // first nnet-output row this side consumes
      xq.load(xseq + (size_t)t0 * D, D, tid);
      if (fwd && xq.has_nan()) bad |= 2;               // a NaN network output: not ok, NaN log-probability
      if (xq.store(xr, xseq + (size_t)t0 * D, D, tid, a.input_is_exp) && fwd) bad |= 2;
    }
    __syncthreads();                                   // red zeroed
    // beta positions that take no constant c(t) (plan.cpp, "states on several lanes"): marked in the sign bit of their leaky
    // probability, which is read as |l| (tile_rows, normalise_row)
    if (!fwd) {
      const int32_t* nc = reinterpret_cast<const int32_t*>(plan + hd->off_no_const);
      for (int j = tid; j < hd->n_no_const; j += kNT) lk[nc[j]] = __uint_as_float(__float_as_uint(lk[nc[j]]) | 0x80000000u);
    }
    if (lane == 0) { red[wave] = p0; red[16 + wave] = p1; }
    __syncthreads();
    tot = block_total(red, lane); wtot = block_total(red + 16, lane);
    const float inv = __builtin_amdgcn_rcpf(tot);
    if (!(tot > 0.f) || !(inv > 0.f)) bad |= 1;
    if (tid == 0) totv[fwd ? 0 : L] = tot;
    normalise_row(fwd, raw, lk, cur, sbuf, (fwd ? 0 : L) * Hp * 4, inv, coef, coef * wtot, H, Hp, tid);
  }
  __syncthreads();

  float4 cl0 = make_float4(0.f, 0.f, 0.f, 0.f);       // coef * leaky probs of the states this thread normalises (alpha)
  if (fwd && tid * 4 < Hp) {
    const float4 l = *reinterpret_cast<const float4*>(lk + tid * 4);
    cl0 = make_float4(coef * l.x, coef * l.y, coef * l.z, coef * l.w);
  }
  // ---- general frames.  alpha: step j produces alpha'(j+1) from alpha'(j) and x(j), j = 0..L-1
  //                        beta:  step j produces beta(t) from beta(t+1) and x(t), t = L-1-j, j = 0..L-2
#ifdef PYCHAIN_PROFILE_PHASES
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
#define PH_ADD(i, t0) ph[i] += PH_T() - (t0)
#else
#define PH_ADD(i, t0) (void)(t0)
#endif
  // One frame step.  VOFF = byte offset of the nnet-output buffer this step gathers from (double-
  // buffered form: even steps read buffer 0 and fill buffer 1, odd steps the reverse; the frame
  // loop is unrolled by two IN SOURCE ORDER - a branch between two inlined copies of the arc loop
  // makes the optimiser hoist their common address arithmetic above the branch and spill it).
#define PYCHAIN_REC_STEP(J, VOFF)                                                                          \
  do {                                                                                                      \
    const int j = (J);                                                                                      \
    unsigned long long pt = PH_T();                                                                         \
    const int tn = fwd ? j + 1 : L - 2 - j;          /* nnet-output row of the NEXT step */                \
    const bool have_next = fwd ? (tn < L) : (tn >= 1);                                                      \
    const float* xrow_next = xseq + (size_t)(have_next ? tn : 0) * D;                                       \
    if (kWithX && have_next) {                       /* in flight during the arc work */                    \
      if constexpr (VEC == 4 && XCH > 0) xq.load_row(xbuf, have_next ? tn : 0, D, tid);                      \
      else xq.load(xrow_next, D, tid);                                                                      \
    }                                                                                                       \
    float s0 = 0.f, s1 = 0.f;                                                                               \
    if (kWithArcs)                                                                                          \
      tile_rows<R, 0, VOFF>(arcs, groups, tail_slots, lane, cur, xr + (VOFF) / 4, raw, nullptr, fwd ? nullptr : lk, s0, s1); \
    else { s0 = 1.f; s1 = 1.f; }                                                                            \
    /* double-buffered: the other buffer was last read in the previous step, which every wave has left */   \
    if (DB && kWithX && have_next) {                                                                        \
      if (fwd && xq.has_nan()) bad |= 2;                                                                    \
      xq.store(xr + (kXOff - (VOFF)) / 4, xrow_next, D, tid, a.input_is_exp);                               \
    }                                                                                                       \
    if (kArcsOnly) { if (s0 == 12345.f) raw[tid] = s0; if (kArcsOnly == 2) __syncthreads(); break; }        \
    PH_ADD(0, pt); pt = PH_T();                                                                             \
    s0 = wave_sum(s0);                                                                                      \
    if (!fwd) s1 = wave_sum(s1);                                                                            \
    if (lane == 0) { red[wave] = s0; red[16 + wave] = s1; }                                                 \
    PH_ADD(1, pt); pt = PH_T();                                                                             \
    __syncthreads();                                 /* every gather of this frame is done */               \
    PH_ADD(2, pt); pt = PH_T();                                                                             \
    tot = block_total(red, lane);                                                                           \
    wtot = fwd ? 0.f : block_total(red + 16, lane);                                                         \
    const float inv = __builtin_amdgcn_rcpf(tot);                                                           \
    if (!(tot > 0.f) || !(inv > 0.f)) bad |= 1;                                                             \
    const int tstore = fwd ? j + 1 : L - 1 - j;                                                             \
    if (tid == 0) totv[tstore] = tot;                /* the scale divided out of this frame (den_finish_kernel) */ \
    const bool do_store = fwd ? (tstore < L) : true;                                                        \
    if (kWithNorm)                                                                                          \
      normalise_row(fwd, raw, lk, cur, sbuf, do_store ? tstore * Hp * 4 : -1, inv, coef, coef * wtot, H, Hp, tid, true, cl0); \
    PH_ADD(3, pt); pt = PH_T();                                                                             \
    if (!DB && kWithX && have_next) {                                                                       \
      if (fwd && xq.has_nan()) bad |= 2;                                                                    \
      if (xq.store(xr, xrow_next, D, tid, a.input_is_exp) && fwd) bad |= 2;   /* (rows staged without registers) */ \
    }                                                                                                       \
    PH_ADD(4, pt); pt = PH_T();                                                                             \
    __syncthreads();                                                                                        \
    PH_ADD(5, pt);                                                                                          \
  } while (0)
  // (PYCHAIN_EXP_*: ablation builds for timing only - results are wrong)
#ifdef PYCHAIN_EXP_NO_X
  constexpr bool kWithX = false;
#else
  constexpr bool kWithX = true;
#endif
#ifdef PYCHAIN_EXP_NO_ARCS
  constexpr bool kWithArcs = false;
#else
  constexpr bool kWithArcs = true;
#endif
#ifdef PYCHAIN_EXP_NO_NORM
  constexpr bool kWithNorm = false;
#else
  constexpr bool kWithNorm = true;
#endif
#ifdef PYCHAIN_EXP_ARCS_ONLY
  constexpr int kArcsOnly = PYCHAIN_EXP_ARCS_ONLY;
#else
  constexpr int kArcsOnly = 0;
#endif
  // Progress signal of the gated schedule: once the steps below seg_bound[s] are done, every wave waits for
  // its own row stores (device-scope write-through, normalise_row: the L2s of the XCDs are not coherent with
  // one another and the occupancy kernel runs on all of them), then one thread counts the workgroup in.
  int next_sig = 0;
  int next_bound = a.sig_n > 0 ? a.seg_bound[0] : 0x7fffffff;
#define PYCHAIN_REC_SIGNAL(DONE)                                                                            \
  while ((DONE) >= next_bound) {                                                                            \
    __builtin_amdgcn_s_waitcnt(0);                     /* this wave's row stores are acknowledged */          \
    __syncthreads();                                                                                        \
    if (tid == 0) __hip_atomic_fetch_add(a.progress + next_sig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    next_sig++;                                                                                             \
    next_bound = next_sig < a.sig_n ? a.seg_bound[next_sig] : 0x7fffffff;                                   \
  }
  if constexpr (DB) {
    // segments start at even steps (seg bounds are multiples of 32), so step parity = buffer parity
    for (int jj = j_begin; jj < j_end; jj += 2) {      // (the macro declares `j`)
      PYCHAIN_REC_STEP(jj, 0);
      if (jj + 1 < j_end) PYCHAIN_REC_STEP(jj + 1, kXOff);
      PYCHAIN_REC_SIGNAL(jj + 2);                      // bounds are even
    }
  } else {
    for (int jj = j_begin; jj < j_end; jj++) { PYCHAIN_REC_STEP(jj, 0); PYCHAIN_REC_SIGNAL(jj + 1); }
  }
  PYCHAIN_REC_SIGNAL(next_sig < a.sig_n ? 0x7ffffffe : 0);   // a sequence shorter than a bound is done with it now
#undef PYCHAIN_REC_SIGNAL
#undef PYCHAIN_REC_STEP
#ifdef PYCHAIN_PROFILE_PHASES
  if (lane == 0 && (b == 0))
    printf("dir %d wave %d steps %d cycles/step: arcs %llu wsum %llu bar1 %llu update %llu xstore %llu bar2 %llu\n", (int)fwd, wave,
           nsteps, ph[0] / max(1, j_end - j_begin), ph[1] / max(1, j_end - j_begin), ph[2] / max(1, j_end - j_begin),
           ph[3] / max(1, j_end - j_begin), ph[4] / max(1, j_end - j_begin), ph[5] / max(1, j_end - j_begin));
#endif

  if (fwd) {
    // ComputeTotLogLike, chain-computation.cc:209-230: log sum_i alpha'(L,i) final(i) + sum_t log tot(t)
    const float* fin = reinterpret_cast<const float*>(plan + hd->off_final_a);
    float f = 0.f;
    for (int i = tid; i < Hp; i += kNT) f += cur[i] * fin[i];
    f = wave_sum(f);
    if (lane == 0) red[wave] = f;
    if (tid == 0) red[16] = 0.f;
    __syncthreads();
    if (bad & 2) red[16] = 1.f;                        // somebody staged a NaN network output
    __syncthreads();
    const float fs = block_total(red, lane);
    if (tid == 0) {
      a.fin_dot[b] = red[16] != 0.f ? __builtin_nanf("") : fs;       // den_finish_kernel: objf = sum_t log tot(t) + log of this
      if (!(fs > 0.f)) bad |= 1;
    }
  }
  if (bad && lane == 0) atomicAdd(a.bad, 1);
}

}  // namespace

// the recursion launch of a call that runs neither as a lazy nor as a pair kernel (launch_den)
template <int VEC, int XCH>
static hipError_t launch_rec_vx(const DenArgs& a, int hint, size_t lds_rec, hipStream_t st) {
  const dim3 grid(2 * a.B);
  // <4, 1> (D <= 4096, D % 4 == 0) double-buffers the nnet-output row: gathered operands end at 32 KiB + 4 Hp
  constexpr bool DB = VEC == 4 && XCH == 1;
  switch (pick_r(a, hint & 1023, DB ? 2 * (kXOff / 4) + a.Hp : a.Hp + ((a.D + 3) & ~3))) {
    case 0: return launch_one(den_recursion_kernel<VEC, XCH, 0, DB>, a, grid, lds_rec, st);
    case 16: return launch_one(den_recursion_kernel<VEC, XCH, 16, DB>, a, grid, lds_rec, st);
    case 32: return launch_one(den_recursion_kernel<VEC, XCH, 32, DB>, a, grid, lds_rec, st);
    case PLAN_RESIDENT_FIT: return launch_one(den_recursion_kernel<VEC, XCH, PLAN_RESIDENT_FIT, DB>, a, grid, lds_rec, st);
    default: return launch_one(den_recursion_kernel<VEC, XCH, kMaxResident, DB>, a, grid, lds_rec, st);
  }
}
hipError_t launch_den_rec2b(const DenArgs& a, int hint, size_t lds_rec, hipStream_t st) {
  const int D = a.D;
  if (D % 4 == 0) {
    if (D <= 4 * 1 * kNT) return launch_rec_vx<4, 1>(a, hint, lds_rec, st);
    if (D <= 4 * 2 * kNT) return launch_rec_vx<4, 2>(a, hint, lds_rec, st);
    if (D <= 4 * 3 * kNT) return launch_rec_vx<4, 3>(a, hint, lds_rec, st);
    if (D <= 4 * 4 * kNT) return launch_rec_vx<4, 4>(a, hint, lds_rec, st);
  } else if (D <= 4 * kNT) {
    return launch_rec_vx<1, 4>(a, hint, lds_rec, st);
  }
  return launch_rec_vx<1, 0>(a, hint, lds_rec, st);
}

}  // namespace pychain_hip
