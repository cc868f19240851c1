// den_kernels.hip - denominator (probability domain, leaky-HMM) forward-backward
// for gfx950.  Hand-written for CDNA4; structurally unrelated to the reference's
// one-launch-per-frame, thread-per-(sequence,state) CUDA kernels
// (pytorch_binding/src/chain-kernels.cu:97-245).
//
// Decomposition:
//
//   launch 1  den_recursion_kernel   2B persistent workgroups, one per (sequence, direction).
//             Block b < B walks the alpha recursion of sequence b forward in time, block B+b
//             walks the beta recursion of the same sequence backward in time, CONCURRENTLY:
//             the beta pass does not wait for the alpha pass because both carry their own
//             per-frame normaliser ("arbitrary_scale" in chain-computation.h:91-98 - any
//             per-frame scale gives the same posteriors).  The state vector of the previous
//             frame and the exp'd nnet-output row live in LDS; every wave keeps its share of
//             the arcs in VGPRs for the whole launch (LDS address of both operands + the
//             probability), so the per-frame inner loop is 2 ds_read + 2 VALU per arc;
//             per-state sums are lane-private (one state per lane), per-frame totals are
//             wave64 DPP reductions + one LDS hop.  Every normalised alpha'(t,.) / beta(t,.)
//             row is streamed to HBM once.
//   launch 2  den_gamma2_kernel / den_gamma_kernel   time-parallel over all (sequence,
//             frame-chunk) pairs:
//             gamma(t,n) = x(t,n) * sum_{arcs with pdf n} p * alpha'(t,src) * beta(t+1,dst),
//             normalised so that each live frame sums to one (the invariant the reference
//             checks at chain-computation.cc:381-390).  Arcs are grouped by pdf-id, so the
//             occupancy is a lane-private sum: no atomics, deterministic, exact (the
//             reference's CUDA path adds stochastically-thresholded atomics,
//             chain-kernels.cu:53-87; parity target is its exact CPU path).  The shipped form
//             evaluates two frames together (float2-interleaved operands, ds_read_b64 gathers,
//             packed fma) and can fold the numerator's occupancies in; the one-frame form is
//             the fallback for graphs that do not fit it.
//
// What bounds these kernels (profiles/): the recursion is a chain of T
// dependent frame steps per workgroup; a step is the LDS gather time of the frame's arcs
// (conflict-free by construction of the plan) plus barrier-separated serial phases, not HBM;
// the occupancy pass is time-parallel and runs near the HBM roofline.
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

#ifdef PYCHAIN_PROFILE_PHASES
#define PH_T() __builtin_readcyclecounter()
#else
#define PH_T() 0ull
#endif

constexpr int kNW = PLAN_REC_WAVES;        // waves per workgroup (both kernels)
constexpr int kNT = kNW * 64;              // threads per workgroup
static_assert(PLAN_REC_WAVES == PLAN_GAM_WAVES, "one workgroup shape for both kernels");
static_assert(kNW <= 16, "block totals are reduced inside one 16-lane DPP row");
constexpr int kMaxResident = PLAN_REC_WAVES > 12 ? PLAN_RESIDENT_2 : 44;   // slot-rows per wave kept in VGPRs
static_assert(PLAN_REC_WAVES <= 12 || (PLAN_RESIDENT_0 == 16 && PLAN_RESIDENT_1 == 32), "the plan compiler sizes its slack for these loop lengths");

// ---- one frame of a tile plan: out[row] = sum_k p_k * U[i0_k] * V[i1_k] ------------------
// The first R slot-rows of a wave are held in registers as ABSOLUTE LDS byte addresses of
// the two operands plus the arc probability (R = 0: everything is streamed from the
// L2-resident plan).  The plan of a wave is loop-invariant over frames, so this is loaded
// ONCE per workgroup and the per-frame inner loop touches only LDS.
#ifndef PYCHAIN_ARC_PACKED
#define PYCHAIN_ARC_PACKED (PLAN_REC_WAVES > 12)   // 16 waves: 128 VGPRs/lane -> 2 registers per slot-row
#endif
template <int R>
struct ArcRegs {
#if PYCHAIN_ARC_PACKED
  // 2 VGPRs per slot-row: both absolute LDS byte addresses packed 16:16, and the probability.
  uint32_t pk[R > 0 ? R : 1];
#else
  // 3 VGPRs per slot-row: the two absolute LDS byte addresses and the probability.
  uint32_t o0[R > 0 ? R : 1];
  uint32_t o1[R > 0 ? R : 1];
#endif
  float p[R > 0 ? R : 1];
  __device__ __forceinline__ void load(const int nslot_rows, const uint2* __restrict__ wave_slots,
                                       uint32_t lds_u, uint32_t lds_v) {
#pragma unroll
    for (int s = 0; s < R; s++) {
      uint2 a = make_uint2(0u, 0u);                    // rows past the plan: p = 0, harmless addresses
      if (s < nslot_rows) a = wave_slots[s * 64];
      uint32_t a0 = lds_u + ((a.x & 0xffffu) << 2), a1 = lds_v + ((a.x >> 16) << 2);
#ifdef PYCHAIN_EXP_NOCONFLICT      // timing experiment: lane-linear gathers (wrong results)
      a0 = lds_u + (((threadIdx.x & 63) + 64 * (s & 7)) << 2); a1 = lds_v + (((threadIdx.x & 63) + 64 * (s & 7)) << 2);
#endif
#if PYCHAIN_ARC_PACKED
      pk[s] = a0 | (a1 << 16);
#else
      o0[s] = a0; o1[s] = a1;
      asm volatile("" : "+v"(o0[s]), "+v"(o1[s]));    // opaque: no re-derivation from the packed word per frame
#endif
      p[s] = __uint_as_float(a.y);
    }
  }
  // the two gathered operands of slot-row s
  // Opaque per frame and chunk: otherwise the optimiser hoists both unpacked addresses of every
  // slot-row out of the frame loop (3 VGPRs per arc instead of 2).  One statement per chunk:
  // every inline asm costs a hazard s_nop.
  template <int N>
  __device__ __forceinline__ void opaque(int s) {
#if PYCHAIN_ARC_PACKED
    if constexpr (N == 4) asm volatile("" : "+v"(pk[s]), "+v"(pk[s + 1]), "+v"(pk[s + 2]), "+v"(pk[s + 3]));
    else for (int k = 0; k < N; k++) asm volatile("" : "+v"(pk[s + k]));
#endif
  }
  template <int VOFF = 0>         // VOFF: compile-time byte offset of operand V (folds into the ds_read offset field)
  __device__ __forceinline__ void gather(int s, float& u, float& v) {
#if PYCHAIN_ARC_PACKED
    const uint32_t a0 = pk[s] & 0xffffu, a1 = pk[s] >> 16;
#else
    const uint32_t a0 = o0[s], a1 = o1[s];
#endif
#ifndef PYCHAIN_EXP_NOLDS
    u = lds_abs(a0); v = lds_abs(a1 + VOFF);
#else
    u = __uint_as_float(a0); v = __uint_as_float(a1);
#endif
  }
};

#ifndef PYCHAIN_CHUNK
#define PYCHAIN_CHUNK 4      // slot-rows gathered ahead per step of the software pipeline
#endif

// The wave's group table lives in registers: lane i of `base` / `n` = output base and
// slot-row count of the wave's i-th group (read back with v_readlane), `endmask` bit s =
// resident slot-row s closes a group.  The frame loop issues NO memory instruction for
// bookkeeping.  (A table in global memory costs a vmcnt wait per group, and vmcnt is
// in-order: it would also wait for the nnet-output prefetch from HBM.)
struct GroupRegs {
  int base, n;
  unsigned long long endmask;
  uint32_t endmask2;        // ... slot-rows 64 .. 95 (the 8-wave recursion keeps up to 80)
  uint32_t chunkmask;       // bit c = chunk c of the resident slot-rows contains a group end
  int ngroups, nslots;      // of this wave
  int tail_g, tail_rem;     // group / slot-rows left in it when the streamed tail (slot-row R) starts
  template <int R>
  __device__ __forceinline__ void load(const WaveEntry we, const GroupEntry* __restrict__ gtab, int lane) {
    ngroups = __builtin_amdgcn_readfirstlane(we.ngroups);
    nslots = __builtin_amdgcn_readfirstlane(we.nslot_rows);
    const int first = __builtin_amdgcn_readfirstlane(we.first_group);
    base = 0; n = 0;
    if (lane < ngroups) { const GroupEntry e = gtab[first + lane]; base = e.out_base; n = e.nslots; }
    endmask = 0ull; endmask2 = 0u; tail_g = 0; tail_rem = 0;
    int cum = 0;
    bool tail_set = false;
    for (int gi = 0; gi < ngroups; gi++) {
      const int cnt = __builtin_amdgcn_readlane(n, gi);
      if (cnt > 0) {
        if (!tail_set && cum + cnt > R) { tail_g = gi; tail_rem = cum + cnt - (cum > R ? cum : R); tail_set = true; }
        cum += cnt;
        if (cum - 1 < R && cum - 1 < 64) endmask |= 1ull << (cum - 1);
        else if (cum - 1 < R && cum - 1 < 96) endmask2 |= 1u << (cum - 1 - 64);
      }
    }
    chunkmask = 0u;
    for (int c = 0; c * PYCHAIN_CHUNK < 64; c++)
      if ((endmask >> (c * PYCHAIN_CHUNK)) & ((1ull << PYCHAIN_CHUNK) - 1ull)) chunkmask |= 1u << c;
    for (int c = 0; c * PYCHAIN_CHUNK < 32; c++)
      if ((endmask2 >> (c * PYCHAIN_CHUNK)) & ((1u << PYCHAIN_CHUNK) - 1u)) chunkmask |= 1u << (c + 64 / PYCHAIN_CHUNK);
  }
};

// MODE 0: out[out_base+lane] = acc (recursions).  MODE 1: out[row_map[out_base+lane]] = acc
// (occupancy pass: plan order -> natural pdf order, row_map in LDS, -1 = padding row).
template <int MODE>
__device__ __forceinline__ void tile_store(float acc, int pos, float* __restrict__ out, const int* __restrict__ row_map) {
  if constexpr (MODE == 0) {
    out[pos] = acc;
  } else {
    const int nat = row_map[pos];
    if (nat >= 0) out[nat] = acc;
  }
}

// s_waitcnt on lgkmcnt only (gfx9 encoding: vmcnt[3:0] expcnt[6:4] lgkmcnt[11:8] vmcnt_hi[15:14])
#define PYCHAIN_WAIT_LGKM(n) __builtin_amdgcn_s_waitcnt(0xC07F | ((n) << 8))

// Issue priority of a wave falls as it progresses through the chunks of a frame: the waves of a SIMD
// that are behind catch up, so all of them finish the arc phase together.  With equal priorities the
// arbiter serves the oldest wave first and the youngest runs its last chunks alone, latency-bound
// (measured per wave: 3000 / 4000 / 4700 / 5300 cycles; with this: recursion 4.02 -> 3.69 ms).
template <int NC>
__device__ __forceinline__ void wave_priority_by_progress(int c) {
#ifndef PYCHAIN_EXP_NOPRIO
  // highest for the first half of the chunks, then stepping down to 0 on the last one (measured best of
  // four schedules: equal quarters 3.76 ms, front-loaded 3.78, this 3.69, two levels 3.84)
  constexpr int N = NC > 0 ? NC : 1;
#if defined(PYCHAIN_PRIO_TABLE)                        /* experiments: eight hex digits, the level of each eighth of the chunks */
  auto level = [](int cc) { return (int)((PYCHAIN_PRIO_TABLE >> (4 * (7 - cc * 8 / N))) & 0xfu); };
#elif !defined(PYCHAIN_PRIO_SCHED) || PYCHAIN_PRIO_SCHED == 0
  auto level = [](int cc) { return cc * 2 / N == 0 ? 3 : max(0, 2 - (cc - N / 2) * 6 / N); };
#elif PYCHAIN_PRIO_SCHED == 2                          /* experiments */
  auto level = [](int cc) { return cc * 2 / N == 0 ? 0 : min(3, 1 + (cc - N / 2) * 6 / N); };
#elif PYCHAIN_PRIO_SCHED == 3
  auto level = [](int cc) { return cc * 4 / N >= 3 ? 0 : 3; };
#else
  auto level = [](int cc) { return 3 - cc * 4 / N; };
#endif
  const int lvl = level(c), prev = c > 0 ? level(c - 1) : -1;
  if (NC >= 4 && lvl != prev) {
    switch (lvl) {                               // (s_setprio takes an immediate)
      case 3: __builtin_amdgcn_s_setprio(3); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      default: __builtin_amdgcn_s_setprio(0); break;
    }
  }
#endif
}

// One frame of a tile plan.  The resident loop is written for instruction count (the arc phase
// is bound by LDS gather cycles, then by instructions issued - DESIGN.md §4):
// per chunk of kChunk slot-rows 2 unpack + 2 ds_read + mul + fma per slot-row, ONE s_waitcnt and
// ONE s_bitcmp/s_cbranch pair.  Nothing but `acc` is carried through the chunks: a group end
// (a few per frame, out of line) finds its group by a popcount of the end mask and adds to the
// row sums in place.
template <int R, int MODE, int VOFF = 0>
__device__ __forceinline__ void tile_rows(ArcRegs<R>& ar, const GroupRegs& gr,
                                          const uint2* __restrict__ tail_slots, int lane,
                                          const float* __restrict__ U, const float* __restrict__ V,
                                          float* __restrict__ out, const int* __restrict__ row_map,
                                          const float* __restrict__ wvec, float& s0, float& s1) {
  float acc = 0.f;
#ifndef PYCHAIN_CHUNK
#define PYCHAIN_CHUNK 4
#endif
  constexpr int kChunk = PYCHAIN_CHUNK;
  static_assert(R % kChunk == 0 && 32 % kChunk == 0 && R <= 64, "whole chunks; a chunk never straddles the mask words");
  constexpr int NC = R / kChunk;
  // Opaque per call: otherwise the optimiser precomputes per-slot-row lane masks outside the
  // frame loop and spills them.
  uint32_t m_lo = (uint32_t)gr.endmask, m_hi = (uint32_t)(gr.endmask >> 32), cm = gr.chunkmask;
  asm volatile("" : "+s"(m_lo), "+s"(m_hi), "+s"(cm));
  // Software pipeline: the gathers of chunk c+1 are issued BEFORE chunk c is consumed; LDS
  // returns in order, so one wait for "all but the newest 2*kChunk" covers the whole chunk.
  // Rows past the wave's plan carry p = 0 and valid addresses: no bound check.
  float ub[2][kChunk], vb[2][kChunk];
  if (R > 0) {
    ar.template opaque<kChunk>(0);
#pragma unroll
    for (int k = 0; k < kChunk; k++) ar.template gather<VOFF>(k, ub[0][k], vb[0][k]);
  }
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int cb = c & 1;
    wave_priority_by_progress<NC>(c);
    if (c + 1 < NC) {
      ar.template opaque<kChunk>((c + 1) * kChunk);
#pragma unroll
      for (int k = 0; k < kChunk; k++) ar.template gather<VOFF>((c + 1) * kChunk + k, ub[cb ^ 1][k], vb[cb ^ 1][k]);
    }
#if !defined(PYCHAIN_EXP_NOLDS) && !defined(PYCHAIN_EXP_NOWAIT)
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < NC) PYCHAIN_WAIT_LGKM(2 * kChunk); else PYCHAIN_WAIT_LGKM(0);
    __builtin_amdgcn_sched_barrier(0);
#endif
    // the common case unconditionally; a chunk with a group end (a few per frame) redoes it
    float nacc = acc;
#pragma unroll
    for (int k = 0; k < kChunk; k++) nacc = fmaf(ar.p[c * kChunk + k] * ub[cb][k], vb[cb][k], nacc);   // (p*u) rounded, then fused with v
    if (__builtin_expect(((cm >> c) & 1u) != 0u, 0)) {
      nacc = acc;
#pragma unroll
      for (int k = 0; k < kChunk; k++) {
        const int sidx = c * kChunk + k;
        nacc = fmaf(ar.p[sidx] * ub[cb][k], vb[cb][k], nacc);
        if (((sidx < 32 ? m_lo : m_hi) >> (sidx & 31)) & 1u) {
          // group index = number of group ends before this slot-row
          const uint32_t lo_before = sidx < 32 ? (m_lo & ((1u << (sidx & 31)) - 1u)) : m_lo;
          const uint32_t hi_before = sidx < 32 ? 0u : (m_hi & ((1u << (sidx & 31)) - 1u));
          const int g = __builtin_popcount(lo_before) + __builtin_popcount(hi_before);
          const int pos = __builtin_amdgcn_readlane(gr.base, g) + lane;
          tile_store<MODE>(nacc, pos, out, row_map);
          if constexpr (MODE == 0) {
            // row sums, updated IN PLACE (tied asm operands): a plain `s0 += nacc` makes s0/s1 loop-carried
            // values of the chunk chain and costs register copies on the common path of every chunk
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(nacc));
            if (wvec) { const float wv = wvec[pos]; asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s1) : "v"(nacc), "v"(wv)); }
          }
          nacc = 0.f;
        }
      }
    }
    acc = nacc;
  }
  int g = __builtin_popcount(m_lo) + __builtin_popcount(m_hi);     // groups closed by the resident rows
  if (gr.nslots > R) {                           // plan larger than the register budget: stream the tail
    const uint2* sp = tail_slots;
    g = gr.tail_g;
    int cur_base = __builtin_amdgcn_readlane(gr.base, g & 63);
    int remaining = gr.tail_rem;
    for (int s = R; s < gr.nslots; s++) {
      const uint2 a = *sp;
      sp += 64;
      acc = fmaf(__uint_as_float(a.y) * U[a.x & 0xffffu], V[a.x >> 16], acc);
      if (--remaining == 0) {
        tile_store<MODE>(acc, cur_base + lane, out, row_map);
        if constexpr (MODE == 0) { s0 += acc; if (wvec) s1 += acc * wvec[cur_base + lane]; }
        acc = 0.f;
        g++;
        cur_base = __builtin_amdgcn_readlane(gr.base, g & 63);
        remaining = __builtin_amdgcn_readlane(gr.n, g & 63);
      }
    }
  }
  for (; g < gr.ngroups; g++)                    // trailing groups whose rows have no arcs: zeros
    tile_store<MODE>(0.f, __builtin_amdgcn_readlane(gr.base, g & 63) + lane, out, row_map);
}

// Normalise the frame's raw sums into the gather operand and stream the row to HBM, 16 bytes
// per lane:  alpha: v = raw/tot + coef*leaky   (AlphaSum/AlphaDash, chain-computation.cc:97-110,178-194)
//            beta:  v = (raw + coef*sum_i leaky_i raw_i)/sum_i raw_i  (Beta, :313-330; unit-sum scale)
// The row also goes to the trajectory store behind `sbuf` (byte offset row_off, < 0 = not stored), with a
// device-scope write-through store (sc1): the occupancy kernel may read it on another XCD while this
// kernel is still running (gated schedule), and the row is not read again here, so it need not stay in
// this XCD's L2.
constexpr int kStoreDeviceScope = 16;    // cache-policy operand of the buffer store: sc1
// `cl0` = coef * leaky probs of this thread's first four states (constant over the frames: kept in
// registers by the frame loop, `have_cl0`, instead of read from LDS and multiplied in every frame).
__device__ __forceinline__ void normalise_row(bool fwd, const float* raw, const float* lk, float* cur, XBuf sbuf, int row_off,
                                              float inv, float coef, float add, int H, int Hp, int tid,
                                              bool have_cl0 = false, float4 cl0 = make_float4(0.f, 0.f, 0.f, 0.f)) {
  for (int i = tid * 4; i < Hp; i += kNT * 4) {
    const float4 r = *reinterpret_cast<const float4*>(raw + i);
    float4 v;
    if (fwd) {
      float4 cl;
      if (have_cl0 && i == tid * 4) cl = cl0;
      else {
        const float4 l = *reinterpret_cast<const float4*>(lk + i);
        cl = make_float4(coef * l.x, coef * l.y, coef * l.z, coef * l.w);
      }
      v = make_float4(r.x * inv + cl.x, r.y * inv + cl.y, r.z * inv + cl.z, r.w * inv + cl.w);
    } else {
      // (positions >= H are padding: nothing gathers them and the occupancy pass skips them, so they are
      // allowed to carry add * inv instead of zero - masking costs 8 VALU per thread on the critical path)
      const float ai = add * inv;                     // (r + add) * inv as one fma per element
      v = make_float4(__builtin_fmaf(r.x, inv, ai), __builtin_fmaf(r.y, inv, ai), __builtin_fmaf(r.z, inv, ai), __builtin_fmaf(r.w, inv, ai));
    }
    *reinterpret_cast<float4*>(cur + i) = v;
    if (row_off >= 0) {
      u32x4 q;
      q.x = __float_as_uint(v.x); q.y = __float_as_uint(v.y); q.z = __float_as_uint(v.z); q.w = __float_as_uint(v.w);
      __builtin_amdgcn_raw_buffer_store_b128(q, sbuf, i * 4, row_off, kStoreDeviceScope);
    }
  }
}

// natural log on the v_log_f32 unit (1 ulp of log2): all lanes, no divergent libm call
__device__ __forceinline__ float fast_log(float v) { return __builtin_amdgcn_logf(v) * 0.693147182464599609375f; }

// The reference's `ok` (BetaGeneralFrameDebug, chain-computation.cc:345-391): alpha'.beta' and the frame's
// derivative sum within 5 % of 1.  With free per-frame scales the same statement reads
// log G(t) + la[t] + lb[t+2] = log P (DenArgs::la); true = violated (also for NaN).
// The occupancy kernels only record G(t) (the recursions may still be running when an overlapped occupancy
// launch evaluates frame 0 of a short sequence); den_finish_kernel compares after the last launch of the call.
__device__ __forceinline__ void den_record_frame_total(const DenArgs& a, int b, int t, float frame_total) {
  a.gtot[(size_t)b * a.T + t] = frame_total;
}

// objf and the invariant check from the stored per-frame totals (DenArgs::tot_a).  One workgroup per sequence.
//   rows of den_recursion_lazy_kernel: alpha row t carries prod_{tau<t} tot(tau); the beta row frame t's occupancy
//       reads, b(t+1,.) + c(t+1), carries prod_{tau>=t+2} n(tau); objf = sum_{t<L} log tot(t) + log fin_dot
//   rows of den_recursion_kernel (normalised): alpha'(t)/tot(t) carries prod_{tau<=t} tot(tau); beta(t+1) carries
//       prod_{tau>=t+1} n(tau); objf = sum_{t<=L} log tot(t) + log fin_dot
constexpr int kFinNT = 256;
__global__ __launch_bounds__(kFinNT) void den_finish_kernel(const DenArgs a) {
  __shared__ double sh[kFinNT];
  __shared__ double sh_total_a, sh_total_b;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int L = seq_len(a.lengths, b, a.T);
  const float* ta = a.tot_a + (size_t)b * (a.T + 2);
  const float* tb = a.tot_b + (size_t)b * (a.T + 2);
  const int na = a.lazy ? L : L + 1;                       // alpha totals 0 .. na-1 make up the log-probability
  // (the math library's log: this is off every critical path and feeds a value compared at 1e-4 relative)
  double sa = 0.0, sb = 0.0;
  for (int t = tid; t < na; t += kFinNT) sa += log((double)ta[t]);
  for (int t = tid + 1; t <= L; t += kFinNT) sb += log((double)tb[t]);
  sh[tid] = sa;
  __syncthreads();
  for (int o = kFinNT / 2; o > 0; o >>= 1) { if (tid < o) sh[tid] += sh[tid + o]; __syncthreads(); }
  if (tid == 0) sh_total_a = sh[0];
  __syncthreads();
  sh[tid] = sb;
  __syncthreads();
  for (int o = kFinNT / 2; o > 0; o >>= 1) { if (tid < o) sh[tid] += sh[tid + o]; __syncthreads(); }
  if (tid == 0) sh_total_b = sh[0];
  __syncthreads();
  const double logp = sh_total_a + log((double)a.fin_dot[b]);
  const float objf = (float)logp;
  int bad = 0;
  if (tid == 0) {
    __hip_atomic_store(a.objf + b, objf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read by the last workgroup: loss_out)
    if (!(objf - objf == 0.f)) bad = 1;                    // -inf (the graph cannot end), NaN
  }
  if (a.check) {
    // frame t: PA(t) = log-scale of its alpha row, SB(t) = log-scale of the beta row it reads
    //   lazy: PA(t) = sum_{tau<t} log tot(tau),  SB(t) = sum_{tau>=t+2} log n(tau)
    //   else: PA(t) = sum_{tau<=t},              SB(t) = sum_{tau>=t+1}
    const float* g = a.gtot + (size_t)b * a.T;
    if (!a.check_all) {
      if (tid == 0) {
        const double pa = a.lazy ? 0.0 : log((double)ta[0]);
        const double sbt = sh_total_b - (a.lazy && L >= 1 ? log((double)tb[1]) : 0.0);   // lazy: tau >= 2; else tau >= 1
        const double est = log((double)g[0]) + pa + sbt;
        if (!(fabs(est - logp) <= 0.0487901642)) bad = 1;  // log(1.05); NaN counts
      }
    } else {
      // every frame: thread `tid` walks frames tid*per .. with running sums started from its chunk's prefix
      const int per = (L + kFinNT - 1) / kFinNT;
      const int t0 = tid * per, t1 = min(t0 + per, L);
      double ca = 0.0, cb = 0.0;                            // sum_{tau<t0} log tot(tau), sum_{tau<=t0} log n(tau) (tau >= 1)
      for (int t = 0; t < t0; t++) ca += log((double)ta[t]);
      for (int t = 1; t <= t0 && t <= L; t++) cb += log((double)tb[t]);
      for (int t = t0; t < t1; t++) {
        const double lt = log((double)ta[t]);
        // cb = sum_{1<=tau<=t} log n(tau)
        const double pa = a.lazy ? ca : ca + lt;
        const double upto = a.lazy ? cb + (t + 1 <= L ? log((double)tb[t + 1]) : 0.0) : cb;   // sum_{tau<=t+1} / sum_{tau<=t}
        const double est = log((double)g[t]) + pa + (sh_total_b - upto);
        if (!(fabs(est - logp) <= 0.0487901642)) bad = 1;
        ca += lt;
        if (t + 1 <= L) cb += log((double)tb[t + 1]);
      }
    }
  }
  if (bad) atomicAdd(a.bad, 1);
  if (a.loss_out == nullptr) return;
  // ---- step totals by the workgroup that finishes last (DenArgs::loss_out).  Every workgroup publishes its sequence's
  // objective and its `bad` increment with the release half of the counter increment; the last one acquires them all.
  __shared__ int s_last;
  __syncthreads();                                           // (tid 0 wrote objf[b] and counted into bad above)
  if (tid == 0)
    s_last = __hip_atomic_fetch_add(a.finish_count, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  double acc = 0.0, frames = 0.0;
  for (int i = tid; i < a.B; i += kFinNT) {
    acc += (double)__hip_atomic_load(a.objf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.loss_num_objf) acc -= (double)a.loss_num_objf[i];
    frames += (double)seq_len(a.lengths, i, a.T);
  }
  sh[tid] = acc;
  __syncthreads();
  for (int o = kFinNT / 2; o > 0; o >>= 1) { if (tid < o) sh[tid] += sh[tid + o]; __syncthreads(); }
  const double total = sh[0];
  __syncthreads();
  sh[tid] = frames;
  __syncthreads();
  for (int o = kFinNT / 2; o > 0; o >>= 1) { if (tid < o) sh[tid] += sh[tid + o]; __syncthreads(); }
  if (tid == 0) {
    double t = total * (double)a.loss_scale;                 // -(num - den) [* 1/frames], pychain/loss.py:100-104, rounded once
    if (a.loss_norm_dev) t /= (double)*a.loss_norm_dev;
    int nbad = 0;
    for (int i = 0; i < a.bad_words; i++) nbad += __hip_atomic_load(a.bad + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a.loss_out[0] = (float)t; a.loss_out[1] = (float)sh[0]; a.loss_out[2] = (float)nbad; a.loss_out[3] = (float)total;
  }
}

// block total of per-wave partials: red[16] in LDS (entries >= kNW stay zero)
__device__ __forceinline__ float block_total(const float* red, int lane) { return dpp_row_sum(red[lane & 15]); }

// ------------------------------------------------------------------------------------
// launch 1: alpha and beta recursions
// ------------------------------------------------------------------------------------
// DB: the nnet-output row is double-buffered in LDS (two 16 KiB regions, D <= 4096), so the row of
// the next frame is exp'd and stored by each wave right after ITS arc work - while slower waves
// still gather - instead of by all waves at once between the two barriers.
constexpr int kXOff = 16384;
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
  const int j_begin = a.seg_begin, j_end = min(a.seg_end, nsteps);
  if (j_begin > 0 && j_begin >= nsteps) return;      // this sequence finished in an earlier segment
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
  int bad = (fwd && a.seg_begin == 0 && seq_len_bad(a.lengths, b, a.T)) ? 1 : 0;   // bit 0 not ok, bit 1 a NaN network output (den_lazy.inc.h)
  float tot, wtot;
  XRow<kNT, VEC, XCH> xq;
  if (tid < 32) red[tid] = 0.f;
  if (j_begin == 0) {
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
    if (lane == 0) { red[wave] = p0; red[16 + wave] = p1; }
    __syncthreads();
    tot = block_total(red, lane); wtot = block_total(red + 16, lane);
    const float inv = __builtin_amdgcn_rcpf(tot);
    if (!(tot > 0.f) || !(inv > 0.f)) bad |= 1;
    if (tid == 0) totv[fwd ? 0 : L] = tot;
    normalise_row(fwd, raw, lk, cur, sbuf, (fwd ? 0 : L) * Hp * 4, inv, coef, coef * wtot, H, Hp, tid);
  } else {
    // ---- resume a later time segment: the state vector is the row the previous segment stored last
    const float* row = store + (size_t)(fwd ? j_begin : L - j_begin) * Hp;
    for (int i = tid * 4; i < Hp; i += kNT * 4) {
      *reinterpret_cast<float4*>(cur + i) = *reinterpret_cast<const float4*>(row + i);
      *reinterpret_cast<float4*>(lk + i) = *reinterpret_cast<const float4*>(leaky_g + i);
    }
    const int t0 = fwd ? j_begin : L - 1 - j_begin;
    xq.load(xseq + (size_t)t0 * D, D, tid);
    xq.store(xr, xseq + (size_t)t0 * D, D, tid, a.input_is_exp);   // j_begin is even: buffer 0
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

  if (j_end >= nsteps && fwd) {
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

#include "den_lazy.inc.h"
#include "den_pair.inc.h"

// Which recursion segment makes frame t of a length-L sequence computable: its alpha'(t) row
// exists once the forward recursion has run t steps, its beta(t+1) row once the backward
// recursion has run L-1-t steps; segment s covers steps [seg_bound[s-1], seg_bound[s]).
// The frames of one occupancy launch, as a range of `need` = max(t, L-1-t): (lo, hi].  Read once per
// workgroup (seg_bound[] lives in the kernel arguments: dependent scalar loads, and the frame loops ask
// several times per frame).
struct LaunchFrames {
  int lo, hi;
  __host__ __device__ __forceinline__ explicit LaunchFrames(const DenArgs& a) : lo(-1), hi(0x7fffffff) {
    if (a.gam_nseg > 0) {
      if (a.gam_seg > 0) lo = a.seg_bound[a.gam_seg - 1];
      if (a.gam_seg < a.gam_nseg - 1) hi = a.seg_bound[a.gam_seg];
    }
  }
  __host__ __device__ __forceinline__ bool has(int t, int L) const {
    const int need = t > L - 1 - t ? t : L - 1 - t;
    return need > lo && need <= hi;
  }
};
__device__ __forceinline__ int den_next_frame(int t, int t_end, int L, const LaunchFrames& lf) {
  while (t < t_end && !lf.has(t, L)) t++;
  return t;
}

// Occupancy launches of the segments after the first only hold the outer frames of a sequence: a left
// range [L-1-hi, L-1-lo) and a right range (lo, hi] (lo, hi = ends of the previous and of this segment).
// Their grid is sized for those ranges (den_compact_grid_x) and block k takes the k-th chunk of
// frames_per_block frames that meets them, instead of one workgroup per chunk of [0, T) of which most
// would find nothing to do (an idle workgroup still needs a whole free CU to start).  The first launch
// keeps the plain mapping: it also zeroes the padding of every chunk.  Returns -1: no chunk for this block.
__host__ __device__ __forceinline__ int den_chunk_of_block(int k, int L, const DenArgs& a) {
  if (a.gam_nseg == 0 || a.gam_seg == 0) return k;
  const int fpb = a.frames_per_block;
  const int lo = a.seg_bound[a.gam_seg - 1];
  const int hi = a.gam_seg == a.gam_nseg - 1 ? 0x3fffffff : a.seg_bound[a.gam_seg];
  const int a0 = L - 1 - hi > 0 ? L - 1 - hi : 0, a1 = L - 1 - lo;   // left frames [a0, a1)
  const int b0 = lo + 1, b1 = hi < L - 1 ? hi : L - 1;          // right frames [b0, b1]
  const int nl = a1 > a0 ? (a1 - 1) / fpb - a0 / fpb + 1 : 0;
  if (k < nl) return a0 / fpb + k;
  if (b1 < b0) return -1;
  int c0 = b0 / fpb;
  if (nl > 0 && c0 == (a1 - 1) / fpb) c0++;                     // that chunk went to the left range's last block
  const int c = c0 + (k - nl);
  return c <= b1 / fpb ? c : -1;
}
int den_compact_grid_x(const DenArgs& a) {
  const int fpb = a.frames_per_block, full = (a.T + fpb - 1) / fpb;
  if (a.gam_nseg == 0 || a.gam_seg == 0) return full;
  const int lo = a.seg_bound[a.gam_seg - 1];
  const int hi = a.gam_seg == a.gam_nseg - 1 ? a.T : a.seg_bound[a.gam_seg];
  return std::min(full, 2 * ((hi - lo + fpb - 1) / fpb + 1));
}

// ------------------------------------------------------------------------------------
// streamed occupancy pass: the queue of frame ranges (DenArgs::stream)
// ------------------------------------------------------------------------------------
// Frame t of a length-L sequence becomes computable when the alpha recursion has stored row t and the beta recursion row
// t+1, i.e. after max(t, L-1-t) steps: nothing before L/2, then ever faster, from the middle outwards.  The queue hands
// out, per sequence, rings of kStreamWidth frames on either side of the middle in that order - item id = padding of
// sequence id (id < B), else ((ring * B + b) * 2 + side) - so that ids are drawn in the order in which they become ready,
// whatever the lengths (which live on the device).  A workgroup draws an id, skips it if it holds no frame, waits until
// the two recursion workgroups of the sequence have reported the rows it needs (DenArgs::seq_progress) and evaluates it.
// No deadlock: the recursion workgroups were all resident before this kernel was released (its gate waits for every one
// of them to pass T/2), they wait for nobody, and every item only waits for them.
struct StreamItem { int b, lo, hi, L, pad; };
__device__ __forceinline__ bool stream_take(const DenArgs& a, int* slot, StreamItem& it) {
  __syncthreads();                                      // the previous item (and its slot) is done with
  if (threadIdx.x == 0) {
    const int B = a.B, total = B + stream_ring_count(a.T) * 2 * B;
    int b = -1, lo = 0, hi = 0, L = 0, pad = 0;
    for (;;) {
      const int id = atomicAdd(a.stream_next, 1);
      if (id >= total) { b = -1; break; }
      if (id < B) {                                     // frames past the sequence's end: exact zeros
        b = id; L = seq_len(a.lengths, b, a.T); lo = L; hi = a.T; pad = 1;
        if (lo < hi) break;
        continue;
      }
      const int q = id - B, r = q / (2 * B), rem = q - r * 2 * B, side = rem & 1;
      b = rem >> 1; L = seq_len(a.lengths, b, a.T); pad = 0;
      const int half = L / 2;
      int n0, n1;                                                           // the ring: need in [n0, n1)
      stream_ring(a.T, r, n0, n1);
      if (side) { lo = max(n0, half); hi = min(n1, L); }                    // right of the middle: computable after t steps
      else { lo = max(0, L - n1); hi = min(half, L - n0); }                 // left: after L - 1 - t steps
      if (lo >= hi) continue;
      // alpha rows lo .. hi-1 and beta rows lo+1 .. hi
      const int need_a = hi, need_b = L - lo;
      const unsigned long long t0 = wall_clock64();     // 100 MHz
      while (__hip_atomic_load(a.seq_progress + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need_a ||
             __hip_atomic_load(a.seq_progress + B + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need_b) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 2000000000ull) { atomicAdd(a.bad, 1); lo = hi; break; }   // 20 s: the recursion died
      }
      if (lo < hi) break;
    }
    slot[0] = b; slot[1] = lo; slot[2] = hi; slot[3] = L; slot[4] = pad;
  }
  __syncthreads();
  it.b = slot[0]; it.lo = slot[1]; it.hi = slot[2]; it.L = slot[3]; it.pad = slot[4];
  return it.b >= 0;
}
constexpr size_t kStaticLds = 64;       // the occupancy kernels' own static LDS (the queue slot) beside their dynamic segment
constexpr int kLoadDeviceScope = 16;     // cache-policy operand of a buffer load: sc1 (rows another XCD wrote while this kernel runs)

// ------------------------------------------------------------------------------------
// launch 2: occupancies (time-parallel)
// ------------------------------------------------------------------------------------
// STREAM = false: one workgroup per (chunk of frames_per_block frames, sequence), the frames of occupancy launch gam_seg.
// STREAM = true:  persistent workgroups drawing frame ranges from the queue above (one plan for all sequences).
template <int VEC, int XCH, int R, bool STREAM>
__global__ __launch_bounds__(kNT) void den_gamma_kernel(const DenArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ int stream_slot[8];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Hp = a.Hp, D = a.D, Dp = (D + 3) & ~3;
  int b = STREAM ? 0 : blockIdx.y;
  int L = STREAM ? 1 : __builtin_amdgcn_readfirstlane(seq_len(a.lengths, b, a.T));
  int t_first = 0, t_live_end = 0;
  const bool first_launch = a.gam_nseg == 0 || a.gam_seg == 0;
  const LaunchFrames lf(a);
  if constexpr (!STREAM) {
    const int chunk = den_chunk_of_block(blockIdx.x, L, a);
    if (chunk < 0) return;
    const int t_begin = chunk * a.frames_per_block;
    const int t_end = min(t_begin + a.frames_per_block, a.T);
    float* gseq0 = a.grad + (size_t)b * a.T * D;
    if (t_begin >= L) {                               // whole chunk is padding: exact zeros (zeros_like, :58)
      if (first_launch)
        for (size_t i = (size_t)t_begin * D + tid; i < (size_t)t_end * D; i += kNT) gseq0[i] = 0.f;
      return;
    }
    t_live_end = min(t_end, L);
    // padded tail of a chunk that straddles the sequence end
    if (first_launch && t_live_end < t_end)
      for (size_t i = (size_t)t_live_end * D + tid; i < (size_t)t_end * D; i += kNT) gseq0[i] = 0.f;
    t_first = den_next_frame(t_begin, t_live_end, L, lf);
    if (t_first >= t_live_end) return;                // none of this chunk's frames belongs to this launch
  }
  const char* plan = a.plans + (size_t)b * a.plan_stride;
  const PlanHeader* hd = reinterpret_cast<const PlanHeader*>(plan);
  const TilePlan tp = hd->gamma;
  const WaveEntry we = reinterpret_cast<const WaveEntry*>(plan + tp.off_wave_tab)[wave];
  const GroupEntry* gtab = reinterpret_cast<const GroupEntry*>(plan + tp.off_group_tab);
  const uint2* slots = reinterpret_cast<const uint2*>(plan + tp.off_slots);
  const int32_t* row_pdf = reinterpret_cast<const int32_t*>(plan + hd->off_row_pdf);

  float* U = reinterpret_cast<float*>(smem_raw);   // alpha'(t,.)   [Hp]
  float* V = U + Hp;                                // beta(t+1,.)   [Hp]
  float* xr = V + Hp;                               // exp x(t,.)    [Dp]
  float* q = xr + Dp;                               // per-pdf arc sums, natural pdf order [Dp]
  int* rmap = reinterpret_cast<int*>(q + Dp);       // plan row -> pdf-id [ngroups*64]
  float* red = reinterpret_cast<float*>(rmap + tp.ngroups * 64);   // [16]
  const uint32_t lds0 = lds_addr(smem_raw);

  GroupRegs groups;
  groups.load<R>(we, gtab, lane);
  const uint2* wave_slots = slots + (size_t)__builtin_amdgcn_readfirstlane(we.slot_row_begin) * 64 + lane;
  ArcRegs<R> arcs;
  arcs.load(groups.nslots, wave_slots, lds0, lds0 + 4u * (uint32_t)Hp);
  const uint2* tail_slots = wave_slots + (size_t)R * 64;

  if (tid < 16) red[tid] = 0.f;
  for (int i = tid; i < Dp; i += kNT) q[i] = 0.f;   // pdfs without arcs stay zero forever
  for (int i = tid; i < tp.ngroups * 64; i += kNT) rmap[i] = row_pdf[i];
  int bad = 0;
  const float gscale = a.grad_scale_dev ? a.grad_scale * *a.grad_scale_dev : a.grad_scale;
  XRow<kNT, VEC, XCH> xq;
  constexpr int kUV = 1;                             // float4 chunks of U and of V per thread (Hp <= 4*kNT fast path)
  float4 ureg[kUV], vreg[kUV];
  const bool uv_in_regs = Hp <= kUV * 4 * kNT;

  for (;;) {                                         // STREAM: one pass per item of the queue; else exactly one pass
    if constexpr (STREAM) {
      StreamItem it;
      if (!stream_take(a, stream_slot, it)) break;
      b = it.b; L = it.L;
      if (it.pad) {
        float* gz = a.grad + (size_t)b * a.T * D;
        for (size_t i = (size_t)it.lo * D + tid; i < (size_t)it.hi * D; i += kNT) gz[i] = 0.f;
        continue;
      }
      t_first = it.lo; t_live_end = it.hi;
    }
    float* gseq = a.grad + (size_t)b * a.T * D;
    const float* xseq = a.x + (size_t)b * a.T * D;
    const float* aseq = a.alpha_store + (size_t)b * a.T * Hp;
    const float* bseq = a.beta_store + (size_t)b * (a.T + 1) * Hp;
    const XBuf abuf = make_xbuf(aseq, (size_t)a.T * Hp * sizeof(float)), bbuf = make_xbuf(bseq, (size_t)(a.T + 1) * Hp * sizeof(float));
    // Software pipeline over frames: the global loads of frame t+1 (alpha', beta, nnet-output
    // rows) are issued into registers before frame t is evaluated and committed to LDS after
    // the last LDS read of frame t, so HBM latency is off the per-frame critical path.
    // (macros, not lambdas: by-reference captures would put the staging registers on the stack;
    //  STREAM: the state rows may have been written by another XCD while this kernel runs: device-scope loads)
#define GAMMA_PREFETCH(t)                                                                     \
  do {                                                                                        \
    xq.load(xseq + (size_t)(t) * D, D, tid);                                                  \
    if (uv_in_regs) {                                                                         \
      const float* ar_ = aseq + (size_t)(t) * Hp;                                             \
      const float* br_ = bseq + (size_t)((t) + 1) * Hp;                                       \
      _Pragma("unroll") for (int c = 0; c < kUV; c++) {                                       \
        const int i = (c * kNT + tid) * 4;                                                    \
        if (i < Hp) {                                                                         \
          if constexpr (STREAM) {                                                             \
            const u32x4 ua_ = __builtin_amdgcn_raw_buffer_load_b128(abuf, i * 4, (t) * Hp * 4, kLoadDeviceScope);        \
            const u32x4 va_ = __builtin_amdgcn_raw_buffer_load_b128(bbuf, i * 4, ((t) + 1) * Hp * 4, kLoadDeviceScope);  \
            ureg[c] = make_float4(__uint_as_float(ua_.x), __uint_as_float(ua_.y), __uint_as_float(ua_.z), __uint_as_float(ua_.w)); \
            vreg[c] = make_float4(__uint_as_float(va_.x), __uint_as_float(va_.y), __uint_as_float(va_.z), __uint_as_float(va_.w)); \
          } else {                                                                            \
            ureg[c] = *reinterpret_cast<const float4*>(ar_ + i);                              \
            vreg[c] = *reinterpret_cast<const float4*>(br_ + i);                              \
          }                                                                                   \
        }                                                                                     \
      }                                                                                       \
    }                                                                                         \
  } while (0)
#define GAMMA_COMMIT(t)                                                                       \
  do {                                                                                        \
    xq.store(xr, xseq + (size_t)(t) * D, D, tid, a.input_is_exp);                             \
    if (uv_in_regs) {                                                                         \
      _Pragma("unroll") for (int c = 0; c < kUV; c++) {                                       \
        const int i = (c * kNT + tid) * 4;                                                    \
        if (i < Hp) { *reinterpret_cast<float4*>(U + i) = ureg[c];                            \
                      *reinterpret_cast<float4*>(V + i) = vreg[c]; }                          \
      }                                                                                       \
    } else {                                                                                  \
      for (int i = tid * 4; i < Hp; i += kNT * 4) {                                           \
        const u32x4 ua_ = __builtin_amdgcn_raw_buffer_load_b128(abuf, i * 4, (t) * Hp * 4, STREAM ? kLoadDeviceScope : 0);        \
        const u32x4 va_ = __builtin_amdgcn_raw_buffer_load_b128(bbuf, i * 4, ((t) + 1) * Hp * 4, STREAM ? kLoadDeviceScope : 0);  \
        *reinterpret_cast<float4*>(U + i) = make_float4(__uint_as_float(ua_.x), __uint_as_float(ua_.y), __uint_as_float(ua_.z), __uint_as_float(ua_.w)); \
        *reinterpret_cast<float4*>(V + i) = make_float4(__uint_as_float(va_.x), __uint_as_float(va_.y), __uint_as_float(va_.z), __uint_as_float(va_.w)); \
      }                                                                                       \
    }                                                                                         \
  } while (0)
    GAMMA_PREFETCH(t_first);
    GAMMA_COMMIT(t_first);
    __syncthreads();
    for (int t = t_first; t < t_live_end;) {
      float* grow = gseq + (size_t)t * D;
      const int t_next = STREAM ? t + 1 : den_next_frame(t + 1, t_live_end, L, lf);
      const bool have_next = t_next < t_live_end;
      if (have_next) GAMMA_PREFETCH(t_next);
      float s0 = 0.f, s1 = 0.f;
      tile_rows<R, 1>(arcs, groups, tail_slots, lane, U, V, q, rmap, nullptr, s0, s1);
      __syncthreads();
      float g[(VEC * XCH) > 0 ? (VEC * XCH) : 1];
      float part = 0.f;
      if constexpr (XCH > 0) {
#pragma unroll
        for (int c = 0; c < XCH; c++)
#pragma unroll
          for (int k = 0; k < VEC; k++) {
            const int e = (c * kNT + tid) * VEC + k;
            g[c * VEC + k] = e < D ? product_into_sum(xr[e], q[e], part) : 0.f;
          }
      } else {
        for (int e = tid; e < D; e += kNT) part += xr[e] * q[e];
      }
      part = wave_sum(part);
      if (lane == 0) red[wave] = part;
      __syncthreads();                                   // also: every read of U/V/xr of this frame is done
      const float tot = block_total(red, lane);
      const float sc = gscale / tot;
      if (!(tot > 0.f) || !(sc - sc == 0.f)) bad = 1;
      if (a.check && (t == 0 || a.check_all) && tid == 0) den_record_frame_total(a, b, t, tot);
      if constexpr (XCH > 0) {
#pragma unroll
        for (int c = 0; c < XCH; c++) {
          const int e = (c * kNT + tid) * VEC;
          if (e < D) {
            if constexpr (VEC == 4) {
              *reinterpret_cast<float4*>(grow + e) =
                  make_float4(g[c * 4] * sc, g[c * 4 + 1] * sc, g[c * 4 + 2] * sc, g[c * 4 + 3] * sc);
            } else {
              grow[e] = g[c] * sc;
            }
          }
        }
        if (have_next) GAMMA_COMMIT(t_next);
      } else {
        for (int e = tid; e < D; e += kNT) grow[e] = xr[e] * q[e] * sc;
        __syncthreads();                                 // generic-D path re-reads xr/q above
        if (have_next) GAMMA_COMMIT(t_next);
      }
      __syncthreads();   // next frame's operands are in place; q is rewritten by the next frame
      t = t_next;
    }
#undef GAMMA_PREFETCH
#undef GAMMA_COMMIT
    if constexpr (!STREAM) break;
  }
  if (bad && lane == 0) atomicAdd(a.bad, 1);
}

// ------------------------------------------------------------------------------------
// launch 2, two-frame form: frames t and t+1 of a sequence are evaluated TOGETHER.
// The occupancy pass is time-parallel, so its cost is CU time, not latency, and what it
// spends per frame is LDS gather cycles (2 ds_read_b32 per arc), VALU/issue slots and three
// workgroup barriers.  Here alpha'(t,.)/alpha'(t+1,.) and beta(t+1,.)/beta(t+2,.) are
// interleaved in LDS as float2, so ONE ds_read_b64 per operand serves both frames at the
// LDS cost of one ds_read_b32, the arithmetic is packed fp32 (v_pk_mul/v_pk_fma), address
// unpacking and the group bookkeeping are shared, and a pair costs two barriers.  8 waves
// of up to 256 VGPRs: every wave keeps twice the arcs of the 16-wave form in registers.
// The nnet-output rows never enter LDS (each thread multiplies the elements it loaded).
// ------------------------------------------------------------------------------------
constexpr int kNW2 = PLAN_GAM2_WAVES;
constexpr int kNT2 = kNW2 * 64;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const v2f lds_cv2f;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
__device__ __forceinline__ v2f lds_abs2(uint32_t byte_addr) { return *(lds_cv2f*)(byte_addr); }
#pragma clang diagnostic pop

template <int R>
struct ArcRegs2 {
  uint32_t pk[R > 0 ? R : 1];       // absolute LDS byte addresses of the two float2 operands, packed 16:16
  float p[R > 0 ? R : 1];
  __device__ __forceinline__ void load(const int nslot_rows, const uint2* __restrict__ wave_slots,
                                       uint32_t lds_u, uint32_t lds_v) {
#pragma unroll
    for (int s = 0; s < R; s++) {
      uint2 a = make_uint2(0u, 0u);
      if (s < nslot_rows) a = wave_slots[s * 64];
      pk[s] = (lds_u + ((a.x & 0xffffu) << 3)) | ((lds_v + ((a.x >> 16) << 3)) << 16);
      p[s] = __uint_as_float(a.y);
    }
  }
  __device__ __forceinline__ void opaque4(int s) {
    asm volatile("" : "+v"(pk[s]), "+v"(pk[s + 1]), "+v"(pk[s + 2]), "+v"(pk[s + 3]));
  }
  __device__ __forceinline__ void gather(int s, v2f& u, v2f& v) {
    u = lds_abs2(pk[s] & 0xffffu); v = lds_abs2(pk[s] >> 16);
  }
};

// (u.x * p, u.y * p) as two scalar multiplies: a packed multiply by the splat {p, p} makes the
// optimiser keep a 64-bit copy of every arc probability in registers (3 VGPRs per arc instead of 2)
__device__ __forceinline__ v2f scale2(v2f u, float p) { return v2f{u.x * p, u.y * p}; }

__device__ __forceinline__ void tile_store2(v2f acc, int pos, float* __restrict__ q2, const int* __restrict__ row_map) {
  const int nat = row_map[pos];
  if (nat >= 0) *reinterpret_cast<v2f*>(q2 + 2 * nat) = acc;
}

// q2[2*pdf + f] = sum_k p_k * U2[2*i0_k + f] * V2[2*i1_k + f]   (f = 0, 1: the two frames)
// `hook(c)` runs at the start of chunk c (c is a constant after unrolling): the occupancy kernel issues its
// global loads for the next pair there, one per chunk, so that they pass through the CU's vector-memory
// path (64 B per clock) while the gathers keep the LDS busy, instead of in one burst before the arc work.
template <int R, typename Hook>
__device__ __forceinline__ void tile_rows2(ArcRegs2<R>& ar, const GroupRegs& gr, const uint2* __restrict__ tail_slots,
                                           int lane, const float* __restrict__ U2, const float* __restrict__ V2,
                                           float* __restrict__ q2, const int* __restrict__ row_map, Hook hook) {
  constexpr int kChunk = 4;
  static_assert(R % kChunk == 0 && R <= 64 && PYCHAIN_CHUNK == 4, "chunk mask of GroupRegs is built for chunks of 4");
  constexpr int NC = R / kChunk;
  v2f acc = {0.f, 0.f};
  uint32_t m_lo = (uint32_t)gr.endmask, m_hi = (uint32_t)(gr.endmask >> 32), cm = gr.chunkmask;
  asm volatile("" : "+s"(m_lo), "+s"(m_hi), "+s"(cm));
  v2f ub[2][kChunk], vb[2][kChunk];
  if (R > 0) {
    ar.opaque4(0);
#pragma unroll
    for (int k = 0; k < kChunk; k++) ar.gather(k, ub[0][k], vb[0][k]);
  }
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int cb = c & 1;
    wave_priority_by_progress<NC>(c);
    hook(c);
    if (c + 1 < NC) {
      ar.opaque4((c + 1) * kChunk);
#pragma unroll
      for (int k = 0; k < kChunk; k++) ar.gather((c + 1) * kChunk + k, ub[cb ^ 1][k], vb[cb ^ 1][k]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < NC) PYCHAIN_WAIT_LGKM(2 * kChunk); else PYCHAIN_WAIT_LGKM(0);
    __builtin_amdgcn_sched_barrier(0);
    v2f nacc = acc;
#pragma unroll
    for (int k = 0; k < kChunk; k++) nacc = __builtin_elementwise_fma(scale2(ub[cb][k], ar.p[c * kChunk + k]), vb[cb][k], nacc);
    if (__builtin_expect(((cm >> c) & 1u) != 0u, 0)) {
      nacc = acc;
#pragma unroll
      for (int k = 0; k < kChunk; k++) {
        const int sidx = c * kChunk + k;
        nacc = __builtin_elementwise_fma(scale2(ub[cb][k], ar.p[sidx]), vb[cb][k], nacc);
        if (((sidx < 32 ? m_lo : m_hi) >> (sidx & 31)) & 1u) {
          const uint32_t lo_before = sidx < 32 ? (m_lo & ((1u << (sidx & 31)) - 1u)) : m_lo;
          const uint32_t hi_before = sidx < 32 ? 0u : (m_hi & ((1u << (sidx & 31)) - 1u));
          const int g = __builtin_popcount(lo_before) + __builtin_popcount(hi_before);
          tile_store2(nacc, __builtin_amdgcn_readlane(gr.base, g) + lane, q2, row_map);
          nacc = v2f{0.f, 0.f};
        }
      }
    }
    acc = nacc;
  }
  int g = __builtin_popcount(m_lo) + __builtin_popcount(m_hi);
  if (gr.nslots > R) {                           // plan larger than the register budget: stream the tail
    const uint2* sp = tail_slots;
    g = gr.tail_g;
    int cur_base = __builtin_amdgcn_readlane(gr.base, g & 63);
    int remaining = gr.tail_rem;
    for (int s = R; s < gr.nslots; s++) {
      const uint2 a = *sp;
      sp += 64;
      const v2f u = *reinterpret_cast<const v2f*>(U2 + 2 * (a.x & 0xffffu));
      const v2f v = *reinterpret_cast<const v2f*>(V2 + 2 * (a.x >> 16));
      acc = __builtin_elementwise_fma(scale2(u, __uint_as_float(a.y)), v, acc);
      if (--remaining == 0) {
        tile_store2(acc, cur_base + lane, q2, row_map);
        acc = v2f{0.f, 0.f};
        g++;
        cur_base = __builtin_amdgcn_readlane(gr.base, g & 63);
        remaining = __builtin_amdgcn_readlane(gr.n, g & 63);
      }
    }
  }
  for (; g < gr.ngroups; g++)
    tile_store2(v2f{0.f, 0.f}, __builtin_amdgcn_readlane(gr.base, g & 63) + lane, q2, row_map);
}

__device__ __forceinline__ bool den_frame_in_launch(int t, int t_live_end, int L, const LaunchFrames& lf) {
  return t < t_live_end && lf.has(t, L);
}

// STREAM as in den_gamma_kernel: persistent workgroups drawing frame ranges from the queue (one plan for all sequences).
template <int XCH, int R, bool STREAM>
__global__ __launch_bounds__(kNT2) void den_gamma2_kernel(const DenArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ int stream_slot[8];
  constexpr int UVC = 2;                              // float4 chunks of a state row per thread: Hp <= 4 * UVC * kNT2
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Hp = a.Hp, D = a.D, T = a.T, Dp = (D + 3) & ~3;
  int b = STREAM ? 0 : blockIdx.y;
  int L = STREAM ? 1 : __builtin_amdgcn_readfirstlane(seq_len(a.lengths, b, a.T));
  const bool first_launch = a.gam_nseg == 0 || a.gam_seg == 0;
  const LaunchFrames lf(a);
  int t0 = 0, t_live_end = 0, t_lo = 0;               // frames [t_lo, t_live_end) of this pass, t0 = first pair (even)
  if constexpr (!STREAM) {
    const int chunk = den_chunk_of_block(blockIdx.x, L, a);
    if (chunk < 0) return;
    const int t_begin = chunk * a.frames_per_block;               // frames_per_block is even
    const int t_end = min(t_begin + a.frames_per_block, T);
    float* gseq0 = a.grad + (size_t)b * T * D;
    t_live_end = min(t_end, L);
    // pairs (t0, t0+1), t0 even, with at least one frame of this launch
    t0 = t_begin;
    while (t0 < t_live_end && !den_frame_in_launch(t0, t_live_end, L, lf) && !den_frame_in_launch(t0 + 1, t_live_end, L, lf)) t0 += 2;
    // padding of the first launch is exact zeros: a whole chunk past the end, or the tail of one that straddles it
    if (first_launch && max(t_begin, t_live_end) < t_end)
      for (size_t i = (size_t)max(t_begin, t_live_end) * D + tid; i < (size_t)t_end * D; i += kNT2) gseq0[i] = 0.f;
    if (t0 >= t_live_end) return;                     // nothing to evaluate
  }
  const char* plan = a.plans + (size_t)b * a.plan_stride;
  const PlanHeader* hd = reinterpret_cast<const PlanHeader*>(plan);
  const TilePlan tp = hd->gamma2;
  const WaveEntry we = reinterpret_cast<const WaveEntry*>(plan + tp.off_wave_tab)[wave];
  const GroupEntry* gtab = reinterpret_cast<const GroupEntry*>(plan + tp.off_group_tab);
  const uint2* slots = reinterpret_cast<const uint2*>(plan + tp.off_slots);
  const int32_t* row_pdf = reinterpret_cast<const int32_t*>(plan + hd->off_row_pdf);

  float* U2 = reinterpret_cast<float*>(smem_raw);    // [Hp] x {alpha'(t0,.), alpha'(t0+1,.)}
  float* V2 = U2 + 2 * Hp;                            // [Hp] x {beta(t0+1,.), beta(t0+2,.)}
  float* q2 = V2 + 2 * Hp;                            // [Dp] x {frame 0, frame 1} per-pdf arc sums, natural pdf order
  int* rmap = reinterpret_cast<int*>(q2 + 2 * Dp);   // plan row -> pdf-id [ngroups*64]
  float* red = reinterpret_cast<float*>(rmap + tp.ngroups * 64);   // [2][16]
  float* n2 = red + 32;                               // [2][Dp] x {frame 0, frame 1} numerator occupancies (fold only; one buffer per pair parity)
  const uint32_t lds0 = lds_addr(smem_raw);

  GroupRegs groups;
  groups.load<R>(we, gtab, lane);
  const uint2* wave_slots = slots + (size_t)__builtin_amdgcn_readfirstlane(we.slot_row_begin) * 64 + lane;
  ArcRegs2<R> arcs;
  arcs.load(groups.nslots, wave_slots, lds0, lds0 + 8u * (uint32_t)Hp);
  const uint2* tail_slots = wave_slots + (size_t)R * 64;

  if (tid < 32) red[tid] = 0.f;
  for (int i = tid; i < 2 * Dp; i += kNT2) q2[i] = 0.f;          // pdfs without arcs stay zero forever
  for (int i = tid; i < tp.ngroups * 64; i += kNT2) rmap[i] = row_pdf[i];
  int bad = 0;
  const float gscale = a.grad_scale_dev ? a.grad_scale * *a.grad_scale_dev : a.grad_scale;
  const bool fold = a.fold_rows != nullptr;
  const float nscale = a.grad_scale_dev ? a.fold_scale * *a.grad_scale_dev : a.fold_scale;
  int prev_U = 0, prev_b = -1;
  // state rows of a pair: global -> registers (one pair ahead) -> LDS, interleaved
  float4 ua[UVC], ub[UVC], va[UVC], vb[UVC];
  XRow<kNT2, 4, XCH> x0, x1;
#ifdef PYCHAIN_PROFILE_PHASES
  unsigned long long g2ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, g2t = PH_T();
  int g2pairs = 0;
#define G2PH(i) do { const unsigned long long n_ = PH_T(); g2ph[i] += n_ - g2t; g2t = n_; } while (0)
#else
#define G2PH(i) (void)0
#endif

  for (;;) {                                          // STREAM: one pass per item of the queue; else exactly one pass
    if constexpr (STREAM) {
      StreamItem it;
      if (!stream_take(a, stream_slot, it)) break;
      b = it.b; L = it.L;
      if (it.pad) {
        float* gz = a.grad + (size_t)b * T * D;
        for (size_t i = (size_t)it.lo * D + tid; i < (size_t)it.hi * D; i += kNT2) gz[i] = 0.f;
        continue;
      }
      t_lo = it.lo; t_live_end = it.hi; t0 = t_lo & ~1;
    }
    float* gseq = a.grad + (size_t)b * T * D;
    const float* xseq = a.x + (size_t)b * T * D;
    const float* aseq = a.alpha_store + (size_t)b * T * Hp;
    const float* bseq = a.beta_store + (size_t)b * (T + 1) * Hp;
    const XBuf abuf = make_xbuf(aseq, (size_t)T * Hp * sizeof(float)), bbuf = make_xbuf(bseq, (size_t)(T + 1) * Hp * sizeof(float));
    // which frames of [t0, ..) this pass evaluates
    auto wanted = [&](int t) { return STREAM ? (t >= t_lo && t < t_live_end) : den_frame_in_launch(t, t_live_end, L, lf); };
    // numerator fold: thread u (and u + kNT2) owns the u-th distinct pdf of the sequence; the set of
    // touched pdfs is the same in every frame, so n2 needs no clearing between pairs (only between sequences)
    const int U = fold ? a.fold_ucount[b] : 0;
    const int32_t* upd = a.fold_upd + (size_t)b * a.fold_K;
    const float* frows = a.fold_rows + (size_t)b * T * a.fold_K;
    int pd0 = -1, pd1 = -1;
    if (fold) {
      if (b != prev_b) {
        if (!STREAM || prev_b < 0) { for (int i = tid; i < 4 * Dp; i += kNT2) n2[i] = 0.f; }
        else {
          // the previous sequence's pdfs back to zero (the entries it touched, not the whole table)
          const int32_t* pupd = a.fold_upd + (size_t)prev_b * a.fold_K;
          for (int u = tid; u < prev_U; u += kNT2) { const int n = pupd[u]; n2[2 * n] = 0.f; n2[2 * n + 1] = 0.f; n2[2 * Dp + 2 * n] = 0.f; n2[2 * Dp + 2 * n + 1] = 0.f; }
        }
        prev_b = b; prev_U = U;
        __syncthreads();
      }
      if (tid < U) pd0 = upd[tid];
      if (tid + kNT2 < U) pd1 = upd[tid + kNT2];
    }
    // load number n of a pair's state rows: n = 4 * c + {0: alpha'(t), 1: alpha'(t+1), 2: beta(t+1), 3: beta(t+2)}
    // (STREAM: device-scope loads - another XCD may have written the rows while this kernel runs)
    auto row_load = [&](XBuf buf, int row, int i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(buf, i * 4, row * Hp * 4, STREAM ? kLoadDeviceScope : 0);
      return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto prefetch_one = [&](int n, int t) {
      const int c = n >> 2, i = (c * kNT2 + tid) * 4;
      if (c < UVC && i < Hp) {
        switch (n & 3) {
          case 0: ua[c] = row_load(abuf, t, i); break;
          case 1: ub[c] = row_load(abuf, min(t + 1, T - 1), i); break;
          case 2: va[c] = row_load(bbuf, t + 1, i); break;
          default: vb[c] = row_load(bbuf, min(t + 2, T), i); break;
        }
      }
    };
#define GAMMA2_COMMIT()                                                                          \
  do {                                                                                           \
    _Pragma("unroll") for (int c = 0; c < UVC; c++) {                                            \
      const int i = (c * kNT2 + tid) * 4;                                                        \
      if (i < Hp) {                                                                              \
        *reinterpret_cast<float4*>(U2 + 2 * i) = make_float4(ua[c].x, ub[c].x, ua[c].y, ub[c].y); \
        *reinterpret_cast<float4*>(U2 + 2 * i + 4) = make_float4(ua[c].z, ub[c].z, ua[c].w, ub[c].w); \
        *reinterpret_cast<float4*>(V2 + 2 * i) = make_float4(va[c].x, vb[c].x, va[c].y, vb[c].y); \
        *reinterpret_cast<float4*>(V2 + 2 * i + 4) = make_float4(va[c].z, vb[c].z, va[c].w, vb[c].w); \
      }                                                                                          \
    }                                                                                            \
  } while (0)
    for (int n = 0; n < 4 * UVC; n++) prefetch_one(n, t0);
    GAMMA2_COMMIT();
    __syncthreads();
    int npar = 0;
    while (t0 < t_live_end) {
#ifdef PYCHAIN_PROFILE_PHASES
      g2pairs++;
#endif
      const bool valid0 = wanted(t0), valid1 = wanted(t0 + 1);
      int tn = t0 + 2;
      while (tn < t_live_end && !wanted(tn) && !wanted(tn + 1)) tn += 2;
      const bool have_next = tn < t_live_end;
      // this pair's nnet-output rows (used after the arc work) and the next pair's state rows
      x0.load(xseq + (size_t)t0 * D, D, tid);
      x1.load(xseq + (size_t)min(t0 + 1, T - 1) * D, D, tid);
      float r00 = 0.f, r01 = 0.f, r10 = 0.f, r11 = 0.f;  // numerator rows of this pair (in flight during the arc work)
      const float* fr0 = frows + (size_t)t0 * a.fold_K;
      const float* fr1 = frows + (size_t)min(t0 + 1, T - 1) * a.fold_K;
      if (pd0 >= 0) { r00 = fr0[tid]; r10 = fr1[tid]; }
      if (pd1 >= 0) { r01 = fr0[tid + kNT2]; r11 = fr1[tid + kNT2]; }
      G2PH(0);
      // the next pair's state rows are requested inside the arc loop, one load per chunk (a pair that is the
      // last of its pass re-reads its own rows: no branch per chunk)
      const int tpre = have_next ? tn : t0;
      tile_rows2<R>(arcs, groups, tail_slots, lane, U2, V2, q2, rmap, [&](int c) {
        constexpr int NC = R / 4 > 0 ? R / 4 : 1;
        // spread over the first chunks, two chunks apart where the loop is long enough
        constexpr int kStep = NC >= 16 ? 2 : 1;
        if (c % kStep == 0 && c / kStep < 4 * UVC) prefetch_one(c / kStep, tpre);
      });
      if (R / 4 < 4 * UVC * (R / 4 >= 16 ? 2 : 1))          // arc loop shorter than the list of loads: the rest here
        for (int n = (R / 4) / (R / 4 >= 16 ? 2 : 1); n < 4 * UVC; n++) prefetch_one(n, tpre);
      G2PH(1);
      float* n2p = n2 + (npar ? 2 * Dp : 0);             // this buffer was last read two pairs ago
      if (fold) {
        if (pd0 >= 0) *reinterpret_cast<v2f*>(n2p + 2 * pd0) = v2f{r00, r10};
        if (pd1 >= 0) *reinterpret_cast<v2f*>(n2p + 2 * pd1) = v2f{r01, r11};
        for (int u = tid + 2 * kNT2; u < U; u += kNT2) *reinterpret_cast<v2f*>(n2p + 2 * upd[u]) = v2f{fr0[u], fr1[u]};
      }
      npar ^= 1;
      G2PH(2);
      __syncthreads();                                   // q2 (and n2) complete; every gather of this pair is done
      G2PH(3);
      float g0[4 * XCH], g1[4 * XCH];
      float part0 = 0.f, part1 = 0.f;
      auto products = [&](auto mode) {                   // (the mode is uniform: one branch, not a select per element)
#pragma unroll
        for (int c = 0; c < XCH; c++) {
          const int e = (c * kNT2 + tid) * 4;
          float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa;
          if (e < D) { qa = *reinterpret_cast<const float4*>(q2 + 2 * e); qb = *reinterpret_cast<const float4*>(q2 + 2 * e + 4); }
          const float qf0[4] = {qa.x, qa.z, qb.x, qb.z}, qf1[4] = {qa.y, qa.w, qb.y, qb.w};
#pragma unroll
          for (int k = 0; k < 4; k++) {
            g0[c * 4 + k] = e < D ? product_into_sum(clamp_exp(x0.v[c * 4 + k], decltype(mode)::value), qf0[k], part0) : 0.f;
            g1[c * 4 + k] = e < D ? product_into_sum(clamp_exp(x1.v[c * 4 + k], decltype(mode)::value), qf1[k], part1) : 0.f;
          }
        }
      };
      if (a.input_is_exp == kXExpClamp) products(std::integral_constant<int, kXExpClamp>{});
      else if (a.input_is_exp == kXIdentity) products(std::integral_constant<int, kXIdentity>{});
      else products(std::integral_constant<int, kXClamp>{});
      part0 = wave_sum(part0); part1 = wave_sum(part1);
      if (lane == 0) { red[wave] = part0; red[16 + wave] = part1; }
      G2PH(4);
      if (have_next) GAMMA2_COMMIT();                    // U2/V2 are free since the barrier above
      G2PH(5);
      __syncthreads();                                   // totals visible; next operands in place; q2 read
      G2PH(6);
      const float tot0 = block_total(red, lane), tot1 = block_total(red + 16, lane);
      const float sc0 = gscale / tot0, sc1 = gscale / tot1;
      if (valid0 && (!(tot0 > 0.f) || !(sc0 - sc0 == 0.f))) bad = 1;
      if (valid1 && (!(tot1 > 0.f) || !(sc1 - sc1 == 0.f))) bad = 1;
      if (a.check && tid == 0) {
        if (valid0 && (t0 == 0 || a.check_all)) den_record_frame_total(a, b, t0, tot0);
        if (valid1 && a.check_all) den_record_frame_total(a, b, t0 + 1, tot1);
      }
      float* grow0 = gseq + (size_t)t0 * D;
      float* grow1 = grow0 + D;
#pragma unroll
      for (int c = 0; c < XCH; c++) {
        const int e = (c * kNT2 + tid) * 4;
        if (e < D) {
          float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;     // numerator occupancies {f0,f1} x 4 pdfs
          if (fold) { na = *reinterpret_cast<const float4*>(n2p + 2 * e); nb = *reinterpret_cast<const float4*>(n2p + 2 * e + 4); }
          // (g * sc) rounded, then + numerator: bit-identical to the unfused order (occupancy pass, then
          // the numerator accumulated into the stored gradient)
#define G2(gv, scv, nv) mul_add_mul_rn((gv), (scv), (nv), nscale)
          if (valid0) *reinterpret_cast<float4*>(grow0 + e) = make_float4(G2(g0[c * 4], sc0, na.x), G2(g0[c * 4 + 1], sc0, na.z),
                                                                             G2(g0[c * 4 + 2], sc0, nb.x), G2(g0[c * 4 + 3], sc0, nb.z));
          if (valid1) *reinterpret_cast<float4*>(grow1 + e) = make_float4(G2(g1[c * 4], sc1, na.y), G2(g1[c * 4 + 1], sc1, na.w),
                                                                             G2(g1[c * 4 + 2], sc1, nb.y), G2(g1[c * 4 + 3], sc1, nb.w));
#undef G2
        }
      }
      t0 = tn;
      G2PH(7);
    }
#undef GAMMA2_COMMIT
    if constexpr (!STREAM) break;
  }
#ifdef PYCHAIN_PROFILE_PHASES
  if (lane == 0 && b == 0 && (wave == 0 || wave == 7) && g2pairs > 0 && blockIdx.x == 20)
    printf("gamma2 wave %d pairs %d cycles/pair: issue-loads %llu arcs %llu fold %llu bar1 %llu products %llu commit %llu bar2 %llu scale+store %llu\n",
           wave, g2pairs, g2ph[0] / g2pairs, g2ph[1] / g2pairs, g2ph[2] / g2pairs, g2ph[3] / g2pairs, g2ph[4] / g2pairs,
           g2ph[5] / g2pairs, g2ph[6] / g2pairs, g2ph[7] / g2pairs);
#endif
  if (bad && lane == 0) atomicAdd(a.bad, 1);
}

__global__ void den_gate_kernel(const int32_t* progress, int target, int32_t* bad) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();              // 100 MHz
  // (relaxed: an acquire here would invalidate this XCD's L2 on every poll; the kernel that follows in
  // stream order acquires at its start)
  while (__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > 2000000000ull) { atomicAdd(bad, 1); break; }   // 20 s: the recursion died
  }
}

template <typename K>
hipError_t launch_one(K kern, const DenArgs& a, dim3 grid, size_t lds, hipStream_t st, int nthreads = kNT) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(nthreads), lds, st, a);
  return hipGetLastError();
}

// the two-frame occupancy kernel: what it supports, and its launch
inline size_t gamma2_lds_bytes(const DenArgs& a, int gamma_max_groups) {
  return sizeof(float) * (4 * (size_t)a.Hp + (a.fold_rows ? 6 : 2) * (size_t)((a.D + 3) & ~3) + (size_t)gamma_max_groups * 64 + 32);
}
inline bool gamma2_eligible(const DenArgs& a, int rows2, int gamma_max_groups) {
  const bool off = a.knobs.gamma16 != 0;                               // test / tuning option: force the one-frame kernel
  return !off && rows2 > 0 && a.D % 4 == 0 && a.D <= 4 * 2 * kNT2 && a.Hp <= 4032 /* packed 16-bit addresses of float2 */ &&
         a.frames_per_block % 2 == 0 && gamma2_lds_bytes(a, gamma_max_groups) + kStaticLds <= 160 * 1024;
}
// rows = slot-rows per wave the plan needs (0 = unknown: stream everything); plans larger
// than kMaxResident keep the first kMaxResident rows in registers and stream their tail.
inline int pick_r(const DenArgs& a, int rows, int lds_words) {
  if (rows <= 0 || (PYCHAIN_ARC_PACKED && lds_words * 4 > 65535)) return 0;   // packed 16-bit LDS addresses
  if (rows <= 16) return 16;
  if (rows <= 32) return 32;
  if (rows <= PLAN_RESIDENT_FIT && kMaxResident > PLAN_RESIDENT_FIT) return PLAN_RESIDENT_FIT;
  return kMaxResident;
}

template <int XCH, bool STREAM>
hipError_t launch_gamma2(const DenArgs& a, int rows2, size_t lds, dim3 grid, hipStream_t st) {
  if (rows2 <= 16) return launch_one(den_gamma2_kernel<XCH, 16, STREAM>, a, grid, lds, st, kNT2);
  if (rows2 <= 32) return launch_one(den_gamma2_kernel<XCH, 32, STREAM>, a, grid, lds, st, kNT2);
  return launch_one(den_gamma2_kernel<XCH, 64, STREAM>, a, grid, lds, st, kNT2);
}
// the streamed form of the one-frame kernel: float4 rows, every arc of a wave in registers
template <int XCH>
hipError_t launch_gamma_stream(const DenArgs& a, int r, size_t lds, dim3 grid, hipStream_t st) {
  if (r <= 16) return launch_one(den_gamma_kernel<4, XCH, 16, true>, a, grid, lds, st);
  if (r <= 32) return launch_one(den_gamma_kernel<4, XCH, 32, true>, a, grid, lds, st);
  return launch_one(den_gamma_kernel<4, XCH, kMaxResident, true>, a, grid, lds, st);
}
inline bool gamma_stream_shape_ok(const DenArgs& a, int hint, int gamma_max_groups) {
  if (a.plan_stride != 0) return false;
  if (gamma2_eligible(a, (hint >> 20) & 511, gamma_max_groups)) return true;
  const int r = pick_r(a, (hint >> 10) & 1023, 2 * a.Hp);
  return a.D % 4 == 0 && a.D <= 4 * 4 * kNT && r > 0;
}

// The lazy-normalisation recursion (den_lazy.inc.h) in its 16-wave shape serves the shape the benchmarks run: nnet-output
// row and state vector within its fixed LDS map, every arc of a wave in registers, at most LzNarrow::kMaxGroups groups
// per wave (bit 30 of the plan hint), the whole sequence in one launch.
// (`dma`: rows by LDS-direct loads, which take any row length; rows through registers are float4 loads: D % 4 == 0)
inline bool lazy_shape_ok(const DenArgs& a, int hint, bool dma = false) {
  const int rows = hint & 1023;
  return ((hint >> 30) & 1) && (dma || a.D % 4 == 0) && a.D <= (int)LzNarrow::kMaxPdfs && a.Hp <= (int)LzNarrow::kMaxStates && rows > 0 &&
         rows <= kMaxResident && PLAN_REC_WAVES == 16 && a.plan_stride >= 0;
}
hipError_t launch_lazy(const DenArgs& a, int hint, hipStream_t st) {
  const dim3 grid(2 * a.B);
  const int rows = hint & 1023;
  if (rows <= 16) return launch_one(den_recursion_lazy_kernel<16, LzNarrow>, a, grid, kLzBytes, st);
  if (rows <= 32) return launch_one(den_recursion_lazy_kernel<32, LzNarrow>, a, grid, kLzBytes, st);
  if (rows <= PLAN_RESIDENT_FIT) return launch_one(den_recursion_lazy_kernel<PLAN_RESIDENT_FIT, LzNarrow>, a, grid, kLzBytes, st);
  return launch_one(den_recursion_lazy_kernel<kMaxResident, LzNarrow>, a, grid, kLzBytes, st);
}
// ... and in its 8-wave shape (LzWide): the plan's 8-wave dealing joins the 16 waves in pairs, so a wave owns at most
// twice the slot-rows and twice the groups of the 16-wave hint.  Chosen where the 16-wave shape does not fit (D > 4096).
inline bool wide_shape_ok(const DenArgs& a, int hint) {
  const int rows = hint & 1023;
  return ((hint >> 30) & 1) && a.D % 4 == 0 && a.D <= (int)LzWide<5>::kMaxPdfs && a.Hp <= (int)LzWide<5>::kMaxStates && rows > 0 &&
         rows <= 40 && PLAN_REC_WAVES == 16 && a.plan_stride >= 0;
}
template <int XCH>
hipError_t launch_wide_x(const DenArgs& a, int rows, hipStream_t st) {
  const dim3 grid(2 * a.B);
  typedef LzWide<XCH> M;
  if (rows <= 16) return launch_one(den_recursion_lazy_kernel<32, M>, a, grid, M::kBytes, st, M::kWaves * 64);
  if (rows <= 32) return launch_one(den_recursion_lazy_kernel<64, M>, a, grid, M::kBytes, st, M::kWaves * 64);
  return launch_one(den_recursion_lazy_kernel<80, M>, a, grid, M::kBytes, st, M::kWaves * 64);
}
// the 16-wave shape with LDS-direct nnet-output rows (LzDma): D <= 9216, Hp <= 3072
inline bool dma_shape_ok(const DenArgs& a, int hint) {
  const int rows = hint & 1023;
  return ((hint >> 30) & 1) && a.D <= (int)LzDma::kMaxPdfs && a.Hp <= (int)LzDma::kMaxStates && rows > 0 &&
         rows <= kMaxResident && PLAN_REC_WAVES == 16 && a.plan_stride >= 0;
}
template <typename M>
hipError_t launch_dma_m(const DenArgs& a, int rows, hipStream_t st) {
  const dim3 grid(2 * a.B);
  if (rows <= 16) return launch_one(den_recursion_lazy_kernel<16, M>, a, grid, M::kBytes, st);
  if (rows <= 32) return launch_one(den_recursion_lazy_kernel<32, M>, a, grid, M::kBytes, st);
  return launch_one(den_recursion_lazy_kernel<kMaxResident, M>, a, grid, M::kBytes, st);
}
// the plan holds two-copy tiles (hint bit 29) and the shape fits the two-copy map: 32-row loops, rows of up to 4096 pdfs
inline bool two_copy_shape_ok(const DenArgs& a, int hint) {
#ifdef PYCHAIN_NO_SPLIT_ARCS
  (void)a; (void)hint;
  return false;
#endif
  return ((hint >> 29) & 1) && a.knobs.den_two_copy != 0 && lazy_shape_ok(a, hint, true) && (hint & 1023) <= 32 &&
         a.D <= (int)LzNarrowDma2::kMaxPdfs && a.Hp <= (int)LzNarrowDma2::kMaxStates;
}
hipError_t launch_dma(const DenArgs& a, int hint, hipStream_t st) {
#ifndef PYCHAIN_NO_SPLIT_ARCS                          /* (the two-copy map takes arcs in the split form only) */
  if (two_copy_shape_ok(a, hint)) {
    const dim3 grid(2 * a.B);
    if ((hint & 1023) <= 16) return launch_one(den_recursion_lazy_kernel<16, LzNarrowDma2>, a, grid, LzNarrowDma2::kBytes, st);
    return launch_one(den_recursion_lazy_kernel<32, LzNarrowDma2>, a, grid, LzNarrowDma2::kBytes, st);
  }
#endif
  // the map of C1-C3 where the shape fits it, else the one for rows of up to 9216 pdfs
  if (lazy_shape_ok(a, hint, true)) return launch_dma_m<LzNarrowDma>(a, hint & 1023, st);
  return launch_dma_m<LzDma>(a, hint & 1023, st);
}
hipError_t launch_wide(const DenArgs& a, int hint, hipStream_t st) {
  const int rows = hint & 1023;
  if (a.knobs.den_wide == 2 && lazy_shape_ok(a, hint, true))       // experiment: twelve waves (rows of a wave <= 56, checked by the kernel)
    return launch_one(den_recursion_lazy_kernel<56, LzNarrowDma12>, a, dim3(2 * a.B), LzNarrowDma12::kBytes, st, 12 * 64);
  return a.D <= (int)LzWide<2>::kMaxPdfs ? launch_wide_x<2>(a, rows, st) : launch_wide_x<5>(a, rows, st);
}

// Two sequences per workgroup (den_pair.inc.h): one plan for all sequences, nnet-output rows and state vectors
// within its fixed LDS map, every arc of a plan wave in registers, the whole sequence in one launch.
inline bool pair_shape_ok(const DenArgs& a, int hint) {
  const int rows = hint & 1023;
  return a.plan_stride == 0 && a.D % 4 == 0 && a.D <= 4096 && a.Hp <= 4096 && rows > 0 && rows <= kMaxResident &&
         PLAN_REC_WAVES == 16 && a.B >= 2;
}
hipError_t launch_pair(const DenArgs& a, int hint, hipStream_t st) {
  const dim3 grid(2 * ((a.B + 1) / 2));
  const int rows = hint & 1023;
  if (rows <= 16) return launch_one(den_recursion_pair_kernel<16>, a, grid, kPrBytes, st, kPrNT);
  if (rows <= 32) return launch_one(den_recursion_pair_kernel<32>, a, grid, kPrBytes, st, kPrNT);
  return launch_one(den_recursion_pair_kernel<kMaxResident>, a, grid, kPrBytes, st, kPrNT);
}

template <int VEC, int XCH>
hipError_t launch_r(const DenArgs& a, int hint, size_t lds_rec, size_t lds_gam, int gx, hipStream_t st, int gamma_max_groups) {
  hipError_t e = hipSuccess;
  if ((a.phase_mask & 1) && a.pair) {
    e = launch_pair(a, hint, st);
    if (e != hipSuccess) return e;
  } else if ((a.phase_mask & 1) && a.lazy) {
    e = a.wide == 1 ? launch_wide(a, hint, st) : (a.wide == 2 ? launch_dma(a, hint, st) : launch_lazy(a, hint, st));
    if (e != hipSuccess) return e;
  } else if (a.phase_mask & 1) {
    const dim3 grid(2 * a.B);
    // <4, 1> (D <= 4096, D % 4 == 0) double-buffers the nnet-output row: gathered operands end at 32 KiB + 4 Hp
    constexpr bool DB = VEC == 4 && XCH == 1;
    switch (pick_r(a, hint & 1023, DB ? 2 * (kXOff / 4) + a.Hp : a.Hp + ((a.D + 3) & ~3))) {
      case 0: e = launch_one(den_recursion_kernel<VEC, XCH, 0, DB>, a, grid, lds_rec, st); break;
      case 16: e = launch_one(den_recursion_kernel<VEC, XCH, 16, DB>, a, grid, lds_rec, st); break;
      case 32: e = launch_one(den_recursion_kernel<VEC, XCH, 32, DB>, a, grid, lds_rec, st); break;
      case PLAN_RESIDENT_FIT: e = launch_one(den_recursion_kernel<VEC, XCH, PLAN_RESIDENT_FIT, DB>, a, grid, lds_rec, st); break;
      default: e = launch_one(den_recursion_kernel<VEC, XCH, kMaxResident, DB>, a, grid, lds_rec, st); break;
    }
    if (e != hipSuccess) return e;
  }
  if (a.phase_mask & 2) {
    const bool stream = (a.stream & 2) != 0;            // ONE persistent launch over the whole queue (DenArgs::stream)
    const dim3 grid = stream ? dim3(a.stream_blocks) : dim3(gx, a.B);
    if (gamma2_eligible(a, (hint >> 20) & 511, gamma_max_groups)) {
      const size_t lds2 = gamma2_lds_bytes(a, gamma_max_groups);
      const int r2 = (hint >> 20) & 511;
      if (stream) return a.D <= 4 * kNT2 ? launch_gamma2<1, true>(a, r2, lds2, grid, st) : launch_gamma2<2, true>(a, r2, lds2, grid, st);
      return a.D <= 4 * kNT2 ? launch_gamma2<1, false>(a, r2, lds2, grid, st) : launch_gamma2<2, false>(a, r2, lds2, grid, st);
    }
    const int r = pick_r(a, (hint >> 10) & 1023, 2 * a.Hp);
    if (stream) {
      if constexpr (VEC == 4 && XCH > 0) return launch_gamma_stream<XCH>(a, r, lds_gam, grid, st);
      else return hipErrorInvalidValue;                 // (den_stream_eligible said no)
    }
    switch (r) {
      case 0: e = launch_one(den_gamma_kernel<VEC, XCH, 0, false>, a, grid, lds_gam, st); break;
      case 16: e = launch_one(den_gamma_kernel<VEC, XCH, 16, false>, a, grid, lds_gam, st); break;
      case 32: e = launch_one(den_gamma_kernel<VEC, XCH, 32, false>, a, grid, lds_gam, st); break;
      default: e = launch_one(den_gamma_kernel<VEC, XCH, kMaxResident, false>, a, grid, lds_gam, st); break;
    }
  }
  return e;
}

}  // namespace

// Host-side view of the frame -> (occupancy launch, workgroup) mapping for the CPU tests (tests/test_api.py):
// out[0] = grid.x of the launch, out[1 + k] = chunk of block k (or -1); returns 1 if frame t belongs to the launch.
int den_debug_launch_map(int T, int L, int t, int frames_per_block, int nseg, const int* seg_bound, int seg,
                         int* out, int out_len) {
  DenArgs a;
  memset(&a, 0, sizeof(a));
  a.T = T; a.frames_per_block = frames_per_block; a.gam_nseg = nseg; a.gam_seg = seg;
  for (int s = 0; s < nseg && s < 16; s++) a.seg_bound[s] = seg_bound[s];
  const int gx = den_compact_grid_x(a);
  if (out && out_len > 0) out[0] = gx;
  for (int k = 0; out && k < gx && 1 + k < out_len; k++) out[1 + k] = den_chunk_of_block(k, L, a);
  return LaunchFrames(a).has(t, L) && t < L ? 1 : 0;
}

hipError_t launch_den_finish(const DenArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(den_finish_kernel, dim3(a.B), dim3(kFinNT), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_den_gate(const int32_t* progress, int target, int32_t* bad, hipStream_t st) {
  hipLaunchKernelGGL(den_gate_kernel, dim3(1), dim3(64), 0, st, progress, target, bad);
  return hipGetLastError();
}

bool den_lazy_eligible(const DenArgs& a, int resident_slot_rows) { return lazy_shape_ok(a, resident_slot_rows); }
bool den_stream_eligible(const DenArgs& a, int gamma_max_groups, int resident_slot_rows) {
  return (a.lazy || a.pair) && gamma_stream_shape_ok(a, resident_slot_rows, gamma_max_groups);
}
bool den_wide_eligible(const DenArgs& a, int resident_slot_rows) { return wide_shape_ok(a, resident_slot_rows); }
bool den_dma_eligible(const DenArgs& a, int resident_slot_rows) { return lazy_shape_ok(a, resident_slot_rows, true) || dma_shape_ok(a, resident_slot_rows); }
const char* den_recursion_kernel_name(const DenArgs& a, int resident_slot_rows) {
  if (a.pair) return "den_recursion_pair_kernel";
  if (a.lazy && a.wide == 1 && a.knobs.den_wide == 2) return "den_recursion_lazy_kernel<12 waves>";
  if (a.lazy && a.wide == 2 && two_copy_shape_ok(a, resident_slot_rows)) return "den_recursion_lazy_kernel<two copies>";
  if (a.lazy) return a.wide == 1 ? "den_recursion_lazy_kernel<wide>" : (a.wide == 2 ? "den_recursion_lazy_kernel<dma>" : "den_recursion_lazy_kernel");
  return "den_recursion_kernel";
}
const char* den_occupancy_kernel_name(const DenArgs& a, int gamma_max_groups, int resident_slot_rows) {
  return gamma2_eligible(a, (resident_slot_rows >> 20) & 511, gamma_max_groups) ? "den_gamma2_kernel" : "den_gamma_kernel";
}
bool den_pair_eligible(const DenArgs& a, int resident_slot_rows) { return pair_shape_ok(a, resident_slot_rows); }
int den_recursion_blocks(const DenArgs& a) { return a.pair ? 2 * ((a.B + 1) / 2) : 2 * a.B; }

bool den_uses_gamma2(const DenArgs& a, int gamma_max_groups, int resident_slot_rows) {
  return gamma2_eligible(a, (resident_slot_rows >> 20) & 511, gamma_max_groups);
}

hipError_t launch_den(const DenArgs& a, int gamma_max_groups, int resident_slot_rows, hipStream_t st,
                      const char** why) {
  const int Dp = (a.D + 3) & ~3;
  const bool db = a.D % 4 == 0 && a.D <= 4 * kNT;        // the <4, 1> instantiation: two 16 KiB nnet-output buffers
  const size_t lds_rec = sizeof(float) * (3 * (size_t)a.Hp + (db ? 2 * (size_t)(kXOff / 4) : (size_t)Dp) + 32);
  const size_t lds_gam = sizeof(float) * (2 * (size_t)a.Hp + 2 * (size_t)Dp + (size_t)gamma_max_groups * 64 + 16);
  if (lds_rec > 160 * 1024 || lds_gam + kStaticLds > 160 * 1024) {
    *why = "state vector + nnet-output row do not fit the 160 KiB LDS of one CU";
    return hipErrorInvalidValue;
  }
  // rows are addressed as 32-bit byte offsets into one sequence's slab (buffer loads / stores)
  if ((size_t)a.T * a.D * 4 >= (size_t)1 << 31 || ((size_t)a.T + 1) * a.Hp * 4 >= (size_t)1 << 31) {
    *why = "one sequence's nnet-output slab or state trajectory reaches 2 GiB";
    return hipErrorInvalidValue;
  }
  const int gx = den_compact_grid_x(a);
  const int D = a.D, r = resident_slot_rows;
  if (D % 4 == 0) {
    if (D <= 4 * 1 * kNT) return launch_r<4, 1>(a, r, lds_rec, lds_gam, gx, st, gamma_max_groups);
    if (D <= 4 * 2 * kNT) return launch_r<4, 2>(a, r, lds_rec, lds_gam, gx, st, gamma_max_groups);
    if (D <= 4 * 3 * kNT) return launch_r<4, 3>(a, r, lds_rec, lds_gam, gx, st, gamma_max_groups);
    if (D <= 4 * 4 * kNT) return launch_r<4, 4>(a, r, lds_rec, lds_gam, gx, st, gamma_max_groups);
  } else if (D <= 4 * kNT) {
    return launch_r<1, 4>(a, r, lds_rec, lds_gam, gx, st, gamma_max_groups);
  }
  return launch_r<1, 0>(a, r, lds_rec, lds_gam, gx, st, gamma_max_groups);
}

}  // namespace pychain_hip
