// den_common.inc.h - what the denominator translation units share (included inside namespace pychain_hip { namespace {):
// the arcs of a wave in registers, the wave's group table, one frame of a tile plan, the row normalisation of the
// two-barrier kernels, launch helpers.  den_rec.hip (two-barrier recursion), den_lazy.hip (lazy and pair recursions)
// and den_kernels.hip (occupancy kernels, den_finish_kernel, dispatch) are compiled side by side.
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
            // (|wv|: beta's leaky probabilities carry a flag in their sign bit - plan.cpp, "states on several lanes")
            if (wvec) { const float wv = wvec[pos]; asm volatile("v_fma_f32 %0, %1, |%2|, %0" : "+v"(s1) : "v"(nacc), "v"(wv)); }
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
        if constexpr (MODE == 0) { s0 += acc; if (wvec) s1 += acc * __builtin_fabsf(wvec[cur_base + lane]); }
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
      // (r + add) * inv as one fma per element; a position whose leaky probability has the sign bit set takes no constant (a
      // state's lanes after its first - plan.cpp, "states on several lanes" -, marked by the kernel before the first frame)
      const float ai = add * inv;
      const float4 l = *reinterpret_cast<const float4*>(lk + i);
      auto gate = [&](float lv) { return __uint_as_float(__float_as_uint(ai) & ~(uint32_t)((int32_t)__float_as_uint(lv) >> 31)); };
      v = make_float4(__builtin_fmaf(r.x, inv, gate(l.x)), __builtin_fmaf(r.y, inv, gate(l.y)), __builtin_fmaf(r.z, inv, gate(l.z)),
                      __builtin_fmaf(r.w, inv, gate(l.w)));
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
  // (device scope: den_finish_kernel may read it while the occupancy launch is still running - DenArgs::occ_done)
  __hip_atomic_store(a.gtot + (size_t)b * a.T + t, frame_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// block total of per-wave partials: red[16] in LDS (entries >= kNW stay zero)
__device__ __forceinline__ float block_total(const float* red, int lane) { return dpp_row_sum(red[lane & 15]); }

constexpr int kXOff = 16384;             // two-barrier recursion, double-buffered nnet-output row: byte distance of the two buffers

template <typename K>
hipError_t launch_one(K kern, const DenArgs& a, dim3 grid, size_t lds, hipStream_t st, int nthreads = kNT) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(nthreads), lds, st, a);
  return hipGetLastError();
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

