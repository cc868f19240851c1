// den_pair.inc.h - alpha / beta recursions of TWO sequences per workgroup (included by den_lazy.hip inside
// its anonymous namespace, after den_lazy.inc.h).
//
// From B = 128 on the 2B workgroups of den_recursion_kernel fill the chip and nothing overlaps them any more
// ("Batch size").  The denominator graph is shared by all sequences, so two of them can share a
// workgroup: ONE set of arc registers and one address unpack per arc, the two state vectors interleaved in LDS
// as float2 {A, B}, the two nnet-output rows likewise - one ds_read_b64 per operand and one packed multiply +
// fma per arc serve both sequences (an LDS gather costs the same for 8 bytes per lane as for 4).  The
// recursions of B sequences then take B workgroups instead of 2B, 1.4 times as long each (the serial tail of a
// frame doubles: profiles/r02_pair_phase_timers.txt), and the other CUs run the occupancy launches and the
// numerator meanwhile: B = 128 7.58 -> 5.85 ms per fused step.  Chosen by api.hip:den_call_is_pair (a fused loss: B >= 100 on 256 CUs; the denominator alone: B > 128).
//
// The arithmetic of a sequence is EXACTLY that of den_recursion_kernel (state vector normalised in place, two
// barriers per frame; packed fp32 lanes are independent and IEEE): same products, same sums in the same order -
// same workgroup shape, same plan waves, same reductions - so even the totals agree bit for bit, and the tests
// compare the two kernels with torch.equal.
//
// Workgroup: 1024 threads = 16 waves of 128 VGPRs: 40 slot-rows x 2 registers per arc as in the one-sequence
// kernels, and the gather pipeline runs two slot-rows deep instead of four (its buffers are float2 now).
// Workgroup p < P = ceil(B/2) walks alpha of sequences 2p, 2p+1 forward, workgroup P + p their beta backward;
// a sequence runs its own number of steps and is masked out (no stores, no flags) once it is done.
//
// LDS map (absolute byte addresses; the dynamic segment starts at 0, checked):
//   [0, 32K)      state vectors, float2[<= 4096] {A, B} (normalised: the gather operand)
//   [32K, 64K)    nnet-output buffer 0, float2[<= 4096] {exp xA, exp xB}     [64K, 96K)  buffer 1
//   [96K, 128K)   this frame's raw sums, float2      [128K, 144K)  leaky probs      [144K, ..)  partial sums
constexpr int kPrNW = kNW, kPrNT = kNT;
constexpr uint32_t kPrX0 = 32768, kPrX1 = 65536, kPrRaw = 98304, kPrLk = 131072, kPrRed = 147456;
constexpr uint32_t kPrBytes = kPrRed + 8 * 16 * 4;       // red: [A s0, B s0, A s1, B s1, final A, final B, flags, spare][16]
static_assert(kPrNW == 16, "partial sums: red[q][16], one entry per wave");

typedef float lz_v4f __attribute__((ext_vector_type(4)));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
__device__ __forceinline__ void lz_st2(uint32_t byte_addr, lz_v2f v) { *(__attribute__((address_space(3))) lz_v2f*)(byte_addr) = v; }
__device__ __forceinline__ lz_v4f lz_ld4(uint32_t byte_addr) { return *(const __attribute__((address_space(3))) lz_v4f*)(byte_addr); }
__device__ __forceinline__ void lz_st4(uint32_t byte_addr, lz_v4f v) { *(__attribute__((address_space(3))) lz_v4f*)(byte_addr) = v; }
#pragma clang diagnostic pop

// An arc is one aligned 64-bit register pair {p, packed addresses}: v_pk_mul_f32 broadcasts the low half of a source
// pair (op_sel_hi = 0), so the probability multiplies both sequences' operands without being copied into a pair of
// its own (the compiler otherwise keeps {p, p} for every arc: 40 more registers).
template <int R>
struct PairArcs {
  lz_v2f pa[R];                                          // .x = p, .y = 8*i0 | (kPrX0 + 8*i1) << 16 (as bits)
  __device__ __forceinline__ void load(int nslot_rows, const uint2* __restrict__ wave_slots) {
#pragma unroll
    for (int s = 0; s < R; s++) {
      uint2 a = make_uint2(0u, 0u);                    // rows past the plan: p = 0, harmless addresses
      if (s < nslot_rows) a = wave_slots[s * 64];
      pa[s] = lz_v2f{__uint_as_float(a.y), __uint_as_float(((a.x & 0xffffu) << 3) | ((kPrX0 + ((a.x >> 16) << 3)) << 16))};
    }
  }
  __device__ __forceinline__ void opaque2(int s) { asm volatile("" : "+v"(pa[s]), "+v"(pa[s + 1])); }
  __device__ __forceinline__ float p(int s) const { return pa[s].x; }
  template <uint32_t VOFF>
  __device__ __forceinline__ void gather(int s, lz_v2f& u, lz_v2f& v) {
    const uint32_t pk = __float_as_uint(pa[s].y);
    u = lz_ld2(pk & 0xffffu);
    v = lz_ld2((pk >> 16) + VOFF);
  }
};

// One frame of a wave's rows for both sequences: raw[row] = sum_k (p_k u[i0_k]) x[i1_k], lane sums s0 (and, beta, the
// leaky-weighted s1) as tile_rows<R, 0> forms them.  Pipelined in pairs of slot-rows; `pairmask` bit h = rows 2h, 2h+1
// hold a group end (from GroupRegs::endmask, once per workgroup).
template <int R, bool FWD, uint32_t VOFF>
__device__ __forceinline__ void pair_tile(PairArcs<R>& ar, const GroupRegs& gr, uint32_t pairmask, int lane, lz_v2f& s0, lz_v2f& s1) {
  constexpr int kStep = 2;
  static_assert(R % kStep == 0 && R <= 64, "whole pairs; one 32-bit mask");
  constexpr int NC = R / kStep;
  uint32_t m_lo = (uint32_t)gr.endmask, m_hi = (uint32_t)(gr.endmask >> 32), pm = pairmask;
  asm volatile("" : "+s"(m_lo), "+s"(m_hi), "+s"(pm));
  lz_v2f acc = {0.f, 0.f};
  lz_v2f ub[2][kStep], vb[2][kStep];
  ar.opaque2(0);
#pragma unroll
  for (int k = 0; k < kStep; k++) ar.template gather<VOFF>(k, ub[0][k], vb[0][k]);
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int cb = c & 1;
    if ((c & 1) == 0) wave_priority_by_progress<NC / 2>(c / 2);
    if (c + 1 < NC) {
      ar.opaque2((c + 1) * kStep);
#pragma unroll
      for (int k = 0; k < kStep; k++) ar.template gather<VOFF>((c + 1) * kStep + k, ub[cb ^ 1][k], vb[cb ^ 1][k]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < NC) PYCHAIN_WAIT_LGKM(2 * kStep); else PYCHAIN_WAIT_LGKM(0);
    __builtin_amdgcn_sched_barrier(0);
    lz_v2f nacc = acc;
#pragma unroll
    for (int k = 0; k < kStep; k++) {
      lz_v2f pu;
      {
#pragma clang fp contract(off)
        const float pp = ar.p(c * kStep + k);
        pu = lz_v2f{pp, pp} * ub[cb][k];                 // (p*u) rounded, then fused with x: as tile_rows
      }
      nacc = __builtin_elementwise_fma(pu, vb[cb][k], nacc);
    }
    if (__builtin_expect(((pm >> c) & 1u) != 0u, 0)) {         // a pair of rows with a group end (a few per frame) redoes it
      nacc = acc;
#pragma unroll
      for (int k = 0; k < kStep; k++) {
        const int sidx = c * kStep + k;
        lz_v2f pu;
        {
#pragma clang fp contract(off)
          const float pp = ar.p(sidx);
          pu = lz_v2f{pp, pp} * ub[cb][k];
        }
        nacc = __builtin_elementwise_fma(pu, vb[cb][k], nacc);
        if (((sidx < 32 ? m_lo : m_hi) >> (sidx & 31)) & 1u) {
          const uint32_t lo_before = sidx < 32 ? (m_lo & ((1u << (sidx & 31)) - 1u)) : m_lo;
          const uint32_t hi_before = sidx < 32 ? 0u : (m_hi & ((1u << (sidx & 31)) - 1u));
          const int g = __builtin_popcount(lo_before) + __builtin_popcount(hi_before);
          const int pos = __builtin_amdgcn_readlane(gr.base, g) + lane;
          lz_st2(kPrRaw + (uint32_t)pos * 8u, nacc);
          {
#pragma clang fp contract(off)
            s0 = s0 + nacc;
          }
          if constexpr (!FWD) { const float wv = lds_abs(kPrLk + (uint32_t)pos * 4u); s1 = __builtin_elementwise_fma(nacc, lz_v2f{wv, wv}, s1); }
          nacc = lz_v2f{0.f, 0.f};
        }
      }
    }
    acc = nacc;
  }
  for (int g = __builtin_popcount(m_lo) + __builtin_popcount(m_hi); g < gr.ngroups; g++)   // trailing groups whose rows have no arcs: zeros
    lz_st2(kPrRaw + (uint32_t)(__builtin_amdgcn_readlane(gr.base, g & 63) + lane) * 8u, lz_v2f{0.f, 0.f});
}

// XH: 2-byte nnet-output rows (DenArgs::x_half): four elements = 8 bytes per thread and row, converted as they arrive
template <int R, bool fwd, bool XH>
__device__ __forceinline__ void pair_recursion(const DenArgs& a, char* smem_raw, const int p) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the two sequences; an odd batch leaves the last workgroup with one (the other half runs masked on the same data)
  const int bA = 2 * p, bB = min(2 * p + 1, a.B - 1);
  const bool haveB = 2 * p + 1 < a.B;
  const int LA = __builtin_amdgcn_readfirstlane(seq_len(a.lengths, bA, a.T));
  const int LB = __builtin_amdgcn_readfirstlane(seq_len(a.lengths, bB, a.T));
  const int nA = fwd ? LA : LA - 1, nB = haveB ? (fwd ? LB : LB - 1) : 0;
  const int nmax = max(nA, nB);
  const int Hp = a.Hp, D = a.D;
  const char* plan = a.plans;                                      // shared by all sequences (checked by the host)
  const PlanHeader* hd = reinterpret_cast<const PlanHeader*>(plan);
  const TilePlan tp = fwd ? hd->alpha : hd->beta;
  const WaveEntry we = reinterpret_cast<const WaveEntry*>(plan + tp.off_wave_tab)[wave];
  const GroupEntry* gtab = reinterpret_cast<const GroupEntry*>(plan + tp.off_group_tab);
  const uint2* slots = reinterpret_cast<const uint2*>(plan + tp.off_slots);

  float* red = reinterpret_cast<float*>(smem_raw + kPrRed);        // [8][16]
  int bad = lds_addr(smem_raw) != 0u ? 1 : 0;                      // the packed arc addresses are absolute
  if (fwd && (seq_len_bad(a.lengths, bA, a.T) || (haveB && seq_len_bad(a.lengths, bB, a.T)))) bad = 1;

  GroupRegs groups;
  groups.load<R>(we, gtab, lane);
  PairArcs<R> arcs;
  arcs.load(groups.nslots, slots + (size_t)__builtin_amdgcn_readfirstlane(we.slot_row_begin) * 64 + lane);
  if (groups.nslots > R) bad = 1;                                  // the host checks the plan before choosing this kernel
  uint32_t pairmask = 0u;
  for (int h = 0; h < 32; h++) if ((groups.endmask >> (2 * h)) & 3ull) pairmask |= 1u << h;

  const float* leaky_g = reinterpret_cast<const float*>(plan + (fwd ? hd->off_leaky_a : hd->off_leaky_b));
  const float* start_g = reinterpret_cast<const float*>(plan + (fwd ? hd->off_init_a : hd->off_final_b));
  constexpr size_t kXe = XH ? 2 : 4;                                // bytes per nnet-output element
  const bool bf16 = a.x_half == kXBf16;
  const float* xA = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x) + (size_t)bA * a.T * D * kXe);
  const float* xB = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x) + (size_t)bB * a.T * D * kXe);
  const size_t rows_per_seq = fwd ? (size_t)a.T : (size_t)a.T + 1;
  float* storeA = (fwd ? a.alpha_store : a.beta_store) + (size_t)bA * rows_per_seq * Hp;
  float* storeB = (fwd ? a.alpha_store : a.beta_store) + (size_t)bB * rows_per_seq * Hp;
  float* totA = (fwd ? a.tot_a : a.tot_b) + (size_t)bA * (a.T + 2);
  float* totB = (fwd ? a.tot_a : a.tot_b) + (size_t)bB * (a.T + 2);
  const float coef = a.coef;
  const XBuf xbufA = make_xbuf(xA, (size_t)a.T * D * kXe), xbufB = make_xbuf(xB, (size_t)a.T * D * kXe);
  auto load_rows = [&](XRow<kPrNT, 4, 1>& qa, XRow<kPrNT, 4, 1>& qb, int ta, int tb, int t) {
    if constexpr (XH) { qa.load_row_h(xbufA, ta, D, t); qb.load_row_h(xbufB, tb, D, t); }   // (raw: converted where they are used)
    else { qa.load_row(xbufA, ta, D, t); qb.load_row(xbufB, tb, D, t); }
  };
  const XBuf sbufA = make_xbuf(storeA, (size_t)(a.T + 1) * Hp * sizeof(float));
  const XBuf sbufB = make_xbuf(storeB, (size_t)(a.T + 1) * Hp * sizeof(float));

  XRow<kPrNT, 4, 1> xqA, xqB;                                       // D <= 4096: one float4 per thread and row
  const bool x_identity = a.input_is_exp == kXIdentity;             // (kXClamp is the numerator's mode: not here)
  // a NaN network output belongs to ONE sequence: flags per sequence (the alpha workgroups watch, as in den_recursion_kernel)
  bool nanA = false, nanB = false;
  // a row of each sequence, clamped / exp'd, interleaved into the nnet-output buffer at `xoff`
  auto stage_rows = [&](uint32_t xoff, int tq) {
    const int e = tq * 4;
    if (e < D) {
      float va[4], vb[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        va[k] = x_identity ? xqA.v[k] : clamp_exp(xqA.v[k], kXExpClamp);
        vb[k] = x_identity ? xqB.v[k] : clamp_exp(xqB.v[k], kXExpClamp);
      }
      lz_st4(xoff + (uint32_t)e * 8u, lz_v4f{va[0], vb[0], va[1], vb[1]});
      lz_st4(xoff + (uint32_t)e * 8u + 16u, lz_v4f{va[2], vb[2], va[3], vb[3]});
    }
  };
  // v = raw normalised (normalise_row's formulas), to the gather operand and - where the sequence is live - to its row
  // of the trajectory store; four states per thread (Hp <= 4096 = one pass).  In two parts: the raw sums (and the
  // leaky probs) are requested from LDS BEFORE the totals are reduced - they do not depend on them, and the
  // reduction is a latency chain of its own.
  auto normalise_read = [&](int tq, lz_v4f& nr01, lz_v4f& nr23, lz_v4f& nlk) {
    const int i = tq * 4;
    nr01 = nr23 = nlk = lz_v4f{0.f, 0.f, 0.f, 0.f};
    if (i < Hp) {
      nr01 = lz_ld4(kPrRaw + (uint32_t)i * 8u);
      nr23 = lz_ld4(kPrRaw + (uint32_t)i * 8u + 16u);
      if (fwd) nlk = lz_ld4(kPrLk + (uint32_t)i * 4u);
    }
  };
  auto normalise = [&](lz_v2f inv, lz_v2f add, int rowA, int rowB, int tq, lz_v4f nr01, lz_v4f nr23, lz_v4f nlk) {
    const int i = tq * 4;
    if (i < Hp) {
      const lz_v4f r01 = nr01, r23 = nr23;
      float va[4], vb[4];
      const float ra[4] = {r01.x, r01.z, r23.x, r23.z}, rb[4] = {r01.y, r01.w, r23.y, r23.w};
      if (fwd) {
        const lz_v4f l = nlk;
        const float cl[4] = {coef * l.x, coef * l.y, coef * l.z, coef * l.w};
#pragma unroll
        for (int k = 0; k < 4; k++) { va[k] = ra[k] * inv.x + cl[k]; vb[k] = rb[k] * inv.y + cl[k]; }
      } else {
        const float aiA = add.x * inv.x, aiB = add.y * inv.y;
#pragma unroll
        for (int k = 0; k < 4; k++) { va[k] = __builtin_fmaf(ra[k], inv.x, aiA); vb[k] = __builtin_fmaf(rb[k], inv.y, aiB); }
      }
      lz_st4((uint32_t)i * 8u, lz_v4f{va[0], vb[0], va[1], vb[1]});
      lz_st4((uint32_t)i * 8u + 16u, lz_v4f{va[2], vb[2], va[3], vb[3]});
      if (rowA >= 0) {
        u32x4 q; q.x = __float_as_uint(va[0]); q.y = __float_as_uint(va[1]); q.z = __float_as_uint(va[2]); q.w = __float_as_uint(va[3]);
        __builtin_amdgcn_raw_buffer_store_b128(q, sbufA, i * 4, rowA * Hp * 4, kStoreDeviceScope);
      }
      if (rowB >= 0) {
        u32x4 q; q.x = __float_as_uint(vb[0]); q.y = __float_as_uint(vb[1]); q.z = __float_as_uint(vb[2]); q.w = __float_as_uint(vb[3]);
        __builtin_amdgcn_raw_buffer_store_b128(q, sbufB, i * 4, rowB * Hp * 4, kStoreDeviceScope);
      }
    }
  };
  // the 16 per-wave partial sums of quantity q (red[q][16]) -> block total, as den_recursion_kernel forms it
  auto total = [&](int q) { return block_total(red + q * 16, lane); };

  // ---- frame 0 (alpha: chain-computation.cc:92-95,97-110,178-194) / frame L (beta: :232-245,313-330): the same start
  //      vector for both sequences
  if (tid < 128) red[tid] = 0.f;
  {
    float p0 = 0.f, p1 = 0.f;
    for (int i = tid; i < Hp; i += kPrNT) {
      const float l = leaky_g[i], s = start_g[i];
      lz_st1(kPrLk + (uint32_t)i * 4u, l);
      lz_st2(kPrRaw + (uint32_t)i * 8u, lz_v2f{s, s});
      p0 += s; p1 += s * l;
    }
    p0 = wave_sum(p0); p1 = wave_sum(p1);
    load_rows(xqA, xqB, fwd ? 0 : LA - 1, fwd ? 0 : LB - 1, tid);
    if constexpr (XH) { xqA.convert_h(bf16); xqB.convert_h(bf16); }
    if (fwd) { nanA = xqA.has_nan(); nanB = haveB && xqB.has_nan(); }
    stage_rows(kPrX0, tid);
    __syncthreads();                                   // red zeroed
    if (lane == 0) { red[wave] = p0; red[2 * 16 + wave] = p1; }
    __syncthreads();
    lz_v4f nr01, nr23, nlk;
    normalise_read(tid, nr01, nr23, nlk);
    const float tot = total(0), wtot = total(2);
    const float inv = __builtin_amdgcn_rcpf(tot);
    if (!(tot > 0.f) || !(inv > 0.f)) bad = 1;
    if (tid == 0) { totA[fwd ? 0 : LA] = tot; if (haveB) totB[fwd ? 0 : LB] = tot; }
    normalise(lz_v2f{inv, inv}, lz_v2f{coef * wtot, coef * wtot}, fwd ? 0 : LA, haveB ? (fwd ? 0 : LB) : -1, tid, nr01, nr23, nlk);
  }
  __syncthreads();

  // ComputeTotLogLike's last factor, chain-computation.cc:209-230: sum_i alpha'(L,i) final(i) of the sequences that
  // have just taken their last step (the state vector holds alpha'(L,.) then).  Rare path: at most twice per workgroup.
  auto final_dot = [&](bool doA, bool doB) {
    const float* fin = reinterpret_cast<const float*>(plan + hd->off_final_a);
    float fa = 0.f, fb = 0.f;
    for (int i = tid; i < Hp; i += kPrNT) {
      const lz_v2f u = lz_ld2((uint32_t)i * 8u);
      fa += u.x * fin[i]; fb += u.y * fin[i];
    }
    fa = wave_sum(fa); fb = wave_sum(fb);
    if (lane == 0) { red[4 * 16 + wave] = fa; red[5 * 16 + wave] = fb; }
    if (tid == 0) { red[6 * 16] = 0.f; red[6 * 16 + 1] = 0.f; }
    __syncthreads();
    if (nanA) red[6 * 16] = 1.f;                       // somebody staged a NaN network output of sequence A / B
    if (nanB) red[6 * 16 + 1] = 1.f;
    __syncthreads();
    const float sa = total(4), sb = total(5);
    if (tid == 0) {
      if (doA) { a.fin_dot[bA] = red[6 * 16] != 0.f ? __builtin_nanf("") : sa; if (!(sa > 0.f)) bad = 1; }
      if (doB) { a.fin_dot[bB] = red[6 * 16 + 1] != 0.f ? __builtin_nanf("") : sb; if (!(sb > 0.f)) bad = 1; }
    }
    __syncthreads();
  };

  int next_sig = 0;
  int next_bound = a.sig_n > 0 ? a.seg_bound[0] : 0x7fffffff;
#define PYCHAIN_PR_SIGNAL(DONE)                                                                             \
  while ((DONE) >= next_bound) {                                                                            \
    __builtin_amdgcn_s_waitcnt(0);                     /* this wave's row stores are acknowledged */          \
    __syncthreads();                                                                                        \
    if (tid == 0) __hip_atomic_fetch_add(a.progress + next_sig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    next_sig++;                                                                                             \
    next_bound = next_sig < a.sig_n ? a.seg_bound[next_sig] : 0x7fffffff;                                   \
  }

  // -DPYCHAIN_PROFILE_PHASES: cycles per phase of a frame step, per wave (tools/phase_timers.sh pair)
#ifdef PYCHAIN_PROFILE_PHASES
  unsigned long long prph[6] = {0, 0, 0, 0, 0, 0}, prt = 0;
#define PR_PH0() prt = PH_T()
#define PR_PH(i) do { const unsigned long long t_ = PH_T(); prph[i] += t_ - prt; prt = t_; } while (0)
#else
#define PR_PH0() (void)0
#define PR_PH(i) (void)0
#endif
  // One frame step j of both sequences; VOFF = the nnet-output buffer it gathers from (even steps 0, odd steps 1).
#define PYCHAIN_PR_STEP(J, VOFF)                                                                            \
  do {                                                                                                      \
    const int j = (J);                                                                                      \
    /* thread index made opaque per frame: the LDS / buffer offsets derived from it are then recomputed here */ \
    /* with a few VALU instead of living in a dozen VGPRs across the arc loop */                            \
    int tq = tid;                                                                                           \
    asm volatile("" : "+v"(tq));                                                                            \
    const int lq = tq & 63;                                                                                 \
    const bool liveA = j < nA, liveB = j < nB;                                                              \
    PR_PH0();                                                                                               \
    /* nnet-output rows of the NEXT step (a sequence that has none re-reads a row that exists) */            \
    const int tnA = fwd ? j + 1 : LA - 2 - j, tnB = fwd ? j + 1 : LB - 2 - j;                               \
    const bool nextA = fwd ? tnA < LA : tnA >= 1, nextB = haveB && (fwd ? tnB < LB : tnB >= 1);             \
    load_rows(xqA, xqB, min(max(tnA, 0), a.T - 1), min(max(tnB, 0), a.T - 1), tq);                          \
    lz_v2f s0 = {0.f, 0.f}, s1 = {0.f, 0.f};                                                                \
    pair_tile<R, fwd, VOFF>(arcs, groups, pairmask, lq, s0, s1);                                            \
    PR_PH(0);                                          /* arc phase */                                       \
    /* the other buffer was last read in the previous step, which every wave has left */                    \
    if constexpr (XH) { xqA.convert_h(bf16); xqB.convert_h(bf16); }   /* (2-byte rows: raw until here) */      \
    if (fwd) { if (nextA && xqA.has_nan()) nanA = true; if (nextB && xqB.has_nan()) nanB = true; }          \
    stage_rows((VOFF) ? kPrX0 : kPrX1, tq);                                                                 \
    PR_PH(1);                                          /* nnet-output rows clamped / exp'd / stored */       \
    {                                                                                                       \
      const float a0 = wave_sum(s0.x), b0 = wave_sum(s0.y);                                                 \
      if (lane == 0) { red[wave] = a0; red[16 + wave] = b0; }                                               \
      if (!fwd) {                                                                                           \
        const float a1 = wave_sum(s1.x), b1 = wave_sum(s1.y);                                               \
        if (lane == 0) { red[32 + wave] = a1; red[48 + wave] = b1; }                                        \
      }                                                                                                     \
    }                                                                                                       \
    PR_PH(2);                                          /* wave sums */                                       \
    __syncthreads();                                   /* every gather of this frame is done */              \
    PR_PH(3);                                          /* wait at barrier 1 */                               \
    lz_v4f nr01, nr23, nlk;                                                                                 \
    normalise_read(tq, nr01, nr23, nlk);                                                                    \
    const lz_v2f tot = {total(0), total(1)};                                                                \
    lz_v2f wtot = {0.f, 0.f};                                                                               \
    if (!fwd) wtot = lz_v2f{total(2), total(3)};                                                            \
    const lz_v2f inv = {__builtin_amdgcn_rcpf(tot.x), __builtin_amdgcn_rcpf(tot.y)};                        \
    if (liveA && (!(tot.x > 0.f) || !(inv.x > 0.f))) bad = 1;                                               \
    if (liveB && (!(tot.y > 0.f) || !(inv.y > 0.f))) bad = 1;                                               \
    const int tsA = fwd ? j + 1 : LA - 1 - j, tsB = fwd ? j + 1 : LB - 1 - j;                               \
    if (tid == 0) { if (liveA) totA[tsA] = tot.x; if (liveB) totB[tsB] = tot.y; }                           \
    normalise(inv, lz_v2f{coef * wtot.x, coef * wtot.y}, (liveA && (!fwd || tsA < LA)) ? tsA : -1,          \
              (liveB && (!fwd || tsB < LB)) ? tsB : -1, tq, nr01, nr23, nlk);                               \
    PR_PH(4);                                          /* totals, normalise pass, row stores */              \
    __syncthreads();                                                                                        \
    PR_PH(5);                                          /* wait at barrier 2 */                               \
    if (fwd && (j + 1 == nA || j + 1 == nB)) final_dot(j + 1 == nA, haveB && j + 1 == nB);                  \
  } while (0)

  // progress reports of the streamed occupancy pass (DenArgs::seq_progress), one counter per sequence: step j stores row
  // j + 1 (alpha; row 0 went out before the loop) / row L - 1 - j (beta; row L before the loop): j + 2 rows after step j
  int32_t* progA = a.seq_progress + (fwd ? 0 : a.B) + bA;
  int32_t* progB = a.seq_progress + (fwd ? 0 : a.B) + bB;
#define PYCHAIN_PR_REPORT(DONE)                                                                             \
  do {                                                                                                      \
    __builtin_amdgcn_s_waitcnt(0);                                                                          \
    __syncthreads();                                                                                        \
    if (tid == 0) {                                                                                         \
      __hip_atomic_store(progA, min((DONE) + 1, LA), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           \
      if (haveB) __hip_atomic_store(progB, min((DONE) + 1, LB), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    }                                                                                                       \
  } while (0)
  for (int jj = 0; jj < nmax; jj += 2) {
    PYCHAIN_PR_STEP(jj, 0u);
    if (jj + 1 < nmax) PYCHAIN_PR_STEP(jj + 1, kPrX1 - kPrX0);
    PYCHAIN_PR_SIGNAL(jj + 2);                         // bounds are even
    if (a.stream && stream_report_due(a.T, jj + 2) && jj + 2 < nmax) PYCHAIN_PR_REPORT(jj + 2);
  }
  PYCHAIN_PR_SIGNAL(next_sig < a.sig_n ? 0x7ffffffe : 0);   // sequences shorter than a bound are done with it now
  if (a.stream) PYCHAIN_PR_REPORT(0x3fffffff);
#undef PYCHAIN_PR_REPORT
#undef PYCHAIN_PR_SIGNAL
#undef PYCHAIN_PR_STEP
#ifdef PYCHAIN_PROFILE_PHASES
  if (lane == 0 && p == 0) {
    const unsigned long long n = (unsigned long long)max(1, nmax);
    printf("pair dir %d wave %2d rows %2d steps %d cycles/step: arcs %llu x %llu wsums %llu bar1 %llu normalise %llu bar2 %llu\n",
           (int)fwd, wave, groups.nslots, nmax, prph[0] / n, prph[1] / n, prph[2] / n, prph[3] / n, prph[4] / n, prph[5] / n);
  }
#endif
#undef PR_PH
#undef PR_PH0
  if ((bad || nanA || nanB) && lane == 0) atomicAdd(a.bad, 1);
}

template <int R, bool XH = false>
__global__ __launch_bounds__(kPrNT) void den_recursion_pair_kernel(const DenArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int P = (a.B + 1) / 2;
  if (blockIdx.x < (unsigned)P) pair_recursion<R, true, XH>(a, smem_raw, blockIdx.x);
  else pair_recursion<R, false, XH>(a, smem_raw, blockIdx.x - P);
}
