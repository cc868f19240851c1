// den_lazy.inc.h - the alpha / beta recursions with LAZY normalisation (included by den_lazy.hip
// inside its anonymous namespace).  Replaces chain-computation.cc:97-110,150-194 (AlphaSum, AlphaDash,
// AlphaGeneralFrame) and :289-330 (BetaDashGeneralFrame, Beta) like den_recursion_kernel does, with a
// different frame structure: ONE barrier per frame and no normalise pass.
//
// den_recursion_kernel ends every frame with: wave sums -> barrier -> totals -> a pass over the state row
// (LDS -> fma -> LDS + HBM) -> barrier: ~1500 cycles of latency chain per frame after
// ~3400 cycles of arc work.  Here the state vector stays UN-normalised in LDS and the scalars of frame t
// are applied one frame later, at the group ends of frame t+1, by when they have long been reduced:
//
//   alpha (chain-computation.cc:150-194):  a(t+1,j) = sum_k w_k a'(t,src_k) / tot(t),  a'(t,i) = a(t,i) + tot(t) cl(i),
//          cl = coef * leaky, w_k = p_k x(t,pdf_k)
//       =>  a(t+1,j) = [sum_k w_k a(t,src_k)] / tot(t) + [sum_k w_k cl(src_k)]
//     LDS holds float2 {a(t,i), cl(i)}: ONE ds_read_b64 per arc gathers both, the two sums are the two
//     halves of one packed fma, and 1/tot(t) multiplies the first at the group end.
//   beta (:289-330):  b(t,i) = sum_k w_k B(t+1,dst_k),  B(t,i) = (b(t,i) + c(t)) / n(t),  c = coef sum_i leaky_i b(t,i),
//          n(t) = sum_i b(t,i)  (any per-frame scale gives the same posteriors, chain-computation.h:91-98)
//       =>  b(t,i) = ([sum_k w_k b(t+1,dst_k)] + c(t+1) [sum_k w_k]) / n(t+1)
//     LDS holds float2 {b(t+1,i), 1}: the same arc loop serves both directions.
//
// What is streamed to HBM is a'(t,.) = a(t,.) + tot(t) cl(.) and b(t,.) + c(t): the rows the occupancy pass
// gathers from, each in a per-frame scale of its own (any scale gives the same posteriors), so the occupancy
// kernels read them as they are.  tot(t) / c(t) only exist after the frame's barrier: a row goes out ONE FRAME
// LATE, at the end of the next frame, from the state buffer that frame gathered from (one ds_read_b64 per
// group gives {a, cl} / {b, 1}); the last beta row is flushed after the loop.  The per-frame totals are stored
// for den_finish_kernel (log-probability, invariant check).  The state vector and the nnet-output row are both
// double-buffered, so a frame writes only buffers nobody reads until the barrier.
//
// Workgroup shapes that share this code (template parameter MAP; all 128 VGPRs per wave, <= 40 slot-rows per wave):
//   LzNarrow / LzNarrowDma  16 waves, D <= 4096, Hp <= 4096: C3 (rows through registers / by LDS-direct loads);
//   LzDma                   16 waves, rows of up to 9216 pdfs by LDS-direct loads, Hp <= 3072: C4;
//   LzSmall                  4 waves, Hp <= 1024, D <= 4096, over the plan's four-wave dealing (alpha4 / beta4): small graphs
//                            (C1, C2) - a frame of theirs is a few hundred gathers, and sixteen waves walking their loops and
//                            meeting at a barrier for four groups of work cost what a C3-size frame costs.  Measured on C2
//                            and not kept (profiles/r04_c2_small_*): the rows requested two steps ahead into a ring of four
//                            buffers (0.209 against 0.199 ms - a frame is not waiting for its row), a wave leaving the arc loop
//                            after its last real chunk (0.266 ms: the row's clamp / exp then sits in the serial tail).
// (8 and 12 waves with 256 / 168 VGPRs were built and measured in round 3: +46 % / +15 %, profiles/r03_a_time_matrix.txt,
// r03_g_twelve_waves.txt; so was a map with two copies of the nnet-output row: +0.3 %, r03_i_two_copies.txt.)
// LDS maps (absolute byte addresses; the dynamic segment starts at 0, checked).  An arc is two VGPRs as in
// den_recursion_kernel: {kUField + 8*i0 | (kXField + 4*i1) << 16, p}; the buffer of a frame is selected by the ds_read
// OFFSET field (16 bits: buffer base - field base, <= 65535), which costs no instruction.
struct LzNarrow {
  //   [0, 32K)    state buffer 0: float2[<= 4096]       [32K, 64K)  state buffer 1
  //   [64K, 80K)  nnet-output buffer 0 (exp'd row)      [80K, 96K)  nnet-output buffer 1
  //   [96K, ...)  partial sums, beta's leaky probs
  static constexpr int kWaves = 16, kMaxGroups = 4, kXch = 1;
  static constexpr bool kDma = false, kQ = false;
  static constexpr uint32_t kU0 = 0, kU1 = 32768, kX0 = 65536, kX1 = 81920, kUField = 0, kXField = 49152;
  static constexpr uint32_t kRed = 98304, kLk = kRed + 2 * 2 * 64 * 4, kMaxStates = 4096, kMaxPdfs = 4096;
  static constexpr uint32_t kBytes = kLk + kMaxStates * 4;
};
// the same map with the nnet-output rows brought in by LDS-direct loads (lazy_recursion: kDma): the default of C1-C3
struct LzNarrowDma : LzNarrow { static constexpr bool kDma = true; static constexpr int kXch = 0; };
// ... and with ONE-WORD state vectors (template parameter MAP::kQ, "Q" below; plans whose leaky probabilities are all positive and
// whose states sit on one position each - launch hint bit 19).  The second word of the float2 {a, cl} / {b, 1} is a constant of the
// STATE, so it can leave the gathered vector:
//   alpha:  gather a^(i) = a(i) / cl(i) and keep q_k = p_k cl(src_k) in the arc's register instead of p_k:
//           sum_k w_k a(src_k) = sum_k q_k x_k a^(src_k),   sum_k w_k cl(src_k) = sum_k q_k x_k  - no operand at all;
//   beta:   gather b(i); the second sum is sum_k p_k x_k.
// Per arc: TWO ds_read_b32 (state, nnet output) instead of a ds_read_b64 and a ds_read_b32 - a third fewer LDS bytes per gather
// pair, which the counters show to buy NO LDS time (SQ_LDS_IDX_ACTIVE unchanged: a b64 gather occupies the pipe about as long as
// a b32 one; bank conflicts -11 %; VALU +12 % for the group ends: profiles/r06_one_word_states.txt) - and the same 2.5 VALU (per pair of rows: 2 unpacks, v_pk_mul q x, v_pk_fma into {sum1 of row 2i, sum1 of row 2i+1},
// v_pk_add into {sum2, sum2}); a group end writes a^(j) = a(j) / cl(j) (1 / cl from LDS, requested a group ahead like beta's
// leaky probability), the row that leaves for HBM is cl (a^ + tot) = a + tot cl.  Same LDS map; the state buffers use their first
// 16 KiB, alpha keeps cl(.) in the upper half of buffer 0 and 1 / cl(.) where beta keeps its leaky probabilities.
struct LzNarrowDmaQ : LzNarrowDma { static constexpr bool kQ = true; static constexpr uint32_t kCl = kU0 + 16384; };
// 16 waves x 128 VGPRs AND nnet-output rows of up to 9216 pdfs (C4): the rows never pass through registers.  Every wave
// requests its 1 KiB chunks of the NEXT step's raw row with `buffer_load_dwordx4 ... lds` (lane l's 16 bytes land at
// chunk base + 16 l: tools/ubench/ldsdma.hip) at the start of a frame, straight into the buffer the next frame gathers
// from, and clamps / exp's its own chunks IN PLACE at the end of the frame.
struct LzDma {
  static constexpr int kWaves = 16, kMaxGroups = 4, kXch = 0;
  static constexpr bool kDma = true, kQ = false;
  static constexpr uint32_t kX0 = 0, kX1 = 36864, kU0 = 73728, kU1 = 98304, kUField = 32776, kXField = 0;
  static constexpr uint32_t kMaxStates = 3072, kMaxPdfs = 9216;
  static constexpr uint32_t kLk = 122880, kRed = kLk + kMaxStates * 4, kBytes = kRed + 2 * 2 * 64 * 4;
};
template <typename MAP> constexpr bool lz_map_ok() {
  return MAP::kU1 - MAP::kUField <= 65535u && MAP::kU0 >= MAP::kUField && MAP::kX1 - MAP::kXField <= 65535u && MAP::kX0 >= MAP::kXField &&
         MAP::kUField + 8u * (MAP::kMaxStates - 1) <= 65535u && MAP::kXField + 4u * (MAP::kMaxPdfs - 1) <= 65535u &&
         MAP::kBytes <= 160u * 1024u;
}
// Four waves, small graphs: everything below 64 KiB, three workgroups per CU by LDS.
//   [0, 8K) state buffer 0: float2[<= 1024]   [8K, 16K) state buffer 1   [16K, 32K) nnet-output buffer 0   [32K, 48K) buffer 1
//   [48K, ...) partial sums, beta's leaky probs
struct LzSmall {
  static constexpr int kWaves = 4, kMaxGroups = 4, kXch = 0;
  static constexpr bool kDma = true, kQ = false;
  static constexpr uint32_t kU0 = 0, kU1 = 8192, kX0 = 16384, kX1 = 32768, kUField = 0, kXField = 16384;
  static constexpr uint32_t kMaxStates = 1024, kMaxPdfs = 4096;
  static constexpr uint32_t kRed = 49152, kLk = kRed + 2 * 2 * 64 * 4, kBytes = kLk + kMaxStates * 4;
};
// "Crossing" (template parameter XF; pdf-by-state plans only): besides the two state buffers and the two nnet-output rows, two
// accumulator rows of D words (the occupancies a direction emits itself, fixed point) and two landing rows of Hp floats (the
// OTHER direction's row of the frame being emitted).  Arcs are LazyArcsState (32-bit state addresses: no 16-bit limit on the
// state field); the 16-bit addresses are those of a pdf (nnet-output rows AND accumulator rows: one window of 64 KiB from kXField)
// and of a landing-row position.
//   landing 0 (12K + 64 B of zeros)   landing 1 (12K + 64)   x row 0 (16K)   x row 1   accumulator 0 (16K)   accumulator 1
//   state buffer 0: float2[<= 3072] (24K)   state buffer 1   beta's leaky probs (12K)   partial sums   the per-wave sums of a row's
//   products [2][16]   the other side's totals [2], this side's predicted totals [2], the gradient scale   the table of a state's
//   further alpha positions (<= PLAN_MAX_EXTRA_A x 8 bytes): 161 056 bytes in all
struct LzCross {
  static constexpr int kWaves = 16, kMaxGroups = 4, kXch = 0;
  static constexpr bool kDma = true, kQ = false;
  static constexpr uint32_t kMaxStates = 3072, kMaxPdfs = 4096;
  // (64 bytes of zeros behind each landing row: a lane whose position is no state - a group its wave does not own, padding - reads
  // its "other side's value" there, at offset kZero of either row, and needs no mask)
  static constexpr uint32_t kZero = kMaxStates * 4, kLand0 = 0, kLand1 = kZero + 64;
  static constexpr uint32_t kX0 = kLand1 + kZero + 64, kX1 = kX0 + 16384, kXField = kX0, kA0 = kX1 + 16384, kA1 = kA0 + 16384;
  static constexpr uint32_t kU0 = kA1 + 16384, kU1 = kU0 + 8 * kMaxStates, kUField = kU0;
  static constexpr uint32_t kLk = kU1 + 8 * kMaxStates, kRed = kLk + kMaxStates * 4, kXRed = kRed + 2 * 2 * 64 * 4;
  // [2] the other side's totals, [2] this side's predicted totals (by step parity), the gradient scale x 2^-30
  static constexpr uint32_t kOtot = kXRed + 2 * 16 * 4, kPred = kOtot + 8, kScale = kOtot + 16, kExtra = kOtot + 32, kBytes = kExtra + 8 * PLAN_MAX_EXTRA_A;
};
static_assert(LzCross::kBytes <= 160u * 1024u && LzCross::kA1 - LzCross::kXField <= 65535u && LzCross::kXField + 4u * (LzCross::kMaxPdfs - 1) <= 65535u &&
              LzCross::kLand1 + 4u * (LzCross::kMaxStates - 1) <= 65535u && LzCross::kU1 - LzCross::kUField <= 65535u, "ds offset fields are 16 bits");
constexpr int kCrossBand = 24;     // frames either side of a segment's middle that stay with the occupancy launch (DenArgs::xf)
static_assert(lz_map_ok<LzNarrow>() && lz_map_ok<LzDma>() && lz_map_ok<LzSmall>(), "ds_read offset fields are 16 bits");
constexpr uint32_t kLzBytes = LzNarrow::kBytes;
template <typename MAP, bool Q> constexpr uint32_t lz_cl_of() { if constexpr (Q) return MAP::kCl; else return 0u; }

typedef float lz_v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const lz_v2f lz_lds_cv2f;
typedef __attribute__((address_space(3))) float lz_lds_float;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
__device__ __forceinline__ lz_v2f lz_ld2(uint32_t byte_addr) { return *(lz_lds_cv2f*)(byte_addr); }
__device__ __forceinline__ void lz_st1(uint32_t byte_addr, float v) { *(lz_lds_float*)(byte_addr) = v; }
#pragma clang diagnostic pop

// ---- nnet-output rows by LDS-direct loads (MAP::kDma) -----------------------------------------------------------
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
typedef __attribute__((address_space(3))) void lz_lds_void;
#ifndef PYCHAIN_LATE_BACK
#define PYCHAIN_LATE_BACK 2                            /* the late hook of lazy_tile runs this many chunks before the end of the arc phase */
#endif
#ifndef PYCHAIN_Q_REDO
#define PYCHAIN_Q_REDO 1                                /* group ends of the one-word form: 1 row by row, 2 pair by pair (packed) */
#endif
#ifndef PYCHAIN_Q_ALPHA
#define PYCHAIN_Q_ALPHA 1                               /* 0: one-word state vectors in the beta recursions only */
#endif
#ifndef PYCHAIN_Q_ASMLK
#define PYCHAIN_Q_ASMLK 0                               /* the group ends' LDS operand requested with an instruction the compiler does not count */
#endif
#ifndef PYCHAIN_XF_EXP
#define PYCHAIN_XF_EXP 0                               /* timing experiments on the crossing (WRONG RESULTS): 1 no adds, 2 no flush, 4 no landing rows, 8 no emission, 16 no totals */
#endif
#ifndef PYCHAIN_EXP_NO_ROWSTORE
#define PYCHAIN_EXP_NO_ROWSTORE 0                      /* timing experiments (WRONG RESULTS): 1 = the rows do not leave for HBM, 2 = they do, but are not re-read from LDS first */
#endif
typedef float lz_v4 __attribute__((ext_vector_type(4)));
#define PYCHAIN_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)     /* vmcnt(0) only (gfx9 encoding) */
// row t of the sequence behind `buf` -> LDS at byte address xbase, 1 KiB chunks dealt to the NW waves; lanes past the
// row's end re-read its last 16 bytes (a chunk may run past the end of the whole slab otherwise)
template <int NW, int NCH, int AUX = 0>
__device__ __forceinline__ void lz_dma_row(XBuf buf, int t, int D, int wave, int lane, uint32_t xbase) {
  const int row_bytes = D * 4;
  const int soff = __builtin_amdgcn_readfirstlane(t * row_bytes);
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int ch = wave + c * NW;                       // (uniform)
    if (ch * 1024 < row_bytes) {
      // a lane that starts inside the row reads its 16 bytes as they are (D % 4 != 0: the last one runs a few floats
      // into the next row or, at the end of the slab, into the zeros of the buffer's range check); one that starts past
      // the row's end re-reads the row's last 16 bytes
      // (rows of whole 16-byte pieces: the last piece's offset inside the chunk is uniform, one v_min instead of compare / select)
      const int last = __builtin_amdgcn_readfirstlane(min(1008, max(0, row_bytes - 16 - ch * 1024)));
      int voff = min(lane * 16, last);
      if (__builtin_expect((row_bytes & 15) != 0, 0)) {      // (uniform, rare, and kept a BRANCH: the asm keeps it from becoming selects)
        asm volatile("" ::: "memory");
        voff = ch * 1024 + lane * 16 < row_bytes ? lane * 16 : max(0, row_bytes - 16 - ch * 1024);
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(buf, (lz_lds_void*)(xbase + (uint32_t)ch * 1024u), 16, voff,
                                               soff + ch * 1024, 0, AUX);
    }
  }
}
// this wave's chunks of the row at xbase: raw -> clamp / exp, in place; returns true if a NaN was seen
template <int NW, int NCH>
__device__ __forceinline__ bool lz_dma_finish(int D, int wave, int lane, uint32_t xbase, int is_exp) {
  bool nan = false;
  PYCHAIN_WAIT_VM0();                                   // this wave's loads have landed
#ifdef PYCHAIN_EXP_NO_FINISH                            /* timing experiment (WRONG RESULTS): the rows are gathered as they arrived */
  return false;
#endif
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int ch = wave + c * NW;
    if (ch * 1024 < D * 4) {
      const uint32_t addr = xbase + (uint32_t)ch * 1024u + (uint32_t)lane * 16u;
      lz_v4 q = *(const __attribute__((address_space(3))) lz_v4*)(addr);
      if ((D & 3) == 0) {
        nan = nan || __builtin_isunordered(q.x, q.y) || __builtin_isunordered(q.z, q.w);
      } else {                                          // (elements past the row's end belong to the next frame, or to nobody)
        const int e = ch * 256 + lane * 4;
        nan = nan || (e < D && q.x != q.x) || (e + 1 < D && q.y != q.y) || (e + 2 < D && q.z != q.z) || (e + 3 < D && q.w != q.w);
      }
#if PYCHAIN_EXP_OPS == 2
      if (is_exp == kXExpClamp) {
        // (clamp_exp element by element, the product with log2 e as two v_pk_mul_f32: the same bits, two VALU instructions less
        // per wave and frame - each one is ~0.5 % of a frame that sits on its VALU issue)
        const lz_v2f l2e = lz_v2f{1.44269502162933349609375f, 1.44269502162933349609375f};
        const lz_v2f t0 = lz_v2f{__builtin_amdgcn_fmed3f(q.x, -30.f, 30.f), __builtin_amdgcn_fmed3f(q.y, -30.f, 30.f)} * l2e;
        const lz_v2f t1 = lz_v2f{__builtin_amdgcn_fmed3f(q.z, -30.f, 30.f), __builtin_amdgcn_fmed3f(q.w, -30.f, 30.f)} * l2e;
        q = lz_v4{__builtin_amdgcn_exp2f(t0.x), __builtin_amdgcn_exp2f(t0.y), __builtin_amdgcn_exp2f(t1.x), __builtin_amdgcn_exp2f(t1.y)};
      }
#else
      if (is_exp == kXExpClamp) q = lz_v4{clamp_exp(q.x, kXExpClamp), clamp_exp(q.y, kXExpClamp), clamp_exp(q.z, kXExpClamp), clamp_exp(q.w, kXExpClamp)};
#endif
      *(__attribute__((address_space(3))) lz_v4*)(addr) = q;
    }
  }
  return nan;
}
// 2-byte rows (DenArgs::x_half): the chunk a wave owns is the same 1 KiB of the fp32 row buffer = 256 elements = 512 bytes of
// raw row, which lanes 0 .. 31 bring into the UPPER half of the chunk with one LDS-direct load (16 bytes each; the other lanes
// are masked off and write nothing); the finish then has every lane read ITS 8 raw bytes (4 elements) from there and write
// its 16 bytes of fp32 over the chunk from the bottom - the reads of a wave are all issued before its first write, and no
// other wave touches the chunk.  Same chunk-to-wave dealing and the same four elements per lane as the fp32 rows.
// D % 8 == 0 (a loading lane's 8 elements are all inside the row or all past it).
template <int NW, int NCH, int AUX = 0>
__device__ __forceinline__ void lz_dma_row_h(XBuf buf, int t, int D, int wave, int lane, uint32_t xbase) {
  const int row_bytes = D * 2;
  const int soff = __builtin_amdgcn_readfirstlane(t * row_bytes);
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int ch = wave + c * NW;                       // (uniform)
    if (ch * 512 < row_bytes && lane < 32) {
      const int voff = ch * 512 + lane * 16 < row_bytes ? lane * 16 : max(0, row_bytes - 16 - ch * 512);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(buf, (lz_lds_void*)(xbase + (uint32_t)ch * 1024u + 512u), 16, voff,
                                               soff + ch * 512, 0, AUX);
    }
  }
}
template <int NW, int NCH>
__device__ __forceinline__ bool lz_dma_finish_h(int D, int wave, int lane, uint32_t xbase, int is_exp, bool bf16) {
  bool nan = false;
  PYCHAIN_WAIT_VM0();                                   // this wave's loads have landed
  // (the wave's chunk address is formed HERE, frame after frame: formed once before the loop it is one register more than the
  // kernel has, and its reload from scratch sits in every frame - C3-bf16 ran 5 % behind the fp32 rows for it)
  int wv = wave;
  if constexpr (NCH == 1) asm volatile("" : "+v"(wv));   // (the maps of wider rows: measured the other way round)
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int ch = wv + c * NW;
    if (ch * 512 < D * 2) {
      const uint32_t cbase = xbase + (uint32_t)ch * 1024u;
      typedef unsigned int lz_u2 __attribute__((ext_vector_type(2)));
      const lz_u2 r = *(const __attribute__((address_space(3))) lz_u2*)(cbase + 512u + (uint32_t)lane * 8u);
      float f0, f1, f2, f3;
      half2_to_f32(r.x, bf16, f0, f1); half2_to_f32(r.y, bf16, f2, f3);
      nan = nan || __builtin_isunordered(f0, f1) || __builtin_isunordered(f2, f3);
      if (is_exp == kXExpClamp) { f0 = clamp_exp(f0, kXExpClamp); f1 = clamp_exp(f1, kXExpClamp); f2 = clamp_exp(f2, kXExpClamp); f3 = clamp_exp(f3, kXExpClamp); }
      *(__attribute__((address_space(3))) lz_v4*)(cbase + (uint32_t)lane * 16u) = lz_v4{f0, f1, f2, f3};
    }
  }
  return nan;
}
#pragma clang diagnostic pop

template <int R, typename MAP>
struct LazyArcs {
  uint32_t pk[R];
  float p[R];
  __device__ __forceinline__ void load(int nslot_rows, const uint2* __restrict__ wave_slots) {
#pragma unroll
    for (int s = 0; s < R; s++) {
      uint2 a = make_uint2(0u, 0u);                    // rows past the plan: p = 0, harmless addresses
      if (s < nslot_rows) a = wave_slots[s * 64];
      pk[s] = (MAP::kUField + ((a.x & 0xffffu) << 3)) | ((MAP::kXField + ((a.x >> 16) << 2)) << 16);
      p[s] = __uint_as_float(a.y);
    }
  }
  __device__ __forceinline__ void opaque4(int s) {
    asm volatile("" : "+v"(pk[s]), "+v"(pk[s + 1]), "+v"(pk[s + 2]), "+v"(pk[s + 3]));
  }
  template <uint32_t UOFF, uint32_t VOFF>
  __device__ __forceinline__ void gather(int s, lz_v2f& u, float& v) {
    u = lz_ld2((pk[s] & 0xffffu) + UOFF);
    v = lds_abs((pk[s] >> 16) + VOFF);
  }
};

// The same arcs in 2.5 registers each instead of 2 (loops of <= 32 slot-rows: 80 VGPRs, what 40 packed rows take): the
// state address has a register of its own, two rows share one for their nnet-output addresses and hold their
// probabilities as a pair.  Per pair of rows a frame then issues 2 unpacking instructions, ONE v_pk_mul_f32 and 2
// v_pk_fma_f32 - 2.5 VALU instructions per arc against 4 (v_and, v_lshrrev, v_mul, v_pk_fma) for the packed form.  The
// arithmetic and its order are the same: bit-identical results.
template <int R, typename MAP>
struct LazyArcsSplit {
  static_assert(R % 4 == 0 && R <= 32, "2.5 registers per arc fit for loops of up to 32 slot-rows");
  uint32_t ua[R];               // state (b64) address
  uint32_t xp[R / 2];           // nnet-output (b32) addresses of rows 2i | 2i + 1 << 16
  lz_v2f pp[R / 2];
  __device__ __forceinline__ void load(int nslot_rows, const uint2* __restrict__ wave_slots) {
#pragma unroll
    for (int s = 0; s < R; s += 2) {
      uint2 a = make_uint2(0u, 0u), b = make_uint2(0u, 0u);
      if (s < nslot_rows) a = wave_slots[s * 64];
      if (s + 1 < nslot_rows) b = wave_slots[(s + 1) * 64];
      ua[s] = MAP::kUField + ((a.x & 0xffffu) << 3);
      ua[s + 1] = MAP::kUField + ((b.x & 0xffffu) << 3);
      xp[s / 2] = (MAP::kXField + ((a.x >> 16) << 2)) | ((MAP::kXField + ((b.x >> 16) << 2)) << 16);
      pp[s / 2] = lz_v2f{__uint_as_float(a.y), __uint_as_float(b.y)};
      // (opaque: a field base beyond the 16-bit offset field of ds_read would otherwise be split off and re-added per gather)
      asm volatile("" : "+v"(ua[s]), "+v"(ua[s + 1]));
    }
  }
  __device__ __forceinline__ void opaque4(int s) { asm volatile("" : "+v"(xp[s / 2]), "+v"(xp[s / 2 + 1])); }
  // rows s, s + 1 (s even)
  template <uint32_t UOFF, uint32_t VOFF>
  __device__ __forceinline__ void gather2(int s, lz_v2f& u0, lz_v2f& u1, lz_v2f& v) {
    // (both states, then both rows: the C3 recursion is 0.7 % faster than with state, row, state, row)
    u0 = lz_ld2(ua[s] + UOFF);
    u1 = lz_ld2(ua[s + 1] + UOFF);
    v.x = lds_abs((xp[s / 2] & 0xffffu) + VOFF);
    v.y = lds_abs((xp[s / 2] >> 16) + VOFF);
  }
};
template <int R, typename MAP> struct LazyArcsOf { typedef LazyArcs<R, MAP> type; };
#ifndef PYCHAIN_NO_SPLIT_ARCS
template <typename MAP> struct LazyArcsOf<16, MAP> { typedef LazyArcsSplit<16, MAP> type; };
template <typename MAP> struct LazyArcsOf<32, MAP> { typedef LazyArcsSplit<32, MAP> type; };
template <typename MAP> struct LazyArcsOf<24, MAP> { typedef LazyArcsSplit<24, MAP> type; };     // (the four-wave shape: launch_small)
#endif

// One-word state vectors (MAP::kQ): the split form with a b32 state address; alpha's probabilities are q_k = p_k cl(src_k).
template <int R, typename MAP>
struct LazyArcsQ {
  static_assert(R % 4 == 0 && R <= 32, "2.5 registers per arc fit for loops of up to 32 slot-rows");
  uint32_t ua[R];               // state (b32) address
  uint32_t xp[R / 2];           // nnet-output (b32) addresses of rows 2i | 2i + 1 << 16
  lz_v2f pp[R / 2];
  // cl_g: alpha - the leaky probabilities in the numbering of the gathered vector (q = p coef leaky(src)); beta: nullptr
  __device__ __forceinline__ void load(int nslot_rows, const uint2* __restrict__ wave_slots, const float* __restrict__ cl_g, float coef) {
#pragma unroll
    for (int s = 0; s < R; s += 2) {
      uint2 a = make_uint2(0u, 0u), b = make_uint2(0u, 0u);
      if (s < nslot_rows) a = wave_slots[s * 64];
      if (s + 1 < nslot_rows) b = wave_slots[(s + 1) * 64];
      ua[s] = MAP::kUField + ((a.x & 0xffffu) << 2);
      ua[s + 1] = MAP::kUField + ((b.x & 0xffffu) << 2);
      xp[s / 2] = (MAP::kXField + ((a.x >> 16) << 2)) | ((MAP::kXField + ((b.x >> 16) << 2)) << 16);
      float qa = __uint_as_float(a.y), qb = __uint_as_float(b.y);
      if (cl_g) { qa *= coef * cl_g[a.x & 0xffffu]; qb *= coef * cl_g[b.x & 0xffffu]; }
      pp[s / 2] = lz_v2f{qa, qb};
      asm volatile("" : "+v"(ua[s]), "+v"(ua[s + 1]));
    }
  }
  __device__ __forceinline__ void opaque4(int s) { asm volatile("" : "+v"(xp[s / 2]), "+v"(xp[s / 2 + 1])); }
  // rows s, s + 1 (s even): u = {state word of row s, of row s + 1}
  template <uint32_t UOFF, uint32_t VOFF>
  __device__ __forceinline__ void gather2(int s, lz_v2f& u, lz_v2f& v) {
    u.x = lds_abs(ua[s] + UOFF);
    u.y = lds_abs(ua[s + 1] + UOFF);
    v.x = lds_abs((xp[s / 2] & 0xffffu) + VOFF);
    v.y = lds_abs((xp[s / 2] >> 16) + VOFF);
  }
};
template <int R, typename MAP, bool Q = MAP::kQ> struct LazyArcsFor { typedef typename LazyArcsOf<R, MAP>::type type; };
template <int R, typename MAP> struct LazyArcsFor<R, MAP, true> { typedef LazyArcsQ<R, MAP> type; };

// "pdf by state" plans (plan_format.h: PLAN_FLAG_PDF_BY_STATE; template parameter SG of lazy_recursion): every arc entering a state
// carries that state's pdf, so the nnet output leaves the arc loop -
//   alpha:  a(t+1,j) = x(t,pdf_j) * ( [sum_k p_k a(t,src_k)] / tot(t) + [sum_k p_k cl(src_k)] )     x multiplies the row's sum ONCE,
//           where the row's value is formed (one ds_read_b32 per lane and group end instead of one per arc);
//   beta:   b(t,i) = ( [sum_k p_k y(dst_k).x] + c(t+1) [sum_k p_k y(dst_k).y] ) / n(t+1),   y(t+1; j) = x(t,pdf_j) * {b(t+1,j), 1}:
//           the vector beta gathers from is PRE-MULTIPLIED where row j is written (a group end of the frame before), which needs
//           the nnet-output row one frame earlier than the arc loop used to: beta's row pipeline runs a frame ahead.
// Per arc: ONE ds_read_b64 and ONE v_pk_fma_f32 (the probability broadcast by op_sel) - against two gathers and 2.5 VALU; two
// registers per arc (state address, probabilities in pairs).
template <int R, typename MAP>
struct LazyArcsState {
  static_assert(R % 4 == 0 && R <= 40, "two registers per arc: loops of up to 40 slot-rows");
  uint32_t ua[R];               // state (b64) address
  lz_v2f pp[R / 2];
  __device__ __forceinline__ void load(int nslot_rows, const uint2* __restrict__ wave_slots) {
#pragma unroll
    for (int s = 0; s < R; s += 2) {
      uint2 a = make_uint2(0u, 0u), b = make_uint2(0u, 0u);
      if (s < nslot_rows) a = wave_slots[s * 64];
      if (s + 1 < nslot_rows) b = wave_slots[(s + 1) * 64];
      ua[s] = MAP::kUField + ((a.x & 0xffffu) << 3);
      ua[s + 1] = MAP::kUField + ((b.x & 0xffffu) << 3);
      pp[s / 2] = lz_v2f{__uint_as_float(a.y), __uint_as_float(b.y)};
      asm volatile("" : "+v"(ua[s]), "+v"(ua[s + 1]));
    }
  }
  // (opaque per frame and chunk: the splat {p, p} of a probability is loop-invariant, and hoisted out of the frame loop it costs
  // two registers per arc instead of one op_sel)
  __device__ __forceinline__ void opaque4(int s) { asm volatile("" : "+v"(pp[s / 2]), "+v"(pp[s / 2 + 1])); }
};

// what a wave carries from frame to frame besides its arcs
struct LazyWave {
  float inv, c;                 // 1 / total of the previous frame; beta: coef * leaky-weighted sum of the previous frame
  float sprev;                  // the scalar that completes the row in the gather buffer: alpha tot(t), beta c(t)
  float lk_next;                // beta: leaky probability of the lane's row in the group whose end comes next
  float x_next;                 // SG: nnet output of the lane's row in the group whose end comes next (alpha: x(t,pdf_j); beta: x(t-1,pdf_i))
  bool x_ok;                    // SG, beta: the row x(t-1,.) exists (a step follows this one); else the vector is written unmultiplied
};

// A group end inside the arc loop does the least it can: the row's new value into the state buffer the
// next frame gathers from.  Totals and the HBM row are built after the arc phase from those LDS words
// (lazy_frame_end): nothing of it is live in registers while the gather buffers are.
template <bool FWD>
__device__ __forceinline__ float lazy_group_end(const LazyWave& w, lz_v2f acc, uint32_t lds_dst) {
  float val;
  if constexpr (FWD) val = __builtin_fmaf(acc.x, w.inv, acc.y);
  else val = __builtin_fmaf(w.c, acc.y, acc.x) * w.inv;
  lz_st1(lds_dst, val);
  return val;
}

// One frame of a recursion tile, lazy form: gathers from the state buffer at ds_read offset UOFF and the nnet-output
// buffer at VOFF, writes the new values into the state buffer at absolute address UNEXT.  Same software pipeline as
// tile_rows; up to 96 slot-rows per wave (three words of group-end bits).
template <int R, typename MAP, bool FWD, uint32_t UOFF, uint32_t VOFF, uint32_t UNEXT, typename Hook, typename Late>
__device__ __forceinline__ void lazy_tile(LazyArcs<R, MAP>& ar, const GroupRegs& gr, LazyWave& w, int lane, float& s0, float& s1, Hook&& after_first_gathers, Late&& late) {
  constexpr int kChunk = 4;
  static_assert(R % kChunk == 0 && R <= 96 && PYCHAIN_CHUNK == 4, "chunk mask of GroupRegs is built for chunks of 4");
  constexpr int NC = R / kChunk;
  constexpr int kLateChunk = NC >= 4 ? NC - PYCHAIN_LATE_BACK : NC - 1;
  uint32_t m_lo = (uint32_t)gr.endmask, m_hi = (uint32_t)(gr.endmask >> 32), m_2 = gr.endmask2, cm = gr.chunkmask;
  if constexpr (R > 64) asm volatile("" : "+s"(m_lo), "+s"(m_hi), "+s"(m_2), "+s"(cm));
  else asm volatile("" : "+s"(m_lo), "+s"(m_hi), "+s"(cm));
  lz_v2f acc = {0.f, 0.f};
  lz_v2f ub[2][kChunk];
  float vb[2][kChunk];
  ar.opaque4(0);
#pragma unroll
  for (int k = 0; k < kChunk; k++) ar.template gather<UOFF, VOFF>(k, ub[0][k], vb[0][k]);
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int cb = c & 1;
    wave_priority_by_progress<NC>(c);
    if (c + 1 < NC) {
      ar.opaque4((c + 1) * kChunk);
#pragma unroll
      for (int k = 0; k < kChunk; k++) ar.template gather<UOFF, VOFF>((c + 1) * kChunk + k, ub[cb ^ 1][k], vb[cb ^ 1][k]);
    }
    // the previous frame's totals (w.inv, w.c: first needed at the first group end) are reduced HERE, behind the
    // gathers of the first two chunks, instead of between the barrier and the first gather
    if (c == 0) after_first_gathers();
    if (c == kLateChunk) late();                       // (the next frame's nnet-output row: lazy_recursion)
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < NC) PYCHAIN_WAIT_LGKM(2 * kChunk); else PYCHAIN_WAIT_LGKM(0);
    __builtin_amdgcn_sched_barrier(0);
    lz_v2f nacc = acc;
#pragma unroll
    for (int k = 0; k < kChunk; k++) {
      const float wk = ar.p[c * kChunk + k] * vb[cb][k];
      nacc = __builtin_elementwise_fma(lz_v2f{wk, wk}, ub[cb][k], nacc);
    }
    if (__builtin_expect(((cm >> c) & 1u) != 0u, 0)) {         // a chunk with a group end (a few per frame) redoes it
      nacc = acc;
#pragma unroll
      for (int k = 0; k < kChunk; k++) {
        const int sidx = c * kChunk + k;
        const float wk = ar.p[sidx] * vb[cb][k];
        nacc = __builtin_elementwise_fma(lz_v2f{wk, wk}, ub[cb][k], nacc);
        const uint32_t mword = sidx < 32 ? m_lo : (sidx < 64 ? m_hi : m_2);
        if ((mword >> (sidx & 31)) & 1u) {
          const uint32_t below = (1u << (sidx & 31)) - 1u;
          const uint32_t lo_before = sidx < 32 ? (m_lo & below) : m_lo;
          const uint32_t hi_before = sidx < 32 ? 0u : (sidx < 64 ? (m_hi & below) : m_hi);
          int g = __builtin_popcount(lo_before) + __builtin_popcount(hi_before);
          if constexpr (R > 64) if (sidx >= 64) g += __builtin_popcount(m_2 & below);
          const uint32_t pos = (uint32_t)(__builtin_amdgcn_readlane(gr.base, g) + lane);
          const float val = lazy_group_end<FWD>(w, nacc, UNEXT + pos * 8u);
          {                                              // the wave's share of the frame's totals, where the value is formed
            s0 += val;
            // (beta's leaky probabilities in four registers instead of this LDS read: two spills, recursion +6 %: r04_p_*)
            if constexpr (!FWD) {
              if constexpr (MAP::kMaxPdfs <= 4096) {                 // (requested one group end ahead: LazyWave::lk_next)
                s1 = __builtin_fmaf(val, w.lk_next, s1);
                w.lk_next = lds_abs(MAP::kLk + (uint32_t)(__builtin_amdgcn_readlane(gr.base, (g + 1) & 63) + lane) * 4u);
              } else {
                s1 = __builtin_fmaf(val, lds_abs(MAP::kLk + pos * 4u), s1);
              }
            }
          }
          nacc = lz_v2f{0.f, 0.f};
        }
      }
    }
    acc = nacc;
  }
}

// ... and over arcs in the split form (LazyArcsSplit): rows in pairs
template <int R, typename MAP, bool FWD, uint32_t UOFF, uint32_t VOFF, uint32_t UNEXT, typename Hook, typename Late>
__device__ __forceinline__ void lazy_tile(LazyArcsSplit<R, MAP>& ar, const GroupRegs& gr, LazyWave& w, int lane, float& s0, float& s1, Hook&& after_first_gathers, Late&& late) {
  constexpr int kChunk = 4;
  static_assert(PYCHAIN_CHUNK == 4, "chunk mask of GroupRegs is built for chunks of 4");
  constexpr int NC = R / kChunk;
  constexpr int kLateChunk = NC >= 4 ? NC - PYCHAIN_LATE_BACK : NC - 1;
  uint32_t m_lo = (uint32_t)gr.endmask, cm = gr.chunkmask;
  asm volatile("" : "+s"(m_lo), "+s"(cm));
  lz_v2f acc = {0.f, 0.f};
  lz_v2f ub[2][kChunk];
  lz_v2f vb[2][kChunk / 2];
  ar.opaque4(0);
#pragma unroll
  for (int k = 0; k < kChunk; k += 2) ar.template gather2<UOFF, VOFF>(k, ub[0][k], ub[0][k + 1], vb[0][k / 2]);
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int cb = c & 1;
    wave_priority_by_progress<NC>(c);
    if (c + 1 < NC) {
      ar.opaque4((c + 1) * kChunk);
#pragma unroll
      for (int k = 0; k < kChunk; k += 2)
        ar.template gather2<UOFF, VOFF>((c + 1) * kChunk + k, ub[cb ^ 1][k], ub[cb ^ 1][k + 1], vb[cb ^ 1][k / 2]);
    }
    if (c == 0) after_first_gathers();
    if (c == kLateChunk) late();                       // (the next frame's nnet-output row: lazy_recursion)
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < NC) PYCHAIN_WAIT_LGKM(2 * kChunk); else PYCHAIN_WAIT_LGKM(0);
    __builtin_amdgcn_sched_barrier(0);
    lz_v2f wk[kChunk / 2];
#pragma unroll
    for (int k = 0; k < kChunk / 2; k++) wk[k] = ar.pp[c * (kChunk / 2) + k] * vb[cb][k];
    lz_v2f nacc = acc;
#pragma unroll
    for (int k = 0; k < kChunk; k++) {
      const float x = (k & 1) ? wk[k / 2].y : wk[k / 2].x;
      nacc = __builtin_elementwise_fma(lz_v2f{x, x}, ub[cb][k], nacc);
    }
    if (__builtin_expect(((cm >> c) & 1u) != 0u, 0)) {         // a chunk with a group end (a few per frame) redoes it
      nacc = acc;
#pragma unroll
      for (int k = 0; k < kChunk; k++) {
        const int sidx = c * kChunk + k;
        const float x = (k & 1) ? wk[k / 2].y : wk[k / 2].x;
        nacc = __builtin_elementwise_fma(lz_v2f{x, x}, ub[cb][k], nacc);
        if ((m_lo >> sidx) & 1u) {
          const int g = __builtin_popcount(m_lo & ((1u << sidx) - 1u));
          const uint32_t pos = (uint32_t)(__builtin_amdgcn_readlane(gr.base, g) + lane);
          const float val = lazy_group_end<FWD>(w, nacc, UNEXT + pos * 8u);
          {                                              // the wave's share of the frame's totals, where the value is formed
            s0 += val;
            // (beta's leaky probabilities in four registers instead of this LDS read: two spills, recursion +6 %: r04_p_*)
            if constexpr (!FWD) {
              if constexpr (MAP::kMaxPdfs <= 4096) {                 // (requested one group end ahead: LazyWave::lk_next)
                s1 = __builtin_fmaf(val, w.lk_next, s1);
                w.lk_next = lds_abs(MAP::kLk + (uint32_t)(__builtin_amdgcn_readlane(gr.base, (g + 1) & 63) + lane) * 4u);
              } else {
                s1 = __builtin_fmaf(val, lds_abs(MAP::kLk + pos * 4u), s1);
              }
            }
          }
          nacc = lz_v2f{0.f, 0.f};
        }
      }
    }
    acc = nacc;
  }
}

// ... and over arcs with one-word states (LazyArcsQ): the sums of a pair of rows live in the two halves of {acc1.x, acc1.y} /
// {acc2.x, acc2.y} - a row's sums are the halves added up where its group ends
template <int R, typename MAP, bool FWD, uint32_t UOFF, uint32_t VOFF, uint32_t UNEXT, typename Hook, typename Late>
__device__ __forceinline__ void lazy_tile(LazyArcsQ<R, MAP>& ar, const GroupRegs& gr, LazyWave& w, int lane, float& s0, float& s1, Hook&& after_first_gathers, Late&& late) {
  constexpr int kChunk = 4;
  static_assert(PYCHAIN_CHUNK == 4, "chunk mask of GroupRegs is built for chunks of 4");
  constexpr int NC = R / kChunk;
  constexpr int kLateChunk = NC >= 4 ? NC - PYCHAIN_LATE_BACK : NC - 1;
  uint32_t m_lo = (uint32_t)gr.endmask, cm = gr.chunkmask;
  asm volatile("" : "+s"(m_lo), "+s"(cm));
  lz_v2f acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
  lz_v2f ub[2][kChunk / 2];
  lz_v2f vb[2][kChunk / 2];
  ar.opaque4(0);
#pragma unroll
  for (int k = 0; k < kChunk; k += 2) ar.template gather2<UOFF, VOFF>(k, ub[0][k / 2], vb[0][k / 2]);
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int cb = c & 1;
    wave_priority_by_progress<NC>(c);
    if (c + 1 < NC) {
      ar.opaque4((c + 1) * kChunk);
#pragma unroll
      for (int k = 0; k < kChunk; k += 2)
        ar.template gather2<UOFF, VOFF>((c + 1) * kChunk + k, ub[cb ^ 1][k / 2], vb[cb ^ 1][k / 2]);
    }
    if (c == 0) after_first_gathers();
    if (c == kLateChunk) late();                       // (the next frame's nnet-output row: lazy_recursion)
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < NC) PYCHAIN_WAIT_LGKM(2 * kChunk); else PYCHAIN_WAIT_LGKM(0);
    __builtin_amdgcn_sched_barrier(0);
    lz_v2f wk[kChunk / 2];
#pragma unroll
    for (int k = 0; k < kChunk / 2; k++) wk[k] = ar.pp[c * (kChunk / 2) + k] * vb[cb][k];
    lz_v2f n1 = acc1, n2 = acc2;
#pragma unroll
    for (int k = 0; k < kChunk / 2; k++) {
      n1 = __builtin_elementwise_fma(wk[k], ub[cb][k], n1);
      n2 = n2 + wk[k];
    }
    if (__builtin_expect(((cm >> c) & 1u) != 0u, 0)) {         // a chunk with a group end (about every other chunk of a C3 wave)
      bool lk_fresh = false;                                   // (a second group end in this chunk: its operand was requested in it)
      auto group_end = [&](int sidx, float r1, float r2) {
        const int g = __builtin_popcount(m_lo & ((1u << sidx) - 1u));
        const uint32_t pos = (uint32_t)(__builtin_amdgcn_readlane(gr.base, g) + lane);
#if PYCHAIN_Q_ASMLK
        if (lk_fresh) PYCHAIN_WAIT_LGKM(0);
#endif
        float val;
        if constexpr (FWD) {
          val = __builtin_fmaf(r1, w.inv, r2);
          lz_st1(UNEXT + pos * 4u, val * w.lk_next);           // a^(j) = a(j) / cl(j)
          s0 += val;
        } else {
          val = __builtin_fmaf(w.c, r2, r1) * w.inv;
          lz_st1(UNEXT + pos * 4u, val);
          s0 += val;
          s1 = __builtin_fmaf(val, w.lk_next, s1);
        }
        // (alpha: 1 / cl, beta: the leaky probability - of the lane's row in the group whose end comes next)
        const uint32_t lka = MAP::kLk + (uint32_t)(__builtin_amdgcn_readlane(gr.base, (g + 1) & 63) + lane) * 4u;
#if PYCHAIN_Q_ASMLK
        // Requested with an instruction the compiler does not count: it would wait for EVERY LDS operation in flight - the next
        // chunk's gathers - before the next group end reads it, although the wait at the top of that chunk's arithmetic (all but
        // the gathers issued after this) has long covered it.  (An uncounted operation in the queue only makes the compiler's
        // own waits wait for one more.)
        asm volatile("ds_read_b32 %0, %1" : "=v"(w.lk_next) : "v"(lka) : "memory");
        lk_fresh = true;
#else
        w.lk_next = lds_abs(lka);
        (void)lk_fresh;
#endif
      };
#if PYCHAIN_Q_REDO == 2
      n1 = acc1; n2 = acc2;
#pragma unroll
      for (int kp = 0; kp < kChunk / 2; kp++) {
        const int se = c * kChunk + 2 * kp;
        const bool e0 = ((m_lo >> se) & 1u) != 0u, e1 = ((m_lo >> (se + 1)) & 1u) != 0u;   // (uniform)
        if (!e0) {
          n1 = __builtin_elementwise_fma(wk[kp], ub[cb][kp], n1);
          n2 = n2 + wk[kp];
          if (e1) { group_end(se + 1, n1.x + n1.y, n2.x + n2.y); n1 = lz_v2f{0.f, 0.f}; n2 = lz_v2f{0.f, 0.f}; }
        } else {
          group_end(se, __builtin_fmaf(wk[kp].x, ub[cb][kp].x, n1.x + n1.y), n2.x + n2.y + wk[kp].x);
          const float o1 = wk[kp].y * ub[cb][kp].y, o2 = wk[kp].y;
          if (e1) { group_end(se + 1, o1, o2); n1 = lz_v2f{0.f, 0.f}; n2 = lz_v2f{0.f, 0.f}; }
          else { n1 = lz_v2f{0.f, o1}; n2 = lz_v2f{0.f, o2}; }
        }
      }
#else
      float r1 = acc1.x + acc1.y, r2 = acc2.x + acc2.y;
#pragma unroll
      for (int k = 0; k < kChunk; k++) {
        const int sidx = c * kChunk + k;
        const float x = (k & 1) ? wk[k / 2].y : wk[k / 2].x;
        const float u = (k & 1) ? ub[cb][k / 2].y : ub[cb][k / 2].x;
        r1 = __builtin_fmaf(x, u, r1);
        r2 += x;
        if ((m_lo >> sidx) & 1u) { group_end(sidx, r1, r2); r1 = 0.f; r2 = 0.f; }
      }
      n1 = lz_v2f{r1, 0.f}; n2 = lz_v2f{r2, 0.f};
#endif
    }
    acc1 = n1; acc2 = n2;
  }
}

#ifndef PYCHAIN_SG_AHEAD
#define PYCHAIN_SG_AHEAD 2                             /* chunks the gathers of the one-gather loop run ahead of its arithmetic (measured: 1, 2, 3 alike) */
#endif
// ... and over arcs of a "pdf by state" plan (LazyArcsState): ONE gather per arc.  xad01 / xad23 = absolute LDS addresses (nnet-output
// field, 16 bits each) of the pdf of this lane's row in the wave's groups 0 | 1 << 16, 2 | 3 << 16; XOFF = ds_read offset of the
// row buffer the group ends read.
template <int R, typename MAP, bool FWD, uint32_t UOFF, uint32_t XOFF, uint32_t UNEXT, int AHEAD, typename Hook, typename Late>
__device__ __forceinline__ void lazy_tile_sg(LazyArcsState<R, MAP>& ar, const GroupRegs& gr, LazyWave& w, const uint32_t xad01, const uint32_t xad23, int lane,
                                             float& s0, float& s1, Hook&& after_first_gathers, Late&& late) {
  constexpr int kChunk = 4;
  static_assert(PYCHAIN_CHUNK == 4, "chunk mask of GroupRegs is built for chunks of 4");
  constexpr int NC = R / kChunk;
  constexpr int kLateChunk = NC >= 4 ? NC - PYCHAIN_LATE_BACK : NC - 1;
  uint32_t m_lo = (uint32_t)gr.endmask, m_hi = (uint32_t)(gr.endmask >> 32), cm = gr.chunkmask;
  asm volatile("" : "+s"(m_lo), "+s"(m_hi), "+s"(cm));
  lz_v2f acc = {0.f, 0.f};
  // With one gather per arc the loop is no longer bound by the LDS pipe but by how many gathers a wave has in flight: the gathers
  // run kAhead chunks ahead of the arithmetic (the two-gather loops run one: their eight gathers per chunk fill the pipe)
  constexpr int kAhead = AHEAD < NC ? AHEAD : NC - 1, kBuf = kAhead + 1;
  lz_v2f ub[kBuf][kChunk];
#pragma unroll
  for (int c0 = 0; c0 < kAhead; c0++)
#pragma unroll
    for (int k = 0; k < kChunk; k++) ub[c0][k] = lz_ld2(ar.ua[c0 * kChunk + k] + UOFF);
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int cb = c % kBuf;
    wave_priority_by_progress<NC>(c);
    if (c + kAhead < NC) {
#pragma unroll
      for (int k = 0; k < kChunk; k++) ub[(c + kAhead) % kBuf][k] = lz_ld2(ar.ua[(c + kAhead) * kChunk + k] + UOFF);
    }
    if (c == 0) after_first_gathers();
    if (c == kLateChunk) late();
    __builtin_amdgcn_sched_barrier(0);
    {                                                        // all but the gathers of the chunks still ahead
      constexpr int kMaxAhead = kAhead * kChunk;
      const int ahead = (NC - 1 - c < kAhead ? NC - 1 - c : kAhead) * kChunk;
      if (ahead == kMaxAhead) PYCHAIN_WAIT_LGKM(kMaxAhead);
      else if (ahead == kChunk) PYCHAIN_WAIT_LGKM(kChunk);
      else if (ahead == 2 * kChunk) PYCHAIN_WAIT_LGKM(2 * kChunk);
      else PYCHAIN_WAIT_LGKM(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    ar.opaque4(c * kChunk);
    lz_v2f nacc = acc;
#pragma unroll
    for (int k = 0; k < kChunk; k++) {
      const lz_v2f pq = ar.pp[c * (kChunk / 2) + k / 2];
      const float p = (k & 1) ? pq.y : pq.x;
      nacc = __builtin_elementwise_fma(lz_v2f{p, p}, ub[cb][k], nacc);
    }
    if (__builtin_expect(((cm >> c) & 1u) != 0u, 0)) {         // a chunk with a group end (a few per frame) redoes it
      nacc = acc;
#pragma unroll
      for (int k = 0; k < kChunk; k++) {
        const int sidx = c * kChunk + k;
        const lz_v2f pq = ar.pp[sidx / 2];
        const float p = (k & 1) ? pq.y : pq.x;
        nacc = __builtin_elementwise_fma(lz_v2f{p, p}, ub[cb][k], nacc);
        const uint32_t mword = sidx < 32 ? m_lo : m_hi;
        if ((mword >> (sidx & 31)) & 1u) {
          const uint32_t below = (1u << (sidx & 31)) - 1u;
          const int g = __builtin_popcount(sidx < 32 ? (m_lo & below) : m_lo) + (sidx < 32 ? 0 : __builtin_popcount(m_hi & below));
          const uint32_t pos = (uint32_t)(__builtin_amdgcn_readlane(gr.base, g) + lane);
          const float xj = w.x_next;
          float val;
          if constexpr (FWD) {
            val = xj * __builtin_fmaf(nacc.x, w.inv, nacc.y);
            lz_st1(UNEXT + pos * 8u, val);
            s0 += val;
          } else {
            val = __builtin_fmaf(w.c, nacc.y, nacc.x) * w.inv;
            const float xm = w.x_ok ? xj : 1.f;              // (the last step of a direction: nobody gathers what it writes)
            *(__attribute__((address_space(3))) lz_v2f*)(UNEXT + pos * 8u) = lz_v2f{xm * val, xm};
            s0 += val;
            s1 = __builtin_fmaf(val, w.lk_next, s1);
            w.lk_next = lds_abs(MAP::kLk + (uint32_t)(__builtin_amdgcn_readlane(gr.base, (g + 1) & 63) + lane) * 4u);
          }
          {                                                  // the next group end's nnet output, requested a group ahead
            // (two 16-bit addresses per register, picked by the uniform group index: an array indexed by it would live in scratch)
            const int gn = g + 1;
            const uint32_t word = (gn & 2) ? xad23 : xad01;
            const uint32_t xa = (gn & 1) ? (word >> 16) : (word & 0xffffu);
            w.x_next = lds_abs(xa + XOFF);
          }
          nacc = lz_v2f{0.f, 0.f};
        }
      }
    }
    acc = nacc;
  }
}

// One (sequence, direction).  The direction is a template parameter and the kernel branches ONCE, at its
// top: with both directions in one body the register allocator keeps a second copy of every arc register
// across the (uniform) direction branches.
// XM: how the nnet-output rows arrive - kLzRowsF32: fp32, clamped / exp'd here; kLzRowsPre: exp'd ahead by den_exp_rows_kernel
// (DenArgs::ex); kLzRowsHalf: 2-byte rows (DenArgs::x_half), converted, clamped / exp'd here.  A template parameter: the
// kernel has no register to spare for more than one form.
enum { kLzRowsF32 = 0, kLzRowsPre = 1, kLzRowsHalf = 2 };
// TS: time segments (DenArgs::tseg) - a template parameter: the one-segment kernel keeps its registers; NC: see the start vector
// SG: a "pdf by state" plan - one gather per arc (LazyArcsState, lazy_tile_sg)
// XF: "crossing" - after the two directions of a (sequence[, segment]) have met in the middle, each emits the occupancies of its
// own second half itself (LzCross): gamma(t, pdf_j) += a(t+1,j) beta(t+1,j) is ONE product per row and lane for a pdf-by-state plan
template <int R, typename MAP, bool fwd, int XM, bool TS, bool NC, bool SG = false, bool XF = false>
__device__ __forceinline__ void lazy_recursion(const DenArgs& a, char* smem_raw, const int b, const int seg_in = 0) {
  static_assert(!XF || SG, "crossing: pdf-by-state plans");
  constexpr bool PRE = XM == kLzRowsPre, XH = XM == kLzRowsHalf;
  constexpr bool kQd = MAP::kQ && (PYCHAIN_Q_ALPHA || !fwd);   // one-word state vectors in this direction
  static_assert(!SG || (MAP::kDma && MAP::kMaxPdfs <= 4096 && !NC && !PRE), "the one-gather form: LDS-direct rows, <= 4096 pdfs, no split beta positions");
  constexpr int NW = MAP::kWaves, NT = NW * 64, MG = MAP::kMaxGroups;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- time segments (DenArgs::tseg).  The recursion below runs on a VIRTUAL sequence: alpha segment k on the frames [f0, e)
  // of the real one (f0 = s - burn: rows, totals and the nnet-output slab are addressed from f0, so that local row j is real
  // row f0 + j), beta segment k as if the sequence ended at f1 = e + burn and stopping at row s + 1.  Rows / totals outside
  // [s, e) (beta: rows s+1 .. e) are the burn-in: discarded, but for the row next to the segment (DenArgs::splice).
  const int Lb = __builtin_amdgcn_readfirstlane(seq_len(a.lengths, b, a.T));
  const int seg = TS ? seg_in : 0;
  const int tburn = TS ? __builtin_amdgcn_readfirstlane(den_tburn(a)) : 0;   // (the plan's controller may have lengthened it: DenArgs::tstate)
  const int nseg = (TS && a.tseg > 1 && Lb >= 2 * tburn) ? a.tseg : 1;
  if (TS && seg >= nseg) return;                         // (a sequence shorter than two burn-ins is not cut: its segment 0 does everything)
  const int seg_s = TS ? (int)(((long)seg * Lb) / nseg) : 0;
  const int seg_e = (!TS || seg + 1 == nseg) ? Lb : (int)(((long)(seg + 1) * Lb) / nseg);
  const bool seg_last = !TS || (fwd ? seg + 1 == nseg : seg == 0);   // the segment that ends where the sequence's recursion ends
  const int f0 = (TS && fwd && seg > 0) ? max(seg_s - tburn, 0) : 0;                 // alpha: first frame of the virtual sequence
  const int L = !TS ? Lb : (fwd ? seg_e - f0 : (seg + 1 == nseg ? Lb : min(Lb, seg_e + tburn)));   // its length (beta: its last frame + 1)
  const int nsteps = fwd ? L : L - 1 - seg_s;
  // real rows / totals (local numbering): alpha rows >= row_lo, beta rows <= row_hi; the speculated row next to them
  const int row_lo = fwd ? seg_s - f0 : 0, row_hi = fwd ? 0x7fffffff : seg_e;
  const int spec_row = fwd ? row_lo - 1 : seg_e + 1;     // (alpha segment 0 / the last beta segment have none: never met)
  auto row_real = [&](int r) { return !TS || (fwd ? r >= row_lo : r <= row_hi); };
  // alpha totals: index i = tot(f0 + i), real for row_lo <= i < L (the last segment: <= L); beta: n(i), real for i <= row_hi
  auto tot_real = [&](int i) { return !TS || (fwd ? (i >= row_lo && (i < L || seg + 1 == nseg)) : i <= row_hi); };
  const int Hp = a.Hp, D = a.D;
  const char* plan = a.plans + (size_t)b * a.plan_stride;
  const PlanHeader* hd = reinterpret_cast<const PlanHeader*>(plan);
  static_assert(NW == PLAN_REC_WAVES || NW == PLAN_REC4_WAVES, "the plan deals the recursion tiles to 16 and to 4 waves");
  const TilePlan tp = NW == PLAN_REC_WAVES ? (fwd ? hd->alpha : hd->beta) : (fwd ? hd->alpha4 : hd->beta4);
  const bool have_tile = tp.nwaves == NW;                         // (the four-wave dealing is in the plans of small graphs only)
  const WaveEntry we = have_tile ? reinterpret_cast<const WaveEntry*>(plan + tp.off_wave_tab)[wave] : WaveEntry{};
  const GroupEntry* gtab = reinterpret_cast<const GroupEntry*>(plan + tp.off_group_tab);
  const uint2* slots = reinterpret_cast<const uint2*>(plan + tp.off_slots);

  float* red = reinterpret_cast<float*>(smem_raw + MAP::kRed);     // [parity][which][64]
  // bit 0: not ok (a total or normaliser that is not finite-positive, a bad length); bit 1: a NaN network output was
  // staged (kept apart: it also turns the log-probability into NaN, and a later "not ok" must not hide it)
  int bad = lds_addr(smem_raw) != 0u ? 1 : 0;                      // the packed arc addresses are absolute
  // (the fallback behind a segmented launch watches every row itself: what an inner segment of the speculation reported is void)
  if (!TS && fwd && a.redo_if && tid == 0) __hip_atomic_store(a.xnan + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((fwd && seq_len_bad(a.lengths, b, a.T)) || !have_tile) bad |= 1;

  GroupRegs groups;
  groups.load<R>(we, gtab, lane);
  const uint2* wave_slots = slots + (size_t)__builtin_amdgcn_readfirstlane(we.slot_row_begin) * 64 + lane;
  typename std::conditional<SG, LazyArcsState<R, MAP>, typename LazyArcsFor<R, MAP, kQd>::type>::type arcs;
  if constexpr (kQd) arcs.load(groups.nslots, wave_slots, fwd ? reinterpret_cast<const float*>(plan + hd->off_leaky_a) : nullptr, a.coef);
  else arcs.load(groups.nslots, wave_slots);

  const float* leaky_g = reinterpret_cast<const float*>(plan + (fwd ? hd->off_leaky_a : hd->off_leaky_b));
  const float* start_g = reinterpret_cast<const float*>(plan + (fwd ? hd->off_init_a : hd->off_final_b));
  const float* xseq = a.x + ((size_t)b * a.T + f0) * D;
  float* store = fwd ? a.alpha_store + ((size_t)b * a.T + f0) * Hp : a.beta_store + (size_t)b * (a.T + 1) * Hp;
  float* totv = (fwd ? a.tot_a : a.tot_b) + (size_t)b * (a.T + 2) + f0;  // per-frame totals for den_finish_kernel (DenArgs::tot_a)
  const float coef = a.coef;
  constexpr int kDmaCh = ((int)MAP::kMaxPdfs / 256 + NW - 1) / NW;   // 1 KiB chunks of a row one wave may own
  // rows exp'd ahead of this kernel (DenArgs::ex): they arrive ready to gather; a row is requested once its end of the
  // sequence reports it (xprog), which after the first microseconds of a call it always has
  constexpr bool pre = PRE;                             // (a template parameter: the kernel has no register to spare for both forms)
  static_assert(!PRE || MAP::kDma, "rows exp'd ahead arrive by LDS-direct loads");
  static_assert(!XH || MAP::kDma, "2-byte rows arrive by LDS-direct loads");
  const bool bf16 = a.x_half == kXBf16;
  const XBuf xbuf = XH ? make_xbuf(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x) + ((size_t)b * a.T + f0) * D * 2), (size_t)(a.T - f0) * D * 2)
                       : make_xbuf(pre ? a.ex + (size_t)b * a.T * D : xseq, (size_t)(a.T - f0) * D * sizeof(float));
  int ready_lo = 0, ready_hi = 0;                       // ex rows [0, ready_lo) and [L - ready_hi, L) are complete
  // rows an end has complete: its workgroup q has done c_q of the rounds q, q + Q, ...: the first round missing is min_q (q + c_q Q)
  auto rows_of_end = [&](int end) {
    const int32_t* c = a.xprog + ((size_t)end * a.B + b) * kExMaxQ;
    int lead = 0x7fffffff;
    for (int q = 0; q < a.ex_q; q++)
      lead = min(lead, q + a.ex_q * __builtin_amdgcn_readfirstlane(__hip_atomic_load(c + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
    return min(lead, 0x00ffffff) * a.ex_nr;              // (clamped to the end's rows by the comparison below: L - Lh <= Lh <= L)
  };
  auto wait_row = [&](int t) {
    if (t < ready_lo || t >= L - ready_hi) return;
    const int Lh = (L + 1) / 2;                           // end 0 owns rows [0, Lh), end 1 rows [Lh, L)
    const unsigned long long t0 = wall_clock64();         // 100 MHz
    for (;;) {
      ready_lo = min(rows_of_end(0), Lh); ready_hi = min(rows_of_end(1), L - Lh);
      if (t < ready_lo || t >= L - ready_hi) break;
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > 2000000000ull) {          // 20 s: den_exp_rows_kernel died; no later row waits again
        bad |= 1; ready_lo = Lh; ready_hi = L - Lh; break;
      }
    }
  };
  // row t -> the buffer at xbase (LDS-direct); device-scope loads for rows another kernel is writing meanwhile
  auto dma_row = [&](int t, int ln, uint32_t xbase) {
    if constexpr (pre) { wait_row(t); lz_dma_row<NW, kDmaCh, kStoreDeviceScope>(xbuf, t, D, wave, ln, xbase); }
    else if constexpr (XH) lz_dma_row_h<NW, kDmaCh, 0>(xbuf, t, D, wave, ln, xbase);
    else lz_dma_row<NW, kDmaCh, 0>(xbuf, t, D, wave, ln, xbase);
  };
  // this wave's share of the row at xbase: raw -> fp32, clamped / exp'd; true if a NaN was seen
  auto dma_finish = [&](int ln, uint32_t xbase) {
    if constexpr (XH) return lz_dma_finish_h<NW, kDmaCh>(D, wave, ln, xbase, a.input_is_exp, bf16);
    else return lz_dma_finish<NW, kDmaCh>(D, wave, ln, xbase, a.input_is_exp);
  };
  const XBuf sbuf = make_xbuf(store, (size_t)(a.T + 1 - f0) * Hp * sizeof(float));
  // where a burn-in row goes: [0] the speculated row next to the segment, [1] a row nobody reads
  const XBuf spbuf = TS ? make_xbuf(a.splice + ((size_t)(b * 2 + (fwd ? 0 : 1)) * kMaxTimeSegs + seg) * 2 * Hp, 2 * (size_t)Hp * sizeof(float)) : sbuf;

  LazyWave w;
  // group g of this wave: rows base_g .. base_g + 63 (lane l owns row base_g + l)
  int gbase[MG];
#pragma unroll
  for (int g = 0; g < MG; g++) gbase[g] = g < groups.ngroups ? __builtin_amdgcn_readlane(groups.base, g) : 0;
  if (groups.ngroups > MG || groups.nslots > R) bad |= 1;   // the host checks the plan before choosing this kernel
  // SG: where the nnet output of this lane's row of group g sits in a row buffer (absolute address in the nnet-output field)
  uint32_t xad01 = 0u, xad23 = 0u;
  if constexpr (SG) {
    static_assert(!SG || MG <= 4, "four groups per wave");
    const int32_t* pdf_of = reinterpret_cast<const int32_t*>(plan + (fwd ? hd->off_pdf_a : hd->off_pdf_b));
    uint32_t xa[4];
#pragma unroll
    for (int g = 0; g < 4; g++) xa[g] = MAP::kXField + 4u * (uint32_t)((g < MG && g < groups.ngroups) ? pdf_of[gbase[g] + lane] : 0);
    xad01 = xa[0] | (xa[1] << 16); xad23 = xa[2] | (xa[3] << 16);
    asm volatile("" : "+v"(xad01), "+v"(xad23));
  }

  // ---- crossing (XF).  Real frames [xs, xe) of this workgroup (its segment, or the sequence); alpha emits the frames t >= xMa,
  // beta the frames t < xMb, the band [xMb, xMa) stays with the occupancy launch (neither direction ever waits for a LATE row of
  // the other: no deadlock).  The row a hook holds - alpha: a(t+1,.) = row t + 1 of its side, beta: beta(t+1,.) - belongs to frame
  // t; the other side's row t + 1 of the same states was landed (LDS-direct, device scope) during the step before.
  const int xs = TS ? seg_s : 0, xe = TS ? seg_e : Lb;
  const bool xon = XF && a.xf != 0 && (xe - xs >= 4 * kCrossBand + 8);
  const int xMa = ((xs + xe) >> 1) + kCrossBand, xMb = ((xs + xe) >> 1) - kCrossBand;
  uint32_t oad01 = 0u, oad23 = 0u;                       // landing-row addresses of this lane's rows (the state's position on the other side)
  XBuf xf_obuf = sbuf;
  float* xf_grad = nullptr;
  const float* xf_ototv = nullptr;
  int32_t* xf_my = nullptr; const int32_t* xf_peer = nullptr;
  int xf_nextra = 0;
  int xf_peer_need = 0;                                  // the peer's reported steps that make the FIRST row this side lands complete
  uint32_t* const xred = reinterpret_cast<uint32_t*>(smem_raw + (XF ? LzCross::kXRed : 0u));   // [2][16] per-wave sums of a row's products
  if constexpr (XF) {
    // (a position that is no state - a group this wave does not own, a beta position past the graph's states, whose "row" is the
    // leaky constant c(t) - reads the zeros behind the landing row: its product vanishes without a mask)
    const int32_t* perm = reinterpret_cast<const int32_t*>(plan + (fwd ? hd->off_a2b : hd->off_b2a));
    uint32_t oa[4];
#pragma unroll
    for (int g = 0; g < 4; g++)
      oa[g] = (g < MG && g < groups.ngroups && (fwd || gbase[g] + lane < hd->graph_states)) ? 4u * (uint32_t)perm[gbase[g] + lane] : LzCross::kZero;
    oad01 = oa[0] | (oa[1] << 16); oad23 = oa[2] | (oa[3] << 16);
    asm volatile("" : "+v"(oad01), "+v"(oad23));
    if (tid < 32) {
      *reinterpret_cast<uint32_t*>(smem_raw + LzCross::kLand0 + LzCross::kZero + 4 * (tid & 15) + (tid >> 4) * (LzCross::kLand1 - LzCross::kLand0)) = 0u;
      xred[tid] = 0u;
    }
    // (gradient scale x 2^-30, fixed point -> gradient: kept in LDS and read with the tail's other operands - a register held for
    // 1500 frames is one spilled)
    if (tid == 0) {
      *reinterpret_cast<float*>(smem_raw + LzCross::kScale) = (a.grad_scale_dev ? a.grad_scale * *a.grad_scale_dev : a.grad_scale) * (1.0f / 1073741824.0f);
      *reinterpret_cast<float*>(smem_raw + LzCross::kPred) = 1.f; *reinterpret_cast<float*>(smem_raw + LzCross::kPred + 4) = 1.f;
    }
    // alpha lands beta rows (row r of beta_store), beta lands alpha-store rows (index t holds a(t+1,.): DenArgs::sg)
    xf_obuf = fwd ? make_xbuf(a.beta_store + (size_t)b * (a.T + 1) * Hp, (size_t)(a.T + 1) * Hp * sizeof(float))
                  : make_xbuf(a.alpha_store + (size_t)b * a.T * Hp, (size_t)a.T * Hp * sizeof(float));
    xf_grad = a.grad + (size_t)b * a.T * D;
    xf_ototv = (fwd ? a.tot_b : a.tot_a) + (size_t)b * (a.T + 2);
    xf_my = a.xprog + ((size_t)(fwd ? 0 : 1) * a.B + b) * kExMaxQ + seg;
    xf_peer = a.xprog + ((size_t)(fwd ? 1 : 0) * a.B + b) * kExMaxQ + seg;
    // the first row this side lands: alpha: beta row xMa (frame xMa - 1, its dry run); beta: alpha-store index xMb (frame xMb)
    if (fwd) {                                           // beta's hook of step j stores row Lvb - j: P steps done => rows >= Lvb - P + 1
      const int Lvb = (!TS || seg + 1 == nseg) ? Lb : min(Lb, seg_e + tburn);
      xf_peer_need = Lvb - xMa + 1;
    } else {                                             // alpha's hook of local step j stores index f0a + j - 1: P steps => indices <= f0a + P - 2
      const int f0a = (TS && seg > 0) ? max(seg_s - tburn, 0) : 0;
      xf_peer_need = xMb - f0a + 2;
    }
    for (int i = tid; i < 2 * (int)LzCross::kMaxPdfs; i += NT) *reinterpret_cast<uint32_t*>(smem_raw + LzCross::kA0 + 4 * i) = 0u;
    // the table of a state's further ALPHA positions {8 x beta position | 4 x alpha position << 16, 4 x pdf}, one per thread (read
    // from the plan frame after frame, a load from memory would be waited for with every row in flight)
    if constexpr (!fwd) {
      const int32_t* ex = reinterpret_cast<const int32_t*>(plan + hd->off_extra_a);
      const int32_t* pdf_b = reinterpret_cast<const int32_t*>(plan + hd->off_pdf_b);
      xf_nextra = __builtin_amdgcn_readfirstlane(hd->n_extra_a);
      if (tid < xf_nextra) {
        const int pb = ex[2 * tid], pa = ex[2 * tid + 1];
        *reinterpret_cast<uint2*>(smem_raw + LzCross::kExtra + 8 * tid) = make_uint2(8u * (uint32_t)pb | (4u * (uint32_t)pa) << 16, 4u * (uint32_t)pdf_b[pb]);
      }
    }
  }
  // (the add itself: written out, because the compiler puts a wait for EVERY memory operation in flight - the row stores of this
  // step's hook among them - in front of an LDS atomic that follows an LDS-direct load it cannot tell apart from it)
  auto xf_add = [&](uint32_t addr, uint32_t val) { asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(val) : "memory"); };
  // which frames a direction emits; the frame before its first one is a dry run (its products are only summed: the first emitted
  // frame's scale needs the measured total of its neighbour)
  auto xf_emits = [&](int tfr) { return fwd ? (tfr >= xMa && tfr < xe) : (tfr < xMb && tfr >= xs); };
  // One row's share of the crossing, run in the TAIL of a step - behind the arc phase, where the registers of the gathers are free
  // and the LDS pipe is no longer queued up by sixteen waves' gathers (in the hook behind the first gathers every dependent LDS
  // round trip of this chain cost 300+ cycles: profiles/r06_crossing.txt).  `tfr`: the frame of the row the step's hook completed
  // (it sits in the buffer the step gathered from); everything is read first, in one round trip:
  //   * the row, the other side's row of the same states (landed a step ahead) and the other side's total of it;
  //   * the per-wave sums of the PREVIOUS row's products (its measured total G'), the total PREDICTED for it, and - if it was
  //     emitted - its accumulators, which leave for the gradient normalised by the measured total:
  //     G'(t) = G'(t -+ 1) x the other side's total of row t + 1 / this side's total of the neighbouring row - the invariant
  //     den_finish_kernel checks, solved for the next frame;
  // then this row's products go into the accumulator row of their pdfs in fixed point (ds_add_u32: 2^30 / G'_pred).  The store of
  // the flush is the tail's last instruction, and the step's barrier does not wait for it.
  float xf_totb = 1.f;                                   // (what the hook leaves for the tail: this side's total of the row before)
#ifdef PYCHAIN_PROFILE_PHASES
  unsigned long long xft[5] = {0, 0, 0, 0, 0};
#define XF_T(i) do { const unsigned long long t_ = PH_T(); xft[i] += t_ - xt0_; xt0_ = t_; } while (0)
#define XF_T0() unsigned long long xt0_ = PH_T()
#else
#define XF_T(i) (void)0
#define XF_T0() (void)0
#endif
  auto xf_row = [&](int par, int tfr, float tot_before, uint32_t ucur, int tq_, int lq_) {
    const bool emit = xf_emits(tfr), dry = fwd ? tfr == xMa - 1 : tfr == xMb;
    const int tprev = fwd ? tfr - 1 : tfr + 1;           // the row of the step before: emitted => its accumulators leave now
    const bool flush = xf_emits(tprev) && !(fwd && tprev < f0) ;
    if (!(emit || dry || flush) || (PYCHAIN_XF_EXP & 8)) return;
    XF_T0();
    const uint32_t land = par ? LzCross::kLand1 : LzCross::kLand0;
    const uint32_t acc = par ? LzCross::kA1 : LzCross::kA0, acc_prev = par ? LzCross::kA0 : LzCross::kA1;
    // (the packed addresses as the tail sees them: opaque, or every address derived from them is formed once before the frame loop
    // and kept - in registers this kernel does not have)
    uint32_t o01 = oad01, o23 = oad23, x01 = xad01, x23 = xad23;
    asm volatile("" : "+v"(o01), "+v"(o23), "+v"(x01), "+v"(x23));
    auto oaddr = [&](int g) { const uint32_t wd = (g & 2) ? o23 : o01; return (g & 1) ? (wd >> 16) : (wd & 0xffffu); };
    auto xaddr = [&](int g) { const uint32_t wd = (g & 2) ? x23 : x01; return (g & 1) ? (wd >> 16) : (wd & 0xffffu); };
    lz_v2f pr[MG]; float ld[MG];
#pragma unroll
    for (int g = 0; g < MG; g++) pr[g] = lz_ld2(ucur + gbase[g] * 8 + lq_ * 8);          // (gbase = 0 beyond ngroups: times zero below)
#pragma unroll
    for (int g = 0; g < MG; g++) ld[g] = lds_abs(oaddr(g) + land);
    const float xr = __uint_as_float(xred[(par ^ 1) * 16 + (lq_ & 15)]);
    const float ot = lds_abs(LzCross::kOtot + 4u * (uint32_t)par);
    const float pred_prev = lds_abs(LzCross::kPred + 4u * (uint32_t)(par ^ 1));
    const float gscale = lds_abs(LzCross::kScale);
    const bool fl = flush && tq_ * 4 < D && !(PYCHAIN_XF_EXP & 2);
    u32x4 fv = u32x4{0u, 0u, 0u, 0u};
    if (fl) fv = *(const __attribute__((address_space(3))) u32x4*)(acc_prev + 16u * (uint32_t)tq_);
    lz_v2f ex = lz_v2f{0.f, 0.f};
    const bool has_extra = !fwd && tq_ < xf_nextra;          // a state on several ALPHA positions: its beta lane took the first one
    if (has_extra) ex = lz_ld2(LzCross::kExtra + 8u * (uint32_t)tq_);
    const float gprev = dpp_row_sum(xr);                     // the previous row's MEASURED total
    const float rgprev = __builtin_amdgcn_rcpf(gprev);
    XF_T(0);                                                 /* (operands read, the previous row's total) */
    if (emit || dry) {
      const float gpred = gprev * ot * __builtin_amdgcn_rcpf(tot_before);
      const float sc = 1073741824.0f * __builtin_amdgcn_rcpf(gpred);
      if (emit && !(gpred > 0.f && sc - sc == 0.f)) bad |= 1;
      float gs = 0.f;
#pragma unroll
      for (int g = 0; g < MG; g++) {
        if (g < groups.ngroups) {                            // (uniform: sixty-four adds of zero to ONE address would cost 255 cycles)
          const float rowv = fwd ? pr[g].x : __builtin_fmaf(pr[g].x, __builtin_amdgcn_rcpf(pr[g].y), w.sprev);
          const float pq = rowv * ld[g];
          gs += pq;
          if (emit && !(PYCHAIN_XF_EXP & 1)) xf_add(xaddr(g) + (acc - LzCross::kXField), (uint32_t)__builtin_fmaf(pq, sc, 0.5f));
        }
      }
      if constexpr (!fwd) {
        if (xf_nextra > 0) {                                 // (uniform; a few per plan, one per thread)
          float pq = 0.f;
          if (has_extra) {
            const uint32_t e0 = __float_as_uint(ex.x), e1 = __float_as_uint(ex.y);
            const lz_v2f pe = lz_ld2(ucur + (e0 & 0xffffu));
            pq = __builtin_fmaf(pe.x, __builtin_amdgcn_rcpf(pe.y), w.sprev) * lds_abs(land + (e0 >> 16));
            if (emit && !(PYCHAIN_XF_EXP & 1)) xf_add(acc + e1, (uint32_t)__builtin_fmaf(pq, sc, 0.5f));
          }
          gs += pq;
        }
      }
      XF_T(1);                                               /* (products, adds) */
      gs = wave_sum(gs);
      if (lq_ == 0) xred[par * 16 + wave] = __float_as_uint(gs);
      if (tq_ == 0) *reinterpret_cast<float*>(smem_raw + LzCross::kPred + 4 * par) = gpred;
      XF_T(2);                                               /* (the row's sum) */
    }
    if (flush) {
      // (last: the previous row's accumulators leave for the gradient - nothing of this step touches LDS after this store)
      if (fl) {
        const float fs = gscale * pred_prev * rgprev;
        *(__attribute__((address_space(3))) u32x4*)(acc_prev + 16u * (uint32_t)tq_) = u32x4{0u, 0u, 0u, 0u};
        lz_v4 o = lz_v4{(float)fv.x * fs, (float)fv.y * fs, (float)fv.z * fs, (float)fv.w * fs};
        *reinterpret_cast<lz_v4*>(xf_grad + (size_t)tprev * D + 4 * tq_) = o;
      }
      if (a.check && (tprev == 0 || a.check_all) && tq_ == 0) den_record_frame_total(a, b, tprev, gprev);
    }
    XF_T(3);                                                 /* (flush) */
  };
  // what a step lands for the NEXT step's tail: the other side's row of the frame that tail handles (if it emits, or dry-runs)
  auto xf_land = [&](int j, int lq_) {
    const int tnx = fwd ? f0 + j : L - j - 2;              // the frame of the next hook's row
    const bool want = fwd ? (tnx >= xMa - 1 && tnx < xe) : (tnx <= xMb && tnx >= xs);
    if (!want) return;
    if (fwd ? tnx == xMa - 1 : tnx == xMb) {               // the first one: the peer must have stored it (later ones it stored EARLIER)
      const unsigned long long t0 = wall_clock64();        // 100 MHz
      while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(xf_peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < xf_peer_need) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 2000000000ull) { bad |= 1; break; }   // 20 s: the peer died
      }
    }
    const int row = fwd ? tnx + 1 : tnx;                   // beta row / alpha-store index
    if (!(PYCHAIN_XF_EXP & 4)) lz_dma_row<NW, ((int)LzCross::kMaxStates / 256 + NW - 1) / NW, kStoreDeviceScope>(xf_obuf, row, Hp, wave, lq_, ((j + 1) & 1) ? LzCross::kLand1 : LzCross::kLand0);
    // ... and its total, by the same route: a load into a REGISTER would be waited for where the compiler next touches the register -
    // at the top of the arc phase, with every row just requested in flight (measured: +1 ms per call)
    if (!(PYCHAIN_XF_EXP & 16) && wave == 0 && lq_ == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xf_ototv + tnx + 1),
                                       (lz_lds_void*)(LzCross::kOtot + (((j + 1) & 1) ? 4u : 0u)), 4, 0, kStoreDeviceScope);
  };

  // ---- frame 0 (alpha: chain-computation.cc:92-95) / frame L (beta: :232-245): un-normalised start vector
  XRow<NT, 4, MAP::kXch> xq;
  {
    float p0 = 0.f, p1 = 0.f;
    const int t0 = fwd ? 0 : L - 1;
    if constexpr (SG && !fwd) {
      // beta gathers y(j) = x(t,pdf_j) {b(t+1,j), 1}: the start vector needs row L-1 before it can be written (landed in buffer 1,
      // which the first step overwrites), and the first step's group ends need row L-2 in buffer 0
      dma_row(t0, lane, MAP::kX1);
      dma_finish(lane, MAP::kX1);
      __syncthreads();
    }
    for (int i = tid; i < (int)MAP::kMaxStates; i += NT) {
      float s = 0.f, second = fwd ? 0.f : 1.f, l = 0.f;
      if (i < Hp) { s = start_g[i]; l = leaky_g[i]; if (fwd) second = coef * l; }
      if constexpr (SG && !fwd) {
        const int32_t* pdf_b = reinterpret_cast<const int32_t*>(plan + hd->off_pdf_b);
        const float xi = i < Hp ? *reinterpret_cast<const float*>(smem_raw + MAP::kX1 + 4 * pdf_b[i]) : 0.f;
        *reinterpret_cast<lz_v2f*>(smem_raw + MAP::kU0 + 8 * i) = lz_v2f{xi * s, xi};
        *reinterpret_cast<lz_v2f*>(smem_raw + MAP::kU1 + 8 * i) = lz_v2f{0.f, 0.f};
      } else if constexpr (kQd) {
        // one-word states: alpha's vector is a / cl (a position without a state - padding - has cl = 0 and stays 0)
        const float rcl = (fwd && second > 0.f) ? 1.0f / second : 0.f;
        *reinterpret_cast<float*>(smem_raw + MAP::kU0 + 4 * i) = fwd ? s * rcl : s;
        *reinterpret_cast<float*>(smem_raw + MAP::kU1 + 4 * i) = 0.f;
        if (fwd) { *reinterpret_cast<float*>(smem_raw + MAP::kCl + 4 * i) = second; *reinterpret_cast<float*>(smem_raw + MAP::kLk + 4 * i) = rcl; }
      } else {
        *reinterpret_cast<lz_v2f*>(smem_raw + MAP::kU0 + 8 * i) = lz_v2f{s, second};
        *reinterpret_cast<lz_v2f*>(smem_raw + MAP::kU1 + 8 * i) = lz_v2f{0.f, second};
      }
      if (!fwd) *reinterpret_cast<float*>(smem_raw + MAP::kLk + 4 * i) = l;
      p0 += s; p1 += s * l;
    }
    if constexpr (SG && !fwd) {
      if (L - 2 >= seg_s + 1) { dma_row(L - 2, lane, MAP::kX0); dma_finish(lane, MAP::kX0); }
    } else if constexpr (MAP::kDma) {
      dma_row(t0, lane, MAP::kX0);
      if constexpr (pre) PYCHAIN_WAIT_VM0();
      else if (dma_finish(lane, MAP::kX0) && fwd) bad |= 2;
    } else {
      xq.load(xseq + (size_t)t0 * D, D, tid);
      if (fwd && xq.has_nan()) bad |= 2;
      xq.store(reinterpret_cast<float*>(smem_raw + MAP::kX0), xseq + (size_t)t0 * D, D, tid, a.input_is_exp);
    }
    p0 = wave_sum(p0); p1 = wave_sum(p1);
    if (NW < 16) {                                     // partial sums of waves that do not exist: zero, once
      if (tid < 256) red[tid] = 0.f;
      __syncthreads();
    }
    if (lane == 0) { red[wave] = p0; red[64 + wave] = p1; }
    if (tid >= NW && tid < 64) { red[tid] = 0.f; red[64 + tid] = 0.f; }
    __syncthreads();
    // NC: beta positions that take no constant c(t) - a state's lanes after its first (plan.cpp, "states on several lanes"): the
    // second word of their {b, 1} pairs is 0, in both buffers, from before the first frame on.  A template parameter (launch
    // hint bit 28: few plans have such positions): even this loop, outside the frames, moves the register allocation of the
    // frame loop - the 2-byte-row kernel lost 5 % to it (profiles/r05_ab_no_const.txt)
    if constexpr (NC && !fwd) {
      const int32_t* nc = reinterpret_cast<const int32_t*>(plan + hd->off_no_const);
      for (int j = tid; j < hd->n_no_const; j += NT) {
        const uint32_t at = 8u * (uint32_t)nc[j] + 4u;
        lz_st1(MAP::kU0 + at, 0.f); lz_st1(MAP::kU1 + at, 0.f);
      }
    }
    const float tot = wave_sum(red[lane]), wtot = wave_sum(red[64 + lane]);
    w.inv = __builtin_amdgcn_rcpf(tot);
    w.c = coef * wtot;
    if (!(tot > 0.f) || !(w.inv > 0.f)) bad |= 1;
    w.sprev = fwd ? tot : w.c;                         // the start row (alpha row 0 / beta row L) goes out at the end of frame 0
    if (tid == 0 && tot_real(fwd ? 0 : L)) {
      if constexpr (XF) __hip_atomic_store(totv + (fwd ? 0 : L), tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else totv[fwd ? 0 : L] = tot;
    }
    __syncthreads();                                                 // red is rewritten by the first frame (its first 4 NW entries per sum)
  }

  float last_tot = 1.f;
  int next_sig = 0;
  int next_bound = a.sig_n > 0 ? a.seg_bound[0] : 0x7fffffff;
#define PYCHAIN_LZ_SIGNAL(DONE)                                                                             \
  while ((DONE) >= next_bound) {                                                                            \
    __builtin_amdgcn_s_waitcnt(0);                     /* this wave's row stores are acknowledged */          \
    __syncthreads();                                                                                        \
    if (tid == 0) __hip_atomic_fetch_add(a.progress + next_sig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    next_sig++;                                                                                             \
    next_bound = next_sig < a.sig_n ? a.seg_bound[next_sig] : 0x7fffffff;                                   \
  }

  // -DPYCHAIN_PROFILE_PHASES: cycles per phase of a frame step, per wave (s_memtime: the sums live in SGPRs), printed
  // for sequence 0 when the kernel ends (tools/phase_timers.sh; the table of DESIGN.md §4)
#ifdef PYCHAIN_PROFILE_PHASES
  unsigned long long lzph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, lzt = 0;
#define LZ_PH0() lzt = PH_T()
#define LZ_PH(i) do { const unsigned long long t_ = PH_T(); lzph[i] += t_ - lzt; lzt = t_; } while (0)
#define LZ_VMWAIT() do { const unsigned long long t_ = PH_T(); PYCHAIN_WAIT_VM0(); lzph[5] += PH_T() - t_; } while (0)   /* (inside the arc phase) */
#define LZ_HK0() const unsigned long long hk0_ = PH_T()
#define LZ_HK(i) do { lzph[i] += PH_T() - hk0_; } while (0)     /* (inside the hook: cumulative from its start) */
#else
#define LZ_HK0() (void)0
#define LZ_HK(i) (void)0
#define LZ_VMWAIT() (void)0
#define LZ_PH0() (void)0
#define LZ_PH(i) (void)0
#endif
  // Totals of frame step JP (partial sums in red[PARP]): the normaliser of the next frame, the scalar that completes the
  // row JP produced, the total den_finish_kernel reads.  tstore = the row step JP produced (alpha: L is written, never read).
#define PYCHAIN_LZ_TOTALS(R0, R1, JP, FWDC, TQ)                                                              \
  do {                                                                                                      \
    const float tot = wave_sum(R0);                                                                         \
    w.inv = __builtin_amdgcn_rcpf(tot);                                                                     \
    if (!(tot > 0.f) || !(w.inv > 0.f)) bad |= 1;                                                           \
    if (FWDC) w.sprev = tot;                                                                                \
    else { w.c = coef * wave_sum(R1); w.sprev = w.c; }                                                      \
    if ((TQ) == 0 && tot_real((FWDC) ? (JP) + 1 : L - 1 - (JP))) {                                           \
      /* (crossing: the other direction reads this total while the kernel runs - device scope) */           \
      if constexpr (XF) __hip_atomic_store(totv + ((FWDC) ? (JP) + 1 : L - 1 - (JP)), tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
      else totv[(FWDC) ? (JP) + 1 : L - 1 - (JP)] = tot;                                                    \
    }                                                                                                       \
    last_tot = tot;                                                                                         \
  } while (0)
  // One frame step j: alpha produces a(j+1,.) from a(j,.) and x(j); beta produces b(t,.), t = L-1-j, from b(t+1,.) and x(t).
#define PYCHAIN_LZ_STEP(J, PAR, FWDC)                                                                       \
  do {                                                                                                      \
    const int j = (J);                                                                                      \
    /* thread and lane index made opaque per frame: everything derived from them (LDS and buffer offsets) is */ \
    /* then recomputed here with a few VALU instead of living in ~8 VGPRs across the arc loop */            \
    int tq = tid;                                                                                           \
    asm volatile("" : "+v"(tq));                                                                            \
    const int lq = tq & 63;                                                                                 \
    constexpr uint32_t UCUR = (PAR) ? MAP::kU1 : MAP::kU0, UNEXT = (PAR) ? MAP::kU0 : MAP::kU1;             \
    constexpr uint32_t UOFF = UCUR - MAP::kUField;                                                          \
    constexpr uint32_t VOFF = ((PAR) ? MAP::kX1 : MAP::kX0) - MAP::kXField;   /* nnet-output buffer PAR */   \
    /* nnet-output row that lands during this step: the NEXT step's - SG beta: the one after (its rows run a frame ahead) */ \
    const int tn = (FWDC) ? j + 1 : (SG ? L - 3 - j : L - 2 - j);                                           \
    const bool have_next = (FWDC) ? (tn < L) : (tn >= seg_s + 1);   /* beta never consumes row 0 (its segment: row s) */ \
    if constexpr (SG) {                                                                                     \
      w.x_ok = (FWDC) || (L - 2 - j >= seg_s + 1);           /* (beta: a step follows this one) */           \
      w.x_next = lds_abs((xad01 & 0xffffu) + VOFF);          /* the first group end's nnet output */         \
    }                                                                                                       \
    LZ_PH0();                                                                                               \
    if constexpr (MAP::kDma) {                               /* straight into the other buffer, in flight during the arc work */ \
      if (have_next) dma_row(tn, lq, (PAR) ? MAP::kX0 : MAP::kX1);                                          \
    } else if (have_next) xq.load_row(xbuf, tn, D, tq);      /* in flight during the arc work */             \
    if constexpr (XF) { if (xon) xf_land(j, lq); }           /* crossing: the other side's row for the NEXT step's tail */ \
    /* What the hook behind the first gathers needs from LDS - the previous step's partial sums and the row it produced (it */ \
    /* sits in the buffer this step gathers from) - is requested HERE, ahead of the gathers: behind them the reads queue up */ \
    /* after sixteen waves' first two chunks, and a wave that waits for them issues nothing meanwhile (C3 recursion -4 %: */ \
    /* profiles/r04_n_*; registers: live exactly as long as in the hook before) */                          \
    float pre0 = 0.f, pre1 = 0.f;                                                                           \
    if (j > 0) { pre0 = red[((PAR) ^ 1) * 128 + lq]; if (!(FWDC)) pre1 = red[((PAR) ^ 1) * 128 + 64 + lq]; } \
    if constexpr (kQd && PYCHAIN_Q_ASMLK) {                /* (uncounted, like the group ends' own requests: lazy_tile) */ \
      asm volatile("ds_read_b32 %0, %1" : "=v"(w.lk_next) : "v"((uint32_t)(MAP::kLk + gbase[0] * 4 + lq * 4)) : "memory"); \
    } else                                                                                                  \
    if constexpr (MAP::kMaxPdfs <= 4096 && (!(FWDC) || kQd)) w.lk_next = lds_abs(MAP::kLk + gbase[0] * 4 + lq * 4);   \
    constexpr bool kPreRows = MAP::kMaxPdfs <= 4096 && !(XF);       /* (the map of C4 has no registers to spare: the rows are read in the hook) */ \
    lz_v2f prow[MG];                                                                                        \
    if constexpr (kPreRows) {                                                                               \
      _Pragma("unroll") for (int g = 0; g < MG; g++) {       /* (gbase = 0 beyond ngroups) */                \
        if constexpr (kQd) prow[g] = lz_v2f{lds_abs(UCUR + gbase[g] * 4 + lq * 4), (FWDC) ? lds_abs(lz_cl_of<MAP, kQd>() + gbase[g] * 4 + lq * 4) : 1.f}; \
        else prow[g] = lz_ld2(UCUR + gbase[g] * 8 + lq * 8);                                                 \
      }                                                                                                     \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    float s0 = 0.f, s1 = 0.f;                                                                               \
    auto hook_first = [&]() {                                                                                \
      LZ_HK0();                                                                                             \
      float tot_before = 0.f;                                /* (crossing: this side's total of the row before the one in this hook) */ \
      if constexpr (XF) tot_before = last_tot;                                                              \
      if (j > 0) PYCHAIN_LZ_TOTALS(pre0, pre1, j - 1, (FWDC), tq);   /* (step 0: the start vector's, above) */ \
      /* ... and with them the row of the PREVIOUS frame (alpha row j, beta row L - j) is completed and leaves for HBM */ \
      /* (SG alpha - DenArgs::sg: the row the occupancy pass reads for frame t is a(t+1,.) ITSELF, without the leaky term, stored as */ \
      /* row t: step j stores a(j,.) - what step j - 1 produced - as row j - 1, and the last one is flushed after the loop) */ \
      const int trow = (FWDC) ? ((SG) ? j - 1 : j) : L - j;                                                 \
      /* (time segments: a burn-in row goes to the splice buffer - the one next to the segment to be verified, the others */ \
      /* to a row nobody reads; both choices are uniform: a descriptor and an offset in SGPRs) */            \
      const bool real_row = row_real(trow);                                                                 \
      const XBuf obuf = (!TS || real_row) ? sbuf : spbuf;                                                   \
      const int row_off = __builtin_amdgcn_readfirstlane((!TS || real_row) ? trow * Hp * 4 : (trow == spec_row ? 0 : Hp * 4)); \
      const int lane4 = lq * 4;                              /* one VGPR of addresses, the group in the SGPR offset */ \
      /* crossing: the frame this hook's row belongs to; a direction stores only the rows the other one (or the band's occupancy */ \
      /* launch) reads - alpha those of the frames below xMa, beta those from xMb on */                     \
      bool xstore = true;                                                                                   \
      if constexpr (XF) { const int tfr = (FWDC) ? f0 + j - 1 : L - j - 1; xstore = !xon || ((FWDC) ? tfr < xMa : tfr >= xMb); } \
      if constexpr (XF) xf_totb = tot_before;                /* (the crossing's share of this row: the step's tail) */ \
      LZ_HK(6);                                                                                             \
      _Pragma("unroll") for (int g = 0; g < MG; g++) {                                                      \
        if (g < groups.ngroups && PYCHAIN_EXP_NO_ROWSTORE != 1 && !((SG) && (FWDC) && j == 0)) {            \
          const lz_v2f pr = kPreRows ? prow[g] : lz_ld2(UCUR + gbase[g] * 8 + lq * 8);                      \
          /* (SG beta: the buffer holds x {b, 1}: b + c = pr.x / pr.y + c) */                                  \
          /* (one-word states: alpha's buffer holds a / cl, the row is cl (a / cl + tot); beta's b, the row b + c) */ \
          const float rowv = (SG) ? ((FWDC) ? pr.x : __builtin_fmaf(pr.x, __builtin_amdgcn_rcpf(pr.y), w.sprev)) \
                                  : (kQd ? ((FWDC) ? pr.y * (pr.x + w.sprev) : pr.x + w.sprev) : __builtin_fmaf(w.sprev, pr.y, pr.x)); \
          if (xstore) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rowv), obuf, lane4,              \
                                                            row_off + gbase[g] * 4, kStoreDeviceScope);     \
        }                                                                                                   \
      }                                                                                                     \
      LZ_HK(7);                                                                                             \
    };                                                                                                      \
    auto hook_late = [&]() {                                                                                \
      /* LDS-direct rows: the next step's row (requested above, landed by now) is clamped / exp'd in place HERE, late in */ \
      /* the arc phase, where its VALU and LDS work hides behind the gathers of sixteen waves - not in the serial tail */ \
      if constexpr (MAP::kDma) {                                                                            \
        LZ_VMWAIT();                                                                                        \
        if constexpr (pre) PYCHAIN_WAIT_VM0();               /* (ready to gather: it only has to have landed before the barrier) */ \
        else if (have_next && dma_finish(lq, (PAR) ? MAP::kX0 : MAP::kX1) && (FWDC)) bad |= 2; \
        /* (crossing: the row landed for the next tail has no nnet-output row beside it in the last steps - and the step's barrier */ \
        /* does not wait for loads) */                                                                      \
        if constexpr (XF) { if (!have_next) PYCHAIN_WAIT_VM0(); }                                           \
      }                                                                                                     \
    };                                                                                                      \
    if constexpr (SG) lazy_tile_sg<R, MAP, (FWDC), UOFF, VOFF, UNEXT, (XF) ? 1 : PYCHAIN_SG_AHEAD>(arcs, groups, w, xad01, xad23, lq, s0, s1, hook_first, hook_late); \
    else lazy_tile<R, MAP, (FWDC), UOFF, VOFF, UNEXT>(arcs, groups, w, lq, s0, s1, hook_first, hook_late);  \
    LZ_PH(0);                                                /* arc phase */                                 \
    /* rows through registers: the next step's nnet-output row into the other buffer (last read in the previous step) */ \
    if constexpr (!MAP::kDma) {                                                                             \
      if (have_next) {                                                                                      \
        if ((FWDC) && xq.has_nan()) bad |= 2;                /* a NaN network output: not ok, NaN log-probability */ \
        xq.store(reinterpret_cast<float*>(smem_raw + ((PAR) ? MAP::kX0 : MAP::kX1)), xseq, D, tq, a.input_is_exp); \
      }                                                                                                     \
    }                                                                                                       \
    LZ_PH(1);                                                /* (rows through registers: nnet-output row clamped / exp'd / stored) */ \
    /* totals: four row sums per wave before the barrier, the rest of the reduction after it */             \
    {                                                                                                       \
      const float r0 = dpp_row_sum(s0);                                                                     \
      red[(PAR) * 128 + wave * 4 + (lq >> 4)] = r0;                                                         \
      if (!(FWDC)) { const float r1 = dpp_row_sum(s1); red[(PAR) * 128 + 64 + wave * 4 + (lq >> 4)] = r1; } \
    }                                                                                                       \
    LZ_PH(2);                                                /* previous row completed and stored, row sums */ \
    /* (crossing: the step's last LDS work; its one store to memory is the last instruction before the barrier, and the barrier */ \
    /* is then one that does NOT wait for stores to be acknowledged - __syncthreads() does, a memory round trip per emitting frame) */ \
    if constexpr (XF) {                                                                                     \
      LZ_HK0(); if (xon && !((FWDC) && j == 0)) xf_row((PAR), (FWDC) ? f0 + j - 1 : L - j - 1, xf_totb, UCUR, tq, lq); LZ_HK(8); \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                       \
    } else                                                                                                  \
    __syncthreads();                                         /* every gather of this frame is done; the new vector is complete */ \
    LZ_PH(3);                                                /* wait at the barrier */                       \
    /* (this frame's totals: reduced by the next step behind its first gathers, or after the loop) */       \
  } while (0)

  // progress report of the streamed occupancy pass (DenArgs::seq_progress): P rows complete and visible device-wide
  int32_t* my_progress = a.seq_progress + (fwd ? 0 : a.B) + b;
#define PYCHAIN_LZ_REPORT(P)                                                                                \
  do {                                                                                                      \
    __builtin_amdgcn_s_waitcnt(0);                     /* this wave's row stores are acknowledged */          \
    __syncthreads();                                                                                        \
    if (tid == 0) __hip_atomic_store(my_progress, (P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         \
  } while (0)
#define PYCHAIN_XF_REPORT(P)                                                                                \
  do {                                                                                                      \
    __builtin_amdgcn_s_waitcnt(0);                                                                          \
    __syncthreads();                                                                                        \
    if (tid == 0) __hip_atomic_store(xf_my, (P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);               \
  } while (0)
  for (int jj = 0; jj < nsteps; jj += 2) {
    PYCHAIN_LZ_STEP(jj, 0, fwd);
    if (jj + 1 < nsteps) PYCHAIN_LZ_STEP(jj + 1, 1, fwd);
    PYCHAIN_LZ_SIGNAL(jj + 1);                                // (rows lag one step: after jj + 2 steps the rows of steps < jj + 1 are out)
    // step j stores the row of the frame before it (alpha row j, beta row L - j): after steps 0 .. jj + 1, jj + 2 rows
    // (SG alpha stores row j - 1 in step j: one row fewer is out)
    if (a.stream && stream_report_due(a.T, jj + 2) && jj + 2 < nsteps) PYCHAIN_LZ_REPORT(jj + 2 - ((SG && fwd) ? 1 : 0));
    // crossing: how many steps' hooks are done and their rows visible device-wide, for the peer that lands them
    if constexpr (XF) { if (xon && ((jj + 2) & 15) == 0 && jj + 2 < nsteps) PYCHAIN_XF_REPORT(jj + 2); }
  }
  const float tot_before_last = last_tot;
  if (nsteps > 0) {                                                                  // the last step's
    const float r0 = red[((nsteps - 1) & 1) * 128 + lane], r1 = fwd ? 0.f : red[((nsteps - 1) & 1) * 128 + 64 + lane];
    PYCHAIN_LZ_TOTALS(r0, r1, nsteps - 1, fwd, tid);
  }
  // crossing: the last row of either direction belongs to a frame its own side emits - the hook no step is left to run
  bool xf_last_done = false;
  if constexpr (XF) {
    if (xon && nsteps > 0) {
      const uint32_t ul = (nsteps & 1) ? MAP::kU1 : MAP::kU0;
#pragma unroll
      for (int g = 0; g < MG; g++) {
        if (g < groups.ngroups) {
          const lz_v2f u = *reinterpret_cast<const lz_v2f*>(smem_raw + ul + 8 * (gbase[g] + lane));
          const float rowv = fwd ? u.x : __builtin_fmaf(u.x, __builtin_amdgcn_rcpf(u.y), w.sprev);
          // (it is also the TRUE row the neighbouring time segment's speculated one is verified against: stored as ever)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rowv), sbuf, lane * 4, (fwd ? nsteps - 1 : L - nsteps) * Hp * 4 + gbase[g] * 4, kStoreDeviceScope);
        }
      }
      xf_row(nsteps & 1, fwd ? f0 + nsteps - 1 : L - nsteps - 1, tot_before_last, ul, tid, lane);
      __syncthreads();                                       // every wave's products are in the accumulator row
      // (one more pass: a frame outside the direction's range, after an emitted one - its accumulators leave)
      xf_row((nsteps & 1) ^ 1, fwd ? f0 + nsteps : L - nsteps - 2, 1.f, ul, tid, lane);
      xf_last_done = true;
    }
  }
  if constexpr (SG && fwd) {
    // the last row of an SG alpha recursion - a(L,.), row L - 1 (DenArgs::sg) - never saw a next step's hook
    if (nsteps > 0 && !xf_last_done) {
      const uint32_t ul = (nsteps & 1) ? MAP::kU1 : MAP::kU0;
      const int row_off = (nsteps - 1) * Hp * 4;
      for (int g = 0; g < MG; g++)
        if (g < groups.ngroups) {
          const lz_v2f u = *reinterpret_cast<const lz_v2f*>(smem_raw + ul + 8 * (gbase[g] + lane));
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(u.x), sbuf, lane * 4, row_off + gbase[g] * 4, kStoreDeviceScope);
        }
    }
  }
  if constexpr (XF) { if (xon) PYCHAIN_XF_REPORT(0x3fffffff); }   // (a peer still waiting never waits for more)
#undef PYCHAIN_XF_REPORT
  if (!fwd && (!XF || !xf_last_done)) {
    // the last beta row (row L - nsteps: row 1, or the start row if the sequence has one frame) never saw a next frame
    const uint32_t ul = (nsteps & 1) ? MAP::kU1 : MAP::kU0;
    const int row_off = (L - nsteps) * Hp * 4;
    for (int g = 0; g < MG; g++)
      if (g < groups.ngroups) {
        lz_v2f u;
        if constexpr (kQd) u = lz_v2f{*reinterpret_cast<const float*>(smem_raw + ul + 4 * (gbase[g] + lane)), 1.f};
        else u = *reinterpret_cast<const lz_v2f*>(smem_raw + ul + 8 * (gbase[g] + lane));
        const float rowv = SG ? __builtin_fmaf(u.x, __builtin_amdgcn_rcpf(u.y), w.sprev) : __builtin_fmaf(w.sprev, u.y, u.x);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rowv), sbuf, lane * 4,
                                              row_off + gbase[g] * 4, kStoreDeviceScope);
      }
  }
  PYCHAIN_LZ_SIGNAL(next_sig < a.sig_n ? 0x7ffffffe : 0);   // a sequence shorter than a bound is done with it now
  if (a.stream) PYCHAIN_LZ_REPORT(L);                       // every row this direction owes the occupancy pass: alpha 0 .. L-1, beta L .. 1
#undef PYCHAIN_LZ_REPORT
#undef PYCHAIN_LZ_SIGNAL
#undef PYCHAIN_LZ_STEP
#undef PYCHAIN_LZ_TOTALS
#ifdef PYCHAIN_PROFILE_PHASES
  if (lane == 0 && b == 0) {
    const unsigned long long n = (unsigned long long)max(1, nsteps);
    printf("lazy dir %d wave %2d rows %2d steps %d cycles/step: arcs %llu rereads+x %llu rowstore+sums %llu barrier %llu totals %llu vmwait-in-arcs %llu hook: totals %llu rows %llu crossing-tail %llu - %llu\n",
           (int)fwd, wave, groups.nslots, nsteps, lzph[0] / n, lzph[1] / n, lzph[2] / n, lzph[3] / n, lzph[4] / n, lzph[5] / n, lzph[6] / n, lzph[7] / n, lzph[8] / n, lzph[9] / n);
    if constexpr (XF) printf("xf tail dir %d wave %2d: reads %llu products %llu sum %llu flush %llu\n", (int)fwd, wave, xft[0] / n, xft[1] / n, xft[2] / n, xft[3] / n);
  }
#endif
#undef LZ_PH
#undef LZ_VMWAIT
#undef LZ_PH0

  if (TS && fwd && !seg_last) {                            // an inner alpha segment: nothing to total; a NaN it staged is reported
    if (bad & 2) { __hip_atomic_store(a.xnan + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); bad &= ~2; }   // (whichever wave staged it)
    if (bad && lane == 0) atomicAdd(a.redo + 3, 1);      // (a speculative launch counts into a scratch word: below)
    return;
  }
  if constexpr (fwd) {
    // ComputeTotLogLike, chain-computation.cc:209-230: log sum_i a'(L,i) final(i) + sum_{t<L} log tot(t),
    // a'(L,i) = a(L,i) + tot(L) cl(i); the vector of step L-1 sits in buffer (L & 1)
    const float* fin = reinterpret_cast<const float*>(plan + hd->off_final_a);
    const uint32_t ul = (nsteps & 1) ? MAP::kU1 : MAP::kU0;
    float f = 0.f;
    for (int i = tid; i < Hp; i += NT) {
      if constexpr (kQd) {
        f += *reinterpret_cast<const float*>(smem_raw + lz_cl_of<MAP, kQd>() + 4 * i) * (*reinterpret_cast<const float*>(smem_raw + ul + 4 * i) + last_tot) * fin[i];
      } else {
        const lz_v2f u = *reinterpret_cast<const lz_v2f*>(smem_raw + ul + 8 * i);
        f += __builtin_fmaf(last_tot, u.y, u.x) * fin[i];
      }
    }
    f = wave_sum(f);
    __syncthreads();
    if (tid < 64) red[tid] = 0.f;
    if (tid == 64) red[64] = 0.f;
    __syncthreads();
    if (lane == 0) red[wave] = f;
    __syncthreads();
    if (PRE && __hip_atomic_load(a.xnan + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) bad |= 2;   // (den_exp_rows_kernel saw it)
    if (bad & 2) red[64] = 1.f;                       // somebody staged a NaN network output
    __syncthreads();
    const float fs = wave_sum(red[lane]);
    if (tid == 0) {
      a.fin_dot[b] = red[64] != 0.f ? __builtin_nanf("") : fs;       // den_finish_kernel: objf = sum_t log tot(t) + log of this
      if (!(fs > 0.f)) bad |= 1;
    }
  }
  // A segmented launch is speculative: if a splice does not verify, the uncut launch behind it recomputes every row and counts
  // for itself - so this launch counts into a scratch word (DenArgs::redo[3]) that den_finish_kernel merges into `bad` only when
  // nothing was redone (a healthy recomputation used to inherit the speculation's counts, real ones were counted twice)
  if (bad && lane == 0) atomicAdd(TS ? a.redo + 3 : a.bad, 1);
}

// XM: kLzRowsPre - the rows were exp'd ahead by den_exp_rows_kernel (DenArgs::ex); kLzRowsHalf - 2-byte rows (DenArgs::x_half)
template <int R, typename MAP, int XM = kLzRowsF32, bool TS = false, bool NC = false, bool SG = false, bool XF = false>
__global__ __launch_bounds__(MAP::kWaves * 64) void den_recursion_lazy_kernel(const DenArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // (the fallback launch behind a segmented one: runs only if a splice did not verify - DenArgs::redo)
  if (a.redo_if && __hip_atomic_load(a.redo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  if constexpr (TS) {
    if (den_tseg_off(a)) return;                           // the plan is cooling down after misses: the uncut launch behind this one does the work
    const unsigned per_dir = (unsigned)a.B * (unsigned)a.tseg;                        // workgroups per direction: segment-major
    const unsigned r = blockIdx.x < per_dir ? blockIdx.x : blockIdx.x - per_dir;
    if (blockIdx.x < per_dir) lazy_recursion<R, MAP, true, XM, true, NC, SG, XF>(a, smem_raw, (int)(r % (unsigned)a.B), (int)(r / (unsigned)a.B));
    else lazy_recursion<R, MAP, false, XM, true, NC, SG, XF>(a, smem_raw, (int)(r % (unsigned)a.B), (int)(r / (unsigned)a.B));
  } else {
    if (blockIdx.x < (unsigned)a.B) lazy_recursion<R, MAP, true, XM, false, NC, SG, XF>(a, smem_raw, blockIdx.x);
    else lazy_recursion<R, MAP, false, XM, false, NC, SG, XF>(a, smem_raw, blockIdx.x - a.B);
  }
}
