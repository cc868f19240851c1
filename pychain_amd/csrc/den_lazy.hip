// den_lazy.hip - the recursions that run the benchmarks: the lazy-normalisation alpha / beta recursion in its three shapes
// (den_lazy.inc.h) and the two-sequences-per-workgroup recursion (den_pair.inc.h), with their shape predicates and launches.
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
#include "den_lazy.inc.h"
#include "den_pair.inc.h"

// The lazy-normalisation recursion (den_lazy.inc.h) in its 16-wave shape serves the shape the benchmarks run: nnet-output
// row and state vector within its fixed LDS map, every arc of a wave in registers, at most LzNarrow::kMaxGroups groups
// per wave (bit 30 of the plan hint), the whole sequence in one launch.
// (`dma`: rows by LDS-direct loads, which take any row length; rows through registers are float4 loads: D % 4 == 0)
inline bool lazy_shape_ok(const DenArgs& a, int hint, bool dma = false) {
  const int rows = hint & 1023;
  return ((hint >> 30) & 1) && (dma || a.D % 4 == 0) && a.D <= (int)LzNarrow::kMaxPdfs && a.Hp <= (int)LzNarrow::kMaxStates && rows > 0 &&
         rows <= kMaxResident && PLAN_REC_WAVES == 16 && a.plan_stride >= 0;
}
// (launch hint bit 28: the plan has beta positions that take no constant - the kernels' NC form, den_lazy.inc.h)
inline bool hint_no_const(int hint) { return hint >= 0 && ((hint >> 28) & 1); }   // (negative: PYCHAIN_HIP_HINT_GENERAL)
// (launch hint bit 27: a "pdf by state" plan - the one-gather form of the recursions, den_lazy.inc.h: SG.  Taken where the form
// is instantiated: the 16-wave map of C3 with LDS-direct fp32 rows the kernel clamps / exp's itself, loops of up to 32 rows, no
// beta position without the constant; everything else runs such a plan like any other.)
inline bool hint_pdf_by_state(int hint) { return hint >= 0 && ((hint >> 27) & 1); }
inline bool sg_shape_ok(const DenArgs& a, int hint) {
  // (at most 32 slot-rows per wave: with two registers per arc the 40-row loop does not fit 128 registers - it compiled to 300
  // reloads from scratch per frame; such a plan runs the ordinary kernels)
  return hint_pdf_by_state(hint) && (hint & 1023) <= 32 && !hint_no_const(hint) && a.knobs.den_sg != 0 && !a.input_is_exp && !a.x_half && !a.use_ex &&
         a.plan_stride == 0 && a.D <= (int)LzNarrow::kMaxPdfs;
}
template <int R, typename M, bool TS>
hipError_t launch_lz_sg(const DenArgs& a, const dim3 grid, hipStream_t st) {
  return launch_one(den_recursion_lazy_kernel<R, M, kLzRowsF32, TS, false, true>, a, grid, M::kBytes, st, M::kWaves * 64);
}
// ... with the crossing (DenArgs::xf; LzCross: state vectors of up to 3072 positions)
inline bool xf_shape_ok(const DenArgs& a, int hint) {
  // (at most 32 slot-rows per wave: the 40-row form of this kernel spills thirty registers inside its loop)
  return sg_shape_ok(a, hint) && (hint & 1023) <= 32 && a.knobs.den_cross != 0 && !a.fused && a.Hp <= (int)LzCross::kMaxStates && a.D % 4 == 0 &&
         a.D <= (int)LzCross::kMaxPdfs && a.T >= 4 * kCrossBand + 8;
}
template <int R, bool TS>
hipError_t launch_lz_xf(const DenArgs& a, const dim3 grid, hipStream_t st) {
  return launch_one(den_recursion_lazy_kernel<R, LzCross, kLzRowsF32, TS, false, true, true>, a, grid, LzCross::kBytes, st, LzCross::kWaves * 64);
}
// (launch hint bit 19: one position per state on both sides, every leaky probability positive - the one-word state vectors of
// den_lazy.inc.h: MAP::kQ.  Instantiated for the 32-row loop of the 16-wave map with LDS-direct fp32 rows, uncut sequences; coef >= 1e-8
// keeps 1 / (coef leaky) <= 1e20.  Option den_q, off by default: measured, it does not pay - DESIGN.md 3.16.)
inline bool hint_one_word(int hint) { return hint >= 0 && ((hint >> 19) & 1); }
inline bool q_shape_ok(const DenArgs& a, int hint) {
  const int rows = hint & 1023;
  return hint_one_word(hint) && !hint_no_const(hint) && rows > 16 && rows <= 32 && a.knobs.den_q != 0 && a.coef >= 1e-8f && a.coef <= 1.f &&
         a.tseg <= 1 && !a.x_half;
}
template <int R, typename M, int XM, bool TS>
hipError_t launch_lz(const DenArgs& a, const dim3 grid, hipStream_t st, bool nc) {
  if constexpr (!M::kQ) { if (nc) return launch_one(den_recursion_lazy_kernel<R, M, XM, TS, true>, a, grid, M::kBytes, st, M::kWaves * 64); }
  return launch_one(den_recursion_lazy_kernel<R, M, XM, TS, false>, a, grid, M::kBytes, st, M::kWaves * 64);
}
hipError_t launch_lazy(const DenArgs& a, int hint, hipStream_t st) {
  const dim3 grid(2 * a.B);
  const int rows = hint & 1023;
  const bool nc = hint_no_const(hint);
  if (rows <= 16) return launch_lz<16, LzNarrow, kLzRowsF32, false>(a, grid, st, nc);
  if (rows <= 32) return launch_lz<32, LzNarrow, kLzRowsF32, false>(a, grid, st, nc);
  if (rows <= PLAN_RESIDENT_FIT) return launch_lz<PLAN_RESIDENT_FIT, LzNarrow, kLzRowsF32, false>(a, grid, st, nc);
  return launch_lz<kMaxResident, LzNarrow, kLzRowsF32, false>(a, grid, st, nc);
}
// the 16-wave shape with LDS-direct nnet-output rows (LzDma): D <= 9216, Hp <= 3072
inline bool dma_shape_ok(const DenArgs& a, int hint) {
  const int rows = hint & 1023;
  return ((hint >> 30) & 1) && a.D <= (int)LzDma::kMaxPdfs && a.Hp <= (int)LzDma::kMaxStates && rows > 0 &&
         rows <= kMaxResident && PLAN_REC_WAVES == 16 && a.plan_stride >= 0;
}
template <typename M, int XM, bool TS = false>
hipError_t launch_dma_m(const DenArgs& a, int rows, hipStream_t st, bool nc) {
  const dim3 grid(2 * a.B * (TS ? a.tseg : 1));
  if (rows <= 16) return launch_lz<16, M, XM, TS>(a, grid, st, nc);
  if (rows <= 32) return launch_lz<32, M, XM, TS>(a, grid, st, nc);
  return launch_lz<kMaxResident, M, XM, TS>(a, grid, st, nc);
}
// how the rows of this call arrive (lazy_recursion: XM): exp'd ahead (fp32 whatever the input's type), 2-byte, fp32;
// time segments (DenArgs::tseg: never with rows exp'd ahead - they are written from the sequence ends inwards)
template <typename M>
hipError_t launch_dma_x(const DenArgs& a, int rows, hipStream_t st, bool nc) {
  if (a.tseg > 1) return a.x_half ? launch_dma_m<M, kLzRowsHalf, true>(a, rows, st, nc) : launch_dma_m<M, kLzRowsF32, true>(a, rows, st, nc);
  if (a.use_ex) return launch_dma_m<M, kLzRowsPre>(a, rows, st, nc);
  return a.x_half ? launch_dma_m<M, kLzRowsHalf>(a, rows, st, nc) : launch_dma_m<M, kLzRowsF32>(a, rows, st, nc);
}
hipError_t launch_dma(const DenArgs& a, int hint, hipStream_t st) {
  // the map of C3 where the shape fits it, else the one for rows of up to 9216 pdfs
  if (lazy_shape_ok(a, hint, true) && sg_shape_ok(a, hint)) {            // a "pdf by state" plan: one gather per arc
    const dim3 grid(2 * a.B * (a.tseg > 1 ? a.tseg : 1));         // (sg_shape_ok: rows <= 32)
    if (a.xf) {
      return a.tseg > 1 ? launch_lz_xf<32, true>(a, grid, st) : launch_lz_xf<32, false>(a, grid, st);   // (xf_shape_ok: rows <= 32)
    }
    return a.tseg > 1 ? launch_lz_sg<32, LzNarrowDma, true>(a, grid, st) : launch_lz_sg<32, LzNarrowDma, false>(a, grid, st);
  }
  if (lazy_shape_ok(a, hint, true) && q_shape_ok(a, hint)) {         // one-word state vectors (rows <= 32: the 32-row loop)
    const dim3 grid(2 * a.B);                             // (q_shape_ok: fp32 rows, the sequence in one piece)
    typedef LzNarrowDmaQ M;
    if (a.use_ex) return launch_lz<32, M, kLzRowsPre, false>(a, grid, st, false);
    return launch_lz<32, M, kLzRowsF32, false>(a, grid, st, false);
  }
  if (lazy_shape_ok(a, hint, true)) return launch_dma_x<LzNarrowDma>(a, hint & 1023, st, hint_no_const(hint));
  return launch_dma_x<LzDma>(a, hint & 1023, st, hint_no_const(hint));
}
// Four-wave workgroups over the plan's four-wave dealing (hint bit 29: every plan of the call holds alpha4 / beta4, and the
// hint's row count is that dealing's): small graphs, LDS-direct rows.
inline bool small_shape_ok(const DenArgs& a, int hint) {
  const int rows = hint & 1023;
  return ((hint >> 29) & 1) && a.D <= (int)LzSmall::kMaxPdfs && a.Hp <= (int)LzSmall::kMaxStates && rows > 0 && rows <= kMaxResident &&
         a.plan_stride >= 0;
}
// A frame of a small graph is the arc loop's chunks, one dependent LDS round trip each with one wave per SIMD, whatever the
// wave gathers (profiles/r04_c2_small_phase_timers_ring4.txt): the loop is as long as the longest wave's rows, in steps of 8 from 16 on.
template <int PRE>
hipError_t launch_small_p(const DenArgs& a, int hint, hipStream_t st) {
  const dim3 grid(2 * a.B);
  const int rows = hint & 1023;
  typedef LzSmall M;
  const bool nc = hint_no_const(hint);
  if (rows <= 16) return launch_lz<16, M, PRE, false>(a, grid, st, nc);
  if (rows <= 24) return launch_lz<24, M, PRE, false>(a, grid, st, nc);
  if (rows <= 32) return launch_lz<32, M, PRE, false>(a, grid, st, nc);
  return launch_lz<kMaxResident, M, PRE, false>(a, grid, st, nc);
}
hipError_t launch_small(const DenArgs& a, int hint, hipStream_t st) {
  if (a.use_ex) return launch_small_p<kLzRowsPre>(a, hint, st);
  return a.x_half ? launch_small_p<kLzRowsHalf>(a, hint, st) : launch_small_p<kLzRowsF32>(a, hint, st);
}

// Two sequences per workgroup (den_pair.inc.h): one plan for all sequences, nnet-output rows and state vectors
// within its fixed LDS map, every arc of a plan wave in registers, the whole sequence in one launch.
inline bool pair_shape_ok(const DenArgs& a, int hint) {
  const int rows = hint & 1023;
  return a.plan_stride == 0 && a.D % 4 == 0 && a.D <= 4096 && a.Hp <= 4096 && rows > 0 && rows <= kMaxResident &&
         PLAN_REC_WAVES == 16 && a.B >= 2 && !hint_no_const(hint);   // (its normalise pass gives every position the constant c(t))
}
hipError_t launch_pair(const DenArgs& a, int hint, hipStream_t st) {
  const dim3 grid(2 * ((a.B + 1) / 2));
  const int rows = hint & 1023;
  if (a.x_half) {                                        // 2-byte nnet-output rows (DenArgs::x_half)
    if (rows <= 16) return launch_one(den_recursion_pair_kernel<16, true>, a, grid, kPrBytes, st, kPrNT);
    if (rows <= 32) return launch_one(den_recursion_pair_kernel<32, true>, a, grid, kPrBytes, st, kPrNT);
    return launch_one(den_recursion_pair_kernel<kMaxResident, true>, a, grid, kPrBytes, st, kPrNT);
  }
  if (rows <= 16) return launch_one(den_recursion_pair_kernel<16>, a, grid, kPrBytes, st, kPrNT);
  if (rows <= 32) return launch_one(den_recursion_pair_kernel<32>, a, grid, kPrBytes, st, kPrNT);
  return launch_one(den_recursion_pair_kernel<kMaxResident>, a, grid, kPrBytes, st, kPrNT);
}

}  // namespace

bool den_lazy_eligible(const DenArgs& a, int resident_slot_rows) { return lazy_shape_ok(a, resident_slot_rows); }
bool den_small_eligible(const DenArgs& a, int resident_slot_rows) { return small_shape_ok(a, resident_slot_rows); }
bool den_dma_eligible(const DenArgs& a, int resident_slot_rows) { return lazy_shape_ok(a, resident_slot_rows, true) || dma_shape_ok(a, resident_slot_rows); }
bool den_pair_eligible(const DenArgs& a, int resident_slot_rows) { return pair_shape_ok(a, resident_slot_rows); }
bool den_sg_eligible(const DenArgs& a, int resident_slot_rows) { return lazy_shape_ok(a, resident_slot_rows, true) && sg_shape_ok(a, resident_slot_rows); }
bool den_xf_eligible(const DenArgs& a, int resident_slot_rows) { return lazy_shape_ok(a, resident_slot_rows, true) && xf_shape_ok(a, resident_slot_rows); }
bool den_q_eligible(const DenArgs& a, int resident_slot_rows) {
  return lazy_shape_ok(a, resident_slot_rows, true) && !sg_shape_ok(a, resident_slot_rows) && q_shape_ok(a, resident_slot_rows);
}
int den_xf_band() { return kCrossBand; }

// the recursion launch of a call whose DenArgs say pair or lazy (launch_den)
hipError_t launch_den_lazy_family(const DenArgs& a, int hint, hipStream_t st) {
  if (a.pair) return launch_pair(a, hint, st);
  return a.shape == kShapeSmall ? launch_small(a, hint, st) : (a.shape == kShapeDma ? launch_dma(a, hint, st) : launch_lazy(a, hint, st));
}

}  // namespace pychain_hip
