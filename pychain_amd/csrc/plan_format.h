// plan_format.h - device-side layout of a compiled denominator graph ("plan").
//
// A plan holds three *tile plans*.  A tile plan evaluates, once per frame,
//
//     out[row] = sum_{k in row}  p_k * U[i0_k] * V[i1_k]
//
// for every row, where U and V are two vectors resident in LDS:
//
//   alpha plan: row = destination state, U = alpha'(t-1,.) , V = exp x(t-1,.)   (i0 = source state, i1 = pdf)
//   beta  plan: row = source state,      U = beta(t+1,.)   , V = exp x(t,.)     (i0 = dest state,   i1 = pdf)
//   gamma plan: row = pdf-id,            U = alpha'(t,.)   , V = beta(t+1,.)    (i0 = source state, i1 = dest state)
//
// Rows are sorted by descending arc count and cut into groups of 64 (one row per
// lane of a wave64).  A group is stored as `nslots` slot-rows; slot-row j holds
// the j-th arc of each of the 64 rows (zero-probability padding where a row is
// shorter), 8 bytes per lane, lane-contiguous, so a wave reads 512 contiguous
// bytes per slot-row.  Groups are dealt to `nwaves` waves by longest-processing
// -time-first so every wave of the workgroup owns about the same number of
// slot-rows.  States are renumbered by sorted position, so lane l of a group
// writes out[out_base + l]: stores are conflict-free and need no index.
//
// State numbering: the alpha side numbers states by descending in-degree
// (position "pa"), the beta side by descending out-degree ("pb").  Vectors in
// the plan are stored in the numbering of the side that uses them and padded
// with zeros to Hp = roundup(H, 64).
#ifndef PYCHAIN_HIP_PLAN_FORMAT_H_
#define PYCHAIN_HIP_PLAN_FORMAT_H_

#include <stdint.h>

#define PLAN_MAGIC 0x4C504843  // "CHPL"
#define PLAN_VERSION 15
#ifndef PLAN_REC_WAVES
#define PLAN_REC_WAVES 16      // waves per workgroup the alpha/beta plans are scheduled for
#define PLAN_GAM_WAVES 16      // same for the gamma plan
#endif
// slot-rows per wave the recursion kernels keep in registers (den_kernels.hip picks the smallest that fits)
#define PLAN_RESIDENT_0 16
#define PLAN_RESIDENT_1 32
#define PLAN_RESIDENT_2 40
#define PLAN_RESIDENT_FIT 36   // between the last two: reached with per-group slack fitted to the wave budget (plan.cpp: fit_slack)
#define PLAN_GAM2_WAVES 8      // the gamma plan again, scheduled for the two-frame occupancy kernel (8 waves x 256 VGPRs)
#define PLAN_REC4_WAVES 4      // the alpha / beta plans again, dealt to FOUR waves: small graphs (a few hundred states, a few thousand
                               // arcs: BASELINE.json's C2) run their recursions in 256-thread workgroups - sixteen waves meeting at
                               // a barrier for four groups of work are the wrong shape (den_lazy.inc.h: LzSmall)

struct TilePlan {              // all offsets are bytes from the start of the blob
  int32_t ngroups;
  int32_t nwaves;
  int32_t off_wave_tab;        // WaveEntry[nwaves]
  int32_t off_group_tab;       // GroupEntry[ngroups], in wave order
  int32_t off_slots;           // uint2[total_slot_rows * 64]
  int32_t total_slot_rows;
  int32_t max_wave_slot_rows;  // max over waves of the slot-rows a wave owns
  int32_t nrows;               // real rows (states / pdfs with arcs)
};

struct WaveEntry {
  int32_t first_group;         // index into the group table
  int32_t ngroups;
  int32_t slot_row_begin;      // first slot-row of this wave in the slot stream
  int32_t nslot_rows;
};

struct GroupEntry {
  int32_t out_base;            // lane l writes out[out_base + l]
  int32_t nslots;              // slot-rows in this group (may be 0)
};

constexpr int PLAN_MAX_EXTRA_A = 1024;   // a pdf-by-state plan's states on several alpha positions: at most this many further positions
struct PlanHeader {
  int32_t magic, version;
  int32_t H, K, D, Hp;
  int32_t total_bytes;
  int32_t rec_max_wave_groups; // most groups any wave of the alpha / beta plans owns
  TilePlan alpha, beta, gamma;
  int32_t off_init_a;          // float[Hp]  initial_probs, alpha numbering
  int32_t off_leaky_a;         // float[Hp]  leaky_probs,   alpha numbering
  int32_t off_final_a;         // float[Hp]  final_probs,   alpha numbering
  int32_t off_leaky_b;         // float[Hp]  leaky_probs,   beta numbering
  int32_t off_final_b;         // float[Hp]  final_probs,   beta numbering
  int32_t off_row_pdf;         // int32[gamma.ngroups*64] natural pdf-id of each gamma row, -1 = padding
  int32_t rec4_max_wave_groups;  // most groups any wave of alpha4 / beta4 owns (0: those tiles are not in the plan)
  int32_t header_hash;         // FNV-1a over this header with both hash fields zero: checked by pychain_hip_den_plan_info
  TilePlan gamma2;             // same rows as `gamma` (row_pdf applies), dealt to PLAN_GAM2_WAVES waves
  // same rows and slot order as `alpha` / `beta`, dealt to PLAN_REC4_WAVES waves; nwaves == 0: the graph is too large for a
  // four-wave workgroup (a wave would own more than PLAN_RESIDENT_2 slot-rows or more than four groups) and they are left out
  TilePlan alpha4, beta4;
  int32_t payload_hash;        // FNV-1a over bytes [sizeof(PlanHeader), total_bytes): checked by pychain_hip_den_plan_info
  // states on several lanes (plan.cpp): the graph's own state count (H counts POSITIONS of the longer side), and the beta
  // positions that take no constant c(t) - every position of a state after its first: int32[n_no_const] at off_no_const
  int32_t graph_states, off_no_const, n_no_const;
  // "pdf by state" (format 14): every arc ENTERING a state carries that state's pdf - the shape of a chain denominator compiled
  // from a phone LM over HMM topologies, where a pdf belongs to the state a transition enters.  flags bit 0 set: the plan then
  // holds the pdf-id of every position of both numberings (a state without arcs entering it: 0), and the recursions need ONE
  // gather per arc - the state operand; the nnet output multiplies a row's sum once, where the row's value is formed (alpha), or
  // is folded into the vector beta gathers from where that vector is written (den_lazy.inc.h: SG).  The tiles themselves are
  // as for any graph (the arcs keep their pdf operand): every other kernel reads the plan unchanged.
  int32_t flags;
  int32_t off_pdf_a;           // int32[Hp]  pdf-id of the arcs entering the state at this alpha position
  int32_t off_pdf_b;           // int32[Hp]  ... at this beta position
  // ... and the occupancies collapse from a sum over ARCS to a sum over STATES: with a(t+1,j) = x(t,pdf_j) sum_k p_k alpha'(t,src_k)
  // - the value the alpha recursion forms anyway -  gamma(t,n) ~ sum_{j: pdf_j = n} a(t+1,j) beta(t+1,j).  The alpha recursion of
  // such a plan stores a(t+1,.) at row t (instead of alpha'(t,.)), and the occupancy kernels walk these tiles - one pseudo-arc per
  // state position {alpha position, beta position, 1}, rows = pdfs, no nnet-output row read - with the kernels and schedules
  // they have (DenArgs::sg).  gamma_sg: 16 waves, gamma2_sg: 8 waves; at most PLAN_RESIDENT_0 slot-rows per wave.
  int32_t off_row_pdf_sg;      // int32[gamma_sg.ngroups*64] natural pdf-id of each row of the *_sg tiles, -1 = padding
  TilePlan gamma_sg, gamma2_sg;
  // ... and (format 15) each recursion can emit the occupancies of its own second half ("crossing", den_lazy.inc.h: XF): a workgroup holds
  // its side's value of a position and needs the OTHER side's value of the same state -
  int32_t off_a2b;             // int32[Hp]  alpha position -> the state's beta position
  int32_t off_b2a;             // int32[Hp]  beta position  -> the state's FIRST alpha position
  int32_t off_extra_a;         // int32[2 n_extra_a]  {beta position, alpha position}: the alpha positions of a state after its first
  int32_t n_extra_a;           // <= PLAN_MAX_EXTRA_A
};
#define PLAN_FLAG_PDF_BY_STATE 1

// ---- the GENERAL format: graphs the compiled tile plans do not take (more than 65 535 states or pdfs, or vectors that
// do not fit the LDS of one CU).  The reference layout as it is (fstext.cc:49-116), plus the arcs grouped by pdf-id for
// an atomics-free occupancy pass; read by den_general.hip straight from global memory.
#define PLAN_MAGIC_GENERAL 0x47504843  // "CHPG"
struct GeneralPlanHeader {
  int32_t magic, version;
  int32_t H, K, D, Hp;
  int64_t total_bytes;
  int64_t off_a_idx, off_a_arc, off_a_p;   // arcs by destination: int32[H][2] begin/end, int32[K][2] {src, pdf}, float[K]
  int64_t off_b_idx, off_b_arc, off_b_p;   // arcs by source:      int32[H][2],           int32[K][2] {dst, pdf}, float[K]
  int64_t off_g_idx, off_g_arc, off_g_p;   // arcs by pdf-id:      int32[D + 1],          int32[K][2] {src, dst}, float[K]
  int64_t off_leaky, off_init, off_final;  // float[Hp], natural state order
  int32_t payload_hash, reserved[3];     // reserved[0] = FNV-1a over this header with both hash words zero (general_header_hash)
};

#ifdef __cplusplus
#include <stddef.h>
namespace pychain_hip {
inline uint32_t general_payload_hash(const void* blob, size_t total_bytes) {
  const unsigned char* p = (const unsigned char*)blob;
  uint32_t h = 2166136261u;
  for (size_t i = sizeof(GeneralPlanHeader); i < total_bytes; i++) { h ^= p[i]; h *= 16777619u; }
  return h;
}
// Does a graph of these sizes fit the compiled-plan kernels (packed 16-bit LDS addresses; state vector + nnet-output row
// in the 160 KiB LDS of one CU, as launch_den checks)?  Else the plan is built in the general format.
inline bool plan_fits_fast_kernels(int H, int D) {
  if (H > 65535 || D > 65535) return false;
  const size_t Hp = (size_t)(H + 63) / 64 * 64, Dp = (size_t)(D + 3) & ~(size_t)3, gmax = (size_t)(D + 63) / 64;
  const bool db = D % 4 == 0 && D <= 4096;
  const size_t lds_rec = 4 * (3 * Hp + (db ? 8192 : Dp) + 32), lds_gam = 4 * (2 * Hp + 2 * Dp + gmax * 64 + 16);
  return lds_rec <= 160 * 1024 && lds_gam + 64 <= 160 * 1024;     // (+ 64: the occupancy kernels' static LDS)
}
// FNV-1a (32 bit) over the blob behind its header
inline uint32_t plan_payload_hash(const void* blob, size_t total_bytes) {
  const unsigned char* p = (const unsigned char*)blob;
  uint32_t h = 2166136261u;
  for (size_t i = sizeof(PlanHeader); i < total_bytes; i++) { h ^= p[i]; h *= 16777619u; }
  return h;
}
// ... and over the header itself, its two hash fields taken as zero: the kernels follow the header's tile offsets and wave
// tables unchecked, so a cache file with a damaged or foreign header over an intact payload must not pass either
inline uint32_t plan_header_hash(const PlanHeader& hd) {
  PlanHeader c = hd;
  c.header_hash = 0; c.payload_hash = 0;
  const unsigned char* p = (const unsigned char*)&c;
  uint32_t h = 2166136261u;
  for (size_t i = 0; i < sizeof(PlanHeader); i++) { h ^= p[i]; h *= 16777619u; }
  return h;
}
inline uint32_t general_header_hash(const GeneralPlanHeader& hd) {
  GeneralPlanHeader c = hd;
  c.payload_hash = 0; c.reserved[0] = 0;
  const unsigned char* p = (const unsigned char*)&c;
  uint32_t h = 2166136261u;
  for (size_t i = 0; i < sizeof(GeneralPlanHeader); i++) { h ^= p[i]; h *= 16777619u; }
  return h;
}
// every tile of a header lies inside the blob (offsets + sizes against total_bytes)
inline bool tile_in_bounds(const TilePlan& tp, size_t total_bytes) {
  if (tp.nwaves == 0 && tp.ngroups == 0) return true;       // (a tile that is not in the plan)
  if (tp.nwaves < 0 || tp.ngroups < 0 || tp.total_slot_rows < 0 || tp.off_wave_tab < 0 || tp.off_group_tab < 0 || tp.off_slots < 0) return false;
  return (size_t)tp.off_wave_tab + (size_t)tp.nwaves * sizeof(WaveEntry) <= total_bytes &&
         (size_t)tp.off_group_tab + (size_t)tp.ngroups * sizeof(GroupEntry) <= total_bytes &&
         (size_t)tp.off_slots + (size_t)tp.total_slot_rows * 64 * 8 <= total_bytes;
}
inline bool plan_header_in_bounds(const PlanHeader& hd) {
  const size_t n = (size_t)hd.total_bytes;
  if (hd.H <= 0 || hd.Hp < hd.H || hd.D <= 0) return false;
  for (const TilePlan* tp : {&hd.alpha, &hd.beta, &hd.gamma, &hd.gamma2, &hd.alpha4, &hd.beta4})
    if (!tile_in_bounds(*tp, n)) return false;
  for (int32_t off : {hd.off_init_a, hd.off_leaky_a, hd.off_final_a, hd.off_leaky_b, hd.off_final_b})
    if (off < 0 || (size_t)off + (size_t)hd.Hp * 4 > n) return false;
  if (hd.n_no_const < 0 || hd.off_no_const < 0 || (size_t)hd.off_no_const + (size_t)hd.n_no_const * 4 > n) return false;
  if (hd.flags & PLAN_FLAG_PDF_BY_STATE) {
    for (int32_t off : {hd.off_pdf_a, hd.off_pdf_b})
      if (off < 0 || (size_t)off + (size_t)hd.Hp * 4 > n) return false;
    if (!tile_in_bounds(hd.gamma_sg, n) || !tile_in_bounds(hd.gamma2_sg, n)) return false;
    if (hd.off_row_pdf_sg < 0 || (size_t)hd.off_row_pdf_sg + (size_t)hd.gamma_sg.ngroups * 64 * 4 > n) return false;
    for (int32_t off : {hd.off_a2b, hd.off_b2a})
      if (off < 0 || (size_t)off + (size_t)hd.Hp * 4 > n) return false;
    if (hd.n_extra_a < 0 || hd.n_extra_a > PLAN_MAX_EXTRA_A || hd.off_extra_a < 0 || (size_t)hd.off_extra_a + (size_t)hd.n_extra_a * 8 > n) return false;
  }
  return hd.off_row_pdf >= 0 && (size_t)hd.off_row_pdf + (size_t)hd.gamma.ngroups * 64 * 4 <= n;
}
}  // namespace pychain_hip
#endif

#endif
