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
#define PLAN_VERSION 8
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
#define PLAN_REC8_WAVES 8      // the alpha / beta plans again, for the 8-wave ("wide") lazy recursion: the waves of the 16-wave
                               // dealing joined in pairs, so a wave owns at most twice the slot-rows and groups

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
  int32_t reserved1[2];
  TilePlan gamma2;             // same rows as `gamma` (row_pdf applies), dealt to PLAN_GAM2_WAVES waves
  TilePlan alpha8, beta8;      // same rows and slot order as `alpha` / `beta`, dealt to PLAN_REC8_WAVES waves
  int32_t rec8_max_wave_groups;
  int32_t payload_hash;        // FNV-1a over bytes [sizeof(PlanHeader), total_bytes): checked by pychain_hip_den_plan_info
  int32_t reserved2[2];
};

#ifdef __cplusplus
#include <stddef.h>
namespace pychain_hip {
// FNV-1a (32 bit) over the blob behind its header
inline uint32_t plan_payload_hash(const void* blob, size_t total_bytes) {
  const unsigned char* p = (const unsigned char*)blob;
  uint32_t h = 2166136261u;
  for (size_t i = sizeof(PlanHeader); i < total_bytes; i++) { h ^= p[i]; h *= 16777619u; }
  return h;
}
}  // namespace pychain_hip
#endif

#endif
