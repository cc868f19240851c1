// den_kernels.h - launch interface of the denominator kernels (den_kernels.hip).
#ifndef PYCHAIN_HIP_DEN_KERNELS_H_
#define PYCHAIN_HIP_DEN_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace pychain_hip {

enum { kShapeRegs = 0, kShapeDma = 2, kShapeSmall = 3 };
constexpr int kMaxTimeSegs = 4;

struct DenArgs {
  const char* plans;         // device plan(s)
  int64_t plan_stride;       // bytes between per-sequence plans, 0 = shared
  const float* x;            // [B,T,D]   (x_half: 2-byte elements behind this pointer)
  const int64_t* lengths;    // [B]
  float* objf;               // [B]
  float* grad;               // [B,T,D]   (x_half: written in the network output's type)
  // 0: fp32 network output and gradient; kXBf16 / kXF16 (device_utils.h): 2-byte rows read as they are by the kernels that
  // take them (den_call_half_native: lazy recursions with LDS-direct rows, pair recursion, both occupancy kernels, the rows
  // exp'd ahead) and the gradient rounded to the same type where it is written - no up-cast pass, no fp32 copy of [B,T,D]
  int x_half;
  int32_t* bad;              // [1]
  float* alpha_store;        // [B,T,Hp]    alpha'(t,.)/tot(t), alpha numbering
  float* beta_store;         // [B,T+1,Hp]  beta(t,.) (unit sum), beta numbering; row 0 unused
  // 1: the recursions of this call ran as den_recursion_lazy_kernel (den_lazy.inc.h): rows stored as a'(t,.) and
  // b(t,.) + c(t), each in a per-frame scale of its own; 0: as den_recursion_kernel (rows normalised).  The
  // occupancy kernels read either form as it is; den_finish_kernel needs to know which scales were divided out.
  int lazy;
  int fused;                  // the call is a fused loss (a numerator runs beside it): api.hip decides pair / time segments by it
  // 1: the recursions run as den_recursion_pair_kernel (den_pair.inc.h): two sequences per workgroup, ceil(B/2)
  // workgroups per direction; rows as den_recursion_kernel stores them (lazy = 0).  Shared plan only.
  int pair;
  // which shape the lazy recursions run in (den_lazy.inc.h): kShapeRegs = LzNarrow (16 waves, D <= 4096, rows through
  // registers), kShapeDma = LzNarrowDma / LzDma (16 waves, rows by LDS-direct loads, up to 9216 pdfs), kShapeSmall = LzSmall
  // (4 waves over the plan's four-wave dealing: small graphs)
  int shape;
  // 1: a "pdf by state" plan in its own forms (plan_format.h: PLAN_FLAG_PDF_BY_STATE) - the lazy recursions gather ONE operand per arc
  // (den_lazy.inc.h: SG), the alpha store holds a(t+1,.) at row t instead of alpha'(t,.), and the occupancy kernels sum over
  // STATES (the plan's gamma_sg tiles) without reading the nnet-output row.  Decided once per call; a later
  // chain_loss_backward on the same workspaces decides alike (same shape, same options).
  int sg;
  // > 0 ("crossing", den_lazy.inc.h: XF; pdf-by-state plans, calls of the denominator alone): each recursion emits the occupancies of
  // its own second half itself - alpha the frames from (middle + xf) on, beta those below (middle - xf), per time segment - and the
  // occupancy launch of the call handles only the band of 2 xf frames around every middle (and the padding); a sequence or segment
  // shorter than 4 xf + 8 frames is the occupancy launch's whole.  xprog is then the progress table of the crossing
  // ([2][B][kMaxTimeSegs]: steps whose rows are out), which the peer direction's first landing waits for.
  int xf;
  // Per-frame totals.  The recursions divide a total out of every frame (alpha: tot(t) = sum_i a(t,i); beta: its
  // own n(t)) and only STORE it; den_finish_kernel turns the stored totals into the log-probability
  //     objf = sum_t log tot_a(t) + log fin_dot        (ComputeTotLogLike, chain-computation.cc:209-230)
  // and runs the reference's invariant check (BetaGeneralFrameDebug, :345-391: alpha'.beta' and the frame's
  // derivative sum within 5 % of 1, at t == 0 always and on every frame when verbose >= 1), restated for
  // per-frame scales that are free: frame t's un-normalised occupancy total G(t) obeys
  //     log G(t) + [log-scale divided out of the alpha row t] + [... of the beta row t+1] = objf     for every t.
  // No log, no fp64 and no serial sum sits in a recursion's frame loop.
  float* tot_a;              // [B,T+2]  alpha: tot(t), t = 0 .. L (lazy rows: t < L)
  float* tot_b;              // [B,T+2]  beta:  n(t),   t = L .. 1
  float* fin_dot;            // [2][B]   [0]: sum_i alpha'(L,i) final(i) in the scale of the last alpha row (NaN: a NaN network output was seen)
  float* gtot;               // [B,T]  G(t) of the frames to check, written by the occupancy kernels
  int check;                 // the occupancy launches of this call record G(t) and den_finish_kernel checks
  int check_all;             // 0: frame 0 only; 1: every frame (verbose level >= 1)
  int B, T, D, H, Hp;
  int input_is_exp;
  int frames_per_block;      // gamma kernel: frames one workgroup handles
  int phase_mask;            // bit0 recursion launch, bit1 gamma launch (bench aid)
  // Time segmentation (the gated schedule: overlap of the occupancy pass with the recursions): the
  // occupancy launch number gam_seg (of gam_nseg) handles exactly the frames whose alpha' and beta rows exist once every
  // recursion has done its steps below seg_bound[gam_seg].  gam_nseg = 0: all.
  int gam_seg, gam_nseg;
  int seg_bound[16];         // recursion segment s covers steps [seg_bound[s-1], seg_bound[s]) (seg_bound[-1] = 0)
  // Streamed occupancy pass (`stream` = 1): the recursion workgroup of (direction, sequence) publishes in
  // seq_progress[dir * B + b] how many of its rows are complete and visible device-wide - alpha rows 0 .. P-1, beta rows
  // L .. L-P+1 - every kStreamWidth steps and when it ends; ONE persistent occupancy launch takes the frames in rings of
  // kStreamWidth by the step count that makes them computable (max(t, L-1-t)) from the queue counter stream_next and
  // waits for exactly the two counters a ring needs.  Frame t needs alpha row t and beta row t+1: Pa >= t+1, Pb >= L-t.
  int32_t* seq_progress;     // [2][B], zeroed before the recursion launch
  int32_t* stream_next;      // [1], zeroed before the recursion launch
  int stream;                // bit 0: the recursions report progress; bit 1: this occupancy launch is the streamed one
  int stream_blocks;         // its grid (the CU count: the workgroups that do not fit beside the recursions start as those end)
  // Progress signalling (the gated schedule): a recursion workgroup adds 1 to progress[s] once its steps
  // [0, seg_bound[s]) are done and their rows are visible device-wide, s < sig_n; the occupancy launch of
  // segment s is released by den_gate_kernel when progress[s] reaches 2B.  sig_n = 0: no signalling.
  int32_t* progress;         // [16], zeroed before the recursion launch
  int sig_n;
  float coef, grad_scale;
  const float* grad_scale_dev;   // optional device scalar multiplied into grad_scale (upstream autograd gradient)
  // Numerator fold (fused ChainLoss, two-frame occupancy kernel only): grad += fold_scale * occupancy of the
  // numerator, read from compact rows over the sequence's distinct pdf-ids (num_kernels.h).  Null = no fold.
  const float* fold_rows;        // [B,T,fold_K]
  const int32_t* fold_upd;       // [B,fold_K]
  const int32_t* fold_ucount;    // [B]
  int fold_K;
  float fold_scale;
  // Step totals (include/pychain_hip.h: loss_out): the den_finish_kernel workgroup that finishes LAST adds up what the host
  // framework would otherwise compute in half a dozen launch-bound scalar kernels behind it (sums of the per-sequence
  // objectives, the loss arithmetic of pychain/loss.py:100-104, frame and bad counts).  Null = not wanted.
  float* loss_out;               // [8]: loss, frames, bad total, sum den - sum num (unscaled), loss again, 0, 0, 0
  const float* loss_num_objf;    // [B] numerator objectives (written by an earlier launch), or null: denominator only
  float loss_scale;              // loss = (sum den - sum num) * loss_scale [/ *loss_norm_dev]
  const float* loss_norm_dev;    // optional device scalar (the frame count of avg = True when the lengths live on the device)
  int32_t* finish_count;         // [1], zeroed with the progress counters
  int bad_words;                 // bad[0 .. bad_words) are added up into loss_out[2]
  // den_finish_kernel launched BEHIND THE RECURSION on the caller's stream while the streamed occupancy launch is still running
  // on its own: it waits (one thread per workgroup, s_sleep) until occ_done - which every workgroup of that launch counts itself
  // into when the queue is empty and its stores are acknowledged - reaches occ_done_target, instead of for the launch's stream
  // event: the wake-up of a queue that waits for another queue's event costs 12-19 us, and the kernel's own 9 us follow it.
  // (The caller's stream still waits for the event behind it: what comes next needs the gradient.)  0: no waiting.
  int32_t* occ_done;             // [1], zeroed with the progress counters
  int occ_done_target;
  // Rows exp'd ahead of the recursions (den_exp_rows_kernel, run_den_launches): ex[b,t,:] = exp(clamp(x[b,t,:])), written from
  // both ends of every sequence towards its middle (end 0: rows 0 .., end 1: rows L-1 .. downwards) in rounds of ex_nr rows by
  // ex_q workgroups per (sequence, end) on the side stream while the recursions run; xprog[(end * B + b) * kExMaxQ + q] = rounds
  // workgroup q (which takes the rounds q, q + ex_q, ...) has complete and visible device-wide; xnan[b] != 0: a NaN was seen.  The lazy recursions then bring their rows in ready to gather - no pass over the row in LDS, which
  // costs a frame of C3 216 of its ~3500 LDS-busy cycles and a frame of C4 twice that (recursion -4 % / -12 %).  Null: the
  // recursions clamp / exp their rows themselves.
  float* ex;                     // [B,T,D] (workspace)
  int32_t* xprog;                // [2][B][kExMaxQ], zeroed with the progress counters
  int ex_nr, ex_q;               // rows per round, workgroups per end
  int32_t* xnan;                 // [B], zeroed with the progress counters
  int use_ex;                    // this launch reads ex / xprog / xnan
  // Time segments (round 5; den_lazy.inc.h: lazy_recursion): with few sequences the T dependent frames of a
  // (sequence, direction) are the whole step and most CUs idle.  tseg = S > 1: the recursion grid is 2 B S workgroups;
  // workgroup (b, direction, k) produces the rows of time segment k of sequence b, started `tburn` frames outside the segment
  // from the ordinary start vector - a forward / backward filter forgets where it started (profiles/r05_forgetting_table.md:
  // 1e-7 after 192 frames on the benchmark graphs) - with the rows of its burn-in discarded, except the one next to the segment:
  // that one goes to splice[((b * 2 + dir) * kMaxTimeSegs + k) * 2 * Hp] and den_splice_check_kernel compares it with the TRUE row
  // the neighbouring segment stored; any mismatch beyond 4e-6 sets *redo (and counts into respec), and the recursion launch that
  // follows - the ordinary one, launched with redo_if - runs only then.  Sequences shorter than 2 tburn frames run as one segment.
  int tseg, tburn;
  // The burn-in controller of a plan (include/pychain_hip.h: pychain_hip_den_tseg_state): how long a recursion needs
  // to forget where it started depends on the DATA, and a call whose speculated rows do not verify runs its recursions twice.  A
  // caller-owned device blob (int32[16], zeroed once) attached to the plan carries, from call to call IN STREAM ORDER - no host
  // read, no host timing: the decision of call n is a function of the calls before it on the stream - [0] magic, [1] the burn-in
  // in effect, [2] calls left of a cool-down during which the plan is not cut (the segmented launch leaves at once, the uncut
  // launch behind it does the work), [3] calls seen, [4] calls that missed.  Every kernel of a call reads it through
  // den_tburn() / den_tseg_off(); den_finish_kernel's last thread updates it at the END of the call: a miss lengthens the burn-in
  // by half while three of them fit the sequence, else the plan cools down for kTsegCooldown calls.  Null: tburn as given.
  int32_t* tstate;
  float* splice;                 // [B][2][kMaxTimeSegs][2][Hp]: the speculated row next to a segment; a row nobody reads (workspace)
  int32_t* redo;                 // [4]: [0] != 0: a splice did not verify; [1]: how many; [2]: the worst mismatch as float bits (reported); [3]: what the segmented launch counted into `bad` - merged by den_finish_kernel only if [0] == 0 (zeroed with the progress counters)
  int redo_if;                   // this recursion launch is the fallback: its workgroups leave at once unless *redo != 0
  CallKnobs knobs;               // this call's snapshot of the library settings (host side only)
};

constexpr int kTsegMagic = 0x74736567, kTsegCooldown = 500, kTsegStateWords = 16;
// the burn-in in effect for this call / whether the plan is cooling down (DenArgs::tstate; the same answer in every kernel of a
// call: the state changes only at its end)
__device__ __forceinline__ int den_tburn(const DenArgs& a) {
  if (!a.tstate) return a.tburn;
  const int m = __hip_atomic_load(a.tstate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int b = __hip_atomic_load(a.tstate + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (m == kTsegMagic && b >= 1) ? b : a.tburn;
}
__device__ __forceinline__ bool den_tseg_off(const DenArgs& a) {
  if (!a.tstate) return false;
  return __hip_atomic_load(a.tstate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kTsegMagic &&
         __hip_atomic_load(a.tstate + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
}

// true if the recursion of this call runs as den_recursion_lazy_kernel (decided once per call; the occupancy
// launches - also those of a later chain_loss_backward on the same workspace - must be told: DenArgs::lazy)
bool den_lazy_eligible(const DenArgs& a, int resident_slot_rows);
// ... with LDS-direct rows (16 waves), and in four-wave workgroups (DenArgs::shape)
bool den_dma_eligible(const DenArgs& a, int resident_slot_rows);
bool den_small_eligible(const DenArgs& a, int resident_slot_rows);
// true if the occupancy pass of this call can run as ONE persistent launch beside the recursion (DenArgs::stream): the
// recursion kernel of the call reports per-sequence progress (lazy and pair forms), one plan for all sequences
bool den_stream_eligible(const DenArgs& a, int gamma_max_groups, int resident_slot_rows);
// names of the kernels launch_den would run for this call: recursion, occupancy (measurement tools and the
// kernel-selection test label by them)
const char* den_recursion_kernel_name(const DenArgs& a, int resident_slot_rows);
const char* den_occupancy_kernel_name(const DenArgs& a, int gamma_max_groups, int resident_slot_rows);
// ... as den_recursion_pair_kernel (DenArgs::pair); den_pair_blocks: its grid = what a progress counter reaches
bool den_pair_eligible(const DenArgs& a, int resident_slot_rows);
// ... in the one-gather form of a "pdf by state" plan (launch hint bit 27; a.shape == kShapeDma, a.use_ex / a.x_half decided)
bool den_sg_eligible(const DenArgs& a, int resident_slot_rows);
// ... with the crossing (DenArgs::xf); the band half-width it uses
bool den_xf_eligible(const DenArgs& a, int resident_slot_rows);
bool den_q_eligible(const DenArgs& a, int resident_slot_rows);    // one-word state vectors (option den_q; den_lazy.inc.h: MAP::kQ)
int den_xf_band();
// is frame t of a length-L sequence one the occupancy launch of a crossing call evaluates (DenArgs::xf)?  `cut`: the call's rows
// come from its time segments (else from the uncut recursion - also after a splice miss)
__host__ __device__ inline bool den_xf_band_frame(int t, int L, int nseg, int w) {
  int k = (int)(((long)t * nseg) / L);
  if (k >= nseg) k = nseg - 1;
  while (k > 0 && t < (int)(((long)k * L) / nseg)) k--;
  while (k + 1 < nseg && t >= (int)(((long)(k + 1) * L) / nseg)) k++;
  const int s = (int)(((long)k * L) / nseg), e = k + 1 == nseg ? L : (int)(((long)(k + 1) * L) / nseg);
  if (e - s < 4 * w + 8) return true;
  const int mid = (s + e) >> 1;
  return t >= mid - w && t < mid + w;
}
int den_recursion_blocks(const DenArgs& a);
hipError_t launch_den_splice_check(const DenArgs& a, hipStream_t st);
bool den_occupancy_half_ok(const DenArgs& a, int gamma_max_groups, int resident_slot_rows);

// true if launch_den would run the two-frame occupancy kernel (the only one that can fold the numerator in)
bool den_uses_gamma2(const DenArgs& a, int gamma_max_groups, int resident_slot_rows);

// the recursion launch of a call by its kernel family (den_lazy.hip: lazy / pair by a.pair, a.shape; den_rec.hip: the two-
// barrier kernel): called by launch_den
hipError_t launch_den_lazy_family(const DenArgs& a, int hint, hipStream_t st);
hipError_t launch_den_rec2b(const DenArgs& a, int hint, size_t lds_rec, hipStream_t st);

// Enqueues the two launches on `st`.  On failure returns the HIP error and, when the
// shape is unsupported, a reason in *why.
hipError_t launch_den(const DenArgs& a, int gamma_max_groups, int resident_slot_rows, hipStream_t st,
                      const char** why);

// test hook behind pychain_hip_debug_launch_map (host code only)
int den_debug_launch_map(int T, int L, int t, int frames_per_block, int nseg, const int* seg_bound, int seg,
                         int* out, int out_len);

// The kernels for plans in the general format (den_general.hip): the launches a.phase_mask selects, on `st`.
hipError_t launch_den_general(const DenArgs& a, hipStream_t st);

// ex = exp(clamp(x)) from both ends of every sequence towards the middle (DenArgs::ex)
constexpr int kExMaxQ = 4;
void den_exp_rows_shape(const DenArgs& a, int cus, int* nr, int* q);
hipError_t launch_den_exp_rows(const DenArgs& a, hipStream_t st);

// After the last launch of a call: objf from the stored totals + the invariant check (DenArgs::tot_a).
// One workgroup per sequence.
hipError_t launch_den_finish(const DenArgs& a, hipStream_t st);

// Rings of the streamed occupancy pass by the step count `need` = max(t, L-1-t) that makes a frame computable:
// kStreamWidth frames wide (= steps between two progress reports of a recursion) up to kStreamFineSpan steps before the end
// of the longest possible sequence (T), kStreamFineWidth from there on, kStreamLastWidth - ONE pair of frames of the two-frame
// kernel - over the last kStreamLastSpan steps: what is left to evaluate when the recursions end is the last ring of the
// longest sequences, so those rings are thin (C2: the occupancy launch ends 25 -> 18 us after the recursion).
constexpr int kStreamWidth = 16, kStreamFineWidth = 4, kStreamFineSpan = 64, kStreamLastWidth = 2, kStreamLastSpan = 8;
__host__ __device__ inline int stream_fine_begin(int T) { const int f = (T - kStreamFineSpan) / kStreamWidth * kStreamWidth; return f > 0 ? f : 0; }
__host__ __device__ inline int stream_last_begin(int T) {      // a multiple of kStreamFineWidth, >= stream_fine_begin(T)
  const int f = stream_fine_begin(T), l = (T - kStreamLastSpan) / kStreamFineWidth * kStreamFineWidth;
  return l > f ? l : f;
}
__host__ __device__ inline int stream_ring_count(int T) {
  const int f = stream_fine_begin(T), l = stream_last_begin(T);
  return f / kStreamWidth + (l - f) / kStreamFineWidth + (T - l + kStreamLastWidth - 1) / kStreamLastWidth;
}
// ring r covers need in [lo, hi)
__host__ __device__ inline void stream_ring(int T, int r, int& lo, int& hi) {
  const int f = stream_fine_begin(T), l = stream_last_begin(T), nc = f / kStreamWidth, nf = (l - f) / kStreamFineWidth;
  if (r < nc) { lo = r * kStreamWidth; hi = lo + kStreamWidth; }
  else if (r < nc + nf) { lo = f + (r - nc) * kStreamFineWidth; hi = lo + kStreamFineWidth; }
  else { lo = l + (r - nc - nf) * kStreamLastWidth; hi = lo + kStreamLastWidth; }
}
// a recursion reports after `done` steps if that is a ring boundary
__host__ __device__ inline bool stream_report_due(int T, int done) {
  if (done < stream_fine_begin(T)) return (done & (kStreamWidth - 1)) == 0;
  if (done < stream_last_begin(T)) return (done & (kStreamFineWidth - 1)) == 0;
  return (done & (kStreamLastWidth - 1)) == 0;
}

// One wave that waits until *progress >= target (set by the recursion workgroups), so that what follows
// it in stream order starts then; gives up after ~20 s and counts that in *bad.
hipError_t launch_den_gate(const int32_t* progress, int target, int32_t* bad, hipStream_t st);

}  // namespace pychain_hip
#endif
