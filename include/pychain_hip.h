/*
 * pychain_hip.h - C ABI of libpychain_hip.so, the MI355X (gfx950) LF-MMI
 * forward-backward.  This library replaces the reference's native extension
 * `pychain_C` (pytorch_binding/src, built by pytorch_binding/setup.py:4-14)
 * wholesale.  Plain pointers and sizes only: no torch / pybind types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative PYCHAIN_HIP_E* code on
 *     error; pychain_hip_last_error() then holds a message (thread-local);
 *   - "dev" pointers are device memory, "host" pointers host memory;
 *   - all work is ordered on the caller's stream (`void* stream` is a hipStream_t):
 *     a call may fork launches to two library-owned non-blocking side streams
 *     (created once per device on first use - with their events the only hidden
 *     state of the library) and joins them back with events before it returns, so
 *     stream-ordered use is unchanged.  The denominator does this for T >= 256: its
 *     time-parallel occupancy launches run beside the (single) recursion launch,
 *     each released by a one-wave gate kernel that polls a progress counter the
 *     recursion workgroups advance.  Nothing synchronises with the
 *     host and nothing allocates: scratch memory is a caller-provided workspace
 *     sized by the *_workspace_bytes queries;
 *   - float = IEEE fp32, indices int32, lengths int64 (the reference's dtypes,
 *     openfst_binding/src/fstext.cc:81-104, pychain/loss.py:41).
 *
 * Reference boundary being replaced (pytorch_binding/src/pychain.cc):
 *   pychain_C.forward_backward            :26-79   -> pychain_hip_den_*  (probability domain, leaky-HMM)
 *   pychain_C.forward_backward_log_domain :81-129  -> pychain_hip_num_*  (log domain)
 *   pychain_C.set_verbose_level           :134     -> pychain_hip_set_verbose_level
 */
#ifndef PYCHAIN_HIP_H_
#define PYCHAIN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYCHAIN_HIP_ABI_VERSION 17

/* Element type of the network output [B,T,D] - and of the gradient an entry point writes for it (ABI 14).
 * 2-byte rows are read as they are by the kernels and converted where they land; the gradient is rounded (to nearest even)
 * to the same type where it is written: no up-cast pass, no fp32 copy of [B,T,D], no cast of the gradient back.  All
 * arithmetic is fp32 (fp64 for the numerator's log-probabilities) whatever the storage type.  Not every kernel family takes
 * 2-byte rows: ask pychain_hip_*_half_native first (a call that cannot returns PYCHAIN_HIP_EUNSUPPORTED; up-cast then). */
#define PYCHAIN_HIP_F32  0
#define PYCHAIN_HIP_BF16 1
#define PYCHAIN_HIP_F16  2

#define PYCHAIN_HIP_OK            0
#define PYCHAIN_HIP_EINVAL      (-1)  /* bad argument (null pointer, size mismatch, index out of range) */
#define PYCHAIN_HIP_EUNSUPPORTED (-2) /* shape outside what the kernels were built for */
#define PYCHAIN_HIP_EWORKSPACE  (-3)  /* workspace / blob too small */
#define PYCHAIN_HIP_ELAUNCH     (-4)  /* HIP runtime reported a launch error */

int         pychain_hip_abi_version(void);
const char* pychain_hip_last_error(void);
/* base.h:34-42 / pychain.cc:134.  `ok` (here: bad_count == 0) carries the reference's invariant checks
 * (chain-computation.cc:345-391, chain-log-domain-computation.cc:283-304): alpha'.beta' and the frame's
 * derivative sum within 5 % of 1, restated for this library's free per-frame scales as
 *   log G(t) + la[t] + lb[t+2] = log P(sequence)   (denominator; G = un-normalised occupancy total of frame t,
 *   la / lb = log-scales the two recursions have divided out), sum of a frame's occupancies = 1 (numerator),
 * plus: no normaliser or total is non-finite or non-positive.  Level 0 checks frame 0 of every sequence (as the
 * reference does), level >= 1 every frame; at level >= 1 the occupancy launches of the denominator are not
 * overlapped with the recursions (the check needs both finished), results are bit-identical. */
void        pychain_hip_set_verbose_level(int level);
int         pychain_hip_get_verbose_level(void);

/* Test hook (host only, no GPU): how the occupancy launch of time segment `seg` (of `nseg`, ends
 * seg_bound[]) maps workgroups to 32-frame chunks of a length-L sequence.
 * out[0] = grid.x, out[1 + k] = chunk of workgroup k or -1; returns 1 if frame t belongs to the
 * launch, 0 if not, negative on bad arguments.  nseg = 0: the unsegmented launch. */
int         pychain_hip_debug_launch_map(int T, int L, int t, int frames_per_block, int nseg,
                                         const int32_t* seg_bound, int seg, int32_t* out, int out_len);
/* Test hook (host only, no GPU): the rings of the STREAMED occupancy pass for sequences of up to T frames:
 * ring r covers the frames whose step count need = max(t, L-1-t) lies in [out[2r], out[2r+1]); returns the number of rings
 * (pairs beyond out_len / 2 are not written).  report_due (may be NULL): report_due[d] = 1 if a recursion workgroup reports its
 * progress after d steps, d < due_len. */
int         pychain_hip_debug_stream_rings(int T, int32_t* out, int out_len, int32_t* report_due, int due_len);
/* Test hook: `workgroups` workgroups of 1024 threads and 100 KB of LDS (one to a CU) that sleep for `microseconds` on `stream` -
 * a stand-in for the long-lived kernels of an overlapped RCCL all-reduce beside a loss step (tests/test_gpu_robust.py). */
int         pychain_hip_debug_occupy(int workgroups, int microseconds, void* stream);
/* Measurement aid (bench.py): restrict pychain_hip_den_forward_backward to a subset of
 * its launches so each kernel can be bracketed by events on the caller's stream.
 * bit 0 = alpha/beta recursion launch, bit 1 = occupancy launch; default 3 = both.
 * With a partial mask the outputs of the call are NOT meaningful. */
void        pychain_hip_set_den_phase_mask(int mask);
/* Test hook: 0 = the denominator recursions always run as den_recursion_kernel (rows normalised in a
 * pass of their own, two barriers per frame); 1 (default) = as den_recursion_lazy_kernel wherever the
 * shape allows.  Both forms must agree to rounding; the tests compare them. */
void        pychain_hip_set_den_lazy(int on);
/* Which kernels a denominator call with this plan hint (pychain_hip_den_plan_info: info[4]), these sizes and the
 * calling thread's current options would launch: "recursion,occupancy" into buf, e.g.
 * "den_recursion_lazy_kernel<dma>,den_gamma2_kernel".  Recursion: den_recursion_lazy_kernel<small> (4 waves: plans that
 * hold the four-wave dealing), den_recursion_lazy_kernel<dma> (16 waves, nnet-output rows of up to 9216 pdfs by LDS-direct
 * loads), den_recursion_lazy_kernel (16 waves, rows through registers: option den_dma = 0), den_recursion_pair_kernel (two
 * sequences per workgroup, shared plan, B >= 3/8 of the CU count), den_recursion_kernel (everything else),
 * den_general_recursion_kernel (plans in the general format).  Measurement tools and the kernel-selection test label by it.
 * plans_shared: bit 0 = one plan for all sequences (plan stride 0); bit 1 = the call is a fused loss (a numerator runs beside
 * it: two sequences per recursion workgroup from B >= 100 on 256 CUs; the denominator alone: once 2 B workgroups no longer fit). */
int         pychain_hip_den_kernel_names(int resident_slot_rows, int num_states, int num_pdfs, int B, int plans_shared,
                                         char* buf, size_t buf_bytes);
/* Settings.  A call reads them ONCE when it starts (process-wide defaults overlaid with the calling thread's
 * overrides), so another thread changing one cannot tear a call in flight, and nothing on the call path reads the
 * environment.  pychain_hip_set_option sets the process-wide default (value NULL or "" clears it);
 * pychain_hip_set_thread_option overrides it for the calling host thread only (value NULL removes the override,
 * "" = unset for this thread); pychain_hip_get_option copies the value in effect for the calling thread into buf and
 * returns its length (0 = unset).  pychain_hip_set_verbose_level / _set_den_phase_mask / _set_den_lazy are the
 * process-wide "verbose" / "den_phase_mask" / "den_lazy".  The thirteen names - every one selects between SHIPPED kernel
 * families so that the tests can compare them (all give the same results to rounding; most bit for bit), or is a test hook:
 *   "verbose"        base.h:34-42: >= 1 checks the reference's invariant on every frame instead of frame 0
 *   "den_phase_mask" bit 0 recursion launch, bit 1 occupancy launch (measurement aid)
 *   "den_lazy"       "0": the two-barrier recursion (den_recursion_kernel) instead of the lazy-normalisation one
 *   "den_dma"        "0": nnet-output rows of the lazy recursion through registers instead of LDS-direct loads; "2": LDS-direct,
 *                    and the recursions clamp / exp every row themselves instead of gathering from rows den_exp_rows_kernel
 *                    exp'd ahead of them on the side stream (bit-identical); "3": rows exp'd ahead wherever the shape allows - by
 *                    default only four-wave workgroups take them
 *   "den_segments"   n >= 1: the occupancy pass in n gated time segments (1 = after the recursions, no overlap) instead of
 *                    the streamed persistent launch
 *   "den_pair"       "1": two sequences per recursion workgroup wherever the shape allows, "0": never; default: a fused loss from 25/64
 *                    of the CU count in sequences on (B >= 100 on 256 CUs), the denominator alone once 2 B one-sequence
 *                    workgroups no longer fit the chip (B > 128) - bit-identical to den_recursion_kernel
 *   "gamma16"        the one-frame occupancy kernel (and with it the numerator accumulated into the stored gradient
 *                    instead of folded into the occupancy launch) also where the two-frame kernel fits
 *   "debug_corrupt_row" "den,b,t,scale" / "num,b,t,scale": the stored alpha row t of sequence b is scaled between the
 *                    recursions and the occupancy pass, so that the 5 % invariant of chain-computation.cc:363-390 /
 *                    chain-log-domain-computation.cc:289-303 can be seen to fire: `ok` false at t = 0, at any t at verbose >= 1
 *   "num_compat"     "1": the numerator in the reference's own fp32 arithmetic (num_compat.hip) instead of the exact path
 *   "den_tseg"       time segments per (sequence, direction) of the lazy recursions: unset / "-1" automatic (few sequences
 *                    only: pychain_hip_den_time_segments), "0": never, "2" / "4": wherever the shape allows
 *   "den_tburn"      frames a time segment starts outside itself (default 192); see totals[5..7] and pychain_hip_den_tseg_state
 *   "den_sg"         "0": a "pdf by state" plan (every arc entering a state carries one pdf: pychain_hip_den_plan_info, hint bit 27) runs
 *                    the recursions every plan runs instead of their one-gather form (the tests compare the two)
 *   "den_q"          "1": the lazy recursions keep ONE-WORD state vectors where the plan allows them (every leaky probability
 *                    positive, one position per state: pychain_hip_den_plan_info, hint bit 19; fp32 rows, uncut sequences, loops of
 *                    17 .. 32 slot-rows): alpha gathers a / (coef leaky), both directions two ds_read_b32 per arc instead of a
 *                    ds_read_b64 and a ds_read_b32.  Off by default: a third fewer LDS bytes per gather pair, same results to 3e-7 - and
 *                    measured 6 % slower with rows the recursions clamp / exp themselves, 7 % faster with rows exp'd ahead (the
 *                    LDS pipe is as busy as before - a b64 gather occupies it about as long as a b32 one - and the group ends add 12 %
 *                    VALU: DESIGN.md 3.16, profiles/r06_one_word_states.txt)
 *   "den_cross"      "1": the recursions of a pdf-by-state plan emit occupancies themselves (calls of the denominator alone with at
 *                    most one workgroup per CU: each direction emits those of its own second half, the occupancy launch handles the
 *                    band around every middle).  Off by default: same results to 3e-7, but measured slower than the streamed
 *                    occupancy launch it replaces (profiles/r06_crossing.txt).  Each direction waits for rows of
 *                    the other, so every workgroup of the launch must be resident: not beside other work on the same GPU (a
 *                    workgroup that waits 20 s for its peer gives up and the call reports `ok` false)
 *   "chain_slices"   the fused loss with a gradient over a batch larger than the chip: "0" one call, "n" n slices; default automatic
 *   "plan_split"     read by pychain_hip_den_plan_build: "0": no state on more than one lane; default: where it gains a
 *                    shorter register-resident loop
 * Unknown name: EINVAL.  (Kernel variants that measured slower - 8 / 12 waves, two copies of the nnet-output row, a
 * recursion relaunched per segment - are not in the library; their measurements are under profiles/r03_*.) */
int         pychain_hip_set_option(const char* name, const char* value);
int         pychain_hip_set_thread_option(const char* name, const char* value);
int         pychain_hip_get_option(const char* name, char* buf, size_t buf_bytes);

/* ------------------------------------------------------------------------
 * Denominator graph plan  (host side, no GPU work).
 *
 * Compiles ONE probability-domain graph, given in the reference layout
 * (pychain/graph.py:36-44; fstext.cc:49-116), into the device format the
 * kernels consume: three degree-sorted, wave-tiled arc orderings
 *   by destination state (alpha recursion; = reference backward_transitions),
 *   by source state      (beta recursion;  = reference forward_transitions),
 *   by pdf-id            (occupancy pass;  replaces the reference's atomicAdd
 *                         scatter, chain-kernels.cu:53-87,230-240)
 * plus the permuted leaky / initial / final vectors.
 *
 * States on several lanes: a row group of a recursion is as long as its longest row, so one state with many arcs
 * entering (leaving) it would set the loop length of the whole workgroup.  Where it gains a shorter register-resident loop
 * the compiler puts such a state on several POSITIONS of a side's numbering, each collecting a share of its arcs; the
 * arcs that gather the state are repeated once per position (csrc/plan.cpp).  The plan then has more positions than the
 * graph has states: pychain_hip_den_plan_info reports the position count as info[0] - THE VALUE TO PASS AS num_states
 * to the workspace and forward-backward calls - and the graph's own count as info[6].  Results are the same to rounding.
 * Plans passed to one call with a non-zero stride must agree in info[0]: build them with the option "plan_split" = "0"
 * (no state on more than one lane).
 *
 * Graphs the tile kernels do not take - more than 65 535 states or pdfs (packed 16-bit arc addresses), or a state
 * vector + nnet-output row beyond the 160 KiB LDS of one CU - are compiled into a GENERAL format instead (the reference
 * layout plus the arcs grouped by pdf-id); pychain_hip_den_plan_info then reports the launch hint
 * PYCHAIN_HIP_HINT_GENERAL and the denominator runs on kernels that gather from global memory: slow, but complete,
 * like the reference's CPU path (chain-computation.cc:113-176,247-311 has no size limit).
 *
 * Returns the number of bytes the plan needs (> 0).  If `blob` is NULL or
 * `blob_bytes` is too small nothing is written (call once to size, once to
 * fill).  The filled blob is position independent: copy it to the device with
 * any allocator and pass the device address to pychain_hip_den_forward_backward.
 * Negative return = error.
 */
int64_t pychain_hip_den_plan_build(
    const int32_t* forward_transitions,        /* host [K,3] (src,dst,pdf)          */
    const int32_t* forward_transition_indices, /* host [H,2] (begin,end) by source   */
    const float*   forward_transition_probs,   /* host [K]   probabilities           */
    const int32_t* backward_transitions,       /* host [K,3] grouped by destination  */
    const int32_t* backward_transition_indices,/* host [H,2]                         */
    const float*   backward_transition_probs,  /* host [K]                           */
    const float*   leaky_probs,                /* host [H]                           */
    const float*   initial_probs,              /* host [H]                           */
    const float*   final_probs,                /* host [H]                           */
    int num_states, int num_transitions, int num_pdfs,
    void* blob, size_t blob_bytes);

/* Facts about a filled HOST blob that the launcher needs (the blob itself lives on the
 * device at call time and is never read back):
 *   info[0] num_states (positions: see "States on several lanes")  info[1] num_transitions  info[2] num_pdfs  info[3] plan bytes, low 31 bits (info[5]: the rest)
 *   info[4] launch hint: slot-rows per wave (= arcs a wave keeps in registers), three
 *           fields: recursion plans (bits 0-9), occupancy plan for 16 waves (10-18; ABI 17: nine bits)
 *           and for 8 waves (20-27); combine several plans by taking the max of each field
 *           bit 19: (ABI 17) every state sits on one position of either numbering and every leaky probability is positive: the
 *                   lazy recursions may keep ONE-WORD state vectors (alpha gathers a / (coef leaky); option "den_q") (AND)
 *           bit 28: a state sits on several positions of the beta numbering: the lazy recursions' NC form; not for
 *                   den_recursion_pair_kernel (OR)
 *           bit 29: the plan also holds its recursion tiles dealt to FOUR waves (small graphs: 256-thread recursion
 *                   workgroups); the recursion field is then that dealing's row count (AND over several plans)
 *           bit 30: every recursion wave owns few enough row groups for the lazy-normalisation recursions (AND)
 *   info[5] plan bytes >> 31 (general-format plans may exceed 2 GiB)   info[6] the graph's states, info[7] positions
 *   added by states on several lanes (0: none)
 * A blob whose header or payload does not match the checksums in its header, or whose header points outside the blob
 * (a damaged or foreign cache file), is EINVAL.
 */
int pychain_hip_den_plan_info(const void* host_blob, size_t blob_bytes, int32_t info[8]);

/* ------------------------------------------------------------------------
 * Denominator forward-backward on the GPU (replaces pychain.cc:26-79 and all
 * of chain-computation.cc / chain-kernels.cu).
 *
 * plans_dev/plan_stride_bytes: device address of the first plan and the byte
 *   distance between the plans of consecutive sequences; 0 = every sequence
 *   shares one graph (the ChainLoss denominator, pychain/loss.py:99).
 * resident_slot_rows: the launch hint info[4] of pychain_hip_den_plan_info (per-field max
 *   over the passed plans); selects the kernel variant that keeps every wave's arcs
 *   in registers for the whole launch.  0 = unknown (arcs are re-read from L2 each frame;
 *   same results, slower).
 * nnet_output: dev [B,T,D].  input_is_exp = 0: raw network output, clamp(-30,30)
 *   and exp are fused into the kernels (pychain/loss.py:30,43);
 *   input_is_exp = 1: already exp(clamp(.)) as pychain_C.forward_backward gets it.
 * seq_lengths: dev [B] int64, 1 <= len <= T.  Any order (the reference needs
 *   them sorted descending for pack_padded_sequence, loss.py:37-40).
 * objf_per_seq: dev [B], log-probability of each sequence (reference returns
 *   their sum, chain-computation.cc:229).
 * grad: dev [B,T,D], d objf / d nnet_output, every element written (rows
 *   t >= len are zero), multiplied by grad_scale.
 * bad_count: dev int32[1]; zeroed by the call, incremented by every frame or
 *   sequence whose normaliser is not finite-positive (the reference's `ok`
 *   flag, chain-computation.cc:367-390, without a host sync).
 * totals: dev float[PYCHAIN_HIP_TOTALS = 8] or NULL.  The call's last kernel adds up what a caller otherwise computes in several
 *   launch-bound scalar kernels behind it (the reference: `tot_log_prob.sum()`, chain-computation.cc:229):
 *   totals[0] = totals[3] = sum_b objf_per_seq[b] (fp64 accumulation, rounded once), totals[1] = sum_b len_b,
 *   totals[2] = bad_count as a float; totals[4] = totals[0] once more (ABI 14: the scalar a host framework hands out as the
 *   loss, apart from the statistics in [0..3] that it may all-reduce or keep); totals[5] = speculated rows of a time-segmented
 *   call that did not verify (0, or the call ran its recursions again unsegmented), totals[6] = its segments per (sequence,
 *   direction) (1: not segmented), totals[7] = the worst mismatch of a speculated row (max |p - q| / max p; the bound is 4e-6).
 * workspace: the stored alpha' / beta rows (4 B T roundup64(num_states) bytes each), per-frame totals, counters, and - in a
 *   call of the denominator alone - a [B,T,D] buffer for the rows exp'd ahead of the recursions (den_exp_rows_kernel: C4
 *   4.80 -> 4.57 ms): pychain_hip_den_workspace_bytes; pychain_hip_den_workspace_min_bytes is the size without it.
 */
#define PYCHAIN_HIP_TOTALS 8
size_t pychain_hip_den_workspace_bytes(int B, int T, int num_states, int num_pdfs);
/* ... without the [B,T,D] buffer: a workspace of at least this size is accepted everywhere; the rows are then never exp'd
 * ahead (the fused loss never does: its callers pass this size). */
size_t pychain_hip_den_workspace_min_bytes(int B, int T, int num_states, int num_pdfs);
/* 1 if pychain_hip_den_forward_backward with these arguments (and the calling thread's options) would use the [B,T,D] buffer
 * of the full workspace, else 0: a caller that caches its workspace asks before it allocates as much again as the network
 * output (1.3 GB at C3) for calls that never touch it (pair / general / two-barrier kernels, T < 64, exp'd input, a recursion
 * grid that leaves less than a quarter of the chip free). */
/* How many time segments the recursions of a (sequence, direction) of such a call are cut into (1: not cut; `fused`: the
 * call is a fused loss - a quarter of the chip stays with the numerator).  totals[5..7] of the call report how the cut went.
 * A function of the shape, the device's CU count and the options den_tseg / den_tburn only - NOT of the verbose level (a debug
 * run computes what production computes).  A cut call's results differ from the uncut call's within the fp32 noise of two
 * recursions (objf bit-equal, gradient ~2e-7), and which shapes are cut depends on B and the CU count: option den_tseg = "0" is
 * the ONE switch for training that must reproduce bit for bit across batch shapes and devices. */
int pychain_hip_den_time_segments(int64_t plan_stride_bytes, int resident_slot_rows, int num_states, int num_pdfs, int B, int T, int fused);
/* The burn-in controller of a plan (ABI 16).  How long a recursion needs to forget where it started depends on
 * the DATA (peaky network outputs forget slowly), and a call whose speculated rows do not verify runs its recursions twice.
 * state_dev: pychain_hip_den_tseg_state_bytes() (64) bytes of DEVICE memory the caller owns and has zeroed ONCE; attached to the
 * plan by address (NULL detaches; detach before freeing either).  Every time-segmented call on that plan then reads its burn-in
 * from the state and the call's last kernel updates it - in stream order, on the device: after a call that missed, the burn-in
 * is half as long again while three of them fit T; beyond that the plan is not cut for the next 500 calls (the segmented launch
 * leaves at once, the uncut launch behind it does the work) and starts over from the default.  No host read, no host timing:
 * the burn-in of call n is a function of the calls before it on the stream - the same run gives the same bits.  Callers that
 * pin "den_tseg" or "den_tburn" with an option bypass the state.  int32 words: [0] magic, [1] burn-in, [2] cool-down calls left,
 * [3] calls seen, [4] calls that missed.  One state per plan AND stream if calls on one plan overlap on several streams. */
int pychain_hip_den_tseg_state(const void* plans_dev, void* state_dev);
size_t pychain_hip_den_tseg_state_bytes(void);
int pychain_hip_den_uses_row_buffer(int64_t plan_stride_bytes, int resident_slot_rows, int num_states, int num_pdfs,
                                    int B, int T, int input_is_exp);
int pychain_hip_den_forward_backward(
    const void* plans_dev, int64_t plan_stride_bytes, int resident_slot_rows,
    int num_states, int num_pdfs,
    const void* nnet_output, int nnet_output_dtype, int input_is_exp, const int64_t* seq_lengths,
    int B, int T, float leaky_hmm_coefficient, float grad_scale,
    float* objf_per_seq, void* grad /* nnet_output_dtype */, int32_t* bad_count, float* totals,
    void* workspace, size_t workspace_bytes, void* stream);
/* 1 if pychain_hip_den_forward_backward takes 2-byte network outputs for this shape (and the calling thread's options): the
 * lazy recursions with LDS-direct rows or the pair recursion, an occupancy kernel in its float4-chunk forms, rows of a
 * multiple of 8 pdfs, a plan that is not in the general format. */
int pychain_hip_den_half_native(int64_t plan_stride_bytes, int resident_slot_rows, int num_states, int num_pdfs, int B, int T);

/* ------------------------------------------------------------------------
 * Numerator forward-backward, log domain, one graph per sequence (replaces
 * pychain.cc:81-129 and chain-log-domain-computation.cc / -kernels.cu).
 * Graph tensors are the batched, zero-padded ChainGraphBatch tensors
 * (pychain/graph.py:122-175) already on the device; graph_batch_stride = 1
 * for per-sequence graphs, 0 when all sequences share row 0.
 * nnet_output: dev [B,T,D] raw (clamped to [-30,30] in-kernel).
 * grad_mode:
 *   PYCHAIN_HIP_GRAD_LOG    grad[b,t,n] = log occupancy, -inf where zero (what
 *                           forward_backward_log_domain returns, :57,:265)
 *   PYCHAIN_HIP_GRAD_LINEAR grad[b,t,n] = grad_scale * occupancy (loss.py:77 fused)
 *   PYCHAIN_HIP_GRAD_ACCUM  grad[b,t,n] += grad_scale * occupancy on the touched
 *                           pdfs only (fused ChainLoss: pass grad_scale < 0 to
 *                           subtract the numerator from the denominator grad)
 */
#define PYCHAIN_HIP_HINT_GENERAL ((int)0x80000000)   /* launch hint of a plan in the general format (pychain_hip_den_plan_info) */
#define PYCHAIN_HIP_GRAD_LOG    0
#define PYCHAIN_HIP_GRAD_LINEAR 1
#define PYCHAIN_HIP_GRAD_ACCUM  2
/* pychain_hip_cpu_num_forward_backward only, OR-ed into grad_mode: the network output is taken AS IT IS - the contract of
 * pychain_C.forward_backward_log_domain, whose C++ does not clamp (chain-log-domain-computation.cc:137-145; the reference
 * clamps in Python, pychain/loss.py:30).  Without it the host twin applies ChainFunction's clamp(-30, 30) itself. */
#define PYCHAIN_HIP_CPU_NO_CLAMP 0x100
size_t pychain_hip_num_workspace_bytes(int B, int T, int num_states, int num_transitions, int num_pdfs);
int pychain_hip_num_forward_backward(
    const int32_t* forward_transitions,         /* dev [G,K,3] */
    const int32_t* forward_transition_indices,  /* dev [G,H,2] */
    const float*   forward_transition_probs,    /* dev [G,K] log-probs */
    const int32_t* backward_transitions,        /* dev [G,K,3] */
    const int32_t* backward_transition_indices, /* dev [G,H,2] */
    const float*   backward_transition_probs,   /* dev [G,K] */
    const float*   initial_probs,               /* dev [G,H] log */
    const float*   final_probs,                 /* dev [G,H] log */
    int graph_batch_stride,
    const void* nnet_output, int nnet_output_dtype, const int64_t* seq_lengths,
    int B, int T, int num_pdfs, int num_states, int num_transitions,
    int grad_mode, float grad_scale,
    float* objf_per_seq, float* grad /* fp32 whatever nnet_output_dtype */, int32_t* bad_count,
    void* workspace, size_t workspace_bytes, void* stream);
/* 1 if the numerator's recursions read 2-byte network outputs for this shape (the tile kernels' float4-chunk forms; not the
 * general kernels, not option num_compat).  The gradient of pychain_hip_num_forward_backward stays fp32. */
int pychain_hip_num_half_native(int num_states, int num_transitions, int num_pdfs);

/* ------------------------------------------------------------------------
 * Fused ChainLoss (replaces the two ChainFunction calls + the autograd add of
 * pychain/loss.py:97-105 with one pass):
 *
 *   grad[b,t,n] = grad_scale * (gamma_den(b,t,n) - gamma_num(b,t,n))      written ONCE
 *   den_objf[b], num_objf[b] = per-sequence log-probabilities
 *
 * so that loss = -(sum num_objf - sum den_objf) [* grad_scale when the caller folds the
 * 1/sum(lengths) of `avg=True` into grad_scale] and d loss / d nnet_output = grad.
 * Numerically identical to calling the two entry points above and subtracting.
 *
 * Scheduling: the numerator recursion is independent of the denominator until the final
 * subtraction, so it runs CONCURRENTLY on a library-owned non-blocking side stream
 * (fork/join with events on `stream`; created once per device on first use, the only
 * hidden state of this library); the join precedes the sparse subtraction.
 * bad_count: dev int32[2] = {denominator, numerator}.
 * Arguments as in pychain_hip_den_forward_backward / pychain_hip_num_forward_backward.
 */
int pychain_hip_chain_loss_forward_backward(
    /* denominator */
    const void* plans_dev, int64_t plan_stride_bytes, int resident_slot_rows, int den_num_states,
    float leaky_hmm_coefficient,
    /* numerator */
    const int32_t* forward_transitions, const int32_t* forward_transition_indices,
    const float* forward_transition_probs, const int32_t* backward_transitions,
    const int32_t* backward_transition_indices, const float* backward_transition_probs,
    const float* initial_probs, const float* final_probs, int graph_batch_stride,
    int num_num_states, int num_num_transitions,
    /* shared */
    const void* nnet_output, int nnet_output_dtype, const int64_t* seq_lengths, int B, int T, int num_pdfs, float grad_scale,
    float* den_objf_per_seq, float* num_objf_per_seq, void* grad /* nnet_output_dtype */, int32_t* bad_count,
    float loss_scale, const float* loss_norm_dev, float* totals,
    void* den_workspace, size_t den_workspace_bytes, void* num_workspace, size_t num_workspace_bytes,
    void* stream);

/* The same fused loss split at the autograd boundary (pychain/loss.py:27-87: forward returns
 * objf, backward returns input_grad * objf_grad):
 *
 *   _forward   runs the recursions (denominator alpha/beta + numerator alpha/beta, the latter on
 *              a side stream) and yields the per-sequence log-probabilities; the stored
 *              trajectories stay in the two workspaces, which the caller must keep untouched
 *              until _backward.  If `grad` is non-NULL the occupancy passes run as well,
 *              OVERLAPPED with the recursions (the frames whose alpha'/beta rows already exist
 *              are evaluated on the CUs the 2B persistent recursion workgroups leave idle), and
 *              grad = grad_scale * (gamma_den - gamma_num) is written; a caller that later
 *              learns an upstream gradient != 1 applies it with pychain_hip_rescale;
 *   _backward  runs the time-parallel occupancy passes and writes
 *              grad = grad_scale * (*grad_scale_dev) * (gamma_den - gamma_num) ONCE.
 *              grad_scale_dev (device float*, may be NULL = 1) is the upstream scalar gradient,
 *              read on the device: no host sync and no extra pass over [B,T,D] to apply it
 *              (the reference multiplies the stored gradient again, loss.py:85).
 * bad_count: dev int32[2] for _forward, dev int32[2] for _backward (may be the same words).
 * totals (dev float[PYCHAIN_HIP_TOTALS = 8] or NULL; _forward and _forward_backward): the scalars of ChainLoss.forward from the call's last
 *   kernel instead of from half a dozen scalar kernels of the host framework behind it (pychain/loss.py:100-104):
 *   totals[0] = (sum_b den_objf[b] - sum_b num_objf[b]) * loss_scale [/ *loss_norm_dev]  = -(num - den) [/ frames],
 *   totals[1] = sum_b len_b, totals[2] = bad_count[0] + bad_count[1] as a float (what a sharded trainer all-reduces
 *   with the loss), totals[3] = sum den - sum num unscaled, totals[4] = totals[0], totals[5..7] as above.  loss_norm_dev: device
 *   float or NULL; where it is given, a gradient written by the same call is divided by it too (grad = grad_scale /
 *   *loss_norm_dev * (gamma_den - gamma_num): the occupancy launches read its reciprocal on the device - ChainLoss(avg=True)
 *   with the lengths on the device costs no pass over [B,T,D] behind the call).
 * A batch larger than the chip (B >= 7/8 of the CU count, one shared denominator plan, `grad` given): the call runs over
 *   SLICES of about CUs / 2 sequences (B = 256: 2 x 128, 320: 2 x 160, 384: 3 x 128), one after the other on `stream` in the same workspaces (their last 4 KiB hold the
 *   slices' counters), and a last small launch forms bad_count and totals over the whole batch - same per-sequence results,
 *   same totals (fp64 over the per-sequence values, rounded once).  With every CU holding a recursion workgroup the
 *   numerator of ONE call only finds room as they end: B = 256 on 256 CUs 11.1 ms, as two slices 10.2.  Option
 *   "chain_slices": "0" never, "n" that many; pychain_hip_chain_loss_slices says what a call would do.
 */
int pychain_hip_chain_loss_slices(int64_t plan_stride_bytes, int resident_slot_rows, int B);
int pychain_hip_chain_loss_forward(
    const void* plans_dev, int64_t plan_stride_bytes, int resident_slot_rows, int den_num_states,
    float leaky_hmm_coefficient,
    const int32_t* forward_transitions, const int32_t* forward_transition_indices,
    const float* forward_transition_probs, const int32_t* backward_transitions,
    const int32_t* backward_transition_indices, const float* backward_transition_probs,
    const float* initial_probs, const float* final_probs, int graph_batch_stride,
    int num_num_states, int num_num_transitions,
    const void* nnet_output, int nnet_output_dtype, const int64_t* seq_lengths, int B, int T, int num_pdfs,
    float* den_objf_per_seq, float* num_objf_per_seq,
    void* grad /* may be NULL; nnet_output_dtype */, float grad_scale, int32_t* bad_count,
    float loss_scale, const float* loss_norm_dev, float* totals /* may be NULL */,
    void* den_workspace, size_t den_workspace_bytes, void* num_workspace, size_t num_workspace_bytes,
    void* stream);
/* ------------------------------------------------------------------------
 * Host twins (ABI 14): the same two computations on HOST pointers in the reference layout, for callers
 * whose tensors live on the CPU - the reference serves those from its own CPU loops (chain-computation.cc:113-176,247-311;
 * chain-log-domain-computation.cc:123-159,231-271) and code written against it unit-tests its criterion there.  The sequences
 * of the minibatch are dealt to `num_threads` host threads (0 = one per hardware thread); graph_batch_stride 0 = one graph
 * for every sequence, 1 = per-sequence graphs [B,...].  Denominator: fp32 state vectors and gradient, fp64 totals and
 * log-probability; numerator: fp64 log-probabilities, exact log-sum-exp (as the device path; the reference's fp32 LogAdd chain
 * is the device's option num_compat).  bad_count: host int32[1], the reference's `ok` (sequences whose log-probability is not
 * finite or whose frame-0 occupancies do not sum to one within 5 %).  These functions never touch a device, and the device
 * entry points never fall back to them.  pychain_hip_cpu_calls: how many host calls this process has made (the GPU tests
 * check that it stays where it was). */
long pychain_hip_cpu_calls(void);
int pychain_hip_cpu_den_forward_backward(
    const int32_t* forward_transitions, const int32_t* forward_transition_indices, const float* forward_transition_probs,
    const int32_t* backward_transitions, const int32_t* backward_transition_indices, const float* backward_transition_probs,
    const float* leaky_probs, const float* initial_probs, const float* final_probs, int graph_batch_stride,
    const float* nnet_output, int input_is_exp, const int64_t* seq_lengths,
    int B, int T, int num_pdfs, int num_states, int num_transitions,
    float leaky_hmm_coefficient, float grad_scale, float* objf_per_seq, float* grad, int32_t* bad_count, int num_threads);
int pychain_hip_cpu_num_forward_backward(
    const int32_t* forward_transitions, const int32_t* forward_transition_indices, const float* forward_transition_probs,
    const int32_t* backward_transitions, const int32_t* backward_transition_indices, const float* backward_transition_probs,
    const float* initial_probs, const float* final_probs, int graph_batch_stride,
    const float* nnet_output, const int64_t* seq_lengths,
    int B, int T, int num_pdfs, int num_states, int num_transitions, int grad_mode, float grad_scale,
    float* objf_per_seq, float* grad, int32_t* bad_count, int num_threads);

/* data[0..n) *= *scale_dev, skipped on the device when the scalar is exactly 1. */
int pychain_hip_rescale(void* data, int dtype /* PYCHAIN_HIP_F32 / _BF16 / _F16 */, size_t n, const float* scale_dev, void* stream);
/* 1 if the fused calls (_forward with a gradient or without, _forward_backward) take 2-byte network outputs for this shape:
 * both sides do (pychain_hip_den_half_native, pychain_hip_num_half_native) and the gradient is written ONCE, by the two-frame
 * occupancy kernel with the numerator folded in (an accumulation into a 2-byte gradient would round twice).
 * pychain_hip_chain_loss_backward is fp32 only. */
int pychain_hip_chain_loss_half_native(int64_t plan_stride_bytes, int resident_slot_rows, int den_num_states, int num_pdfs,
                                       int B, int T, int num_num_states, int num_num_transitions);
/* out[0] = (sum_b den_objf_per_seq[b] - sum_b num_objf_per_seq[b]) * scale, divided by *norm_dev if norm_dev != NULL:
 * the scalar ChainLoss.forward returns, -(num - den) [/ sum of lengths] (pychain/loss.py:100-104), in one launch
 * (fp64 accumulation, rounded once).  num_objf_per_seq may be NULL (denominator only).  All pointers on the device. */
int pychain_hip_loss_total(const float* den_objf_per_seq, const float* num_objf_per_seq, int B, float scale,
                           const float* norm_dev, float* out, void* stream);
/* NOT after a forward call that wrote its gradient in slices (a batch of >= 7/8 of the CU count with `grad` given:
 * pychain_hip_chain_loss_slices > 1): the workspaces then hold the LAST slice's trajectories only, and this call fails with
 * PYCHAIN_HIP_EUNSUPPORTED instead of writing a wrong gradient (the library remembers, by workspace address, which forward
 * call last wrote a workspace). */
int pychain_hip_chain_loss_backward(
    const void* plans_dev, int64_t plan_stride_bytes, int resident_slot_rows, int den_num_states,
    const int32_t* forward_transitions, const int32_t* forward_transition_indices,
    const float* forward_transition_probs,
    int graph_batch_stride, int num_num_states, int num_num_transitions,
    const void* nnet_output, int nnet_output_dtype /* PYCHAIN_HIP_F32 only */, const int64_t* seq_lengths, int B, int T, int num_pdfs,
    float grad_scale, const float* grad_scale_dev,
    void* grad, int32_t* bad_count,
    void* den_workspace, size_t den_workspace_bytes, void* num_workspace, size_t num_workspace_bytes,
    void* stream);

/* ------------------------------------------------------------------------
 * Batch containers (replaces the Python collation of pychain/graph.py:122-194).
 * The ten tensors of a ChainGraphBatch built from a list of graphs live in ONE buffer, fields in the order
 *   forward_transitions [B,K,3] i32, forward_transition_indices [B,H,2] i32, forward_transition_probs [B,K] f32,
 *   backward_transitions, backward_transition_indices, backward_transition_probs (same shapes),
 *   final_probs [B,H] f32, initial_probs [B,H] f32, leaky_probs [B,H] f32 (absent in the log domain), start_state [B] i64,
 * each field 64-byte aligned: _layout gives their byte offsets / bytes per utterance and returns the buffer size.
 *   _pack        host: rec[b] = {num_transitions, num_states, start_state, then the host addresses of the graph's nine
 *                tensors in field order (leaky = 0 in the log domain)}; pads as the reference does (zeros; -inf for
 *                initial / final probs in the log domain, graph.py:140-145).  Pack into pinned memory and the whole
 *                batch reaches the device with ONE copy.
 *   _reorder     host: out[b] = in[order[b]] for every field (index_select along the batch: graph.py:177-194; B_out may
 *                differ from B_in - sharding selects a subset)
 *   _reorder_dev the same gather between two DEVICE buffers, one launch on `stream`; order_dev: dev int64[B_out]. */
#define PYCHAIN_HIP_BATCH_FIELDS 10
#define PYCHAIN_HIP_BATCH_REC_WORDS 12
int64_t pychain_hip_batch_layout(int B, int K, int H, int log_domain, int64_t offsets[PYCHAIN_HIP_BATCH_FIELDS],
                                 int64_t row_bytes[PYCHAIN_HIP_BATCH_FIELDS]);
int     pychain_hip_batch_pack(int B, int K, int H, int log_domain, const uint64_t* rec, void* out, size_t out_bytes);
int     pychain_hip_batch_reorder(int B_in, int B_out, int K, int H, int log_domain, const void* in, void* out,
                                  const int64_t* order);
int     pychain_hip_batch_reorder_dev(int B_in, int B_out, int K, int H, int log_domain, const void* in_dev, void* out_dev,
                                      const int64_t* order_dev, void* stream);

/* ------------------------------------------------------------------------
 * Graph ingestion without OpenFST (host only, one-time per graph): what the reference's
 * `simplefst` module provides on top of OpenFST (openfst_binding/src/fstext.cc:174-184).
 * Handles are opaque host objects; *_read / *_from_arcs return NULL on error.
 *   pychain_hip_fst_read(file, 0)        StdVectorFst.read       (fstext.cc:178)
 *   pychain_hip_fst_read(file, offset)   StdVectorFst.read_ark   (ReadFstFromArk, fstext.cc:7-16)
 *   pychain_hip_fst_write                StdVectorFst.write      (fstext.cc:177)
 *   pychain_hip_fst_num_states / _start  num_states / start_state (fstext.cc:182-183)
 *   pychain_hip_fst_to_tensors           fst_to_tensor           (FstToTensor, fstext.cc:19-117)
 *   pychain_hip_fst_leaky_probs          set_leaky_probs         (SetLeakyProbs, fstext.cc:120-171)
 * fst_to_tensors outputs are caller-allocated: forward/backward transitions [K,3] int32, probs [K],
 * indices [H,2] int32, final [H]; K = pychain_hip_fst_num_arcs.
 */
void*   pychain_hip_fst_read(const char* filename, int64_t byte_offset);
void*   pychain_hip_fst_from_arcs(int32_t num_states, int32_t start, int64_t num_arcs,
                                  const int32_t* src, const int32_t* dst, const int32_t* ilabel,
                                  const float* weight, const float* final_weight);
void    pychain_hip_fst_free(void* fst);
int     pychain_hip_fst_write(const void* fst, const char* filename);
int32_t pychain_hip_fst_num_states(const void* fst);
int32_t pychain_hip_fst_start(const void* fst);
int64_t pychain_hip_fst_num_arcs(const void* fst);
int     pychain_hip_fst_to_tensors(const void* fst, int log_domain,
                                   int32_t* forward_transitions, float* forward_transition_probs,
                                   int32_t* forward_transition_indices,
                                   int32_t* backward_transitions, float* backward_transition_probs,
                                   int32_t* backward_transition_indices, float* final_probs);
int     pychain_hip_fst_leaky_probs(const void* fst, float* leaky_probs);

#ifdef __cplusplus
}
#endif
#endif  /* PYCHAIN_HIP_H_ */
