// num_kernels.h - launch interface of the numerator kernels (num_kernels.hip).
#ifndef PYCHAIN_HIP_NUM_KERNELS_H_
#define PYCHAIN_HIP_NUM_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pychain_hip {

struct NumArgs {
  const int32_t* fwd_trans; const int32_t* fwd_idx; const float* fwd_probs;
  const int32_t* bwd_trans; const int32_t* bwd_idx; const float* bwd_probs;
  const float* initial; const float* final_;
  const float* x;            // [B,T,D] raw   (x_half: 2-byte elements behind this pointer)
  int x_half;                // 0 fp32; kXBf16 / kXF16 (device_utils.h): the recursions read 2-byte rows as they are (num_fb_kernel)
  const int64_t* lengths;    // [B]
  float* objf;               // [B]
  float* grad;               // [B,T,D]
  int32_t* bad;              // [1]
  double* alpha_ws;          // [B,T+1,H]  alpha(t,h): unnormalised log-probabilities (fp64)
  double* beta_ws;           // [B,T+1,H]  beta(t,h)
  double* logp_ws;           // [B]        sequence log-probability (fp64)
  float* frac_ws;            // [B,T,K]    r_k(t): log-share of forward arc k in beta(t,src_k) (written by the backward pass)
  float* rows_ws;            // [B,T,K]    compact rows: occupancy of the u-th distinct pdf of the sequence
  int32_t* upd_ws;           // [B,K]      the distinct pdf-ids of a sequence's arcs, ascending
  int32_t* ucount_ws;        // [B]        how many
  int32_t* uidx_ws;          // [B,K]      compact row of every arc (index of its pdf in upd_ws), -1 = unused arc
  int graph_stride;          // 1 = per-sequence graphs, 0 = shared
  int B, T, D, H, K;
  int grad_mode;
  int frames_per_block;      // occupancy kernel
  float grad_scale;
  const float* grad_scale_dev;   // optional device scalar multiplied into grad_scale
  // BetaGeneralFrameDebug, chain-log-domain-computation.cc:283-304: a frame's occupancies must sum to 1 within
  // 5 % (and be finite); checked at t == 0 always, on every frame when check_all (verbose level >= 1)
  int check_all;
  // 1: a NaN network output stays a NaN in the staged rows (torch.clamp keeps it, loss.py:30: it reaches the
  // log-probability if an arc of the utterance emits that pdf).  0: the fused loss - there the denominator's alpha
  // workgroups watch every element of every row and turn the loss into NaN, and the row-staging waves, whose
  // instruction count sets the pace of a numerator step, skip the check (num_fb 8 % faster).
  int watch_nan;
  // option debug_corrupt_row "num,b,t,scale": log(scale) is added to the stored alpha(t,.) of utterance b between the
  // recursions and the occupancy pass, so that the 5 % invariant can be seen to fire; corrupt_b < 0: off
  int corrupt_b, corrupt_t;
  float corrupt_log;
  // graphs beyond the tile kernels (num_needs_general): the launches below run num_general.hip instead; gen_acc = the
  // occupancy launch's accumulator rows in the workspace (kNumGeneralBlocks x D 64-bit words)
  int general;
  void* gen_acc;
  // option num_compat = 1 (num_compat.hip): the reference's own fp32 arithmetic - LogAdd with its cut-off, per-frame
  // renormalisation, the reference's term order - instead of the exact fp64 path; compat_ws = per-sequence scratch of
  // compat_stride bytes (sort keys, per-arc occupation log-probabilities, beta rows)
  int compat;
  char* compat_ws;
  size_t compat_stride;
};

// Numerator graphs the tile kernels do not take - more than 65 535 states or pdfs, or state vectors + nnet-output rows +
// arcs beyond the LDS - run on kernels that gather everything from global memory (num_general.hip): slow, complete, like
// the reference's CPU path (chain-log-domain-computation.cc:123-159 has no size limit).
constexpr int kNumGeneralBlocks = 256;
bool num_needs_general(int H, int K, int D);
size_t num_general_acc_bytes(int D);
hipError_t launch_num_general_fb(const NumArgs& a, hipStream_t st);
hipError_t launch_num_general_occ(const NumArgs& a, void* acc, hipStream_t st);

// the numerator in the reference's arithmetic, one workgroup per sequence (NumArgs::compat); phases bit 0: alpha, the
// log-probabilities and beta's start row; bit 1: beta and the gradient (from what bit 0 left in the workspace)
size_t num_compat_stride(int H, int K);
hipError_t launch_num_compat(const NumArgs& a, int phases, hipStream_t st);

size_t num_fb_lds_bytes(int H, int K, int D);
// forward and backward recursions (launch 1, 2B workgroups): reads x + graphs, writes objf, logp_ws, alpha_ws, beta_ws
hipError_t launch_num_fb(const NumArgs& a, hipStream_t st, const char** why);
// option debug_corrupt_row (NumArgs::corrupt_b)
hipError_t launch_num_corrupt(const NumArgs& a, hipStream_t st);
// distinct pdf-ids per sequence (upd_ws, ucount_ws): needed before a compact occupancy launch
hipError_t launch_num_prep(const NumArgs& a, hipStream_t st, const char** why);
// occupancies -> gradient rows (launch 2): reads alpha_ws, beta_ws, logp_ws, x; writes/accumulates grad
// (grad_mode) or, with `compact`, rows_ws
hipError_t launch_num_occ(const NumArgs& a, bool compact, hipStream_t st, const char** why);
// compact rows (rows_ws, upd_ws, ucount_ws) accumulated into grad, scaled by grad_scale
hipError_t launch_num_scatter(const NumArgs& a, hipStream_t st, const char** why);

}  // namespace pychain_hip
#endif
