// num_general.hip - the numerator forward-backward for graphs the tile kernels of num_kernels.hip do not take: more than
// 65 535 states or pdfs (their arcs pack state | pdf << 16), or state vectors + nnet-output rows + arcs beyond the
// 160 KiB LDS of one CU.  The reference's CPU path has no such limits (chain-log-domain-computation.cc:123-159,231-271),
// so neither may the drop-in: these kernels are SLOW - every operand comes from global memory - but complete: any H, K, D
// the int32 indices of the reference layout can express.  Same decomposition, same stored rows (fp64 log-probabilities
// without per-frame renormalisation), same occupancy formula and the same checks as the tile path, so everything around
// the launches (workspace, corrupt-row hook, `ok`) is shared.
//
//   num_general_fb_kernel    2B workgroups, one per (sequence, direction), persistent over the frames; a thread owns states
//                            tid, tid + 1024, ...; the previous row is read back from the trajectory store it was written to
//                            (L1-bypassing loads); the backward pass also writes every arc's log-share r_k(t)
//   num_general_occ_kernel   time-parallel over (sequence, frame): per-arc occupancies merged by pdf-id in 64-bit fixed point
//                            - order-independent, hence deterministic - in a per-workgroup accumulator row in GLOBAL memory
//                            (the tile kernel's lives in LDS), written as log / linear / accumulated gradient rows
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pychain_hip.h"
#include "common.h"
#include "device_utils.h"
#include "num_kernels.h"

namespace pychain_hip {
namespace {
constexpr int kNGT = 1024;
constexpr float kGLog2e = 1.44269504088896340736f, kGLn2 = 0.693147182464599609375f;
constexpr float kGFixScale = 72057594037927936.0f, kGFixInv = 1.0f / 72057594037927936.0f;     // 2^56, as num_kernels.hip

__device__ __forceinline__ float gexp(float d) { return __builtin_amdgcn_exp2f(d * kGLog2e); }   // d <= ~0
__device__ __forceinline__ float glog(float s) { return __builtin_amdgcn_logf(s) * kGLn2; }       // s >= 1

// a double written by another thread of this workgroup before the last barrier: read around the (non-coherent) vector L1
__device__ __forceinline__ double fresh(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// torch.clamp(x, -30, 30) keeps a NaN (pychain/loss.py:30): so does this
__device__ __forceinline__ float gclamp(float x) { return x != x ? x : fminf(fmaxf(x, -30.f), 30.f); }

// log-sum-exp over float64 terms with fp32 transcendentals on small differences (num_kernels.hip: Lse); the sum itself is
// kept in fp64 here - a state of these graphs may have tens of thousands of arcs, and a float sum of 70 000 terms is 2e-5 off
struct GLse {
  double m, s;
  __device__ __forceinline__ void init() { m = -INFINITY; s = 0.0; }
  __device__ __forceinline__ void push(double e) {
    if (e != e) { m = e; return; }                                     // a NaN network output reaches the log-probability
    if (e > m) { s = (m == -INFINITY) ? 1.0 : s * (double)gexp((float)(m - e)) + 1.0; m = e; }
    else if (e != -INFINITY) { s += (double)gexp((float)(e - m)); }
  }
  __device__ __forceinline__ double value() const { return m == -INFINITY ? -INFINITY : m + log(s); }
};

__device__ __forceinline__ double block_max(double v, double* red, int tid) {     // red[17]; every thread gets the result
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double m = red[0];
  for (int w = 1; w < kNGT / 64; w++) m = fmax(m, red[w]);
  return m;
}
__device__ __forceinline__ float block_sumf(float v, double* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = (double)v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < kNGT / 64; w++) s += red[w];
  return (float)s;
}

__global__ __launch_bounds__(kNGT) void num_general_fb_kernel(const NumArgs a) {
  __shared__ double red[32];
  const int tid = threadIdx.x;
  const bool fwd = blockIdx.x < (unsigned)a.B;
  const int b = fwd ? blockIdx.x : blockIdx.x - a.B;
  const int L = seq_len(a.lengths, b, a.T);
  const int H = a.H, K = a.K, D = a.D, T = a.T;
  const size_t g = (size_t)b * a.graph_stride;
  // alpha gathers over the arcs ENTERING a state (backward_transitions: {src, dst, pdf} grouped by dst), beta over the arcs
  // LEAVING it (forward_transitions grouped by src): fstext.cc:49-116
  const int32_t* tr = (fwd ? a.bwd_trans : a.fwd_trans) + g * K * 3;
  const float* pr = (fwd ? a.bwd_probs : a.fwd_probs) + g * K;
  const int2* idx = reinterpret_cast<const int2*>((fwd ? a.bwd_idx : a.fwd_idx) + g * H * 2);
  const float* xseq = a.x + (size_t)b * T * D;
  double* rows = (fwd ? a.alpha_ws : a.beta_ws) + (size_t)b * (T + 1) * H;
  const int other = fwd ? 0 : 1;                       // the arc's other end: source for alpha, destination for beta
  // AlphaFirstFrame :84-90 / BetaLastFrame :192-202 (unnormalised: beta(L,i) = final(i); 1 / P enters the occupancy)
  for (int h = tid; h < H; h += kNGT) rows[(size_t)(fwd ? 0 : L) * H + h] = (double)(fwd ? a.initial : a.final_)[g * H + h];
  __threadfence();
  __syncthreads();
  // fwd: alpha(t,h) = LogSum_k alpha(t-1,src_k) + lp_k + x(t-1,pdf_k), t = 1..L     (:93-159, unnormalised)
  // bwd: beta(t,h)  = LogSum_k lp_k + beta(t+1,dst_k) + x(t,pdf_k),    t = L-1..0   (:204-271, unnormalised)
  for (int s = 1; s <= L; s++) {
    const int t_in = fwd ? s - 1 : L - s + 1, t_out = fwd ? s : L - s, tx = fwd ? s - 1 : L - s;
    const double* prev = rows + (size_t)t_in * H;
    const float* xrow = xseq + (size_t)tx * D;
    double* out = rows + (size_t)t_out * H;
    float* frow = fwd ? nullptr : a.frac_ws + ((size_t)b * T + t_out) * K;
    for (int h = tid; h < H; h += kNGT) {
      const int2 be = idx[h];
      GLse acc; acc.init();
      for (int k = be.x; k < be.y; k++)
        acc.push(fresh(prev + tr[3 * k + other]) + ((double)pr[k] + (double)gclamp(xrow[tr[3 * k + 2]])));
      const double v = acc.value();
      out[h] = v;
      // the backward pass also writes, per arc, its log-share of its source state's beta: r_k(t) = term_k - beta(t,h) <= 0
      // (all the occupancy pass needs from this frame's nnet-output row: num_kernels.hip)
      if (!fwd)
        for (int k = be.x; k < be.y; k++)
          frow[k] = (float)(fresh(prev + tr[3 * k + other]) + ((double)pr[k] + (double)gclamp(xrow[tr[3 * k + 2]])) - v);
    }
    __threadfence();
    __syncthreads();
  }
  if (!fwd) return;
  // total log-probability: LogSum_i alpha(L,i) + final(i)   (ComputeTotLogLike :170-190)
  const double* vL = rows + (size_t)L * H;
  double mx = -INFINITY;
  bool nan = false;
  for (int h = tid; h < H; h += kNGT) { const double e = fresh(vL + h) + (double)a.final_[g * H + h]; nan = nan || e != e; mx = fmax(mx, e); }
  const double gm = block_max(mx, red, tid);
  float se = 0.f;
  if (gm != -INFINITY)
    for (int h = tid; h < H; h += kNGT) se += gexp((float)(fresh(vL + h) + (double)a.final_[g * H + h] - gm));
  const float stot = block_sumf(se, red, tid);
  const float anynan = block_sumf(nan ? 1.f : 0.f, red, tid);
  if (tid == 0) {
    const double logp = anynan != 0.f ? (double)__builtin_nanf("") : (gm == -INFINITY ? -INFINITY : gm + (double)glog(stot));
    const float objf = (float)logp;
    a.objf[b] = objf;
    a.logp_ws[b] = logp;
    if (!(objf - objf == 0.f) || seq_len_bad(a.lengths, b, a.T)) atomicAdd(a.bad, 1);
  }
}

// BetaGeneralFrame :204-271: occupancy of arc k = exp(alpha(t,src) + [lp + x(t,pdf) + beta(t+1,dst)] - logP), the bracket
// being beta(t,src) + r_k(t).  Persistent workgroups over the (sequence, frame) items; `acc` = this workgroup's row of D
// 64-bit fixed-point accumulators in the workspace (zero between frames).
__global__ __launch_bounds__(kNGT) void num_general_occ_kernel(const NumArgs a, unsigned long long* acc_all, int Dp) {
  __shared__ float s_fsum;
  const int tid = threadIdx.x;
  const int K = a.K, D = a.D, T = a.T, H = a.H;
  unsigned long long* acc = acc_all + (size_t)blockIdx.x * Dp;
  for (int n = tid; n < Dp; n += kNGT) acc[n] = 0ull;
  __threadfence();
  __syncthreads();
  const int mode = a.grad_mode;
  const float gscale = a.grad_scale_dev ? a.grad_scale * *a.grad_scale_dev : a.grad_scale;
  const float fill = mode == PYCHAIN_HIP_GRAD_LOG ? -INFINITY : 0.f;
  int bad = 0;
  for (long item = blockIdx.x; item < (long)a.B * T; item += gridDim.x) {
    const int b = (int)(item / T), t = (int)(item % T);
    const int L = seq_len(a.lengths, b, T);
    float* grow = a.grad + ((size_t)b * T + t) * D;
    if (t >= L) {                                     // padded frames: -inf (full_like(-inf), :57) / zero; ACCUM leaves them alone
      if (mode != PYCHAIN_HIP_GRAD_ACCUM) for (int n = tid; n < D; n += kNGT) grow[n] = fill;
      continue;
    }
    const size_t g = (size_t)b * a.graph_stride;
    const int32_t* ft = a.fwd_trans + g * K * 3;
    const int2* fi = reinterpret_cast<const int2*>(a.fwd_idx + g * H * 2);
    const double logp = a.logp_ws[b];
    const double* arow = a.alpha_ws + ((size_t)b * (T + 1) + t) * H;
    const double* brow = a.beta_ws + ((size_t)b * (T + 1) + t) * H;
    const float* frow = a.frac_ws + ((size_t)b * T + t) * K;
    const bool check = t == 0 || a.check_all;         // the reference's `ok` (NumArgs::check_all)
    if (tid == 0) s_fsum = 0.f;
    __syncthreads();
    float fsum = 0.f;
    // arcs that no state indexes (batch padding, pychain/graph.py:132-139) carry no occupancy: walk the states' arc ranges
    for (int h = tid; h < H; h += kNGT) {
      const int2 be = fi[h];
      if (be.y <= be.x) continue;
      const float st = (float)(arow[h] + brow[h] - logp);     // log occupancy of the source state (fp64, rounded once: as the tile kernels)
      for (int k = be.x; k < be.y; k++) {
        const float v = st == -INFINITY ? 0.f : gexp(st + frow[k]);
        fsum += v;
        if (v > 0.f) {
          if (v <= 2.f) atomicAdd(&acc[ft[3 * k + 2]], (unsigned long long)(v * kGFixScale));
          else bad = 1;
        } else if (v != 0.f) {
          bad = 1;                                    // NaN
        }
      }
    }
    if (check) {
      const float ws = wave_sum(fsum);
      if ((tid & 63) == 0) atomicAdd(&s_fsum, ws);
    }
    __threadfence();
    __syncthreads();
    if (check && tid == 0 && !(fabsf(s_fsum - 1.f) <= 0.05f)) bad = 1;
    // every pdf's merged sum: taken (and zeroed for the next frame) with an exchange, which also reads around the L1
    for (int n = tid; n < D; n += kNGT) {
      const unsigned long long u = atomicExch(&acc[n], 0ull);
      const float v = (float)u * kGFixInv;
      if (mode == PYCHAIN_HIP_GRAD_ACCUM) { if (u) grow[n] = mul_add_rn(v, gscale, grow[n]); }
      else grow[n] = mode == PYCHAIN_HIP_GRAD_LOG ? (u ? logf(v) : -INFINITY) : gscale * v;
    }
    __threadfence();
    __syncthreads();
  }
  if (bad && (tid & 63) == 0) atomicAdd(a.bad, 1);
}
}  // namespace

bool num_needs_general(int H, int K, int D) {
  if (H > 65535 || D > 65535) return true;
  // what the tile kernels keep in LDS (num_kernels.hip: num_fb_kernel, num_prep_kernel, num_occ_kernel)
  const size_t Dp = (size_t)(D + 3) & ~(size_t)3;
  return num_fb_lds_bytes(H, K, D) > 160 * 1024 || 4 * (size_t)D + 16 + 4 * 256 + 64 > 160 * 1024 ||
         8 * Dp + 16 * ((size_t)(H + 1) & ~(size_t)1) + 16 + 10 * (size_t)K + 64 > 160 * 1024;
}
size_t num_general_acc_bytes(int D) { return (size_t)kNumGeneralBlocks * (((size_t)D + 3) & ~(size_t)3) * 8; }

hipError_t launch_num_general_fb(const NumArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(num_general_fb_kernel, dim3(2 * a.B), dim3(kNGT), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_num_general_occ(const NumArgs& a, void* acc, hipStream_t st) {
  const long items = (long)a.B * a.T;
  const int nb = (int)(items < kNumGeneralBlocks ? items : kNumGeneralBlocks);
  hipLaunchKernelGGL(num_general_occ_kernel, dim3(nb), dim3(kNGT), 0, st, a, (unsigned long long*)acc, (a.D + 3) & ~3);
  return hipGetLastError();
}
}  // namespace pychain_hip
