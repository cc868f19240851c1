// den_general.hip - the denominator forward-backward for graphs the compiled-plan kernels do not take: more than
// 65 535 states or pdfs (their arcs pack two 16-bit LDS addresses), or a state vector + nnet-output row that does not
// fit the 160 KiB LDS of one CU.  The reference's CPU path has no such limits (chain-computation.cc:113-176,247-311), so
// neither may the drop-in: these kernels are SLOW (every operand is gathered from global memory, L2-resident at best) but
// complete - any H, K, D that fits the int32 indices of the reference layout - and keep the same decomposition, row
// formats and checks as the fast path (den_recursion_kernel's normalised rows, den_finish_kernel, the `ok` invariant),
// so everything downstream of the launches is shared.
//
//   den_general_recursion_kernel   2B workgroups, one per (sequence, direction), persistent over the frames; a thread
//                                  owns states tid, tid + 1024, ...; arcs CSR by destination (alpha) / source (beta) as
//                                  the reference lays them out (fstext.cc:49-116); the previous row is read back from
//                                  the trajectory store it was written to (L1-bypassing loads)
//   den_general_gamma_kernel       time-parallel, arcs CSR by pdf-id: lane-private sums, no atomics, deterministic
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "den_kernels.h"
#include "device_utils.h"
#include "plan_format.h"

namespace pychain_hip {
namespace {
constexpr int kGNT = 1024;

struct GenView {
  const int32_t *a_idx, *b_idx, *g_idx;         // [H][2], [H][2], [D + 1]
  const int2 *a_arc, *b_arc, *g_arc;            // {src, pdf} by destination; {dst, pdf} by source; {src, dst} by pdf
  const float *a_p, *b_p, *g_p;
  const float *leaky, *init, *fin;
  int H, K, D;
};
__device__ __forceinline__ GenView gen_view(const char* plan) {
  const GeneralPlanHeader* h = reinterpret_cast<const GeneralPlanHeader*>(plan);
  GenView v;
  v.H = h->H; v.K = h->K; v.D = h->D;
  v.a_idx = reinterpret_cast<const int32_t*>(plan + h->off_a_idx); v.a_arc = reinterpret_cast<const int2*>(plan + h->off_a_arc);
  v.a_p = reinterpret_cast<const float*>(plan + h->off_a_p);
  v.b_idx = reinterpret_cast<const int32_t*>(plan + h->off_b_idx); v.b_arc = reinterpret_cast<const int2*>(plan + h->off_b_arc);
  v.b_p = reinterpret_cast<const float*>(plan + h->off_b_p);
  v.g_idx = reinterpret_cast<const int32_t*>(plan + h->off_g_idx); v.g_arc = reinterpret_cast<const int2*>(plan + h->off_g_arc);
  v.g_p = reinterpret_cast<const float*>(plan + h->off_g_p);
  v.leaky = reinterpret_cast<const float*>(plan + h->off_leaky); v.init = reinterpret_cast<const float*>(plan + h->off_init);
  v.fin = reinterpret_cast<const float*>(plan + h->off_final);
  return v;
}

// a float written by another thread of this workgroup before the last barrier: read around the (non-coherent) vector L1
__device__ __forceinline__ float load_fresh(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {      // red[17]; every thread gets the total
  v = wave_sum(v);
  __syncthreads();                                    // red free again
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < kGNT / 64; w++) s += red[w];    // (same order in every thread)
  return s;
}

__device__ __forceinline__ float nnet(const float* row, int n, int is_exp, bool& nan) {
  const float r = row[n];
  nan = nan || r != r;
  return clamp_exp(r, is_exp);
}

__global__ __launch_bounds__(kGNT) void den_general_recursion_kernel(const DenArgs a) {
  __shared__ float red[32];
  const int tid = threadIdx.x;
  const bool fwd = blockIdx.x < (unsigned)a.B;
  const int b = fwd ? blockIdx.x : blockIdx.x - a.B;
  const int L = seq_len(a.lengths, b, a.T);
  const GenView g = gen_view(a.plans + (size_t)b * a.plan_stride);
  const int H = g.H, Hp = a.Hp, D = a.D;
  const float* xseq = a.x + (size_t)b * a.T * D;
  float* store = fwd ? a.alpha_store + (size_t)b * a.T * Hp : a.beta_store + (size_t)b * (a.T + 1) * Hp;
  float* totv = (fwd ? a.tot_a : a.tot_b) + (size_t)b * (a.T + 2);
  const float coef = a.coef;
  int bad = (fwd && seq_len_bad(a.lengths, b, a.T)) ? 1 : 0;   // bit 0 not ok, bit 1 a NaN network output (den_lazy.inc.h)
  bool nan = false;

  // frame 0 (alpha: chain-computation.cc:92-110,178-194) / frame L (beta: :232-245,313-330)
  {
    const float* start = fwd ? g.init : g.fin;
    float p0 = 0.f, p1 = 0.f;
    for (int i = tid; i < H; i += kGNT) { p0 += start[i]; p1 += start[i] * g.leaky[i]; }
    const float tot = block_sum(p0, red, tid), wtot = block_sum(p1, red, tid);
    const float inv = 1.f / tot;
    if (!(tot > 0.f) || !(inv > 0.f)) bad |= 1;
    if (tid == 0) totv[fwd ? 0 : L] = tot;
    float* row = store + (size_t)(fwd ? 0 : L) * Hp;
    for (int i = tid; i < Hp; i += kGNT) {
      float v = 0.f;
      if (i < H) v = fwd ? start[i] * inv + coef * g.leaky[i] : (start[i] + coef * wtot) * inv;
      row[i] = v;
    }
    __threadfence();
    __syncthreads();
  }
  // general frames.  alpha: step j produces alpha'(j+1) from alpha'(j) and x(j); beta: step j produces beta(t), t = L-1-j,
  // from beta(t+1) and x(t).  alpha'(L,.) - read by nobody but ComputeTotLogLike below - goes to row 0 of the sequence's
  // BETA store, which the beta recursion never writes (its rows are L .. 1).
  const int nsteps = fwd ? L : L - 1;
  float* last_alpha = a.beta_store + (size_t)b * (a.T + 1) * Hp;
  for (int j = 0; j < nsteps; j++) {
    const int t_in = fwd ? j : L - j, t_out = fwd ? j + 1 : L - 1 - j, tx = fwd ? j : L - 1 - j;
    const float* prev = store + (size_t)t_in * Hp;
    const float* xrow = xseq + (size_t)tx * D;
    float* out = (fwd && t_out == L) ? last_alpha : store + (size_t)t_out * Hp;
    float s0 = 0.f, s1 = 0.f;
    // a NaN anywhere in a live frame's row is a NaN loss, as in the tile kernels (they watch every element they stage) -
    // also on a pdf no arc of this graph carries
    if (fwd) for (int n = tid; n < D; n += kGNT) nan = nan || xrow[n] != xrow[n];
    for (int i = tid; i < H; i += kGNT) {
      const int32_t* idx = (fwd ? g.a_idx : g.b_idx) + 2 * i;
      const int2* arc = fwd ? g.a_arc : g.b_arc;
      const float* pr = fwd ? g.a_p : g.b_p;
      float acc = 0.f;
      for (int k = idx[0]; k < idx[1]; k++) {
        const int2 e = arc[k];
        acc = fmaf(pr[k] * load_fresh(prev + e.x), nnet(xrow, e.y, a.input_is_exp, nan), acc);
      }
      out[i] = acc;                                    // the raw sum waits in the row's own slot for the frame's total
      s0 += acc;
      if (!fwd) s1 += acc * g.leaky[i];
    }
    const float tot = block_sum(s0, red, tid);
    const float wtot = fwd ? 0.f : block_sum(s1, red, tid);
    const float inv = 1.f / tot;
    if (!(tot > 0.f) || !(inv > 0.f)) bad |= 1;
    if (tid == 0) totv[t_out] = tot;
    for (int i = tid; i < H; i += kGNT) {              // (each thread normalises what it wrote)
      const float r = out[i];
      out[i] = fwd ? r * inv + coef * g.leaky[i] : (r + coef * wtot) * inv;
    }
    __threadfence();
    __syncthreads();
  }
  if (fwd) {
    // ComputeTotLogLike, chain-computation.cc:209-230: log sum_i alpha'(L,i) final(i) + sum_t log tot(t)
    const float* last = L > 0 ? last_alpha : store;
    float f = 0.f;
    for (int i = tid; i < H; i += kGNT) f += load_fresh(last + i) * g.fin[i];
    const float fs = block_sum(f, red, tid);
    const float anynan = block_sum(nan ? 1.f : 0.f, red, tid);
    if (tid == 0) {
      a.fin_dot[b] = anynan != 0.f ? __builtin_nanf("") : fs;   // den_finish_kernel: objf = sum_t log tot(t) + log of this
      if (!(fs > 0.f)) bad |= 1;
    }
  }
  if (nan && fwd) bad |= 2;
  if (bad && (tid & 63) == 0) atomicAdd(a.bad, 1);
}

// occupancies of the frames of one chunk: gamma(t,n) = x(t,n) sum_{arcs with pdf n} p alpha'(t,src) beta(t+1,dst),
// normalised to sum one per frame (chain-computation.cc:289-311,381-390)
__global__ __launch_bounds__(kGNT) void den_general_gamma_kernel(const DenArgs a) {
  __shared__ float red[32];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int L = seq_len(a.lengths, b, a.T);
  const GenView g = gen_view(a.plans + (size_t)b * a.plan_stride);
  const int D = a.D, Hp = a.Hp, T = a.T;
  const int t_begin = blockIdx.x * a.frames_per_block, t_end = min(t_begin + a.frames_per_block, T);
  float* gseq = a.grad + (size_t)b * T * D;
  const float gscale = a.grad_scale_dev ? a.grad_scale * *a.grad_scale_dev : a.grad_scale;
  int bad = 0;
  for (int t = t_begin; t < t_end; t++) {
    float* grow = gseq + (size_t)t * D;
    if (t >= L) {                                     // padding: exact zeros (zeros_like, chain-computation.cc:58)
      for (int n = tid; n < D; n += kGNT) grow[n] = 0.f;
      continue;
    }
    const float* al = a.alpha_store + ((size_t)b * T + t) * Hp;
    const float* be = a.beta_store + ((size_t)b * (T + 1) + t + 1) * Hp;
    const float* xrow = a.x + ((size_t)b * T + t) * D;
    float part = 0.f;
    for (int n = tid; n < D; n += kGNT) {
      float q = 0.f;
      for (int k = g.g_idx[n]; k < g.g_idx[n + 1]; k++) {
        const int2 e = g.g_arc[k];
        q = fmaf(g.g_p[k] * al[e.x], be[e.y], q);
      }
      const float v = q > 0.f ? clamp_exp(xrow[n], a.input_is_exp) * q : 0.f;   // pdfs without arcs: exact zero, x never read
      grow[n] = v;                                    // un-normalised, rescaled below by the thread that wrote it
      part += v;
    }
    const float tot = block_sum(part, red, tid);
    const float sc = gscale / tot;
    if (!(tot > 0.f) || !(sc - sc == 0.f)) bad = 1;
    if (a.check && (t == 0 || a.check_all) && tid == 0) a.gtot[(size_t)b * T + t] = tot;
    for (int n = tid; n < D; n += kGNT) grow[n] *= sc;
  }
  if (bad && (tid & 63) == 0) atomicAdd(a.bad, 1);
}
}  // namespace

hipError_t launch_den_general(const DenArgs& a, hipStream_t st) {
  if (a.phase_mask & 1) {
    hipLaunchKernelGGL(den_general_recursion_kernel, dim3(2 * a.B), dim3(kGNT), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (a.phase_mask & 2) {
    const int gx = (a.T + a.frames_per_block - 1) / a.frames_per_block;
    hipLaunchKernelGGL(den_general_gamma_kernel, dim3(gx, a.B), dim3(kGNT), 0, st, a);
    return hipGetLastError();
  }
  return hipSuccess;
}
}  // namespace pychain_hip
