// num_kernels.hip - numerator (log domain, no leaky-HMM) forward-backward for gfx950.
// Replaces chain-log-domain-computation.cc + chain-log-domain-kernels.cu of the
// reference; shares no structure with them (the reference launches one kernel per
// frame with one thread per (sequence, state) and CAS-loop atomicLogAdd).
//
//   launch 1  num_fb_kernel    one persistent workgroup per sequence walks alpha forward and
//             beta backward in time inside the kernel: one state per thread, the utterance's
//             small graph cached in LDS/registers, ONE barrier per frame.  Log-probabilities
//             are carried in float64 WITHOUT per-frame renormalisation (the reference
//             renormalises fp32 values each frame, chain-log-domain-computation.cc:150-158;
//             in fp64 the raw log-probabilities - a few 1e4 in magnitude at T=1500 - keep
//             ~1e-12 absolute accuracy, so no block reduction sits on the sequential path).
//             exp/log act on small differences and run on the fp32 transcendental unit.
//             Output: per-arc occupancies occ[b,t,k] (fp32, linear domain) and the
//             sequence log-probability.  At T=1500 this is ~1e-5 from the fp64 evaluation of
//             the reference's equations, where the reference's own fp32 recursion is ~2e-4.
//   launch 2  num_emit_kernel  time-parallel: merges the per-arc occupancies of a frame by
//             pdf-id (64-bit fixed point in LDS: order-independent, hence deterministic) and
//             writes the gradient in the requested form: log (reference contract, -inf
//             where zero), linear, or accumulated into an existing dense gradient.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pychain_hip.h"
#include "device_utils.h"
#include "num_kernels.h"

namespace pychain_hip {
namespace {

constexpr int kFbNT = 512;                 // num_fb_kernel: one state per thread up to 512 states
constexpr int kFbNW = kFbNT / 64;
constexpr int kEmNT = 256;                 // num_emit_kernel
constexpr float kFixScale = 72057594037927936.0f;          // 2^56
constexpr float kFixInv = 1.0f / 72057594037927936.0f;
constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.693147182464599609375f;

__device__ __forceinline__ float fexp(float d) { return __builtin_amdgcn_exp2f(d * kLog2e); }   // d <= ~0
__device__ __forceinline__ float flog(float s) { return __builtin_amdgcn_logf(s) * kLn2; }       // s >= 1

struct ArcW { uint32_t pk; float lp; };    // pk = state | pdf << 16

// log-sum-exp accumulator over float64 terms with fp32 transcendentals: m = running max,
// s = sum exp(term - m).  value() = m + log s.
struct Lse {
  double m; float s;
  __device__ __forceinline__ void init() { m = -INFINITY; s = 0.f; }
  __device__ __forceinline__ void push(double e) {
    if (e > m) { s = (m == -INFINITY) ? 1.f : s * fexp((float)(m - e)) + 1.f; m = e; }
    else if (e != -INFINITY) { s += fexp((float)(e - m)); }
  }
  __device__ __forceinline__ double value() const { return m == -INFINITY ? -INFINITY : m + (double)flog(s); }
};

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

template <int VEC, int XCH>
__global__ __launch_bounds__(kFbNT) void num_fb_kernel(const NumArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int L = __builtin_amdgcn_readfirstlane((int)a.lengths[b]);
  const int H = a.H, K = a.K, D = a.D, T = a.T, Dp = (D + 3) & ~3;
  const size_t g = (size_t)b * a.graph_stride;

  // ---- LDS: state vectors (fp64, ping-pong), nnet-output rows (ping-pong), arcs, reductions
  char* p = smem_raw;
  const int Hq = (H + 1) & ~1;
  double* va = reinterpret_cast<double*>(p); p += 8 * (size_t)Hq;
  double* vb = reinterpret_cast<double*>(p); p += 8 * (size_t)Hq;
  double* redd = reinterpret_cast<double*>(p); p += 8 * 16;
  float* redf = reinterpret_cast<float*>(p); p += 4 * 16;
  float* xr0 = reinterpret_cast<float*>(p); p += 4 * (size_t)Dp;
  float* xr1 = reinterpret_cast<float*>(p); p += 4 * (size_t)Dp;
  ArcW* in_arc = reinterpret_cast<ArcW*>(p); p += 8 * (size_t)K;      // by destination: (src, pdf, lp)
  ArcW* out_arc = reinterpret_cast<ArcW*>(p); p += 8 * (size_t)K;     // by source:      (dst, pdf, lp)
  {
    const int32_t* bt = a.bwd_trans + g * K * 3; const int32_t* ft = a.fwd_trans + g * K * 3;
    const float* bp = a.bwd_probs + g * K; const float* fp = a.fwd_probs + g * K;
    for (int k = tid; k < K; k += kFbNT) {
      in_arc[k] = ArcW{(uint32_t)bt[3 * k] | ((uint32_t)bt[3 * k + 2] << 16), bp[k]};
      out_arc[k] = ArcW{(uint32_t)ft[3 * k + 1] | ((uint32_t)ft[3 * k + 2] << 16), fp[k]};
    }
  }
  const float* xseq = a.x + (size_t)b * T * D;
  double* aws = a.alpha_ws + (size_t)b * (T + 1) * H;     // alpha(t,h), t = 0..L  (fp64 log-prob)
  float* occ = a.occ_ws + (size_t)b * T * K;              // occ(t,k) per forward arc
  const int2* bi = reinterpret_cast<const int2*>(a.bwd_idx + g * H * 2);
  const int2* fi = reinterpret_cast<const int2*>(a.fwd_idx + g * H * 2);

  // this thread's state(s): h = tid (+ kFbNT, ... for graphs with more than 512 states)
  const int h0 = tid;
  const bool own = h0 < H;
  int2 ibe = make_int2(0, 0), obe = make_int2(0, 0);
  float fin0 = -INFINITY;
  if (own) { ibe = bi[h0]; obe = fi[h0]; fin0 = a.final_[g * H + h0]; }
  XRow<kFbNT, VEC, XCH> xq;
  xq.load(xseq, D, tid);
  // AlphaFirstFrame, chain-log-domain-computation.cc:84-90
  for (int h = tid; h < H; h += kFbNT) { const double v = (double)a.initial[g * H + h]; va[h] = v; aws[h] = v; }
  xq.store(xr0, xseq, D, tid, kXClamp);
  __syncthreads();
  // first two arcs of this thread's state in registers (the common left-to-right case needs no more)
  ArcW i0{0u, -INFINITY}, i1{0u, -INFINITY}, o0{0u, -INFINITY}, o1{0u, -INFINITY};
  if (own) {
    if (ibe.y - ibe.x > 0) i0 = in_arc[ibe.x];
    if (ibe.y - ibe.x > 1) i1 = in_arc[ibe.x + 1];
    if (obe.y - obe.x > 0) o0 = out_arc[obe.x];
    if (obe.y - obe.x > 1) o1 = out_arc[obe.x + 1];
  }

  // ---- forward: alpha(t,h) = LogSum_k alpha(t-1,src_k) + lp_k + x(t-1,pdf_k)   (:93-159, unnormalised)
  for (int t = 1; t <= L; t++) {
    const double* vin = (t & 1) ? va : vb;
    double* vout = (t & 1) ? vb : va;
    const float* xcur = (t & 1) ? xr0 : xr1;        // row t-1
    float* xnext = (t & 1) ? xr1 : xr0;
    const bool have_next = t < L;
    const float* xrow_next = xseq + (size_t)(have_next ? t : 0) * D;
    if (have_next) xq.load(xrow_next, D, tid);
    if (own) {
      Lse acc; acc.init();
      const double e0 = vin[i0.pk & 0xffffu] + ((double)i0.lp + (double)xcur[i0.pk >> 16]);
      const double e1 = vin[i1.pk & 0xffffu] + ((double)i1.lp + (double)xcur[i1.pk >> 16]);
      acc.push(e0); acc.push(e1);
      for (int k = ibe.x + 2; k < ibe.y; k++) {
        const ArcW w = in_arc[k];
        acc.push(vin[w.pk & 0xffffu] + ((double)w.lp + (double)xcur[w.pk >> 16]));
      }
      const double v = acc.value();
      vout[h0] = v;
      aws[(size_t)t * H + h0] = v;
    }
    for (int h = h0 + kFbNT; h < H; h += kFbNT) {             // graphs with more than 512 states
      const int2 be = bi[h];
      Lse acc; acc.init();
      for (int k = be.x; k < be.y; k++) {
        const ArcW w = in_arc[k];
        acc.push(vin[w.pk & 0xffffu] + ((double)w.lp + (double)xcur[w.pk >> 16]));
      }
      const double v = acc.value();
      vout[h] = v;
      aws[(size_t)t * H + h] = v;
    }
    if (have_next) xq.store(xnext, xrow_next, D, tid, kXClamp);
    __syncthreads();
  }

  // ---- total log-probability: LogSum_i alpha(L,i) + final(i)   (ComputeTotLogLike :170-190)
  const double* vL = (L & 1) ? vb : va;
  double* bnext = (L & 1) ? va : vb;        // beta(L) goes to the buffer alpha(L) does not occupy
  double mx = -INFINITY;
  for (int h = tid; h < H; h += kFbNT) mx = fmax(mx, vL[h] + (double)a.final_[g * H + h]);
  mx = wave_max(mx);
  if (tid < 16) { redd[tid] = -INFINITY; redf[tid] = 0.f; }
  __syncthreads();
  if (lane == 0) redd[wave] = mx;
  __syncthreads();
  double gm = -INFINITY;
#pragma unroll
  for (int w = 0; w < kFbNW; w++) gm = fmax(gm, redd[w]);
  float se = 0.f;
  if (gm != -INFINITY)
    for (int h = tid; h < H; h += kFbNT) se += fexp((float)(vL[h] + (double)a.final_[g * H + h] - gm));
  se = wave_sum(se);
  if (lane == 0) redf[wave] = se;
  __syncthreads();
  const float stot = dpp_row_sum(redf[lane & 15]);
  const double logp = gm == -INFINITY ? -INFINITY : gm + (double)flog(stot);
  if (tid == 0) {
    const float objf = (float)logp;
    a.objf[b] = objf;
    if (!(objf - objf == 0.f)) atomicAdd(a.bad, 1);
  }
  // BetaLastFrame :192-202 (unnormalised: beta(L,i) = final(i); the 1/P factor enters the occupancy)
  for (int h = tid; h < H; h += kFbNT) bnext[h] = (double)a.final_[g * H + h];
  {
    const float* xrow = xseq + (size_t)(L - 1) * D;
    xq.load(xrow, D, tid);
    xq.store(xr0, xrow, D, tid, kXClamp);
  }
  double a_cur = own ? aws[(size_t)(L - 1) * H + h0] : 0.0;   // alpha(t,h0): written by this very thread
  __syncthreads();

  // ---- backward: beta(t,h) = LogSum_k lp_k + beta(t+1,dst_k) + x(t,pdf_k);  occ = exp(alpha + term - logP)
  double* bcur = (L & 1) ? vb : va;
  int step = 0;
  for (int t = L - 1; t >= 0; t--, step++) {
    const float* xcur = (step & 1) ? xr1 : xr0;
    float* xnext = (step & 1) ? xr0 : xr1;
    const bool have_next = t > 0;
    const float* xrow_next = xseq + (size_t)(have_next ? t - 1 : 0) * D;
    if (have_next) xq.load(xrow_next, D, tid);
    double a_next = 0.0;
    if (have_next && own) a_next = aws[(size_t)(t - 1) * H + h0];
    float* orow = occ + (size_t)t * K;
    if (own) {
      Lse acc; acc.init();
      const double base = a_cur - logp;
      const double e0 = bnext[o0.pk & 0xffffu] + ((double)o0.lp + (double)xcur[o0.pk >> 16]);
      const double e1 = bnext[o1.pk & 0xffffu] + ((double)o1.lp + (double)xcur[o1.pk >> 16]);
      acc.push(e0); acc.push(e1);
      if (obe.y - obe.x > 0) orow[obe.x] = fexp((float)(base + e0));
      if (obe.y - obe.x > 1) orow[obe.x + 1] = fexp((float)(base + e1));
      for (int k = obe.x + 2; k < obe.y; k++) {
        const ArcW w = out_arc[k];
        const double e = bnext[w.pk & 0xffffu] + ((double)w.lp + (double)xcur[w.pk >> 16]);
        acc.push(e);
        orow[k] = fexp((float)(base + e));
      }
      bcur[h0] = acc.value();
    }
    for (int h = h0 + kFbNT; h < H; h += kFbNT) {
      const int2 be = fi[h];
      const double base = aws[(size_t)t * H + h] - logp;
      Lse acc; acc.init();
      for (int k = be.x; k < be.y; k++) {
        const ArcW w = out_arc[k];
        const double e = bnext[w.pk & 0xffffu] + ((double)w.lp + (double)xcur[w.pk >> 16]);
        acc.push(e);
        orow[k] = fexp((float)(base + e));
      }
      bcur[h] = acc.value();
    }
    if (have_next) xq.store(xnext, xrow_next, D, tid, kXClamp);
    a_cur = a_next;
    __syncthreads();
    double* tmp = bnext; bnext = bcur; bcur = tmp;
  }
}

// ------------------------------------------------------------------------------------
// launch 2: per-arc occupancies -> gradient rows (time-parallel)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kEmNT) void num_emit_kernel(const NumArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int L = (int)a.lengths[b];
  const int K = a.K, D = a.D, T = a.T, Dp = (D + 3) & ~3;
  const int t_begin = blockIdx.x * a.frames_per_block;
  const int t_end = min(t_begin + a.frames_per_block, T);
  float* gseq = a.grad + (size_t)b * T * D;
  const int mode = a.grad_mode;
  const float gscale = a.grad_scale_dev ? a.grad_scale * *a.grad_scale_dev : a.grad_scale;
  const float fill = mode == PYCHAIN_HIP_GRAD_LOG ? -INFINITY : 0.f;
  const int t_live_end = min(t_end, L);
  if (t_begin < L) {
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem_raw);   // [Dp]
    int* s_kmax_p = reinterpret_cast<int*>(acc + Dp);                              // [4] (all LDS dynamic: base stays 16-B aligned)
    uint16_t* pdf = reinterpret_cast<uint16_t*>(s_kmax_p + 4);                     // [K]
    const size_t g = (size_t)b * a.graph_stride;
    const int32_t* ft = a.fwd_trans + g * K * 3;
    const int2* fi = reinterpret_cast<const int2*>(a.fwd_idx + g * a.H * 2);
    // arcs that no state indexes (batch padding, pychain/graph.py:132-139) carry no occupancy
    int kmax = 0;
    for (int h = tid; h < a.H; h += kEmNT) kmax = max(kmax, fi[h].y);
    if (tid == 0) *s_kmax_p = 0;
    __syncthreads();
    atomicMax(s_kmax_p, kmax);
    for (int n = tid; n < Dp; n += kEmNT) acc[n] = 0ull;
    for (int k = tid; k < K; k += kEmNT) pdf[k] = (uint16_t)ft[3 * k + 2];
    __syncthreads();
    const int Kused = *s_kmax_p;
    const float* occ = a.occ_ws + (size_t)b * T * K;
    int bad = 0;
    for (int t = t_begin; t < t_live_end; t++) {
      const float* orow = occ + (size_t)t * K;
      float* grow = gseq + (size_t)t * D;
      for (int k = tid; k < Kused; k += kEmNT) {
        const float v = orow[k];
        if (v > 0.f) {
          if (v <= 2.f) atomicAdd(&acc[pdf[k]], (unsigned long long)(v * kFixScale));
          else bad = 1;
        } else if (v != 0.f) {
          bad = 1;                                  // NaN
        }
      }
      __syncthreads();
      if (mode == PYCHAIN_HIP_GRAD_ACCUM) {
        for (int k = tid; k < Kused; k += kEmNT) {
          const int n = pdf[k];
          const unsigned long long u = atomicExch(&acc[n], 0ull);   // exactly one arc per pdf sees the merged sum
          if (u) grow[n] += gscale * ((float)u * kFixInv);
        }
      } else {
        for (int n = tid; n < D; n += kEmNT) {
          const unsigned long long u = acc[n];
          if (u) acc[n] = 0ull;
          const float v = (float)u * kFixInv;
          grow[n] = mode == PYCHAIN_HIP_GRAD_LOG ? (u ? logf(v) : -INFINITY) : gscale * v;
        }
      }
      __syncthreads();
    }
    if (bad) atomicAdd(a.bad, 1);
  }
  // padded frames: -inf (full_like(-inf), :57) / zero; ACCUM leaves them alone
  if (mode != PYCHAIN_HIP_GRAD_ACCUM) {
    const int t0 = max(t_begin, t_live_end);
    for (size_t i = (size_t)t0 * D + tid; i < (size_t)t_end * D; i += kEmNT) gseq[i] = fill;
  }
}

template <int VEC, int XCH>
hipError_t launch_fb(const NumArgs& a, size_t lds, hipStream_t st) {
  auto k = num_fb_kernel<VEC, XCH>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(a.B), dim3(kFbNT), lds, st, a);
  return hipGetLastError();
}

}  // namespace

size_t num_fb_lds_bytes(int H, int K, int D) {
  const size_t Dp = (D + 3) & ~3, Hq = (H + 1) & ~1;
  return 16 * Hq + 8 * 16 + 4 * 16 + 8 * Dp + 16 * (size_t)K + 64;
}
size_t num_emit_lds_bytes(int K, int D) {
  const size_t Dp = (D + 3) & ~3;
  return 8 * Dp + 16 + 2 * (size_t)K + 64;
}

hipError_t launch_num_fb(const NumArgs& a, hipStream_t st, const char** why) {
  const size_t lds = num_fb_lds_bytes(a.H, a.K, a.D);
  if (lds > 160 * 1024) {
    *why = "numerator graph + nnet-output rows do not fit the 160 KiB LDS of one CU";
    return hipErrorInvalidValue;
  }
  const int D = a.D;
  if (D % 4 == 0) {
    if (D <= 4 * 2 * kFbNT) return launch_fb<4, 2>(a, lds, st);
    if (D <= 4 * 8 * kFbNT) return launch_fb<4, 8>(a, lds, st);
  } else if (D <= 8 * kFbNT) {
    return launch_fb<1, 8>(a, lds, st);
  }
  return launch_fb<1, 0>(a, lds, st);
}

hipError_t launch_num_emit(const NumArgs& a, hipStream_t st, const char** why) {
  const size_t lds = num_emit_lds_bytes(a.K, a.D);
  if (lds > 160 * 1024) {
    *why = "pdf accumulators do not fit the 160 KiB LDS of one CU";
    return hipErrorInvalidValue;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(num_emit_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const int gx = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  hipLaunchKernelGGL(num_emit_kernel, dim3(gx, a.B), dim3(kEmNT), lds, st, a);
  return hipGetLastError();
}

}  // namespace pychain_hip
