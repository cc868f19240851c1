// num_kernels.hip - numerator (log domain, no leaky-HMM) forward-backward for gfx950.
// Replaces chain-log-domain-computation.cc + chain-log-domain-kernels.cu of the
// reference; shares no structure with them (the reference launches one kernel per
// frame with one thread per (sequence, state) and CAS-loop atomicLogAdd).
//
// One persistent workgroup per sequence.  The utterance's small graph is cached in LDS
// once, the time loops run inside the kernel, the per-frame state vectors stay in LDS,
// one state per thread; the frame's log-sum-exp is a wave64 shuffle reduction of
// (max, sum) pairs + one LDS hop.  Occupancies are accumulated in LDS as 64-bit
// fixed point (2^-56 resolution), which makes the per-pdf sums order-independent:
// deterministic without sorting, no float atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pychain_hip.h"
#include "device_utils.h"
#include "num_kernels.h"

namespace pychain_hip {
namespace {

constexpr int kNumNT = 256;
constexpr int kNumNW = kNumNT / 64;
constexpr float kMinLogDiff = -15.9423847198486328125f;   // log(FLT_EPSILON), base.h:12
constexpr float kFixScale = 72057594037927936.0f;          // 2^56
constexpr float kFixInv = 1.0f / 72057594037927936.0f;

// base.h:14-32 (same cut-off: the smaller term is dropped below log(FLT_EPSILON))
__device__ __forceinline__ float log_add(float x, float y) {
  const float mx = fmaxf(x, y), mn = fminf(x, y);
  const float d = mn - mx;                       // <= 0, or NaN for (-inf) - (-inf)
  return (d >= kMinLogDiff) ? mx + log1pf(expf(d)) : mx;
}

struct MS { float m, s; };                        // running (max, sum exp(v - max))
__device__ __forceinline__ MS ms_merge(MS a, MS b) {
  const float M = fmaxf(a.m, b.m);
  if (M == -INFINITY) return MS{-INFINITY, 0.f};
  return MS{M, a.s * expf(a.m - M) + b.s * expf(b.m - M)};
}
__device__ __forceinline__ MS ms_push(MS a, float v) { return ms_merge(a, MS{v, v == -INFINITY ? 0.f : 1.f}); }

// block-wide log-sum-exp of per-thread (m,s) pairs; red = float[2*kNumNW] in LDS.
// Contains two barriers; every thread returns the same value.
__device__ __forceinline__ float block_lse(MS v, float* red, int lane, int wave) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MS other{__shfl_xor(v.m, o, 64), __shfl_xor(v.s, o, 64)};
    v = ms_merge(v, other);
  }
  __syncthreads();                                // red may still be read from the previous call
  if (lane == 0) { red[wave] = v.m; red[kNumNW + wave] = v.s; }
  __syncthreads();
  MS t{-INFINITY, 0.f};
#pragma unroll
  for (int w = 0; w < kNumNW; w++) t = ms_merge(t, MS{red[w], red[kNumNW + w]});
  return t.m == -INFINITY ? -INFINITY : t.m + logf(t.s);
}

template <int VEC, int XCH>
__global__ __launch_bounds__(kNumNT) void num_kernel(const NumArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int L = (int)a.lengths[b];
  const int H = a.H, K = a.K, D = a.D, T = a.T, Dp = (D + 3) & ~3;
  const size_t g = (size_t)b * a.graph_stride;

  // ---- carve LDS
  char* p = smem_raw;
  unsigned long long* gam = reinterpret_cast<unsigned long long*>(p); p += sizeof(unsigned long long) * (size_t)Dp;
  float* xr0 = reinterpret_cast<float*>(p); p += 4 * (size_t)Dp;
  float* xr1 = reinterpret_cast<float*>(p); p += 4 * (size_t)Dp;
  const int Hq = (H + 3) & ~3;
  float* va = reinterpret_cast<float*>(p); p += 4 * (size_t)Hq;     // state vector ping
  float* vb = reinterpret_cast<float*>(p); p += 4 * (size_t)Hq;     // state vector pong
  float* arow = reinterpret_cast<float*>(p); p += 4 * (size_t)Hq;   // alpha(t,.) during the backward pass
  float* red = reinterpret_cast<float*>(p); p += 4 * 2 * kNumNW;
  const int2* in_be; const uint32_t* in_pk; const float* in_lp;
  const int2* out_be; const uint32_t* out_pk; const float* out_lp;
  {
    const int32_t* bt = a.bwd_trans + g * K * 3; const int32_t* ft = a.fwd_trans + g * K * 3;
    const int2* bi = reinterpret_cast<const int2*>(a.bwd_idx + g * H * 2);
    const int2* fi = reinterpret_cast<const int2*>(a.fwd_idx + g * H * 2);
    const float* bp = a.bwd_probs + g * K; const float* fp = a.fwd_probs + g * K;
    // graph -> LDS, packed (state | pdf << 16)
    int2* l_in_be = reinterpret_cast<int2*>(p); p += 8 * (size_t)H;
    int2* l_out_be = reinterpret_cast<int2*>(p); p += 8 * (size_t)H;
    uint32_t* l_in_pk = reinterpret_cast<uint32_t*>(p); p += 4 * (size_t)K;
    uint32_t* l_out_pk = reinterpret_cast<uint32_t*>(p); p += 4 * (size_t)K;
    float* l_in_lp = reinterpret_cast<float*>(p); p += 4 * (size_t)K;
    float* l_out_lp = reinterpret_cast<float*>(p); p += 4 * (size_t)K;
    for (int h = tid; h < H; h += kNumNT) { l_in_be[h] = bi[h]; l_out_be[h] = fi[h]; }
    for (int k = tid; k < K; k += kNumNT) {
      l_in_pk[k] = (uint32_t)bt[3 * k] | ((uint32_t)bt[3 * k + 2] << 16);      // (src, pdf)
      l_out_pk[k] = (uint32_t)ft[3 * k + 1] | ((uint32_t)ft[3 * k + 2] << 16); // (dst, pdf)
      l_in_lp[k] = bp[k]; l_out_lp[k] = fp[k];
    }
    in_be = l_in_be; out_be = l_out_be; in_pk = l_in_pk; out_pk = l_out_pk; in_lp = l_in_lp; out_lp = l_out_lp;
  }
  const float* init = a.initial + g * H;
  const float* fin = a.final_ + g * H;
  const float* xseq = a.x + (size_t)b * T * D;
  float* gseq = a.grad + (size_t)b * T * D;
  float* aws = a.alpha_ws + (size_t)b * (T + 1) * H;      // alpha(t,h), t = 0..L
  float* lws = a.logtot_ws + (size_t)b * (T + 1);         // alpha-sum(t) (log), t = 0..L

  for (int n = tid; n < Dp; n += kNumNT) gam[n] = 0ull;
  // AlphaFirstFrame, chain-log-domain-computation.cc:84-90 (alpha-sum(0) = 0 by fiat)
  for (int h = tid; h < H; h += kNumNT) { const float v = init[h]; va[h] = v; aws[h] = v; }
  if (tid == 0) lws[0] = 0.f;
  XRow<kNumNT, VEC, XCH> xq;
  xq.load(xseq, D, tid);
  xq.store(xr0, xseq, D, tid, kXClamp);
  __syncthreads();

  // ---- forward: AlphaGeneralFrame :93-159
  float logtot_prev = 0.f;
  double logsum = 0.0;                    // sum_{t<L} alpha-sum(t), :179-189
  for (int t = 1; t <= L; t++) {
    const float* vin = (t & 1) ? va : vb;
    float* vout = (t & 1) ? vb : va;
    const float* xcur = (t & 1) ? xr0 : xr1;        // row t-1
    float* xnext = (t & 1) ? xr1 : xr0;
    const bool have_next = t < L;
    const float* xrow_next = xseq + (size_t)(have_next ? t : 0) * D;
    if (have_next) xq.load(xrow_next, D, tid);
    MS ms{-INFINITY, 0.f};
    for (int h = tid; h < H; h += kNumNT) {
      const int2 be = in_be[h];
      float v = -INFINITY;
      for (int k = be.x; k < be.y; k++) {
        const uint32_t pk = in_pk[k];
        v = log_add(v, vin[pk & 0xffffu] + in_lp[k] + xcur[pk >> 16]);
      }
      v -= logtot_prev;
      vout[h] = v;
      aws[(size_t)t * H + h] = v;
      ms = ms_push(ms, v);
    }
    if (have_next) xq.store(xnext, xrow_next, D, tid, kXClamp);
    const float logtot = block_lse(ms, red, lane, wave);   // barriers inside publish vout / xnext
    if (tid == 0) {
      lws[t] = logtot;
      if (t < L && logtot != -INFINITY) logsum += (double)logtot;
    }
    logtot_prev = logtot;
  }

  // ---- ComputeTotLogLike :170-190 and BetaLastFrame :192-202
  const float* vL = (L & 1) ? vb : va;
  float* bnext = (L & 1) ? va : vb;        // beta(L) goes to the buffer alpha(L) does not occupy
  MS ms{-INFINITY, 0.f};
  for (int h = tid; h < H; h += kNumNT) ms = ms_push(ms, vL[h] + fin[h]);
  const float last = block_lse(ms, red, lane, wave);
  int bad = 0;
  if (tid == 0) {
    const float objf = (float)(logsum + (double)last);
    a.objf[b] = objf;
    if (!(objf - objf == 0.f)) bad = 1;
  }
  for (int h = tid; h < H; h += kNumNT) bnext[h] = fin[h] - last;
  // first backward frame needs x(L-1) and alpha(L-1)
  {
    const float* xrow = xseq + (size_t)(L - 1) * D;
    xq.load(xrow, D, tid);
    xq.store(xr0, xrow, D, tid, kXClamp);
    for (int h = tid; h < H; h += kNumNT) arow[h] = aws[(size_t)(L - 1) * H + h];
  }
  __syncthreads();

  // ---- backward: BetaGeneralFrame :204-271
  const float scale = a.grad_scale;
  float* bcur = (L & 1) ? vb : va;
  int step = 0;
  float inv_scale = lws[L - 1];                       // alpha-sum(t), :246 (prefetched one frame ahead)
  for (int t = L - 1; t >= 0; t--, step++) {
    const float* xcur = (step & 1) ? xr1 : xr0;
    float* xnext = (step & 1) ? xr0 : xr1;
    const bool have_next = t > 0;
    const float* xrow_next = xseq + (size_t)(have_next ? t - 1 : 0) * D;
    if (have_next) xq.load(xrow_next, D, tid);
    float anext[4];                                   // alpha(t-1,.) prefetch (H <= 4*NT fast path)
    const bool areg = H <= 4 * kNumNT;
    if (have_next && areg) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int h = c * kNumNT + tid;
        anext[c] = h < H ? aws[(size_t)(t - 1) * H + h] : 0.f;
      }
    }
    float inv_scale_next = 0.f;
    if (have_next) inv_scale_next = lws[t - 1];
    for (int h = tid; h < H; h += kNumNT) {
      const int2 be = out_be[h];
      const float ah = arow[h];
      float tot = -INFINITY;
      for (int k = be.x; k < be.y; k++) {
        const uint32_t pk = out_pk[k];
        const uint32_t pdf = pk >> 16;
        const float vf = out_lp[k] + bnext[pk & 0xffffu] + xcur[pdf] - inv_scale;
        tot = log_add(tot, vf);
        const float occ = expf(vf + ah);             // posterior of this arc at frame t, in [0,1]
        if (occ > 0.f) {
          if (occ <= 2.f) atomicAdd(&gam[pdf], (unsigned long long)(occ * kFixScale));
          else bad = 1;
        } else if (occ != 0.f) {
          bad = 1;                                    // NaN
        }
      }
      bcur[h] = tot;
    }
    __syncthreads();
    // frame's occupancy row -> HBM, accumulators reset
    float* grow = gseq + (size_t)t * D;
    for (int n = tid; n < D; n += kNumNT) {
      const unsigned long long u = gam[n];
      if (u != 0ull) gam[n] = 0ull;
      const float v = (float)u * kFixInv;
      if (a.grad_mode == PYCHAIN_HIP_GRAD_LOG) grow[n] = u ? logf(v) : -INFINITY;
      else if (a.grad_mode == PYCHAIN_HIP_GRAD_LINEAR) grow[n] = scale * v;
      else if (u) grow[n] += scale * v;
    }
    if (have_next) {
      xq.store(xnext, xrow_next, D, tid, kXClamp);
      if (areg) {
#pragma unroll
        for (int c = 0; c < 4; c++) { const int h = c * kNumNT + tid; if (h < H) arow[h] = anext[c]; }
      } else {
        for (int h = tid; h < H; h += kNumNT) arow[h] = aws[(size_t)(t - 1) * H + h];
      }
    }
    __syncthreads();
    inv_scale = inv_scale_next;
    float* tmp = bnext; bnext = bcur; bcur = tmp;
  }
  // padded frames: -inf (full_like(-inf), :57) / zero; ACCUM leaves them alone
  if (a.grad_mode != PYCHAIN_HIP_GRAD_ACCUM) {
    const float fill = a.grad_mode == PYCHAIN_HIP_GRAD_LOG ? -INFINITY : 0.f;
    for (size_t i = (size_t)L * D + tid; i < (size_t)T * D; i += kNumNT) gseq[i] = fill;
  }
  if (bad) atomicAdd(a.bad, 1);
}

template <int VEC, int XCH>
hipError_t launch_variant(const NumArgs& a, size_t lds, hipStream_t st) {
  auto k = num_kernel<VEC, XCH>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(a.B), dim3(kNumNT), lds, st, a);
  return hipGetLastError();
}

}  // namespace

size_t num_lds_bytes(int H, int K, int D) {
  const size_t Dp = (D + 3) & ~3, Hq = (H + 3) & ~3;
  return 8 * Dp + 8 * Dp + 12 * Hq + 4 * 2 * kNumNW + 16 * (size_t)H + 16 * (size_t)K + 64;
}

hipError_t launch_num(const NumArgs& a, hipStream_t st, const char** why) {
  const size_t lds = num_lds_bytes(a.H, a.K, a.D);
  if (lds > 160 * 1024) {
    *why = "numerator graph + nnet-output row do not fit the 160 KiB LDS of one CU";
    return hipErrorInvalidValue;
  }
  const int D = a.D;
  if (D % 4 == 0) {
    if (D <= 4 * 4 * kNumNT) return launch_variant<4, 4>(a, lds, st);
    if (D <= 4 * 12 * kNumNT) return launch_variant<4, 12>(a, lds, st);
  } else if (D <= 4 * kNumNT) {
    return launch_variant<1, 4>(a, lds, st);
  }
  return launch_variant<1, 0>(a, lds, st);
}

}  // namespace pychain_hip
