// num_kernels.hip - numerator (log domain, no leaky-HMM) forward-backward for gfx950.
// Replaces chain-log-domain-computation.cc + chain-log-domain-kernels.cu of the
// reference; shares no structure with them (the reference launches one kernel per
// frame with one thread per (sequence, state) and CAS-loop atomicLogAdd).
//
//   launch 1  num_fb_kernel    2B persistent workgroups, one per (sequence, direction): block b
//             walks alpha forward in time, block B+b walks beta backward, CONCURRENTLY - the
//             log-probabilities are carried in float64 WITHOUT per-frame renormalisation (the
//             reference renormalises fp32 values each frame, chain-log-domain-computation.cc:
//             150-158; in fp64 the raw log-probabilities - a few 1e4 in magnitude at T=1500 -
//             keep ~1e-12 absolute accuracy), so neither direction needs the other's
//             normaliser and no block reduction sits on the sequential path.  One state per
//             thread, the utterance's small graph cached in LDS/registers, ONE barrier per
//             frame; exp/log act on small differences and run on the fp32 transcendental unit.
//             Output: every alpha(t,.) and beta(t,.) row (fp64) and the sequence log-probability.
//   launch 2  num_occ_kernel   time-parallel over (sequence, frame chunk): occupancy of every
//             arc  exp(alpha(t,src) + lp + x(t,pdf) + beta(t+1,dst) - logP), merged by pdf-id
//             in 64-bit fixed point in LDS (order-independent, hence deterministic), written
//             in the requested form: log (reference contract, -inf where zero), linear,
//             accumulated into an existing dense gradient, or as compact rows over the
//             sequence's distinct pdf-ids (what the fused ChainLoss folds into the
//             denominator's occupancy pass).  At T=1500 the result is ~1e-5 from the fp64
//             evaluation of the reference's equations; the reference's own fp32 recursion ~2e-4.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pychain_hip.h"
#include "device_utils.h"
#include "common.h"
#include "num_kernels.h"

namespace pychain_hip {
namespace {

constexpr int kFbNT = 512;                 // num_fb_kernel: one state per thread up to 512 states
constexpr int kFbNW = kFbNT / 64;
constexpr int kOcNT = 256;                 // num_occ_kernel, num_prep_kernel
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

// LD > 0: the workgroup has LD extra threads (whole waves) that do nothing but stage nnet-output rows
// (request the row of step s+2, clamp and store the row of step s+1 to LDS, barrier).  Their code has no
// branch between a load and its use and no stores to memory, so the compiler can wait for exactly the
// loads it needs (s_waitcnt vmcnt(XCH)); in the LD = 0 form the same wave also issues the row and
// log-share stores under divergent branches, every wait degenerates to vmcnt(0), and each step lasts a
// round trip to HBM (1461 us for T = 1500; XCH then counts chunks of kFbNT threads).
constexpr int kFbLd = 128;
// XH: 2-byte network output (NumArgs::x_half; float4-chunk forms only): four elements = 8 bytes per thread and chunk, converted
// as they arrive
template <int VEC, int XCH, int LD, bool XH = false>
__global__ __launch_bounds__(kFbNT + LD) void num_fb_kernel(const NumArgs a) {
  static_assert(!XH || (VEC == 4 && XCH > 0), "2-byte rows: chunks of four elements through registers");
  constexpr size_t kXe = XH ? 2 : 4;
  const bool bf16 = a.x_half == kXBf16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool fwd = blockIdx.x < (unsigned)a.B;
  const int b = fwd ? blockIdx.x : blockIdx.x - a.B;
  const int L = __builtin_amdgcn_readfirstlane(seq_len(a.lengths, b, a.T));
  const int H = a.H, K = a.K, D = a.D, T = a.T, Dp = (D + 3) & ~3;
  const size_t g = (size_t)b * a.graph_stride;

  // ---- LDS: state vectors (fp64, ping-pong), nnet-output rows (ping-pong), this direction's arcs
  char* p = smem_raw;
  const int Hq = (H + 1) & ~1;
  double* va = reinterpret_cast<double*>(p); p += 8 * (size_t)Hq;
  double* vb = reinterpret_cast<double*>(p); p += 8 * (size_t)Hq;
  double* redd = reinterpret_cast<double*>(p); p += 8 * 16;
  float* redf = reinterpret_cast<float*>(p); p += 4 * 16;
  float* xr0 = reinterpret_cast<float*>(p); p += 4 * (size_t)Dp;
  float* xr1 = reinterpret_cast<float*>(p); p += 4 * (size_t)Dp;
  ArcW* arc = reinterpret_cast<ArcW*>(p); p += 8 * (size_t)K;   // fwd: by destination (src, pdf, lp); bwd: by source (dst, pdf, lp)
  {
    const int32_t* tr = (fwd ? a.bwd_trans : a.fwd_trans) + g * K * 3;
    const float* pr = (fwd ? a.bwd_probs : a.fwd_probs) + g * K;
    for (int k = tid; k < K; k += kFbNT + LD)
      arc[k] = ArcW{(uint32_t)tr[3 * k + (fwd ? 0 : 1)] | ((uint32_t)tr[3 * k + 2] << 16), pr[k]};
  }
  const float* xseq = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x) + (size_t)b * T * D * kXe);
  const XBuf xbuf = make_xbuf(xseq, (size_t)T * D * kXe);
  // row `t` of the sequence into a staging register set (by pointer / through the buffer descriptor)
  auto xload = [&](auto& xr, int t, int ti) {
    if constexpr (XH) xr.load_h(reinterpret_cast<const char*>(xseq) + (size_t)t * D * 2, D, ti);      // (raw: converted in stage())
    else xr.load(xseq + (size_t)t * D, D, ti);
  };
  auto xload_row = [&](auto& xr, int t, int ti) {
    if constexpr (XH) xr.load_row_h(xbuf, t, D, ti);
    else xr.load_row(xbuf, t, D, ti);
  };
  double* rows = (fwd ? a.alpha_ws : a.beta_ws) + (size_t)b * (T + 1) * H;     // row t = alpha(t,.) / beta(t,.), fp64 log-prob
  const int2* idx = reinterpret_cast<const int2*>((fwd ? a.bwd_idx : a.fwd_idx) + g * H * 2);

  // this thread's state(s): h = tid (+ kFbNT, ... for graphs with more than 512 states)
  const int h0 = tid;
  const bool own = h0 < H && tid < kFbNT;
  int2 be = make_int2(0, 0);
  if (own) be = idx[h0];
  XRow<LD ? LD : kFbNT, VEC, XCH> xq;
  const int xt = LD ? tid - kFbNT : tid;            // this thread's index among the threads that stage rows
  // a row, clamped, into LDS; NaNs kept only where nobody else watches for them (NumArgs::watch_nan)
  const bool watch_nan = a.watch_nan != 0;
  auto stage = [&](auto& xr, float* lds, const float* row, int t) {
    if constexpr (XH) xr.convert_h(bf16);            // (2-byte rows stay raw in their registers until they are used: here)
    if constexpr (XCH > 0) {
      if (watch_nan) xr.store(lds, row, D, t, kXClamp);
      else xr.template store_mode<kXClamp>(lds, D, t);
    } else {
      xr.store(lds, row, D, t, kXClamp);
    }
  };
  {
    const float* xrow = XH ? nullptr : xseq + (size_t)(fwd ? 0 : L - 1) * D;    // (only the register-less form reads it again)
    if (!LD || tid >= kFbNT) xload(xq, fwd ? 0 : L - 1, xt);
    // AlphaFirstFrame :84-90 / BetaLastFrame :192-202 (unnormalised: beta(L,i) = final(i); 1/P enters the occupancy)
    if (tid < kFbNT)
      for (int h = tid; h < H; h += kFbNT) {
        const double v = (double)(fwd ? a.initial : a.final_)[g * H + h];
        va[h] = v; rows[(size_t)(fwd ? 0 : L) * H + h] = v;
      }
    if (!LD || tid >= kFbNT) stage(xq, xr0, xrow, xt);
  }
  __syncthreads();
  if constexpr (LD > 0) {
    if (tid >= kFbNT) {
      // ---- row-staging waves: step s gathers from xr0 (s odd) / xr1 (s even); rows past the last one are
      // re-reads of a valid row into a buffer nobody gathers from (no branch around a load or its use)
      static_assert(VEC == 4 && XCH > 0, "row-staging waves use the float4 buffer-load form");
      XRow<LD, VEC, XCH> xq2;
      auto row_of_step = [&](int s) { return fwd ? min(s, L) - 1 : max(L - s, 0); };
      xload_row(xq, row_of_step(2), xt);
      for (int s = 1; s <= L; s += 2) {
        xload_row(xq2, row_of_step(s + 2), xt);
        stage(xq, xr1, nullptr, xt);                 // row of step s+1
        __syncthreads();
        if (s + 1 <= L) {
          xload_row(xq, row_of_step(s + 3), xt);
          stage(xq2, xr0, nullptr, xt);              // row of step s+2
          __syncthreads();
        }
      }
      return;
    }
  }
  // first two arcs of this thread's state in registers (the common left-to-right case needs no more)
  ArcW w0{0u, -INFINITY}, w1{0u, -INFINITY};
  if (own) {
    if (be.y - be.x > 0) w0 = arc[be.x];
    if (be.y - be.x > 1) w1 = arc[be.x + 1];
  }

  // fwd: alpha(t,h) = LogSum_k alpha(t-1,src_k) + lp_k + x(t-1,pdf_k), t = 1..L     (:93-159, unnormalised)
  // bwd: beta(t,h)  = LogSum_k lp_k + beta(t+1,dst_k) + x(t,pdf_k),    t = L-1..0   (:204-271, unnormalised)
  for (int s = 1; s <= L; s++) {
    const double* vin = (s & 1) ? va : vb;
    double* vout = (s & 1) ? vb : va;
    const float* xcur = (s & 1) ? xr0 : xr1;
    float* xnext = (s & 1) ? xr1 : xr0;
    const bool have_next = s < L;
    const int t_next = have_next ? (fwd ? s : L - 1 - s) : 0;
    const float* xrow_next = XH ? nullptr : xseq + (size_t)t_next * D;
    const size_t trow = (size_t)(fwd ? s : L - s) * H;
    if (!LD && have_next) {
      // buffer form: no address VGPR is written per step, so the load does not wait for this step's row stores
      if constexpr (VEC == 4 && XCH > 0) xload_row(xq, t_next, tid);
      else xq.load(xrow_next, D, tid);
    }
    // bwd also writes, per arc, its log-share of its source state's beta:  r_k(t) = term_k - beta(t,h) <= 0.
    // That is all the occupancy pass needs from this frame's nnet-output row:
    //   occupancy = exp(alpha(t,src) + beta(t,src) - logP + r_k(t)),  a small fp32 number instead of
    // a second, scattered read of the row.
    float* frow = fwd ? nullptr : a.frac_ws + ((size_t)b * T + (L - s)) * K;
    if (own) {
      Lse acc; acc.init();
      const double e0 = vin[w0.pk & 0xffffu] + ((double)w0.lp + (double)xcur[w0.pk >> 16]);
      const double e1 = vin[w1.pk & 0xffffu] + ((double)w1.lp + (double)xcur[w1.pk >> 16]);
      acc.push(e0); acc.push(e1);
      for (int k = be.x + 2; k < be.y; k++) {
        const ArcW w = arc[k];
        acc.push(vin[w.pk & 0xffffu] + ((double)w.lp + (double)xcur[w.pk >> 16]));
      }
      const double v = acc.value();
      vout[h0] = v;
      rows[trow + h0] = v;
      if (!fwd) {
        if (be.y - be.x > 0) frow[be.x] = (float)(e0 - v);
        if (be.y - be.x > 1) frow[be.x + 1] = (float)(e1 - v);
        for (int k = be.x + 2; k < be.y; k++) {
          const ArcW w = arc[k];
          frow[k] = (float)(vin[w.pk & 0xffffu] + ((double)w.lp + (double)xcur[w.pk >> 16]) - v);
        }
      }
    }
    for (int h = h0 + kFbNT; h < H; h += kFbNT) {             // graphs with more than 512 states
      const int2 e2 = idx[h];
      Lse acc; acc.init();
      for (int k = e2.x; k < e2.y; k++) {
        const ArcW w = arc[k];
        acc.push(vin[w.pk & 0xffffu] + ((double)w.lp + (double)xcur[w.pk >> 16]));
      }
      const double v = acc.value();
      vout[h] = v;
      rows[trow + h] = v;
      if (!fwd)
        for (int k = e2.x; k < e2.y; k++) {
          const ArcW w = arc[k];
          frow[k] = (float)(vin[w.pk & 0xffffu] + ((double)w.lp + (double)xcur[w.pk >> 16]) - v);
        }
    }
    if (!LD && have_next) stage(xq, xnext, xrow_next, tid);
    __syncthreads();
  }
  if (!fwd) return;

  // ---- total log-probability: LogSum_i alpha(L,i) + final(i)   (ComputeTotLogLike :170-190)
  const double* vL = (L & 1) ? vb : va;
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
    a.logp_ws[b] = logp;
    if (!(objf - objf == 0.f) || seq_len_bad(a.lengths, b, a.T)) atomicAdd(a.bad, 1);
  }
}

// ------------------------------------------------------------------------------------
// the distinct pdf-ids of every sequence's numerator graph, ascending: upd[b][u], ucount[b]
// (compact rows are indexed by u).  One workgroup per sequence, once per call.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kOcNT) void num_prep_kernel(const NumArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int K = a.K, D = a.D;
  int* used = reinterpret_cast<int*>(smem_raw);          // [D]  1 = some arc emits this pdf
  int* s_misc = used + D;                                // [0] = used arcs, [1..] scan scratch
  const size_t g = (size_t)b * a.graph_stride;
  const int32_t* ft = a.fwd_trans + g * K * 3;
  const int2* fi = reinterpret_cast<const int2*>(a.fwd_idx + g * a.H * 2);
  int kmax = 0;
  for (int h = tid; h < a.H; h += kOcNT) kmax = max(kmax, fi[h].y);
  if (tid == 0) s_misc[0] = 0;
  for (int n = tid; n < D; n += kOcNT) used[n] = 0;
  __syncthreads();
  atomicMax(&s_misc[0], kmax);
  __syncthreads();
  const int Kused = s_misc[0];
  for (int k = tid; k < Kused; k += kOcNT) used[ft[3 * k + 2]] = 1;
  __syncthreads();
  // exclusive scan of `used` in chunks of kOcNT consecutive pdfs per thread
  const int per = (D + kOcNT - 1) / kOcNT;
  int cnt = 0;
  for (int n = tid * per; n < min(D, (tid + 1) * per); n++) cnt += used[n];
  int* scan = s_misc + 4;
  scan[tid] = cnt;
  __syncthreads();
  if (tid == 0) { int run = 0; for (int i = 0; i < kOcNT; i++) { const int c = scan[i]; scan[i] = run; run += c; } a.ucount_ws[b] = run; }
  __syncthreads();
  int u = scan[tid];
  int32_t* upd = a.upd_ws + (size_t)b * K;
  for (int n = tid * per; n < min(D, (tid + 1) * per); n++) if (used[n]) { upd[u] = n; used[n] = u++; }
  __syncthreads();
  // every arc's row in the compact layout (index of its pdf among the distinct ones); unused arcs: -1
  int32_t* uidx = a.uidx_ws + (size_t)b * K;
  for (int k = tid; k < K; k += kOcNT) uidx[k] = k < Kused ? used[ft[3 * k + 2]] : -1;
}

// ------------------------------------------------------------------------------------
// launch 2: occupancies from the stored alpha / beta rows -> gradient rows (time-parallel)
// ------------------------------------------------------------------------------------
constexpr int kGradCompact = 3;            // internal mode: rows_ws[b,t,u] = occupancy of the u-th distinct pdf
__global__ __launch_bounds__(kOcNT) void num_occ_kernel(const NumArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int L = seq_len(a.lengths, b, a.T);
  const int K = a.K, D = a.D, T = a.T, H = a.H, Dp = (D + 3) & ~3;
  const int t_begin = blockIdx.x * a.frames_per_block;
  const int t_end = min(t_begin + a.frames_per_block, T);
  float* gseq = a.grad + (size_t)b * T * D;
  const int mode = a.grad_mode;
  const float gscale = a.grad_scale_dev ? a.grad_scale * *a.grad_scale_dev : a.grad_scale;
  const float fill = mode == PYCHAIN_HIP_GRAD_LOG ? -INFINITY : 0.f;
  const int t_live_end = min(t_end, L);
  if (t_begin < L) {
    char* p = smem_raw;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(p); p += 8 * (size_t)Dp;     // [Dp]
    double* arow = reinterpret_cast<double*>(p); p += 8 * (size_t)((H + 1) & ~1);                  // alpha(t,.)
    double* brow = reinterpret_cast<double*>(p); p += 8 * (size_t)((H + 1) & ~1);                  // beta(t+1,.)
    int* s_kmax_p = reinterpret_cast<int*>(p);
    float* s_fsum = reinterpret_cast<float*>(p + 8); p += 16;
    uint32_t* sd = reinterpret_cast<uint32_t*>(p); p += 4 * (size_t)K;                             // src | dst << 16
    float* lp = reinterpret_cast<float*>(p); p += 4 * (size_t)K;
    uint16_t* pdf = reinterpret_cast<uint16_t*>(p);                                                // [K]
    const size_t g = (size_t)b * a.graph_stride;
    const int32_t* ft = a.fwd_trans + g * K * 3;
    const float* fp = a.fwd_probs + g * K;
    const int2* fi = reinterpret_cast<const int2*>(a.fwd_idx + g * H * 2);
    // arcs that no state indexes (batch padding, pychain/graph.py:132-139) carry no occupancy
    int kmax = 0;
    for (int h = tid; h < H; h += kOcNT) kmax = max(kmax, fi[h].y);
    if (tid == 0) *s_kmax_p = 0;
    __syncthreads();
    atomicMax(s_kmax_p, kmax);
    for (int n = tid; n < Dp; n += kOcNT) acc[n] = 0ull;
    for (int k = tid; k < K; k += kOcNT) {
      sd[k] = (uint32_t)ft[3 * k] | ((uint32_t)ft[3 * k + 1] << 16);
      pdf[k] = (uint16_t)ft[3 * k + 2];
      lp[k] = fp[k];
    }
    __syncthreads();
    const int Kused = *s_kmax_p;
    const double logp = a.logp_ws[b];
    const double* aws = a.alpha_ws + (size_t)b * (T + 1) * H;
    const double* bws = a.beta_ws + (size_t)b * (T + 1) * H;
    const float* fseq = a.frac_ws + (size_t)b * T * K;
    const int U = mode == kGradCompact ? a.ucount_ws[b] : 0;
    const int32_t* upd = a.upd_ws + (size_t)b * K;
    int bad = 0;
    // Software pipeline over frames: the rows and nnet-output values of frame t+1 are loaded into
    // registers while frame t is evaluated (kNR row elements and kNX arcs per thread are staged;
    // larger graphs read the rest directly).
    constexpr int kNR = 2, kNX = 4;
    double pa[kNR], pb[kNR];
    float px[kNX];
#define NUM_OCC_PREFETCH(t)                                                                        \
    do {                                                                                           \
      _Pragma("unroll") for (int i = 0; i < kNR; i++) {                                            \
        const int h = tid + i * kOcNT;                                                             \
        if (h < H) { pa[i] = aws[(size_t)(t) * H + h]; pb[i] = bws[(size_t)(t) * H + h]; }         \
      }                                                                                            \
      _Pragma("unroll") for (int i = 0; i < kNX; i++) {                                            \
        const int k = tid + i * kOcNT;                                                             \
        if (k < Kused) px[i] = fseq[(size_t)(t) * K + k];                                          \
      }                                                                                            \
    } while (0)
    NUM_OCC_PREFETCH(t_begin);
    for (int t = t_begin; t < t_live_end; t++) {
#pragma unroll
      for (int i = 0; i < kNR; i++) { const int h = tid + i * kOcNT; if (h < H) { arow[h] = pa[i]; brow[h] = pb[i]; } }
      for (int h = tid + kNR * kOcNT; h < H; h += kOcNT) { arow[h] = aws[(size_t)t * H + h]; brow[h] = bws[(size_t)t * H + h]; }
      float xcur[kNX];
      float fsum = 0.f;                                 // this thread's share of the frame's occupancy total
#pragma unroll
      for (int i = 0; i < kNX; i++) xcur[i] = px[i];
      if (t + 1 < t_live_end) NUM_OCC_PREFETCH(t + 1);
      const bool check = t == 0 || a.check_all;         // the reference's `ok` (NumArgs::check_all)
      if (check && tid == 0) *s_fsum = 0.f;
      __syncthreads();
      const float* frow = fseq + (size_t)t * K;
      float* grow = gseq + (size_t)t * D;
      // BetaGeneralFrame :204-271: occupancy = exp(alpha(t,src) + [lp + x(t,pdf) + beta(t+1,dst)] - logP), the
      // bracket being beta(t,src) + r_k(t) with the arc's log-share r from the backward pass (num_fb_kernel)
#define NUM_OCC_ARC(k, rk)                                                                         \
      do {                                                                                         \
        const int src = sd[k] & 0xffffu;                                                           \
        const int n = pdf[k];                                                                      \
        /* log occupancy of the source state: formed in fp64, rounded ONCE to fp32 (as num_occ_wave_kernel keeps it in LDS) */ \
        const float st = (float)(arow[src] + brow[src] - logp);                                    \
        const float v = st == -INFINITY ? 0.f : fexp(st + (rk));                                   \
        fsum += v;                                                                                 \
        if (v > 0.f) {                                                                             \
          if (v <= 2.f) atomicAdd(&acc[n], (unsigned long long)(v * kFixScale));                   \
          else bad = 1;                                                                            \
        } else if (v != 0.f) {                                                                     \
          bad = 1;                                  /* NaN */                                      \
        }                                                                                          \
      } while (0)
#pragma unroll
      for (int i = 0; i < kNX; i++) { const int k = tid + i * kOcNT; if (k < Kused) NUM_OCC_ARC(k, xcur[i]); }
      for (int k = tid + kNX * kOcNT; k < Kused; k += kOcNT) NUM_OCC_ARC(k, frow[k]);
#undef NUM_OCC_ARC
      if (check) {
        const float ws = wave_sum(fsum);
        if ((tid & 63) == 0) atomicAdd(s_fsum, ws);
      }
      __syncthreads();
      if (check && tid == 0 && !(fabsf(*s_fsum - 1.f) <= 0.05f)) bad = 1;
      if (mode == PYCHAIN_HIP_GRAD_ACCUM) {
        for (int k = tid; k < Kused; k += kOcNT) {
          const int n = pdf[k];
          const unsigned long long u = atomicExch(&acc[n], 0ull);   // exactly one arc per pdf sees the merged sum
          if (u) grow[n] = mul_add_rn((float)u * kFixInv, gscale, grow[n]);   // (no fma: same bits as the folded form)
        }
      } else if (mode == kGradCompact) {
        float* crow = a.rows_ws + ((size_t)b * T + t) * K;
        for (int u = tid; u < U; u += kOcNT) {
          const int n = upd[u];
          const unsigned long long v = acc[n];
          if (v) acc[n] = 0ull;
          crow[u] = (float)v * kFixInv;
        }
      } else {
        for (int n = tid; n < D; n += kOcNT) {
          const unsigned long long u = acc[n];
          if (u) acc[n] = 0ull;
          const float v = (float)u * kFixInv;
          grow[n] = mode == PYCHAIN_HIP_GRAD_LOG ? (u ? logf(v) : -INFINITY) : gscale * v;
        }
      }
      __syncthreads();
    }
#undef NUM_OCC_PREFETCH
    if (bad) atomicAdd(a.bad, 1);
  }
  // padded frames: -inf (full_like(-inf), :57) / zero; ACCUM and the compact rows leave them alone
  if (mode == PYCHAIN_HIP_GRAD_LOG || mode == PYCHAIN_HIP_GRAD_LINEAR) {
    const int t0 = max(t_begin, t_live_end);
    for (size_t i = (size_t)t0 * D + tid; i < (size_t)t_end * D; i += kOcNT) gseq[i] = fill;
  }
}

// ------------------------------------------------------------------------------------
// launch 2, compact form for the fused ChainLoss: ONE WAVE PER FRAME, no workgroup barrier in the
// frame loop.  The merge accumulator is indexed by the arc's compact row (num_prep_kernel), so a
// wave needs U <= K 64-bit words instead of one per pdf; rows, accumulator and arc tables of four
// waves fit LDS several times over, and a CU keeps a dozen frames in flight.
// ------------------------------------------------------------------------------------
constexpr int kOwFrames = 8;               // consecutive frames per wave
__global__ __launch_bounds__(kOcNT) void num_occ_wave_kernel(const NumArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int L = __builtin_amdgcn_readfirstlane(seq_len(a.lengths, b, a.T));
  const int K = a.K, T = a.T, H = a.H, Hq = (H + 3) & ~3;
  const int t_wg = blockIdx.x * (4 * kOwFrames);
  if (t_wg >= L) return;
  char* p = smem_raw;
  uint32_t* su = reinterpret_cast<uint32_t*>(p); p += 4 * (size_t)K;                 // src | compact row << 16
  p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 15) & ~uintptr_t(15));
  // per wave: the log-occupancy of every state of the frame (ONE fp32 row: alpha + beta - logP is formed in fp64 from the
  // prefetched rows and rounded once - a number of magnitude < 100 whose rounding moves an occupancy by < 1e-6 of itself)
  // and the merge accumulator; 4 KiB + 8 K bytes per wave let four workgroups share a CU (two while the fp64 rows sat here)
  const size_t per_wave = 4 * (size_t)Hq + 8 * (size_t)K;
  float* strow = reinterpret_cast<float*>(p + wave * per_wave);
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(strow + Hq);       // [U]
  const size_t g = (size_t)b * a.graph_stride;
  const int32_t* ft = a.fwd_trans + g * K * 3;
  const int32_t* uidx = a.uidx_ws + (size_t)b * K;
  const int U = a.ucount_ws[b];
  int kused = 0;
  for (int k = tid; k < K; k += kOcNT) {
    const int u = uidx[k];
    su[k] = (uint32_t)ft[3 * k] | ((uint32_t)max(u, 0) << 16);
    if (u >= 0) kused = k + 1;
  }
  for (int u = lane; u < U; u += 64) acc[u] = 0ull;
  // used arcs are a prefix (fstext.cc:30-116 lays arcs out state by state): its length, over the workgroup
  __shared__ int s_kused;
  if (tid == 0) s_kused = 0;
  __syncthreads();
  atomicMax(&s_kused, kused);
  __syncthreads();
  const int Kused = s_kused;
  const double logp = a.logp_ws[b];
  const double* aws = a.alpha_ws + (size_t)b * (T + 1) * H;
  const double* bws = a.beta_ws + (size_t)b * (T + 1) * H;
  const float* fseq = a.frac_ws + (size_t)b * T * K;
  int bad = 0;
  const int t0 = t_wg + wave * kOwFrames, t1 = min(t0 + kOwFrames, L);
  // Software pipeline: a wave has nobody to hide its own global latencies behind, so the rows and the
  // log-shares of frame t+1 are loaded into registers while frame t is evaluated (kWX arcs and
  // kWR row elements per lane are staged; larger graphs read the rest directly).
  constexpr int kWX = 16, kWR = 8;
  float px[kWX];
  double pa[kWR], pb[kWR];
#define NUM_OCCW_PREFETCH(t)                                                                       \
  do {                                                                                             \
    const float* fr_ = fseq + (size_t)(t) * K;                                                     \
    _Pragma("unroll") for (int i = 0; i < kWX; i++) {                                              \
      const int k = lane + 64 * i;                                                                 \
      if (k < Kused) px[i] = fr_[k];                                                               \
    }                                                                                              \
    _Pragma("unroll") for (int i = 0; i < kWR; i++) {                                              \
      const int h = lane + 64 * i;                                                                 \
      if (h < H) { pa[i] = aws[(size_t)(t) * H + h]; pb[i] = bws[(size_t)(t) * H + h]; }           \
    }                                                                                              \
  } while (0)
  // occupancy = exp(alpha(t,src) + beta(t,src) - logP + r_k(t)), see num_occ_kernel
#define NUM_OCCW_ARC(k, rk)                                                                        \
  do {                                                                                             \
    const uint32_t w_ = su[k];                                                                     \
    const float st = strow[w_ & 0xffffu];                                                          \
    const float v = st == -INFINITY ? 0.f : fexp(st + (rk));                                       \
    fsum += v;                                                                                     \
    if (v > 0.f) {                                                                                 \
      if (v <= 2.f) atomicAdd(&acc[w_ >> 16], (unsigned long long)(v * kFixScale));                \
      else bad = 1;                                                                                \
    } else if (v != 0.f) {                                                                         \
      bad = 1;                                  /* NaN */                                          \
    }                                                                                              \
  } while (0)
  if (t0 < t1) NUM_OCCW_PREFETCH(t0);
  for (int t = t0; t < t1; t++) {
    const float* frow = fseq + (size_t)t * K;
#pragma unroll
    for (int i = 0; i < kWR; i++) { const int h = lane + 64 * i; if (h < H) strow[h] = (float)(pa[i] + pb[i] - logp); }
    for (int h = lane + 64 * kWR; h < H; h += 64) strow[h] = (float)(aws[(size_t)t * H + h] + bws[(size_t)t * H + h] - logp);
    float xc[kWX];
    float fsum = 0.f;                                   // this lane's share of the frame's occupancy total
#pragma unroll
    for (int i = 0; i < kWX; i++) xc[i] = px[i];
    if (t + 1 < t1) NUM_OCCW_PREFETCH(t + 1);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < kWX; i++) { const int k = lane + 64 * i; if (k < Kused) NUM_OCCW_ARC(k, xc[i]); }
    for (int k = lane + 64 * kWX; k < Kused; k += 64) NUM_OCCW_ARC(k, frow[k]);
    if (t == 0 || a.check_all) {                        // the reference's `ok`: the frame's occupancies sum to 1 within 5 %
      const float ftot = wave_sum(fsum);
      if (!(fabsf(ftot - 1.f) <= 0.05f)) bad = 1;
    }
    __builtin_amdgcn_wave_barrier();
    float* crow = a.rows_ws + ((size_t)b * T + t) * K;
    for (int u = lane; u < U; u += 64) {
      const unsigned long long v = acc[u];
      acc[u] = 0ull;
      crow[u] = (float)v * kFixInv;
    }
    __builtin_amdgcn_wave_barrier();
  }
#undef NUM_OCCW_ARC
#undef NUM_OCCW_PREFETCH
  if (bad) atomicAdd(a.bad, 1);
}

// ------------------------------------------------------------------------------------
// compact rows -> an existing dense gradient: grad[b,t,pdf_u] += grad_scale * rows[b,t,u]
// (time-parallel; the fused ChainLoss uses it when the occupancy pass cannot fold the numerator in)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kOcNT) void num_scatter_kernel(const NumArgs a) {
  const int tid = threadIdx.x, b = blockIdx.y;
  const int L = seq_len(a.lengths, b, a.T), K = a.K, T = a.T, D = a.D;
  const int t_begin = blockIdx.x * a.frames_per_block;
  const int t_end = min(min(t_begin + a.frames_per_block, T), L);
  const int U = a.ucount_ws[b];
  const int32_t* upd = a.upd_ws + (size_t)b * K;
  const float gscale = a.grad_scale_dev ? a.grad_scale * *a.grad_scale_dev : a.grad_scale;
  for (int u = tid; u < U; u += kOcNT) {
    const int n = upd[u];
    for (int t = t_begin; t < t_end; t++) {
      const float v = a.rows_ws[((size_t)b * T + t) * K + u];
      float* gp = a.grad + ((size_t)b * T + t) * D + n;
      if (v != 0.f) *gp = mul_add_rn(v, gscale, *gp);      // same bits as the ACCUM mode of num_occ_kernel
    }
  }
}

template <int VEC, int XCH, int LD, bool XH>
hipError_t launch_fb_x(const NumArgs& a, size_t lds, hipStream_t st) {
  auto k = num_fb_kernel<VEC, XCH, LD, XH>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(2 * a.B), dim3(kFbNT + LD), lds, st, a);
  return hipGetLastError();
}
template <int VEC, int XCH, int LD = 0>
hipError_t launch_fb(const NumArgs& a, size_t lds, hipStream_t st) {
  if (a.x_half) {                                      // 2-byte rows: the float4-chunk forms only (num_half_native)
    if constexpr (VEC == 4 && XCH > 0) return launch_fb_x<VEC, XCH, LD, true>(a, lds, st);
    else return hipErrorInvalidValue;
  }
  return launch_fb_x<VEC, XCH, LD, false>(a, lds, st);
}

}  // namespace

size_t num_fb_lds_bytes(int H, int K, int D) {
  const size_t Dp = (D + 3) & ~3, Hq = (H + 1) & ~1;
  return 16 * Hq + 8 * 16 + 4 * 16 + 8 * Dp + 8 * (size_t)K + 64;
}
size_t num_occ_lds_bytes(int H, int K, int D) {
  const size_t Dp = (D + 3) & ~3, Hq = (H + 1) & ~1;
  return 8 * Dp + 16 * Hq + 16 + 10 * (size_t)K + 64;
}

hipError_t launch_num_fb(const NumArgs& a, hipStream_t st, const char** why) {
  if (a.general) return launch_num_general_fb(a, st);
  const size_t lds = num_fb_lds_bytes(a.H, a.K, a.D);
  if (lds > 160 * 1024) {
    *why = "numerator graph + nnet-output rows do not fit the 160 KiB LDS of one CU";
    return hipErrorInvalidValue;
  }
  if (a.x_half && (a.D % 4 != 0 || a.D > 4 * 8 * kFbNT)) {
    *why = "2-byte network outputs need rows of a multiple of four pdfs within the register-staged forms";
    return hipErrorInvalidValue;
  }
  if ((size_t)a.T * a.D * 4 >= (size_t)1 << 31) {      // rows are addressed as 32-bit byte offsets (buffer loads)
    *why = "one sequence's nnet-output slab reaches 2 GiB";
    return hipErrorInvalidValue;
  }
  const int D = a.D;
  if (D % 4 == 0) {
    if (D <= 4 * 4 * kFbLd) return launch_fb<4, 4, kFbLd>(a, lds, st);       // (row-staging waves: 0.95 ms against 1.46 ms without them at C3)
    if (D <= 4 * 8 * kFbLd) return launch_fb<4, 8, kFbLd>(a, lds, st);
    if (D <= 4 * 2 * kFbNT) return launch_fb<4, 2>(a, lds, st);
    if (D <= 4 * 8 * kFbNT) return launch_fb<4, 8>(a, lds, st);
  } else if (D <= 8 * kFbNT) {
    return launch_fb<1, 8>(a, lds, st);
  }
  return launch_fb<1, 0>(a, lds, st);
}

namespace {
__global__ void num_corrupt_kernel(double* row, int n, double add) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) row[i] += add;
}
}  // namespace
hipError_t launch_num_corrupt(const NumArgs& a, hipStream_t st) {
  double* row = a.alpha_ws + ((size_t)a.corrupt_b * (a.T + 1) + a.corrupt_t) * a.H;
  hipLaunchKernelGGL(num_corrupt_kernel, dim3(1), dim3(256), 0, st, row, a.H, (double)a.corrupt_log);
  return hipGetLastError();
}

hipError_t launch_num_prep(const NumArgs& a, hipStream_t st, const char** why) {
  if (a.general) return hipSuccess;                          // (compact rows are a tile-path form)
  const size_t lds = 4 * (size_t)a.D + 16 + 4 * kOcNT + 64;
  if (lds > 160 * 1024) { *why = "pdf table does not fit the 160 KiB LDS of one CU"; return hipErrorInvalidValue; }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(num_prep_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(num_prep_kernel, dim3(a.B), dim3(kOcNT), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_num_scatter(const NumArgs& a, hipStream_t st, const char** why) {
  (void)why;
  const int gx = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  hipLaunchKernelGGL(num_scatter_kernel, dim3(gx, a.B), dim3(kOcNT), 0, st, a);
  return hipGetLastError();
}

size_t num_occ_wave_lds_bytes(int H, int K) {
  const size_t Hq = (H + 3) & ~3;
  return 4 * (size_t)K + 16 + 4 * (4 * Hq + 8 * (size_t)K) + 64;
}

hipError_t launch_num_occ(const NumArgs& a, bool compact, hipStream_t st, const char** why) {
  if (a.general) {
    if (compact) { *why = "compact occupancy rows are not produced for graphs on the general numerator kernels"; return hipErrorInvalidValue; }
    return launch_num_general_occ(a, a.gen_acc, st);
  }
  if (compact && a.D <= 65535 && a.K <= 32767 && num_occ_wave_lds_bytes(a.H, a.K) <= 64 * 1024) {
    const size_t lds = num_occ_wave_lds_bytes(a.H, a.K);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(num_occ_wave_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int gx = (a.T + 4 * kOwFrames - 1) / (4 * kOwFrames);
    hipLaunchKernelGGL(num_occ_wave_kernel, dim3(gx, a.B), dim3(kOcNT), lds, st, a);
    return hipGetLastError();
  }
  const size_t lds = num_occ_lds_bytes(a.H, a.K, a.D);
  if (lds > 160 * 1024) {
    *why = "pdf accumulators + numerator graph do not fit the 160 KiB LDS of one CU";
    return hipErrorInvalidValue;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(num_occ_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  NumArgs b = a;
  if (compact) b.grad_mode = kGradCompact;
  const int gx = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  hipLaunchKernelGGL(num_occ_kernel, dim3(gx, a.B), dim3(kOcNT), lds, st, b);
  return hipGetLastError();
}

}  // namespace pychain_hip
