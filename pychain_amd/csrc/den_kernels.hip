// den_kernels.hip - denominator (probability domain, leaky-HMM) forward-backward
// for gfx950.  Hand-written for CDNA4; structurally unrelated to the reference's
// one-launch-per-frame, thread-per-(sequence,state) CUDA kernels
// (pytorch_binding/src/chain-kernels.cu:97-245).
//
// Decomposition:
//
//   launch 1  den_recursion_kernel   2B persistent workgroups, one per (sequence, direction).
//             Block b < B walks the alpha recursion of sequence b forward in time, block B+b
//             walks the beta recursion of the same sequence backward in time, CONCURRENTLY:
//             the beta pass does not wait for the alpha pass because both carry their own
//             per-frame normaliser ("arbitrary_scale" in chain-computation.h:91-98 - any
//             per-frame scale gives the same posteriors).  The state vector of the previous
//             frame and the exp'd nnet-output row live in LDS; arcs are read wave-tiled and
//             coalesced from the compiled plan; per-state sums are lane-private (one state
//             per lane), per-frame totals are wave64 shuffle reductions + one LDS hop.
//             Every normalised alpha'(t,.) / beta(t,.) row is streamed to HBM once.
//   launch 2  den_gamma_kernel       time-parallel over all (sequence, frame-chunk) pairs:
//             gamma(t,n) = x(t,n) * sum_{arcs with pdf n} p * alpha'(t,src) * beta(t+1,dst),
//             normalised so that each live frame sums to one (the invariant the reference
//             checks at chain-computation.cc:381-390).  Arcs are grouped by pdf-id, so the
//             occupancy is a lane-private sum: no atomics, deterministic, exact (the
//             reference's CUDA path adds stochastically-thresholded atomics,
//             chain-kernels.cu:53-87; parity target is its exact CPU path).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "den_kernels.h"
#include "device_utils.h"
#include "plan_format.h"

namespace pychain_hip {

namespace {

constexpr int kNT = PLAN_REC_WAVES * 64;   // threads per workgroup (both kernels)
constexpr int kNW = PLAN_REC_WAVES;
static_assert(PLAN_REC_WAVES == PLAN_GAM_WAVES, "one workgroup shape for both kernels");
constexpr int kMaxResident = 40;           // slot-rows per wave kept in VGPRs (2 VGPRs each) at 16 waves/CU

// ---- one frame of a tile plan: out[row] = sum_k p_k * U[i0_k] * V[i1_k] ------------------
// The first R slot-rows of a wave are held in registers (R = 0: everything is streamed
// from the L2-resident plan).  The plan of a wave is loop-invariant over frames, so its arcs are loaded ONCE per
// workgroup into VGPRs (2 per slot-row) and the per-frame inner loop touches only LDS.
// Slot-rows beyond R (plans larger than the register budget) are streamed as above.
template <int R>
struct ArcRegs {
  // 2 VGPRs per slot-row: the two LDS byte offsets packed 16:16 (so this variant needs
  // 4*Hp and 4*D below 65536) and the arc probability.
  uint32_t pk[R > 0 ? R : 1];
  float p[R > 0 ? R : 1];
  __device__ __forceinline__ void load(const WaveEntry we, const uint2* __restrict__ slots, int lane) {
    const uint2* sp = slots + (size_t)we.slot_row_begin * 64 + lane;
#pragma unroll
    for (int s = 0; s < R; s++) {
      uint2 a = make_uint2(0u, 0u);
      if (s < we.nslot_rows) a = sp[s * 64];
      pk[s] = ((a.x & 0xffffu) << 2) | ((a.x >> 16) << 18);
      p[s] = __uint_as_float(a.y);
    }
  }
};

__device__ __forceinline__ float lds_at(const float* base, uint32_t byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

// MODE 0: out[out_base+lane] = acc (recursions).  MODE 1: out[row_map[out_base+lane]] = acc
// (occupancy pass: plan order -> natural pdf order, row_map in LDS, -1 = padding row).
#define PYCHAIN_TILE_FLUSH()                                                   \
  do {                                                                         \
    const int pos = cur_base + lane;                                           \
    if constexpr (MODE == 0) {                                                 \
      out[pos] = acc;                                                          \
      s0 += acc;                                                               \
      if (wvec) s1 += acc * wvec[pos];                                         \
    } else {                                                                   \
      const int n = row_map[pos];                                              \
      if (n >= 0) out[n] = acc;                                                \
    }                                                                          \
    acc = 0.f;                                                                 \
    g++;                                                                       \
    cur_base = nxt_base; remaining = nxt_n;                                    \
    if (g + 1 < ng) { nxt_base = gt[g + 1].out_base; nxt_n = gt[g + 1].nslots; } \
    else { nxt_base = 0; nxt_n = 0; }                                          \
  } while (0)

template <int R, int MODE>
__device__ __forceinline__ void tile_rows(ArcRegs<R>& ar, const WaveEntry we,
                                          const GroupEntry* __restrict__ gtab, const uint2* __restrict__ slots,
                                          int lane, const float* U, const float* V, float* out,
                                          const int* row_map, const float* wvec, float& s0, float& s1) {
  const GroupEntry* gt = gtab + we.first_group;
  const int ng = we.ngroups, ns = we.nslot_rows;
  int g = 0;
  int cur_base = 0, remaining = 0, nxt_base = 0, nxt_n = 0;   // one group ahead: its scalar load is off the critical path
  if (ng > 0) { cur_base = gt[0].out_base; remaining = gt[0].nslots; }
  if (ng > 1) { nxt_base = gt[1].out_base; nxt_n = gt[1].nslots; }
  float acc = 0.f;
#pragma unroll
  for (int s = 0; s < R; s++) {
    if (s < ns) {
      // Opaque to the optimiser: without this it hoists the two unpacked offsets of every
      // slot out of the frame loop and the arcs cost 3 VGPRs each instead of 2.
      asm volatile("" : "+v"(ar.pk[s]));
      acc = fmaf(ar.p[s] * lds_at(U, ar.pk[s] & 0xffffu), lds_at(V, ar.pk[s] >> 16), acc);
      if (--remaining == 0) PYCHAIN_TILE_FLUSH();
    }
  }
  if (ns > R) {
    const uint2* sp = slots + ((size_t)we.slot_row_begin + R) * 64 + lane;
    for (int s = R; s < ns; s++) {
      const uint2 a = *sp;
      sp += 64;
      acc = fmaf(__uint_as_float(a.y) * U[a.x & 0xffffu], V[a.x >> 16], acc);
      if (--remaining == 0) PYCHAIN_TILE_FLUSH();
    }
  }
  while (g < ng) PYCHAIN_TILE_FLUSH();           // trailing groups whose rows have no arcs: zeros
}

// ------------------------------------------------------------------------------------
// launch 1: alpha and beta recursions
// ------------------------------------------------------------------------------------
template <int VEC, int XCH, int R>
__global__ __launch_bounds__(kNT) void den_recursion_kernel(const DenArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool fwd = blockIdx.x < (unsigned)a.B;
  const int b = fwd ? blockIdx.x : blockIdx.x - a.B;
  const int L = (int)a.lengths[b];
  const int Hp = a.Hp, H = a.H, D = a.D, Dp = (D + 3) & ~3;
  const char* plan = a.plans + (size_t)b * a.plan_stride;
  const PlanHeader* hd = reinterpret_cast<const PlanHeader*>(plan);
  const TilePlan tp = fwd ? hd->alpha : hd->beta;
  const WaveEntry we = reinterpret_cast<const WaveEntry*>(plan + tp.off_wave_tab)[wave];
  const GroupEntry* gtab = reinterpret_cast<const GroupEntry*>(plan + tp.off_group_tab);
  const uint2* slots = reinterpret_cast<const uint2*>(plan + tp.off_slots);
  ArcRegs<R> arcs;
  arcs.load(we, slots, lane);

  float* vec0 = reinterpret_cast<float*>(smem_raw);
  float* vec1 = vec0 + Hp;
  float* lk = vec1 + Hp;             // leaky probs in this side's numbering
  float* xr0 = lk + Hp;
  float* xr1 = xr0 + Dp;
  float* red = xr1 + Dp;             // [2*kNW]

  const float* leaky_g = reinterpret_cast<const float*>(plan + (fwd ? hd->off_leaky_a : hd->off_leaky_b));
  const float* start_g = reinterpret_cast<const float*>(plan + (fwd ? hd->off_init_a : hd->off_final_b));
  const float* xseq = a.x + (size_t)b * a.T * D;
  float* store = fwd ? a.alpha_store + (size_t)b * a.T * Hp : a.beta_store + (size_t)b * (a.T + 1) * Hp;
  const float coef = a.coef;

  // ---- frame 0 (alpha) / frame L (beta): chain-computation.cc:92-95,97-110,178-194 / :232-245,313-330
  float p0 = 0.f, p1 = 0.f;
  for (int i = tid; i < Hp; i += kNT) {
    const float l = leaky_g[i], s = start_g[i];
    lk[i] = l; vec0[i] = s;
    p0 += s; p1 += s * l;
  }
  p0 = wave_sum(p0); p1 = wave_sum(p1);
  if (lane == 0) { red[wave] = p0; red[kNW + wave] = p1; }
  XRow<kNT, VEC, XCH> xq;
  {
    const int t0 = fwd ? 0 : L - 1;                 // first nnet-output row this side consumes
    xq.load(xseq + (size_t)t0 * D, D, tid);
    xq.store(xr0, xseq + (size_t)t0 * D, D, tid, a.input_is_exp);
  }
  __syncthreads();
  float tot = 0.f, wtot = 0.f;
#pragma unroll
  for (int w = 0; w < kNW; w++) { tot += red[w]; wtot += red[kNW + w]; }
  double logsum = 0.0;                 // sum_t log tot-alpha(t), chain-computation.cc:216-229
  int bad = 0;
  {
    const float inv = 1.f / tot;
    if (!(tot > 0.f) || !(inv > 0.f)) bad = 1;
    if (fwd && tid == 0) logsum += (double)logf(tot);
    float* row = store + (size_t)(fwd ? 0 : L) * Hp;
    for (int i = tid; i < Hp; i += kNT) {
      float v = fwd ? vec0[i] * inv + coef * lk[i]                    // alpha'(0)/tot(0)
                    : (i < H ? (vec0[i] + coef * wtot) * inv : 0.f);  // beta(L), unit sum
      vec0[i] = v;
      row[i] = v;
    }
  }
  __syncthreads();

  // ---- general frames.  alpha: step j produces alpha'(j+1) from alpha'(j) and x(j), j = 0..L-1
  //                        beta:  step j produces beta(t) from beta(t+1) and x(t), t = L-1-j, j = 0..L-2
  const int nsteps = fwd ? L : L - 1;
  for (int j = 0; j < nsteps; j++) {
    const float* vin = (j & 1) ? vec1 : vec0;
    float* vout = (j & 1) ? vec0 : vec1;
    const float* xcur = (j & 1) ? xr1 : xr0;
    float* xnext = (j & 1) ? xr0 : xr1;
    const int tn = fwd ? j + 1 : L - 2 - j;          // nnet-output row of the NEXT step
    const bool have_next = fwd ? (tn < L) : (tn >= 1);
    const float* xrow_next = xseq + (size_t)(have_next ? tn : 0) * D;
    if (have_next) xq.load(xrow_next, D, tid);       // in flight during the arc work

    float s0 = 0.f, s1 = 0.f;
    tile_rows<R, 0>(arcs, we, gtab, slots, lane, vin, xcur, vout, nullptr, fwd ? nullptr : lk, s0, s1);
    s0 = wave_sum(s0);
    if (!fwd) s1 = wave_sum(s1);
    if (lane == 0) { red[wave] = s0; red[kNW + wave] = s1; }
    __syncthreads();
    tot = 0.f; wtot = 0.f;
#pragma unroll
    for (int w = 0; w < kNW; w++) { tot += red[w]; wtot += red[kNW + w]; }
    const float inv = 1.f / tot;
    if (!(tot > 0.f) || !(inv > 0.f)) bad = 1;
    if (fwd && tid == 0) logsum += (double)logf(tot);
    const int tstore = fwd ? j + 1 : L - 1 - j;
    const bool do_store = fwd ? (tstore < L) : true;
    float* row = store + (size_t)(do_store ? tstore : 0) * Hp;
    for (int i = tid; i < Hp; i += kNT) {
      float v = fwd ? vout[i] * inv + coef * lk[i]
                    : (i < H ? (vout[i] + coef * wtot) * inv : 0.f);
      vout[i] = v;
      if (do_store) row[i] = v;
    }
    if (have_next) xq.store(xnext, xrow_next, D, tid, a.input_is_exp);
    __syncthreads();
  }

  if (fwd) {
    // ComputeTotLogLike, chain-computation.cc:209-230: log sum_i alpha'(L,i) final(i) + sum_t log tot(t)
    const float* vL = (nsteps & 1) ? vec1 : vec0;
    const float* fin = reinterpret_cast<const float*>(plan + hd->off_final_a);
    float f = 0.f;
    for (int i = tid; i < Hp; i += kNT) f += vL[i] * fin[i];
    f = wave_sum(f);
    if (lane == 0) red[wave] = f;
    __syncthreads();
    if (tid == 0) {
      float fs = 0.f;
      for (int w = 0; w < kNW; w++) fs += red[w];
      const float objf = (float)(logsum + (double)logf(fs));
      a.objf[b] = objf;
      if (!(fs > 0.f) || !(objf - objf == 0.f)) bad = 1;
    }
  }
  if (bad && lane == 0) atomicAdd(a.bad, 1);
}

// ------------------------------------------------------------------------------------
// launch 2: occupancies (time-parallel)
// ------------------------------------------------------------------------------------
template <int VEC, int XCH, int R>
__global__ __launch_bounds__(kNT) void den_gamma_kernel(const DenArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int L = (int)a.lengths[b];
  const int Hp = a.Hp, D = a.D, Dp = (D + 3) & ~3;
  const int t_begin = blockIdx.x * a.frames_per_block;
  const int t_end = min(t_begin + a.frames_per_block, a.T);
  float* gseq = a.grad + (size_t)b * a.T * D;
  if (t_begin >= L) {                               // whole chunk is padding: exact zeros (zeros_like, :58)
    for (size_t i = (size_t)t_begin * D + tid; i < (size_t)t_end * D; i += kNT) gseq[i] = 0.f;
    return;
  }
  const char* plan = a.plans + (size_t)b * a.plan_stride;
  const PlanHeader* hd = reinterpret_cast<const PlanHeader*>(plan);
  const TilePlan tp = hd->gamma;
  const WaveEntry we = reinterpret_cast<const WaveEntry*>(plan + tp.off_wave_tab)[wave];
  const GroupEntry* gtab = reinterpret_cast<const GroupEntry*>(plan + tp.off_group_tab);
  const uint2* slots = reinterpret_cast<const uint2*>(plan + tp.off_slots);
  const int32_t* row_pdf = reinterpret_cast<const int32_t*>(plan + hd->off_row_pdf);
  ArcRegs<R> arcs;
  arcs.load(we, slots, lane);

  float* U = reinterpret_cast<float*>(smem_raw);   // alpha'(t,.)   [Hp]
  float* V = U + Hp;                                // beta(t+1,.)   [Hp]
  float* xr = V + Hp;                               // exp x(t,.)    [Dp]
  float* q = xr + Dp;                               // per-pdf arc sums, natural pdf order [Dp]
  int* rmap = reinterpret_cast<int*>(q + Dp);       // plan row -> pdf-id [ngroups*64]
  float* red = reinterpret_cast<float*>(rmap + tp.ngroups * 64);   // [kNW]

  const float* xseq = a.x + (size_t)b * a.T * D;
  const float* aseq = a.alpha_store + (size_t)b * a.T * Hp;
  const float* bseq = a.beta_store + (size_t)b * (a.T + 1) * Hp;

  for (int i = tid; i < Dp; i += kNT) q[i] = 0.f;   // pdfs without arcs stay zero forever
  for (int i = tid; i < tp.ngroups * 64; i += kNT) rmap[i] = row_pdf[i];
  int bad = 0;
  const int t_live_end = min(t_end, L);
  for (int t = t_begin; t < t_live_end; t++) {
    float* grow = gseq + (size_t)t * D;
    XRow<kNT, VEC, XCH> xq;
    xq.load(xseq + (size_t)t * D, D, tid);
    const float* ar = aseq + (size_t)t * Hp;
    const float* br = bseq + (size_t)(t + 1) * Hp;
    for (int i = tid * 4; i < Hp; i += kNT * 4) {
      *reinterpret_cast<float4*>(U + i) = *reinterpret_cast<const float4*>(ar + i);
      *reinterpret_cast<float4*>(V + i) = *reinterpret_cast<const float4*>(br + i);
    }
    xq.store(xr, xseq + (size_t)t * D, D, tid, a.input_is_exp);
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
    tile_rows<R, 1>(arcs, we, gtab, slots, lane, U, V, q, rmap, nullptr, s0, s1);
    __syncthreads();
    float g[(VEC * XCH) > 0 ? (VEC * XCH) : 1];
    float part = 0.f;
    if constexpr (XCH > 0) {
#pragma unroll
      for (int c = 0; c < XCH; c++)
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          const int e = (c * kNT + tid) * VEC + k;
          g[c * VEC + k] = e < D ? xr[e] * q[e] : 0.f;
          part += g[c * VEC + k];
        }
    } else {
      for (int e = tid; e < D; e += kNT) part += xr[e] * q[e];
    }
    part = wave_sum(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < kNW; w++) tot += red[w];
    const float sc = a.grad_scale / tot;
    if (!(tot > 0.f) || !(sc - sc == 0.f)) bad = 1;
    if constexpr (XCH > 0) {
#pragma unroll
      for (int c = 0; c < XCH; c++) {
        const int e = (c * kNT + tid) * VEC;
        if (e < D) {
          if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(grow + e) =
                make_float4(g[c * 4] * sc, g[c * 4 + 1] * sc, g[c * 4 + 2] * sc, g[c * 4 + 3] * sc);
          } else {
            grow[e] = g[c] * sc;
          }
        }
      }
    } else {
      for (int e = tid; e < D; e += kNT) grow[e] = xr[e] * q[e] * sc;
    }
    __syncthreads();   // U/V/xr/q are rewritten by the next frame
  }
  // padded tail of a chunk that straddles the sequence end
  if (t_live_end < t_end)
    for (size_t i = (size_t)t_live_end * D + tid; i < (size_t)t_end * D; i += kNT) gseq[i] = 0.f;
  if (bad && lane == 0) atomicAdd(a.bad, 1);
}

template <typename K>
hipError_t launch_one(K kern, const DenArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(kNT), lds, st, a);
  return hipGetLastError();
}

// rows = slot-rows per wave the plan needs (0 = unknown: stream everything); plans larger
// than kMaxResident keep the first kMaxResident rows in registers and stream their tail.
// The packed-offset register format needs 4*Hp and 4*D < 65536.
inline int pick_r(const DenArgs& a, int rows) {
  if (rows <= 0 || a.Hp * 4 > 65535 || a.D * 4 > 65535) return 0;
  if (rows <= 16) return 16;
  if (rows <= 32) return 32;
  return kMaxResident;
}

template <int VEC, int XCH>
hipError_t launch_r(const DenArgs& a, int hint, size_t lds_rec, size_t lds_gam, int gx, hipStream_t st) {
  hipError_t e = hipSuccess;
  if (a.phase_mask & 1) {
    const dim3 grid(2 * a.B);
    switch (pick_r(a, hint & 0xffff)) {
      case 0: e = launch_one(den_recursion_kernel<VEC, XCH, 0>, a, grid, lds_rec, st); break;
      case 16: e = launch_one(den_recursion_kernel<VEC, XCH, 16>, a, grid, lds_rec, st); break;
      case 32: e = launch_one(den_recursion_kernel<VEC, XCH, 32>, a, grid, lds_rec, st); break;
      default: e = launch_one(den_recursion_kernel<VEC, XCH, kMaxResident>, a, grid, lds_rec, st); break;
    }
    if (e != hipSuccess) return e;
  }
  if (a.phase_mask & 2) {
    const dim3 grid(gx, a.B);
    switch (pick_r(a, (hint >> 16) & 0x7fff)) {
      case 0: e = launch_one(den_gamma_kernel<VEC, XCH, 0>, a, grid, lds_gam, st); break;
      case 16: e = launch_one(den_gamma_kernel<VEC, XCH, 16>, a, grid, lds_gam, st); break;
      case 32: e = launch_one(den_gamma_kernel<VEC, XCH, 32>, a, grid, lds_gam, st); break;
      default: e = launch_one(den_gamma_kernel<VEC, XCH, kMaxResident>, a, grid, lds_gam, st); break;
    }
  }
  return e;
}

}  // namespace

hipError_t launch_den(const DenArgs& a, int gamma_max_groups, int resident_slot_rows, hipStream_t st,
                      const char** why) {
  const int Dp = (a.D + 3) & ~3;
  const size_t lds_rec = sizeof(float) * (3 * (size_t)a.Hp + 2 * (size_t)Dp + 2 * kNW);
  const size_t lds_gam = sizeof(float) * (2 * (size_t)a.Hp + 2 * (size_t)Dp + (size_t)gamma_max_groups * 64 + kNW);
  if (lds_rec > 160 * 1024 || lds_gam > 160 * 1024) {
    *why = "state vector + nnet-output row do not fit the 160 KiB LDS of one CU";
    return hipErrorInvalidValue;
  }
  const int gx = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  const int D = a.D, r = resident_slot_rows;
  if (D % 4 == 0) {
    if (D <= 4 * kNT) return launch_r<4, 1>(a, r, lds_rec, lds_gam, gx, st);
    if (D <= 8 * kNT) return launch_r<4, 2>(a, r, lds_rec, lds_gam, gx, st);
    if (D <= 16 * kNT) return launch_r<4, 4>(a, r, lds_rec, lds_gam, gx, st);
  } else if (D <= 4 * kNT) {
    return launch_r<1, 4>(a, r, lds_rec, lds_gam, gx, st);
  }
  return launch_r<1, 0>(a, r, lds_rec, lds_gam, gx, st);
}

}  // namespace pychain_hip
