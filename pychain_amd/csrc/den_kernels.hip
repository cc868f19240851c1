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
//             frame and the exp'd nnet-output row live in LDS; every wave keeps its share of
//             the arcs in VGPRs for the whole launch (LDS address of both operands + the
//             probability), so the per-frame inner loop is 2 ds_read + 2 VALU per arc;
//             per-state sums are lane-private (one state per lane), per-frame totals are
//             wave64 DPP reductions + one LDS hop.  Every normalised alpha'(t,.) / beta(t,.)
//             row is streamed to HBM once.
//   launch 2  den_gamma2_kernel / den_gamma_kernel   time-parallel over all (sequence,
//             frame-chunk) pairs:
//             gamma(t,n) = x(t,n) * sum_{arcs with pdf n} p * alpha'(t,src) * beta(t+1,dst),
//             normalised so that each live frame sums to one (the invariant the reference
//             checks at chain-computation.cc:381-390).  Arcs are grouped by pdf-id, so the
//             occupancy is a lane-private sum: no atomics, deterministic, exact (the
//             reference's CUDA path adds stochastically-thresholded atomics,
//             chain-kernels.cu:53-87; parity target is its exact CPU path).  The shipped form
//             evaluates two frames together (float2-interleaved operands, ds_read_b64 gathers,
//             packed fma) and can fold the numerator's occupancies in; the one-frame form is
//             the fallback for graphs that do not fit it.
//
// What bounds these kernels (profiles/): the recursion is a chain of T
// dependent frame steps per workgroup; a step is the LDS gather time of the frame's arcs
// (conflict-free by construction of the plan) plus barrier-separated serial phases, not HBM;
// the occupancy pass is time-parallel and runs near the HBM roofline.
#include <hip/hip_runtime.h>
#include <cstring>
#include <type_traits>
#include <stdint.h>

#include "common.h"
#include "den_kernels.h"
#include "device_utils.h"
#include "plan_format.h"

namespace pychain_hip {

namespace {

#include "den_common.inc.h"
// objf and the invariant check from the stored per-frame totals (DenArgs::tot_a).  One workgroup per sequence.
//   rows of den_recursion_lazy_kernel: alpha row t carries prod_{tau<t} tot(tau); the beta row frame t's occupancy
//       reads, b(t+1,.) + c(t+1), carries prod_{tau>=t+2} n(tau); objf = sum_{t<L} log tot(t) + log fin_dot
//   rows of den_recursion_kernel (normalised): alpha'(t)/tot(t) carries prod_{tau<=t} tot(tau); beta(t+1) carries
//       prod_{tau>=t+1} n(tau); objf = sum_{t<=L} log tot(t) + log fin_dot
constexpr int kFinNT = 256;
// block sums of two doubles per thread with ONE barrier: ds_bpermute butterflies inside a wave, then every thread adds the
// kFinNT / 64 wave partials in the same order (the kernel sits in the serial tail of every step: its latency chain counts)
__device__ __forceinline__ void fin_block_sum2(double& x, double& y, double (*part)[kFinNT / 64], int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { x += __shfl_xor(x, o); y += __shfl_xor(y, o); }
  if ((tid & 63) == 0) { part[0][tid >> 6] = x; part[1][tid >> 6] = y; }
  __syncthreads();
  x = 0.0; y = 0.0;
#pragma unroll
  for (int w = 0; w < kFinNT / 64; w++) { x += part[0][w]; y += part[1][w]; }
}
__global__ __launch_bounds__(kFinNT) void den_finish_kernel(const DenArgs a) {
  __shared__ double part[2][kFinNT / 64], part2[2][kFinNT / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int L = seq_len(a.lengths, b, a.T);
  const float* ta = a.tot_a + (size_t)b * (a.T + 2);
  const float* tb = a.tot_b + (size_t)b * (a.T + 2);
  const int na = a.lazy ? L : L + 1;                       // alpha totals 0 .. na-1 make up the log-probability
  // sum_t log tot(t) as the log of a product: a thread multiplies the mantissas of its totals (fp64: exact to 2^-53 per factor,
  // no overflow - a mantissa lies in [0.5, 1)) and adds up their exponents, so it evaluates ONE fp64 log for its 2 x T / 256
  // totals instead of one each.  A total that is zero, negative, infinite or NaN keeps its meaning: the product turns
  // 0 / negative / inf / NaN and so does the log.
  auto log_of_product = [](const float* v, int first, int end) -> double {
    double m = 1.0; long ex = 0;
    for (int t = first; t < end; t += kFinNT) {
      const float x = v[t];
      if (x > 0.f && x < __builtin_inff()) { int e; m *= (double)__builtin_frexpf(x, &e); ex += e; }
      else m *= (double)x;
      if (m < 0x1p-900) { m *= 0x1p+800; ex -= 800; }      // (> 900 factors per thread: T > 230 000)
    }
    return log(m) + (double)ex * 0.69314718055994530942;
  };
  // (what thread 0 needs after the sums: requested before them)
  // (time segments: an inner alpha segment that staged a NaN network output says so in xnan - the last one owns fin_dot)
  const float fin_dot = (a.tseg > 1 && __hip_atomic_load(a.xnan + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ? __builtin_nanf("") : a.fin_dot[b];
  const float ta0 = ta[0], tb1 = L >= 1 ? tb[1] : 1.f;
  double sh_total_a = log_of_product(ta, tid, na), sh_total_b = log_of_product(tb, tid + 1, L + 1);
  fin_block_sum2(sh_total_a, sh_total_b, part, tid);
  const double logp = sh_total_a + log((double)fin_dot);
  const float objf = (float)logp;
  // (frame 0's side of the check, below)   lazy: PA(0) = 0, SB(0) = sum_{tau>=2} log n(tau);   else: log tot(0), sum_{tau>=1}
  // (DenArgs::sg: the occupancy totals are sums of a(t+1,.) beta(t+1,.) - the alpha factor already divided by tot(t): PA(t) then
  // runs over tau <= t as for normalised rows, SB(t) as for the lazy beta rows)
  const bool pa_incl = !a.lazy || a.sg;
  const double pa_sb0 = (pa_incl ? log((double)ta0) : 0.0) + sh_total_b - (a.lazy && L >= 1 ? log((double)tb1) : 0.0);
  // Everything above needs the recursions only; frame 0's occupancy total - the other side of the check - comes from the
  // occupancy launch, which may still be running (DenArgs::occ_done): the sequence's side is left for the last workgroup.
  int bad = 0;
  if (tid == 0) {
    __hip_atomic_store(a.objf + b, objf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read by the last workgroup)
    __hip_atomic_store(a.fin_dot + a.B + b, (float)(pa_sb0 - logp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!(objf - objf == 0.f)) bad = 1;                    // -inf (the graph cannot end), NaN
  }
  if (a.check && a.check_all) {
    // every frame (verbose level >= 1; never with an overlapped schedule): thread `tid` walks frames tid*per .. with running
    // sums started from its chunk's prefix
    //   frame t: PA(t) = log-scale of its alpha row, SB(t) = log-scale of the beta row it reads
    //   lazy: PA(t) = sum_{tau<t} log tot(tau),  SB(t) = sum_{tau>=t+2} log n(tau)
    //   else: PA(t) = sum_{tau<=t},              SB(t) = sum_{tau>=t+1}
    const float* g = a.gtot + (size_t)b * a.T;
    const int per = (L + kFinNT - 1) / kFinNT;
    const int t0 = tid * per, t1 = min(t0 + per, L);
    double ca = 0.0, cb = 0.0;                            // sum_{tau<t0} log tot(tau), sum_{tau<=t0} log n(tau) (tau >= 1)
    for (int t = 0; t < t0; t++) ca += log((double)ta[t]);
    for (int t = 1; t <= t0 && t <= L; t++) cb += log((double)tb[t]);
    for (int t = t0; t < t1; t++) {
      const double lt = log((double)ta[t]);
      // cb = sum_{1<=tau<=t} log n(tau)
      const double pa = pa_incl ? ca + lt : ca;
      const double upto = a.lazy ? cb + (t + 1 <= L ? log((double)tb[t + 1]) : 0.0) : cb;   // sum_{tau<=t+1} / sum_{tau<=t}
      const double est = log((double)g[t]) + pa + (sh_total_b - upto);
      if (!(fabs(est - logp) <= 0.0487901642)) bad = 1;   // log(1.05); NaN counts
      ca += lt;
      if (t + 1 <= L) cb += log((double)tb[t + 1]);
    }
  }
  if (bad) atomicAdd(a.bad, 1);
  // ---- the workgroup that finishes last: the frame-0 check of every sequence and the step totals (DenArgs::loss_out).  Every
  // workgroup publishes its sequence's values and its `bad` increment with the release half of the counter increment.
  __shared__ int s_last;
  __syncthreads();                                           // (tid 0 wrote objf[b] and counted into bad above)
  if (tid == 0)
    s_last = __hip_atomic_fetch_add(a.finish_count, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  if (a.occ_done_target) {                                   // ONE thread of the call waits for the occupancy launch
    if (tid == 0) {
      const unsigned long long t0 = wall_clock64();          // 100 MHz
      while (__hip_atomic_load(a.occ_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.occ_done_target) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 2000000000ull) { atomicAdd(a.bad, 1); break; }   // 20 s
      }
    }
    __syncthreads();
  }
  double acc = 0.0, frames = 0.0, nfail = 0.0, unused = 0.0;
  for (int i = tid; i < a.B; i += kFinNT) {
    const float oi = __hip_atomic_load(a.objf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.check && !a.check_all) {
      // log G(0) + PA(0) + SB(0) = log P within log(1.05) (NaN counts; a sequence whose objective is not finite is counted already)
      const float g0 = __hip_atomic_load(a.gtot + (size_t)i * a.T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float di = __hip_atomic_load(a.fin_dot + a.B + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (oi - oi == 0.f && !(fabs(log((double)g0) + (double)di) <= 0.0487901642)) nfail += 1.0;
    }
    acc += (double)oi;
    if (a.loss_num_objf)                                   // (written by the numerator's kernels on another stream: device-scope read)
      acc -= (double)__hip_atomic_load(a.loss_num_objf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    frames += (double)seq_len(a.lengths, i, a.T);
  }
  fin_block_sum2(acc, frames, part2, tid);
  __syncthreads();
  fin_block_sum2(nfail, unused, part2, tid);
  if (tid == 0) {
    if (nfail > 0.0) atomicAdd(a.bad, (int)nfail);
    // what the segmented (speculative) recursion launch counted stands only if its rows were kept (DenArgs::redo[3])
    if (a.tseg > 1 && __hip_atomic_load(a.redo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      const int spec = __hip_atomic_load(a.redo + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (spec > 0) atomicAdd(a.bad, spec);
    }
    // the plan's burn-in controller (DenArgs::tstate): this call's outcome decides the next call's burn-in - here, at the end of the
    // call, in stream order; nothing on the host reads it
    const bool ts_off = a.tseg > 1 && den_tseg_off(a);
    if (a.tseg > 1 && a.tstate) {
      const bool init = __hip_atomic_load(a.tstate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kTsegMagic;
      int burn = den_tburn(a), off = init ? a.tstate[2] : 0, calls = init ? a.tstate[3] : 0, misses = init ? a.tstate[4] : 0;
      calls++;
      if (off > 0) off--;
      else if (__hip_atomic_load(a.redo + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) {
        misses++;
        const int grown = (3 * burn + 1) / 2;
        if (3 * grown <= a.T) burn = grown;                  // (a burn-in beyond a third of the sequence: the cut no longer pays)
        else { off = kTsegCooldown; burn = a.tburn; }
      }
      a.tstate[1] = burn; a.tstate[2] = off; a.tstate[3] = calls; a.tstate[4] = misses;
      __hip_atomic_store(a.tstate, kTsegMagic, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a.loss_out) {
      double t = acc * (double)a.loss_scale;                 // -(num - den) [* 1/frames], pychain/loss.py:100-104, rounded once
      if (a.loss_norm_dev) t /= (double)__hip_atomic_load(a.loss_norm_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int nbad = 0;
      for (int i = 0; i < a.bad_words; i++) nbad += __hip_atomic_load(a.bad + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.loss_out[0] = (float)t; a.loss_out[1] = (float)frames; a.loss_out[2] = (float)nbad; a.loss_out[3] = (float)acc;
      a.loss_out[4] = (float)t;                              // (a second copy: the scalar a caller hands out, apart from the statistics)
      // [5]: time segments (DenArgs::tseg) - how many speculated rows did not verify (> 0: the call ran its recursions again, whole)
      a.loss_out[5] = a.tseg > 1 ? (float)__hip_atomic_load(a.redo + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
      a.loss_out[6] = (float)((a.tseg > 1 && !ts_off) ? a.tseg : 1);
      a.loss_out[7] = a.tseg > 1 ? __uint_as_float((unsigned int)__hip_atomic_load(a.redo + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.f;
    }
  }
}


// Which recursion segment makes frame t of a length-L sequence computable: its alpha'(t) row
// exists once the forward recursion has run t steps, its beta(t+1) row once the backward
// recursion has run L-1-t steps; segment s covers steps [seg_bound[s-1], seg_bound[s]).
// The frames of one occupancy launch, as a range of `need` = max(t, L-1-t): (lo, hi].  Read once per
// workgroup (seg_bound[] lives in the kernel arguments: dependent scalar loads, and the frame loops ask
// several times per frame).
struct LaunchFrames {
  int lo, hi;
  __host__ __device__ __forceinline__ explicit LaunchFrames(const DenArgs& a) : lo(-1), hi(0x7fffffff) {
    if (a.gam_nseg > 0) {
      if (a.gam_seg > 0) lo = a.seg_bound[a.gam_seg - 1];
      if (a.gam_seg < a.gam_nseg - 1) hi = a.seg_bound[a.gam_seg];
    }
  }
  __host__ __device__ __forceinline__ bool has(int t, int L) const {
    const int need = t > L - 1 - t ? t : L - 1 - t;
    return need > lo && need <= hi;
  }
};
// "crossing" calls (DenArgs::xf): the occupancy launch evaluates only the band around the middle of every time segment - the
// recursions emitted the rest themselves.  The segments are those whose rows stand: the call's own, or - after a splice miss, when
// the uncut launch recomputed everything - the whole sequence.
struct XfBand {
  int w, nseg_cut, tburn;
  __device__ __forceinline__ explicit XfBand(const DenArgs& a) : w(a.xf), nseg_cut(1), tburn(0) {
    if (w && a.tseg > 1 && __hip_atomic_load(a.redo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && !den_tseg_off(a)) { nseg_cut = a.tseg; tburn = den_tburn(a); }
  }
  __device__ __forceinline__ bool has(int t, int L) const {
    if (!w) return true;
    return den_xf_band_frame(t, L, (nseg_cut > 1 && L >= 2 * tburn) ? nseg_cut : 1, w);
  }
};
__device__ __forceinline__ int den_next_frame(int t, int t_end, int L, const LaunchFrames& lf, const XfBand& xb) {
  while (t < t_end && !(lf.has(t, L) && xb.has(t, L))) t++;
  return t;
}

// Occupancy launches of the segments after the first only hold the outer frames of a sequence: a left
// range [L-1-hi, L-1-lo) and a right range (lo, hi] (lo, hi = ends of the previous and of this segment).
// Their grid is sized for those ranges (den_compact_grid_x) and block k takes the k-th chunk of
// frames_per_block frames that meets them, instead of one workgroup per chunk of [0, T) of which most
// would find nothing to do (an idle workgroup still needs a whole free CU to start).  The first launch
// keeps the plain mapping: it also zeroes the padding of every chunk.  Returns -1: no chunk for this block.
__host__ __device__ __forceinline__ int den_chunk_of_block(int k, int L, const DenArgs& a) {
  if (a.gam_nseg == 0 || a.gam_seg == 0) return k;
  const int fpb = a.frames_per_block;
  const int lo = a.seg_bound[a.gam_seg - 1];
  const int hi = a.gam_seg == a.gam_nseg - 1 ? 0x3fffffff : a.seg_bound[a.gam_seg];
  const int a0 = L - 1 - hi > 0 ? L - 1 - hi : 0, a1 = L - 1 - lo;   // left frames [a0, a1)
  const int b0 = lo + 1, b1 = hi < L - 1 ? hi : L - 1;          // right frames [b0, b1]
  const int nl = a1 > a0 ? (a1 - 1) / fpb - a0 / fpb + 1 : 0;
  if (k < nl) return a0 / fpb + k;
  if (b1 < b0) return -1;
  int c0 = b0 / fpb;
  if (nl > 0 && c0 == (a1 - 1) / fpb) c0++;                     // that chunk went to the left range's last block
  const int c = c0 + (k - nl);
  return c <= b1 / fpb ? c : -1;
}
int den_compact_grid_x(const DenArgs& a) {
  const int fpb = a.frames_per_block, full = (a.T + fpb - 1) / fpb;
  if (a.gam_nseg == 0 || a.gam_seg == 0) return full;
  const int lo = a.seg_bound[a.gam_seg - 1];
  const int hi = a.gam_seg == a.gam_nseg - 1 ? a.T : a.seg_bound[a.gam_seg];
  return std::min(full, 2 * ((hi - lo + fpb - 1) / fpb + 1));
}

// ------------------------------------------------------------------------------------
// streamed occupancy pass: the queue of frame ranges (DenArgs::stream)
// ------------------------------------------------------------------------------------
// Frame t of a length-L sequence becomes computable when the alpha recursion has stored row t and the beta recursion row
// t+1, i.e. after max(t, L-1-t) steps: nothing before L/2, then ever faster, from the middle outwards.  The queue hands
// out, per sequence, rings of kStreamWidth frames on either side of the middle in that order - item id = padding of
// sequence id (id < B), else ((ring * B + b) * 2 + side) - so that ids are drawn in the order in which they become ready,
// whatever the lengths (which live on the device).  A workgroup draws an id, skips it if it holds no frame, waits until
// the two recursion workgroups of the sequence have reported the rows it needs (DenArgs::seq_progress) and evaluates it.
// No deadlock: the recursion workgroups were all resident before this kernel was released (its gate waits for every one
// of them to pass T/2), they wait for nobody, and every item only waits for them.
struct StreamItem { int b, lo, hi, L, pad; };
__device__ __forceinline__ bool stream_take(const DenArgs& a, int* slot, StreamItem& it) {
  __syncthreads();                                      // the previous item (and its slot) is done with
  if (threadIdx.x == 0) {
    const int B = a.B, total = B + stream_ring_count(a.T) * 2 * B;
    int b = -1, lo = 0, hi = 0, L = 0, pad = 0;
    for (;;) {
      const int id = atomicAdd(a.stream_next, 1);
      if (id >= total) { b = -1; break; }
      if (id < B) {                                     // frames past the sequence's end: exact zeros
        b = id; L = seq_len(a.lengths, b, a.T); lo = L; hi = a.T; pad = 1;
        if (lo < hi) break;
        continue;
      }
      const int q = id - B, r = q / (2 * B), rem = q - r * 2 * B, side = rem & 1;
      b = rem >> 1; L = seq_len(a.lengths, b, a.T); pad = 0;
      const int half = L / 2;
      int n0, n1;                                                           // the ring: need in [n0, n1)
      stream_ring(a.T, r, n0, n1);
      if (side) { lo = max(n0, half); hi = min(n1, L); }                    // right of the middle: computable after t steps
      else { lo = max(0, L - n1); hi = min(half, L - n0); }                 // left: after L - 1 - t steps
      if (lo >= hi) continue;
      // alpha rows lo .. hi-1 and beta rows lo+1 .. hi
      const int need_a = hi, need_b = L - lo;
      const unsigned long long t0 = wall_clock64();     // 100 MHz
      while (__hip_atomic_load(a.seq_progress + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need_a ||
             __hip_atomic_load(a.seq_progress + B + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need_b) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 2000000000ull) { atomicAdd(a.bad, 1); lo = hi; break; }   // 20 s: the recursion died
      }
      if (lo < hi) break;
    }
    slot[0] = b; slot[1] = lo; slot[2] = hi; slot[3] = L; slot[4] = pad;
  }
  __syncthreads();
  it.b = slot[0]; it.lo = slot[1]; it.hi = slot[2]; it.L = slot[3]; it.pad = slot[4];
  if (it.b < 0 && a.occ_done_target) {                  // the queue is empty: this workgroup is done (DenArgs::occ_done)
    __builtin_amdgcn_s_waitcnt(0);                      // ... and its stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(a.occ_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return it.b >= 0;
}
constexpr size_t kStaticLds = 64;       // the occupancy kernels' own static LDS (the queue slot) beside their dynamic segment
constexpr int kLoadDeviceScope = 16;     // cache-policy operand of a buffer load: sc1 (rows another XCD wrote while this kernel runs)

// ------------------------------------------------------------------------------------
// launch 2: occupancies (time-parallel)
// ------------------------------------------------------------------------------------
// STREAM = false: one workgroup per (chunk of frames_per_block frames, sequence), the frames of occupancy launch gam_seg.
// STREAM = true:  persistent workgroups drawing frame ranges from the queue above (one plan for all sequences).
// XH: 2-byte network output and gradient (DenArgs::x_half; float4 chunks only: VEC == 4, XCH > 0)
// `n` elements of a gradient slab from element i0 on -> exact zeros, whatever the element size (2-byte: n and i0 even - D % 4 == 0)
template <bool XH>
__device__ __forceinline__ void grad_zero(float* gseq, size_t i0, size_t i1, int tid, int nthreads) {
  if constexpr (XH) {
    uint32_t* g32 = reinterpret_cast<uint32_t*>(gseq);
    for (size_t i = i0 / 2 + tid; i < i1 / 2; i += nthreads) g32[i] = 0u;
  } else {
    for (size_t i = i0 + tid; i < i1; i += nthreads) gseq[i] = 0.f;
  }
}
template <int VEC, int XCH, int R, bool STREAM, bool XH = false>
__global__ __launch_bounds__(kNT) void den_gamma_kernel(const DenArgs a) {
  static_assert(!XH || (VEC == 4 && XCH > 0), "2-byte rows: chunks of four elements through registers");
  constexpr size_t kXe = XH ? 2 : 4;                   // bytes per nnet-output / gradient element
  const bool bf16 = a.x_half == kXBf16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ int stream_slot[8];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Hp = a.Hp, D = a.D, Dp = (D + 3) & ~3;
  int b = STREAM ? 0 : blockIdx.y;
  int L = STREAM ? 1 : __builtin_amdgcn_readfirstlane(seq_len(a.lengths, b, a.T));
  int t_first = 0, t_live_end = 0;
  const bool first_launch = a.gam_nseg == 0 || a.gam_seg == 0;
  const LaunchFrames lf(a);
  const XfBand xband(a);
  if constexpr (!STREAM) {
    const int chunk = den_chunk_of_block(blockIdx.x, L, a);
    if (chunk < 0) return;
    const int t_begin = chunk * a.frames_per_block;
    const int t_end = min(t_begin + a.frames_per_block, a.T);
    float* gseq0 = reinterpret_cast<float*>(reinterpret_cast<char*>(a.grad) + (size_t)b * a.T * D * kXe);
    if (t_begin >= L) {                               // whole chunk is padding: exact zeros (zeros_like, :58)
      if (first_launch) grad_zero<XH>(gseq0, (size_t)t_begin * D, (size_t)t_end * D, tid, kNT);
      return;
    }
    t_live_end = min(t_end, L);
    // padded tail of a chunk that straddles the sequence end
    if (first_launch && t_live_end < t_end) grad_zero<XH>(gseq0, (size_t)t_live_end * D, (size_t)t_end * D, tid, kNT);
    t_first = den_next_frame(t_begin, t_live_end, L, lf, xband);
    if (t_first >= t_live_end) return;                // none of this chunk's frames belongs to this launch
  }
  const char* plan = a.plans + (size_t)b * a.plan_stride;
  const PlanHeader* hd = reinterpret_cast<const PlanHeader*>(plan);
  // DenArgs::sg ("pdf by state" plans, plan_format.h: gamma_sg): the alpha store holds a(t+1,.) at row t and the occupancies are a
  // sum over STATES - the tile over states, and no nnet-output row (its rows are "read" as zeros: exp(0) = 1 multiplies)
  const bool sg = a.sg != 0;
  const TilePlan tp = sg ? hd->gamma_sg : hd->gamma;
  const WaveEntry we = reinterpret_cast<const WaveEntry*>(plan + tp.off_wave_tab)[wave];
  const GroupEntry* gtab = reinterpret_cast<const GroupEntry*>(plan + tp.off_group_tab);
  const uint2* slots = reinterpret_cast<const uint2*>(plan + tp.off_slots);
  const int32_t* row_pdf = reinterpret_cast<const int32_t*>(plan + (sg ? hd->off_row_pdf_sg : hd->off_row_pdf));

  float* U = reinterpret_cast<float*>(smem_raw);   // alpha'(t,.)   [Hp]
  float* V = U + Hp;                                // beta(t+1,.)   [Hp]
  float* xr = V + Hp;                               // exp x(t,.)    [Dp]
  float* q = xr + Dp;                               // per-pdf arc sums, natural pdf order [Dp]
  int* rmap = reinterpret_cast<int*>(q + Dp);       // plan row -> pdf-id [ngroups*64]
  float* red = reinterpret_cast<float*>(rmap + tp.ngroups * 64);   // [16]
  const uint32_t lds0 = lds_addr(smem_raw);

  GroupRegs groups;
  groups.load<R>(we, gtab, lane);
  const uint2* wave_slots = slots + (size_t)__builtin_amdgcn_readfirstlane(we.slot_row_begin) * 64 + lane;
  ArcRegs<R> arcs;
  arcs.load(groups.nslots, wave_slots, lds0, lds0 + 4u * (uint32_t)Hp);
  const uint2* tail_slots = wave_slots + (size_t)R * 64;

  if (tid < 16) red[tid] = 0.f;
  for (int i = tid; i < Dp; i += kNT) q[i] = 0.f;   // pdfs without arcs stay zero forever
  for (int i = tid; i < tp.ngroups * 64; i += kNT) rmap[i] = row_pdf[i];
  int bad = 0;
  const float gscale = a.grad_scale_dev ? a.grad_scale * *a.grad_scale_dev : a.grad_scale;
  XRow<kNT, VEC, XCH> xq;
  constexpr int kUV = 1;                             // float4 chunks of U and of V per thread (Hp <= 4*kNT fast path)
  float4 ureg[kUV], vreg[kUV];
  const bool uv_in_regs = Hp <= kUV * 4 * kNT;

  for (;;) {                                         // STREAM: one pass per item of the queue; else exactly one pass
    if constexpr (STREAM) {
      StreamItem it;
      if (!stream_take(a, stream_slot, it)) break;
      b = it.b; L = it.L;
      if (it.pad) {
        grad_zero<XH>(reinterpret_cast<float*>(reinterpret_cast<char*>(a.grad) + (size_t)b * a.T * D * kXe), (size_t)it.lo * D,
                      (size_t)it.hi * D, tid, kNT);
        continue;
      }
      t_first = it.lo; t_live_end = it.hi;
    }
    float* gseq = reinterpret_cast<float*>(reinterpret_cast<char*>(a.grad) + (size_t)b * a.T * D * kXe);
    const float* xseq = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x) + (size_t)b * a.T * D * kXe);
    const float* aseq = a.alpha_store + (size_t)b * a.T * Hp;
    const float* bseq = a.beta_store + (size_t)b * (a.T + 1) * Hp;
    const XBuf abuf = make_xbuf(aseq, (size_t)a.T * Hp * sizeof(float)), bbuf = make_xbuf(bseq, (size_t)(a.T + 1) * Hp * sizeof(float));
    // Software pipeline over frames: the global loads of frame t+1 (alpha', beta, nnet-output
    // rows) are issued into registers before frame t is evaluated and committed to LDS after
    // the last LDS read of frame t, so HBM latency is off the per-frame critical path.
    // (macros, not lambdas: by-reference captures would put the staging registers on the stack;
    //  STREAM: the state rows may have been written by another XCD while this kernel runs: device-scope loads)
#define GAMMA_PREFETCH(t)                                                                     \
  do {                                                                                        \
    if constexpr (XH) xq.load_h(reinterpret_cast<const char*>(xseq) + (size_t)(t) * D * 2, D, tid);   /* (raw) */        \
    else if (sg) { _Pragma("unroll") for (int i_ = 0; i_ < ((VEC * XCH) > 0 ? (VEC * XCH) : 1); i_++) xq.v[i_] = 0.f; }  \
    else xq.load(xseq + (size_t)(t) * D, D, tid);                                             \
    if (uv_in_regs) {                                                                         \
      const float* ar_ = aseq + (size_t)(t) * Hp;                                             \
      const float* br_ = bseq + (size_t)((t) + 1) * Hp;                                       \
      _Pragma("unroll") for (int c = 0; c < kUV; c++) {                                       \
        const int i = (c * kNT + tid) * 4;                                                    \
        if (i < Hp) {                                                                         \
          if constexpr (STREAM) {                                                             \
            const u32x4 ua_ = __builtin_amdgcn_raw_buffer_load_b128(abuf, i * 4, (t) * Hp * 4, kLoadDeviceScope);        \
            const u32x4 va_ = __builtin_amdgcn_raw_buffer_load_b128(bbuf, i * 4, ((t) + 1) * Hp * 4, kLoadDeviceScope);  \
            ureg[c] = make_float4(__uint_as_float(ua_.x), __uint_as_float(ua_.y), __uint_as_float(ua_.z), __uint_as_float(ua_.w)); \
            vreg[c] = make_float4(__uint_as_float(va_.x), __uint_as_float(va_.y), __uint_as_float(va_.z), __uint_as_float(va_.w)); \
          } else {                                                                            \
            ureg[c] = *reinterpret_cast<const float4*>(ar_ + i);                              \
            vreg[c] = *reinterpret_cast<const float4*>(br_ + i);                              \
          }                                                                                   \
        }                                                                                     \
      }                                                                                       \
    }                                                                                         \
  } while (0)
#define GAMMA_COMMIT(t)                                                                       \
  do {                                                                                        \
    if constexpr (XH) xq.convert_h(bf16);                                                     \
    xq.store(xr, XH ? nullptr : xseq + (size_t)(t) * D, D, tid, a.input_is_exp);              \
    if (uv_in_regs) {                                                                         \
      _Pragma("unroll") for (int c = 0; c < kUV; c++) {                                       \
        const int i = (c * kNT + tid) * 4;                                                    \
        if (i < Hp) { *reinterpret_cast<float4*>(U + i) = ureg[c];                            \
                      *reinterpret_cast<float4*>(V + i) = vreg[c]; }                          \
      }                                                                                       \
    } else {                                                                                  \
      for (int i = tid * 4; i < Hp; i += kNT * 4) {                                           \
        const u32x4 ua_ = __builtin_amdgcn_raw_buffer_load_b128(abuf, i * 4, (t) * Hp * 4, STREAM ? kLoadDeviceScope : 0);        \
        const u32x4 va_ = __builtin_amdgcn_raw_buffer_load_b128(bbuf, i * 4, ((t) + 1) * Hp * 4, STREAM ? kLoadDeviceScope : 0);  \
        *reinterpret_cast<float4*>(U + i) = make_float4(__uint_as_float(ua_.x), __uint_as_float(ua_.y), __uint_as_float(ua_.z), __uint_as_float(ua_.w)); \
        *reinterpret_cast<float4*>(V + i) = make_float4(__uint_as_float(va_.x), __uint_as_float(va_.y), __uint_as_float(va_.z), __uint_as_float(va_.w)); \
      }                                                                                       \
    }                                                                                         \
  } while (0)
    GAMMA_PREFETCH(t_first);
    GAMMA_COMMIT(t_first);
    __syncthreads();
    for (int t = t_first; t < t_live_end;) {
      float* grow = reinterpret_cast<float*>(reinterpret_cast<char*>(gseq) + (size_t)t * D * kXe);
      const int t_next = STREAM ? t + 1 : den_next_frame(t + 1, t_live_end, L, lf, xband);
      const bool have_next = t_next < t_live_end;
      if (have_next) GAMMA_PREFETCH(t_next);
      float s0 = 0.f, s1 = 0.f;
      tile_rows<R, 1>(arcs, groups, tail_slots, lane, U, V, q, rmap, nullptr, s0, s1);
      __syncthreads();
      float g[(VEC * XCH) > 0 ? (VEC * XCH) : 1];
      float part = 0.f;
      if constexpr (XCH > 0) {
#pragma unroll
        for (int c = 0; c < XCH; c++)
#pragma unroll
          for (int k = 0; k < VEC; k++) {
            const int e = (c * kNT + tid) * VEC + k;
            g[c * VEC + k] = e < D ? product_into_sum(xr[e], q[e], part) : 0.f;
          }
      } else {
        for (int e = tid; e < D; e += kNT) part += xr[e] * q[e];
      }
      part = wave_sum(part);
      if (lane == 0) red[wave] = part;
      __syncthreads();                                   // also: every read of U/V/xr of this frame is done
      const float tot = block_total(red, lane);
      const float sc = gscale / tot;
      if (!(tot > 0.f) || !(sc - sc == 0.f)) bad = 1;
      if (a.check && (t == 0 || a.check_all) && tid == 0) den_record_frame_total(a, b, t, tot);
      if constexpr (XCH > 0) {
#pragma unroll
        for (int c = 0; c < XCH; c++) {
          const int e = (c * kNT + tid) * VEC;
          if (e < D) {
            if constexpr (XH) {
              *reinterpret_cast<uint2*>(reinterpret_cast<char*>(grow) + (size_t)e * 2) =
                  make_uint2(pack_half2(g[c * 4] * sc, g[c * 4 + 1] * sc, bf16), pack_half2(g[c * 4 + 2] * sc, g[c * 4 + 3] * sc, bf16));
            } else if constexpr (VEC == 4) {
              *reinterpret_cast<float4*>(grow + e) =
                  make_float4(g[c * 4] * sc, g[c * 4 + 1] * sc, g[c * 4 + 2] * sc, g[c * 4 + 3] * sc);
            } else {
              grow[e] = g[c] * sc;
            }
          }
        }
        if (have_next) GAMMA_COMMIT(t_next);
      } else {
        for (int e = tid; e < D; e += kNT) grow[e] = xr[e] * q[e] * sc;
        __syncthreads();                                 // generic-D path re-reads xr/q above
        if (have_next) GAMMA_COMMIT(t_next);
      }
      __syncthreads();   // next frame's operands are in place; q is rewritten by the next frame
      t = t_next;
    }
#undef GAMMA_PREFETCH
#undef GAMMA_COMMIT
    if constexpr (!STREAM) break;
  }
  if (bad && lane == 0) atomicAdd(a.bad, 1);
}

// ------------------------------------------------------------------------------------
// launch 2, two-frame form: frames t and t+1 of a sequence are evaluated TOGETHER.
// The occupancy pass is time-parallel, so its cost is CU time, not latency, and what it
// spends per frame is LDS gather cycles (2 ds_read_b32 per arc), VALU/issue slots and three
// workgroup barriers.  Here alpha'(t,.)/alpha'(t+1,.) and beta(t+1,.)/beta(t+2,.) are
// interleaved in LDS as float2, so ONE ds_read_b64 per operand serves both frames at the
// LDS cost of one ds_read_b32, the arithmetic is packed fp32 (v_pk_mul/v_pk_fma), address
// unpacking and the group bookkeeping are shared, and a pair costs two barriers.  8 waves
// of up to 256 VGPRs: every wave keeps twice the arcs of the 16-wave form in registers.
// The nnet-output rows never enter LDS (each thread multiplies the elements it loaded).
// ------------------------------------------------------------------------------------
constexpr int kNW2 = PLAN_GAM2_WAVES;
constexpr int kNT2 = kNW2 * 64;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const v2f lds_cv2f;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
__device__ __forceinline__ v2f lds_abs2(uint32_t byte_addr) { return *(lds_cv2f*)(byte_addr); }
#pragma clang diagnostic pop

template <int R>
struct ArcRegs2 {
  uint32_t pk[R > 0 ? R : 1];       // absolute LDS byte addresses of the two float2 operands, packed 16:16
  float p[R > 0 ? R : 1];
  __device__ __forceinline__ void load(const int nslot_rows, const uint2* __restrict__ wave_slots,
                                       uint32_t lds_u, uint32_t lds_v) {
#pragma unroll
    for (int s = 0; s < R; s++) {
      uint2 a = make_uint2(0u, 0u);
      if (s < nslot_rows) a = wave_slots[s * 64];
      pk[s] = (lds_u + ((a.x & 0xffffu) << 3)) | ((lds_v + ((a.x >> 16) << 3)) << 16);
      p[s] = __uint_as_float(a.y);
    }
  }
  __device__ __forceinline__ void opaque4(int s) {
    asm volatile("" : "+v"(pk[s]), "+v"(pk[s + 1]), "+v"(pk[s + 2]), "+v"(pk[s + 3]));
  }
  __device__ __forceinline__ void gather(int s, v2f& u, v2f& v) {
    u = lds_abs2(pk[s] & 0xffffu); v = lds_abs2(pk[s] >> 16);
  }
};

// (u.x * p, u.y * p) as two scalar multiplies: a packed multiply by the splat {p, p} makes the
// optimiser keep a 64-bit copy of every arc probability in registers (3 VGPRs per arc instead of 2)
__device__ __forceinline__ v2f scale2(v2f u, float p) { return v2f{u.x * p, u.y * p}; }

__device__ __forceinline__ void tile_store2(v2f acc, int pos, float* __restrict__ q2, const int* __restrict__ row_map) {
  const int nat = row_map[pos];
  if (nat >= 0) *reinterpret_cast<v2f*>(q2 + 2 * nat) = acc;
}

// q2[2*pdf + f] = sum_k p_k * U2[2*i0_k + f] * V2[2*i1_k + f]   (f = 0, 1: the two frames)
// `hook(c)` runs at the start of chunk c (c is a constant after unrolling): the occupancy kernel issues its
// global loads for the next pair there, one per chunk, so that they pass through the CU's vector-memory
// path (64 B per clock) while the gathers keep the LDS busy, instead of in one burst before the arc work.
template <int R, typename Hook>
__device__ __forceinline__ void tile_rows2(ArcRegs2<R>& ar, const GroupRegs& gr, const uint2* __restrict__ tail_slots,
                                           int lane, const float* __restrict__ U2, const float* __restrict__ V2,
                                           float* __restrict__ q2, const int* __restrict__ row_map, Hook hook) {
  constexpr int kChunk = 4;
  static_assert(R % kChunk == 0 && R <= 64 && PYCHAIN_CHUNK == 4, "chunk mask of GroupRegs is built for chunks of 4");
  constexpr int NC = R / kChunk;
  v2f acc = {0.f, 0.f};
  uint32_t m_lo = (uint32_t)gr.endmask, m_hi = (uint32_t)(gr.endmask >> 32), cm = gr.chunkmask;
  asm volatile("" : "+s"(m_lo), "+s"(m_hi), "+s"(cm));
  v2f ub[2][kChunk], vb[2][kChunk];
  if (R > 0) {
    ar.opaque4(0);
#pragma unroll
    for (int k = 0; k < kChunk; k++) ar.gather(k, ub[0][k], vb[0][k]);
  }
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int cb = c & 1;
    wave_priority_by_progress<NC>(c);
    hook(c);
    if (c + 1 < NC) {
      ar.opaque4((c + 1) * kChunk);
#pragma unroll
      for (int k = 0; k < kChunk; k++) ar.gather((c + 1) * kChunk + k, ub[cb ^ 1][k], vb[cb ^ 1][k]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < NC) PYCHAIN_WAIT_LGKM(2 * kChunk); else PYCHAIN_WAIT_LGKM(0);
    __builtin_amdgcn_sched_barrier(0);
    v2f nacc = acc;
#pragma unroll
    for (int k = 0; k < kChunk; k++) nacc = __builtin_elementwise_fma(scale2(ub[cb][k], ar.p[c * kChunk + k]), vb[cb][k], nacc);
    if (__builtin_expect(((cm >> c) & 1u) != 0u, 0)) {
      nacc = acc;
#pragma unroll
      for (int k = 0; k < kChunk; k++) {
        const int sidx = c * kChunk + k;
        nacc = __builtin_elementwise_fma(scale2(ub[cb][k], ar.p[sidx]), vb[cb][k], nacc);
        if (((sidx < 32 ? m_lo : m_hi) >> (sidx & 31)) & 1u) {
          const uint32_t lo_before = sidx < 32 ? (m_lo & ((1u << (sidx & 31)) - 1u)) : m_lo;
          const uint32_t hi_before = sidx < 32 ? 0u : (m_hi & ((1u << (sidx & 31)) - 1u));
          const int g = __builtin_popcount(lo_before) + __builtin_popcount(hi_before);
          tile_store2(nacc, __builtin_amdgcn_readlane(gr.base, g) + lane, q2, row_map);
          nacc = v2f{0.f, 0.f};
        }
      }
    }
    acc = nacc;
  }
  int g = __builtin_popcount(m_lo) + __builtin_popcount(m_hi);
  if (gr.nslots > R) {                           // plan larger than the register budget: stream the tail
    const uint2* sp = tail_slots;
    g = gr.tail_g;
    int cur_base = __builtin_amdgcn_readlane(gr.base, g & 63);
    int remaining = gr.tail_rem;
    for (int s = R; s < gr.nslots; s++) {
      const uint2 a = *sp;
      sp += 64;
      const v2f u = *reinterpret_cast<const v2f*>(U2 + 2 * (a.x & 0xffffu));
      const v2f v = *reinterpret_cast<const v2f*>(V2 + 2 * (a.x >> 16));
      acc = __builtin_elementwise_fma(scale2(u, __uint_as_float(a.y)), v, acc);
      if (--remaining == 0) {
        tile_store2(acc, cur_base + lane, q2, row_map);
        acc = v2f{0.f, 0.f};
        g++;
        cur_base = __builtin_amdgcn_readlane(gr.base, g & 63);
        remaining = __builtin_amdgcn_readlane(gr.n, g & 63);
      }
    }
  }
  for (; g < gr.ngroups; g++)
    tile_store2(v2f{0.f, 0.f}, __builtin_amdgcn_readlane(gr.base, g & 63) + lane, q2, row_map);
}

__device__ __forceinline__ bool den_frame_in_launch(int t, int t_live_end, int L, const LaunchFrames& lf) {
  return t < t_live_end && lf.has(t, L);
}

// STREAM as in den_gamma_kernel: persistent workgroups drawing frame ranges from the queue (one plan for all sequences).
template <int XCH, int R, bool STREAM, bool XH = false>
__global__ __launch_bounds__(kNT2) void den_gamma2_kernel(const DenArgs a) {
  constexpr size_t kXe = XH ? 2 : 4;                   // bytes per nnet-output / gradient element (DenArgs::x_half)
  const bool bf16 = a.x_half == kXBf16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ int stream_slot[8];
  constexpr int UVC = 2;                              // float4 chunks of a state row per thread: Hp <= 4 * UVC * kNT2
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Hp = a.Hp, D = a.D, T = a.T, Dp = (D + 3) & ~3;
  int b = STREAM ? 0 : blockIdx.y;
  int L = STREAM ? 1 : __builtin_amdgcn_readfirstlane(seq_len(a.lengths, b, a.T));
  const bool first_launch = a.gam_nseg == 0 || a.gam_seg == 0;
  const LaunchFrames lf(a);
  const XfBand xband(a);
  int t0 = 0, t_live_end = 0, t_lo = 0;               // frames [t_lo, t_live_end) of this pass, t0 = first pair (even)
  if constexpr (!STREAM) {
    const int chunk = den_chunk_of_block(blockIdx.x, L, a);
    if (chunk < 0) return;
    const int t_begin = chunk * a.frames_per_block;               // frames_per_block is even
    const int t_end = min(t_begin + a.frames_per_block, T);
    float* gseq0 = reinterpret_cast<float*>(reinterpret_cast<char*>(a.grad) + (size_t)b * T * D * kXe);
    t_live_end = min(t_end, L);
    // pairs (t0, t0+1), t0 even, with at least one frame of this launch
    t0 = t_begin;
    while (t0 < t_live_end && !(den_frame_in_launch(t0, t_live_end, L, lf) && xband.has(t0, L)) &&
           !(den_frame_in_launch(t0 + 1, t_live_end, L, lf) && xband.has(t0 + 1, L))) t0 += 2;
    // padding of the first launch is exact zeros: a whole chunk past the end, or the tail of one that straddles it
    if (first_launch && max(t_begin, t_live_end) < t_end)
      grad_zero<XH>(gseq0, (size_t)max(t_begin, t_live_end) * D, (size_t)t_end * D, tid, kNT2);
    if (t0 >= t_live_end) return;                     // nothing to evaluate
  }
  const char* plan = a.plans + (size_t)b * a.plan_stride;
  const PlanHeader* hd = reinterpret_cast<const PlanHeader*>(plan);
  const bool sg = a.sg != 0;                          // (the tile over states of a "pdf by state" plan: den_gamma_kernel)
  const TilePlan tp = sg ? hd->gamma2_sg : hd->gamma2;
  const WaveEntry we = reinterpret_cast<const WaveEntry*>(plan + tp.off_wave_tab)[wave];
  const GroupEntry* gtab = reinterpret_cast<const GroupEntry*>(plan + tp.off_group_tab);
  const uint2* slots = reinterpret_cast<const uint2*>(plan + tp.off_slots);
  const int32_t* row_pdf = reinterpret_cast<const int32_t*>(plan + (sg ? hd->off_row_pdf_sg : hd->off_row_pdf));

  float* U2 = reinterpret_cast<float*>(smem_raw);    // [Hp] x {alpha'(t0,.), alpha'(t0+1,.)}
  float* V2 = U2 + 2 * Hp;                            // [Hp] x {beta(t0+1,.), beta(t0+2,.)}
  float* q2 = V2 + 2 * Hp;                            // [Dp] x {frame 0, frame 1} per-pdf arc sums, natural pdf order
  int* rmap = reinterpret_cast<int*>(q2 + 2 * Dp);   // plan row -> pdf-id [ngroups*64]
  float* red = reinterpret_cast<float*>(rmap + tp.ngroups * 64);   // [2][16]
  float* n2 = red + 32;                               // [2][Dp] x {frame 0, frame 1} numerator occupancies (fold only; one buffer per pair parity)
  const uint32_t lds0 = lds_addr(smem_raw);

  GroupRegs groups;
  groups.load<R>(we, gtab, lane);
  const uint2* wave_slots = slots + (size_t)__builtin_amdgcn_readfirstlane(we.slot_row_begin) * 64 + lane;
  ArcRegs2<R> arcs;
  arcs.load(groups.nslots, wave_slots, lds0, lds0 + 8u * (uint32_t)Hp);
  const uint2* tail_slots = wave_slots + (size_t)R * 64;

  if (tid < 32) red[tid] = 0.f;
  for (int i = tid; i < 2 * Dp; i += kNT2) q2[i] = 0.f;          // pdfs without arcs stay zero forever
  for (int i = tid; i < tp.ngroups * 64; i += kNT2) rmap[i] = row_pdf[i];
  int bad = 0;
  const float gscale = a.grad_scale_dev ? a.grad_scale * *a.grad_scale_dev : a.grad_scale;
  const bool fold = a.fold_rows != nullptr;
  const float nscale = a.grad_scale_dev ? a.fold_scale * *a.grad_scale_dev : a.fold_scale;
  int prev_U = 0, prev_b = -1;
  // state rows of a pair: global -> registers (one pair ahead) -> LDS, interleaved
  float4 ua[UVC], ub[UVC], va[UVC], vb[UVC];
  XRow<kNT2, 4, XCH> x0, x1;
#ifdef PYCHAIN_PROFILE_PHASES
  unsigned long long g2ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, g2t = PH_T();
  int g2pairs = 0;
#define G2PH(i) do { const unsigned long long n_ = PH_T(); g2ph[i] += n_ - g2t; g2t = n_; } while (0)
#else
#define G2PH(i) (void)0
#endif

  for (;;) {                                          // STREAM: one pass per item of the queue; else exactly one pass
    if constexpr (STREAM) {
      StreamItem it;
      if (!stream_take(a, stream_slot, it)) break;
      b = it.b; L = it.L;
      if (it.pad) {
        grad_zero<XH>(reinterpret_cast<float*>(reinterpret_cast<char*>(a.grad) + (size_t)b * T * D * kXe), (size_t)it.lo * D,
                      (size_t)it.hi * D, tid, kNT2);
        continue;
      }
      t_lo = it.lo; t_live_end = it.hi; t0 = t_lo & ~1;
    }
    float* gseq = reinterpret_cast<float*>(reinterpret_cast<char*>(a.grad) + (size_t)b * T * D * kXe);
    const float* xseq = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x) + (size_t)b * T * D * kXe);
    const float* aseq = a.alpha_store + (size_t)b * T * Hp;
    const float* bseq = a.beta_store + (size_t)b * (T + 1) * Hp;
    const XBuf abuf = make_xbuf(aseq, (size_t)T * Hp * sizeof(float)), bbuf = make_xbuf(bseq, (size_t)(T + 1) * Hp * sizeof(float));
    // which frames of [t0, ..) this pass evaluates
    auto wanted = [&](int t) { return STREAM ? (t >= t_lo && t < t_live_end) : (den_frame_in_launch(t, t_live_end, L, lf) && xband.has(t, L)); };
    // numerator fold: thread u (and u + kNT2) owns the u-th distinct pdf of the sequence; the set of
    // touched pdfs is the same in every frame, so n2 needs no clearing between pairs (only between sequences)
    const int U = fold ? a.fold_ucount[b] : 0;
    const int32_t* upd = a.fold_upd + (size_t)b * a.fold_K;
    const float* frows = a.fold_rows + (size_t)b * T * a.fold_K;
    int pd0 = -1, pd1 = -1;
    if (fold) {
      if (b != prev_b) {
        if (!STREAM || prev_b < 0) { for (int i = tid; i < 4 * Dp; i += kNT2) n2[i] = 0.f; }
        else {
          // the previous sequence's pdfs back to zero (the entries it touched, not the whole table)
          const int32_t* pupd = a.fold_upd + (size_t)prev_b * a.fold_K;
          for (int u = tid; u < prev_U; u += kNT2) { const int n = pupd[u]; n2[2 * n] = 0.f; n2[2 * n + 1] = 0.f; n2[2 * Dp + 2 * n] = 0.f; n2[2 * Dp + 2 * n + 1] = 0.f; }
        }
        prev_b = b; prev_U = U;
        __syncthreads();
      }
      if (tid < U) pd0 = upd[tid];
      if (tid + kNT2 < U) pd1 = upd[tid + kNT2];
    }
    // load number n of a pair's state rows: n = 4 * c + {0: alpha'(t), 1: alpha'(t+1), 2: beta(t+1), 3: beta(t+2)}
    // (STREAM: device-scope loads - another XCD may have written the rows while this kernel runs)
    auto row_load = [&](XBuf buf, int row, int i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(buf, i * 4, row * Hp * 4, STREAM ? kLoadDeviceScope : 0);
      return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto prefetch_one = [&](int n, int t) {
      const int c = n >> 2, i = (c * kNT2 + tid) * 4;
      if (c < UVC && i < Hp) {
        switch (n & 3) {
          case 0: ua[c] = row_load(abuf, t, i); break;
          case 1: ub[c] = row_load(abuf, min(t + 1, T - 1), i); break;
          case 2: va[c] = row_load(bbuf, t + 1, i); break;
          default: vb[c] = row_load(bbuf, min(t + 2, T), i); break;
        }
      }
    };
#define GAMMA2_COMMIT()                                                                          \
  do {                                                                                           \
    _Pragma("unroll") for (int c = 0; c < UVC; c++) {                                            \
      const int i = (c * kNT2 + tid) * 4;                                                        \
      if (i < Hp) {                                                                              \
        *reinterpret_cast<float4*>(U2 + 2 * i) = make_float4(ua[c].x, ub[c].x, ua[c].y, ub[c].y); \
        *reinterpret_cast<float4*>(U2 + 2 * i + 4) = make_float4(ua[c].z, ub[c].z, ua[c].w, ub[c].w); \
        *reinterpret_cast<float4*>(V2 + 2 * i) = make_float4(va[c].x, vb[c].x, va[c].y, vb[c].y); \
        *reinterpret_cast<float4*>(V2 + 2 * i + 4) = make_float4(va[c].z, vb[c].z, va[c].w, vb[c].w); \
      }                                                                                          \
    }                                                                                            \
  } while (0)
    for (int n = 0; n < 4 * UVC; n++) prefetch_one(n, t0);
    GAMMA2_COMMIT();
    __syncthreads();
    int npar = 0;
    while (t0 < t_live_end) {
#ifdef PYCHAIN_PROFILE_PHASES
      g2pairs++;
#endif
      const bool valid0 = wanted(t0), valid1 = wanted(t0 + 1);
      int tn = t0 + 2;
      while (tn < t_live_end && !wanted(tn) && !wanted(tn + 1)) tn += 2;
      const bool have_next = tn < t_live_end;
      // this pair's nnet-output rows (used after the arc work) and the next pair's state rows
      if constexpr (XH) {
        x0.load_h(reinterpret_cast<const char*>(xseq) + (size_t)t0 * D * 2, D, tid);            // (raw: converted behind the arc work)
        x1.load_h(reinterpret_cast<const char*>(xseq) + (size_t)min(t0 + 1, T - 1) * D * 2, D, tid);
      } else if (sg) {                                   // no nnet-output row: exp(0) = 1 multiplies
#pragma unroll
        for (int i_ = 0; i_ < 4 * XCH; i_++) { x0.v[i_] = 0.f; x1.v[i_] = 0.f; }
      } else {
        x0.load(xseq + (size_t)t0 * D, D, tid);
        x1.load(xseq + (size_t)min(t0 + 1, T - 1) * D, D, tid);
      }
      float r00 = 0.f, r01 = 0.f, r10 = 0.f, r11 = 0.f;  // numerator rows of this pair (in flight during the arc work)
      const float* fr0 = frows + (size_t)t0 * a.fold_K;
      const float* fr1 = frows + (size_t)min(t0 + 1, T - 1) * a.fold_K;
      if (pd0 >= 0) { r00 = fr0[tid]; r10 = fr1[tid]; }
      if (pd1 >= 0) { r01 = fr0[tid + kNT2]; r11 = fr1[tid + kNT2]; }
      G2PH(0);
      // the next pair's state rows are requested inside the arc loop, one load per chunk (a pair that is the
      // last of its pass re-reads its own rows: no branch per chunk)
      const int tpre = have_next ? tn : t0;
      tile_rows2<R>(arcs, groups, tail_slots, lane, U2, V2, q2, rmap, [&](int c) {
        constexpr int NC = R / 4 > 0 ? R / 4 : 1;
        // spread over the first chunks, two chunks apart where the loop is long enough
        constexpr int kStep = NC >= 16 ? 2 : 1;
        if (c % kStep == 0 && c / kStep < 4 * UVC) prefetch_one(c / kStep, tpre);
      });
      if (R / 4 < 4 * UVC * (R / 4 >= 16 ? 2 : 1))          // arc loop shorter than the list of loads: the rest here
        for (int n = (R / 4) / (R / 4 >= 16 ? 2 : 1); n < 4 * UVC; n++) prefetch_one(n, tpre);
      G2PH(1);
      float* n2p = n2 + (npar ? 2 * Dp : 0);             // this buffer was last read two pairs ago
      if (fold) {
        if (pd0 >= 0) *reinterpret_cast<v2f*>(n2p + 2 * pd0) = v2f{r00, r10};
        if (pd1 >= 0) *reinterpret_cast<v2f*>(n2p + 2 * pd1) = v2f{r01, r11};
        for (int u = tid + 2 * kNT2; u < U; u += kNT2) *reinterpret_cast<v2f*>(n2p + 2 * upd[u]) = v2f{fr0[u], fr1[u]};
      }
      npar ^= 1;
      G2PH(2);
      __syncthreads();                                   // q2 (and n2) complete; every gather of this pair is done
      G2PH(3);
      float g0[4 * XCH], g1[4 * XCH];
      float part0 = 0.f, part1 = 0.f;
      auto products = [&](auto mode) {                   // (the mode is uniform: one branch, not a select per element)
#pragma unroll
        for (int c = 0; c < XCH; c++) {
          const int e = (c * kNT2 + tid) * 4;
          float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa;
          if (e < D) { qa = *reinterpret_cast<const float4*>(q2 + 2 * e); qb = *reinterpret_cast<const float4*>(q2 + 2 * e + 4); }
          const float qf0[4] = {qa.x, qa.z, qb.x, qb.z}, qf1[4] = {qa.y, qa.w, qb.y, qb.w};
#pragma unroll
          for (int k = 0; k < 4; k++) {
            g0[c * 4 + k] = e < D ? product_into_sum(clamp_exp(x0.v[c * 4 + k], decltype(mode)::value), qf0[k], part0) : 0.f;
            g1[c * 4 + k] = e < D ? product_into_sum(clamp_exp(x1.v[c * 4 + k], decltype(mode)::value), qf1[k], part1) : 0.f;
          }
        }
      };
      if constexpr (XH) { x0.convert_h(bf16); x1.convert_h(bf16); }
      if (a.input_is_exp == kXExpClamp) products(std::integral_constant<int, kXExpClamp>{});
      else if (a.input_is_exp == kXIdentity) products(std::integral_constant<int, kXIdentity>{});
      else products(std::integral_constant<int, kXClamp>{});
      part0 = wave_sum(part0); part1 = wave_sum(part1);
      if (lane == 0) { red[wave] = part0; red[16 + wave] = part1; }
      G2PH(4);
      if (have_next) GAMMA2_COMMIT();                    // U2/V2 are free since the barrier above
      G2PH(5);
      __syncthreads();                                   // totals visible; next operands in place; q2 read
      G2PH(6);
      const float tot0 = block_total(red, lane), tot1 = block_total(red + 16, lane);
      const float sc0 = gscale / tot0, sc1 = gscale / tot1;
      if (valid0 && (!(tot0 > 0.f) || !(sc0 - sc0 == 0.f))) bad = 1;
      if (valid1 && (!(tot1 > 0.f) || !(sc1 - sc1 == 0.f))) bad = 1;
      if (a.check && tid == 0) {
        if (valid0 && (t0 == 0 || a.check_all)) den_record_frame_total(a, b, t0, tot0);
        if (valid1 && a.check_all) den_record_frame_total(a, b, t0 + 1, tot1);
      }
      float* grow0 = reinterpret_cast<float*>(reinterpret_cast<char*>(gseq) + (size_t)t0 * D * kXe);
      float* grow1 = reinterpret_cast<float*>(reinterpret_cast<char*>(grow0) + (size_t)D * kXe);
#pragma unroll
      for (int c = 0; c < XCH; c++) {
        const int e = (c * kNT2 + tid) * 4;
        if (e < D) {
          float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;     // numerator occupancies {f0,f1} x 4 pdfs
          if (fold) { na = *reinterpret_cast<const float4*>(n2p + 2 * e); nb = *reinterpret_cast<const float4*>(n2p + 2 * e + 4); }
          // (g * sc) rounded, then + numerator: bit-identical to the unfused order (occupancy pass, then
          // the numerator accumulated into the stored gradient)
#define G2(gv, scv, nv) mul_add_mul_rn((gv), (scv), (nv), nscale)
          if constexpr (XH) {                            // the same fp32 values, rounded once to the network output's type
            if (valid0) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(grow0) + (size_t)e * 2) =
                make_uint2(pack_half2(G2(g0[c * 4], sc0, na.x), G2(g0[c * 4 + 1], sc0, na.z), bf16),
                           pack_half2(G2(g0[c * 4 + 2], sc0, nb.x), G2(g0[c * 4 + 3], sc0, nb.z), bf16));
            if (valid1) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(grow1) + (size_t)e * 2) =
                make_uint2(pack_half2(G2(g1[c * 4], sc1, na.y), G2(g1[c * 4 + 1], sc1, na.w), bf16),
                           pack_half2(G2(g1[c * 4 + 2], sc1, nb.y), G2(g1[c * 4 + 3], sc1, nb.w), bf16));
          } else {
          if (valid0) *reinterpret_cast<float4*>(grow0 + e) = make_float4(G2(g0[c * 4], sc0, na.x), G2(g0[c * 4 + 1], sc0, na.z),
                                                                             G2(g0[c * 4 + 2], sc0, nb.x), G2(g0[c * 4 + 3], sc0, nb.z));
          if (valid1) *reinterpret_cast<float4*>(grow1 + e) = make_float4(G2(g1[c * 4], sc1, na.y), G2(g1[c * 4 + 1], sc1, na.w),
                                                                             G2(g1[c * 4 + 2], sc1, nb.y), G2(g1[c * 4 + 3], sc1, nb.w));
          }
#undef G2
        }
      }
      t0 = tn;
      G2PH(7);
    }
#undef GAMMA2_COMMIT
    if constexpr (!STREAM) break;
  }
#ifdef PYCHAIN_PROFILE_PHASES
  if (lane == 0 && b == 0 && (wave == 0 || wave == 7) && g2pairs > 0 && blockIdx.x == 20)
    printf("gamma2 wave %d pairs %d cycles/pair: issue-loads %llu arcs %llu fold %llu bar1 %llu products %llu commit %llu bar2 %llu scale+store %llu\n",
           wave, g2pairs, g2ph[0] / g2pairs, g2ph[1] / g2pairs, g2ph[2] / g2pairs, g2ph[3] / g2pairs, g2ph[4] / g2pairs,
           g2ph[5] / g2pairs, g2ph[6] / g2pairs, g2ph[7] / g2pairs);
#endif
  if (bad && lane == 0) atomicAdd(a.bad, 1);
}

__global__ void den_gate_kernel(const int32_t* progress, int target, int32_t* bad) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();              // 100 MHz
  // (relaxed: an acquire here would invalidate this XCD's L2 on every poll; the kernel that follows in
  // stream order acquires at its start)
  while (__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > 2000000000ull) { atomicAdd(bad, 1); break; }   // 20 s: the recursion died
  }
}

// the two-frame occupancy kernel: what it supports, and its launch
inline size_t gamma2_lds_bytes(const DenArgs& a, int gamma_max_groups) {
  return sizeof(float) * (4 * (size_t)a.Hp + (a.fold_rows ? 6 : 2) * (size_t)((a.D + 3) & ~3) + (size_t)gamma_max_groups * 64 + 32);
}
inline bool gamma2_eligible(const DenArgs& a, int rows2, int gamma_max_groups) {
  const bool off = a.knobs.gamma16 != 0;                               // test / tuning option: force the one-frame kernel
  return !off && rows2 > 0 && a.D % 4 == 0 && a.D <= 4 * 2 * kNT2 && a.Hp <= 4032 /* packed 16-bit addresses of float2 */ &&
         a.frames_per_block % 2 == 0 && gamma2_lds_bytes(a, gamma_max_groups) + kStaticLds <= 160 * 1024;
}
template <int XCH, bool STREAM, bool XH>
hipError_t launch_gamma2_x(const DenArgs& a, int rows2, size_t lds, dim3 grid, hipStream_t st) {
  if (rows2 <= 16) return launch_one(den_gamma2_kernel<XCH, 16, STREAM, XH>, a, grid, lds, st, kNT2);
  if (rows2 <= 32) return launch_one(den_gamma2_kernel<XCH, 32, STREAM, XH>, a, grid, lds, st, kNT2);
  return launch_one(den_gamma2_kernel<XCH, 64, STREAM, XH>, a, grid, lds, st, kNT2);
}
template <int XCH, bool STREAM>
hipError_t launch_gamma2(const DenArgs& a, int rows2, size_t lds, dim3 grid, hipStream_t st) {
  return a.x_half ? launch_gamma2_x<XCH, STREAM, true>(a, rows2, lds, grid, st) : launch_gamma2_x<XCH, STREAM, false>(a, rows2, lds, grid, st);
}
// the streamed form of the one-frame kernel: float4 rows, every arc of a wave in registers
template <int XCH, bool XH>
hipError_t launch_gamma_stream_x(const DenArgs& a, int r, size_t lds, dim3 grid, hipStream_t st) {
  if (r <= 16) return launch_one(den_gamma_kernel<4, XCH, 16, true, XH>, a, grid, lds, st);
  if (r <= 32) return launch_one(den_gamma_kernel<4, XCH, 32, true, XH>, a, grid, lds, st);
  return launch_one(den_gamma_kernel<4, XCH, kMaxResident, true, XH>, a, grid, lds, st);
}
template <int XCH>
hipError_t launch_gamma_stream(const DenArgs& a, int r, size_t lds, dim3 grid, hipStream_t st) {
  return a.x_half ? launch_gamma_stream_x<XCH, true>(a, r, lds, grid, st) : launch_gamma_stream_x<XCH, false>(a, r, lds, grid, st);
}
inline bool gamma_stream_shape_ok(const DenArgs& a, int hint, int gamma_max_groups) {
  if (a.plan_stride != 0) return false;
  if (gamma2_eligible(a, (hint >> 20) & 127, gamma_max_groups)) return true;
  const int r = pick_r(a, (hint >> 10) & 511, 2 * a.Hp);
  return a.D % 4 == 0 && a.D <= 4 * 4 * kNT && r > 0;
}

template <int VEC, int XCH>
hipError_t launch_r(const DenArgs& a, int hint, size_t lds_rec, size_t lds_gam, int gx, hipStream_t st, int gamma_max_groups) {
  hipError_t e = hipSuccess;
  if (a.phase_mask & 1) {                                // (den_lazy.hip / den_rec.hip)
    e = (a.pair || a.lazy) ? launch_den_lazy_family(a, hint, st) : launch_den_rec2b(a, hint, lds_rec, st);
    if (e != hipSuccess) return e;
  }
  if (a.phase_mask & 2) {
    const bool stream = (a.stream & 2) != 0;            // ONE persistent launch over the whole queue (DenArgs::stream)
    const dim3 grid = stream ? dim3(a.stream_blocks) : dim3(gx, a.B);
    if (gamma2_eligible(a, (hint >> 20) & 127, gamma_max_groups)) {
      const size_t lds2 = gamma2_lds_bytes(a, gamma_max_groups);
      const int r2 = a.sg ? PLAN_RESIDENT_0 : (hint >> 20) & 127;       // (the tile over states: at most PLAN_RESIDENT_0 rows per wave)
      if (stream) return a.D <= 4 * kNT2 ? launch_gamma2<1, true>(a, r2, lds2, grid, st) : launch_gamma2<2, true>(a, r2, lds2, grid, st);
      return a.D <= 4 * kNT2 ? launch_gamma2<1, false>(a, r2, lds2, grid, st) : launch_gamma2<2, false>(a, r2, lds2, grid, st);
    }
    const int r = a.sg ? PLAN_RESIDENT_0 : pick_r(a, (hint >> 10) & 511, 2 * a.Hp);
    if (stream) {
      if constexpr (VEC == 4 && XCH > 0) return launch_gamma_stream<XCH>(a, r, lds_gam, grid, st);
      else return hipErrorInvalidValue;                 // (den_stream_eligible said no)
    }
    if (a.x_half) {                                      // 2-byte rows: the float4-chunk forms only (den_call_half_native)
      if constexpr (VEC == 4 && XCH > 0) {
        switch (r) {
          case 0: return hipErrorInvalidValue;
          case 16: return launch_one(den_gamma_kernel<4, XCH, 16, false, true>, a, grid, lds_gam, st);
          case 32: return launch_one(den_gamma_kernel<4, XCH, 32, false, true>, a, grid, lds_gam, st);
          default: return launch_one(den_gamma_kernel<4, XCH, kMaxResident, false, true>, a, grid, lds_gam, st);
        }
      } else return hipErrorInvalidValue;
    }
    switch (r) {
      case 0: e = launch_one(den_gamma_kernel<VEC, XCH, 0, false>, a, grid, lds_gam, st); break;
      case 16: e = launch_one(den_gamma_kernel<VEC, XCH, 16, false>, a, grid, lds_gam, st); break;
      case 32: e = launch_one(den_gamma_kernel<VEC, XCH, 32, false>, a, grid, lds_gam, st); break;
      default: e = launch_one(den_gamma_kernel<VEC, XCH, kMaxResident, false>, a, grid, lds_gam, st); break;
    }
  }
  return e;
}

}  // namespace

// Host-side view of the frame -> (occupancy launch, workgroup) mapping for the CPU tests (tests/test_api.py):
// out[0] = grid.x of the launch, out[1 + k] = chunk of block k (or -1); returns 1 if frame t belongs to the launch.
int den_debug_launch_map(int T, int L, int t, int frames_per_block, int nseg, const int* seg_bound, int seg,
                         int* out, int out_len) {
  DenArgs a;
  memset(&a, 0, sizeof(a));
  a.T = T; a.frames_per_block = frames_per_block; a.gam_nseg = nseg; a.gam_seg = seg;
  for (int s = 0; s < nseg && s < 16; s++) a.seg_bound[s] = seg_bound[s];
  const int gx = den_compact_grid_x(a);
  if (out && out_len > 0) out[0] = gx;
  for (int k = 0; out && k < gx && 1 + k < out_len; k++) out[1 + k] = den_chunk_of_block(k, L, a);
  return LaunchFrames(a).has(t, L) && t < L ? 1 : 0;
}

namespace {
// ---- nnet-output rows exp'd ahead of the recursions (DenArgs::ex) --------------------------------------------------
// Workgroup (sequence b, end e, q) walks the half of the sequence at that end from the end towards the middle in ROUNDS of NR
// rows, taking the rounds q, q + Q, q + 2Q, ...: loads, clamp / exp (the recursions' own clamp_exp: bit-identical rows),
// device-scope stores and - once those are acknowledged - the count of its complete rounds.  A round is an HBM round trip each
// way (~10 us measured); Q workgroups x NR rows per round keep an end 3x ahead of a recursion that consumes a row per ~2 us
// (one workgroup per end, 4 rows per round: 2.5 us per row, and the recursions waited for it: profiles/r04_t_*).
constexpr int kExNT = 512;
template <int CH, bool XH = false>                     // XH: 2-byte network output (DenArgs::x_half); the rows written are fp32 either way
__global__ __launch_bounds__(kExNT) void den_exp_rows_kernel(const DenArgs a) {
  const bool bf16 = a.x_half == kXBf16;
  constexpr int NR = CH <= 2 ? 8 : 4;
  const int tid = threadIdx.x;
  const int Q = a.ex_q;
  const int b = blockIdx.x % a.B, end = (blockIdx.x / a.B) & 1, qi = blockIdx.x / (2 * a.B);
  const int L = __builtin_amdgcn_readfirstlane(seq_len(a.lengths, b, a.T)), D = a.D;
  const int Lh = (L + 1) / 2, n = end ? L - Lh : Lh;             // end 0 owns rows [0, Lh), end 1 rows [Lh, L) from L-1 down
  const XBuf xin = XH ? make_xbuf(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x) + (size_t)b * a.T * D * 2), (size_t)a.T * D * 2)
                      : make_xbuf(a.x + (size_t)b * a.T * D, (size_t)a.T * D * sizeof(float));
  const XBuf xout = make_xbuf(a.ex + (size_t)b * a.T * D, (size_t)a.T * D * sizeof(float));
  int32_t* prog = a.xprog + ((size_t)end * a.B + b) * kExMaxQ + qi;
  bool nan = false;
  int done = 0;
  for (int r0 = qi * NR; r0 < n; r0 += Q * NR) {
    u32x4 q[NR][CH];
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const int t = end ? L - 1 - (r0 + k) : r0 + k;
      const int soff = __builtin_amdgcn_readfirstlane(t * D * (XH ? 2 : 4));
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const int e = (c * kExNT + tid) * 4;
        if (r0 + k < n && e < D) {
          if constexpr (XH) {
            typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
            const u32x2_ h = __builtin_amdgcn_raw_buffer_load_b64(xin, e * 2, soff, 0);
            q[k][c].x = h.x; q[k][c].y = h.y;
          } else q[k][c] = __builtin_amdgcn_raw_buffer_load_b128(xin, e * 4, soff, 0);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const int t = end ? L - 1 - (r0 + k) : r0 + k;
      const int soff = __builtin_amdgcn_readfirstlane(t * D * 4);
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const int e = (c * kExNT + tid) * 4;
        if (r0 + k < n && e < D) {
          float x0, x1, x2, x3;
          if constexpr (XH) { half2_to_f32(q[k][c].x, bf16, x0, x1); half2_to_f32(q[k][c].y, bf16, x2, x3); }
          else { x0 = __uint_as_float(q[k][c].x); x1 = __uint_as_float(q[k][c].y); x2 = __uint_as_float(q[k][c].z); x3 = __uint_as_float(q[k][c].w); }
          nan = nan || __builtin_isunordered(x0, x1) || __builtin_isunordered(x2, x3);
          u32x4 o;
          o.x = __float_as_uint(clamp_exp(x0, kXExpClamp)); o.y = __float_as_uint(clamp_exp(x1, kXExpClamp));
          o.z = __float_as_uint(clamp_exp(x2, kXExpClamp)); o.w = __float_as_uint(clamp_exp(x3, kXExpClamp));
          __builtin_amdgcn_raw_buffer_store_b128(o, xout, e * 4, soff, kStoreDeviceScope);
        }
      }
    }
    if (nan) __hip_atomic_store(a.xnan + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);                       // this wave's stores are acknowledged
    __syncthreads();
    done++;
    if (tid == 0) __hip_atomic_store(prog, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace
// rounds of the rows exp'd ahead: rows per round and workgroups per end of a sequence (DenArgs::ex_nr, ex_q)
void den_exp_rows_shape(const DenArgs& a, int cus, int* nr, int* q) {
  const int ch = (a.D / 4 + kExNT - 1) / kExNT;
  *nr = ch <= 2 ? 8 : 4;
  const int per_end = cus / (2 * a.B);                  // as many workgroups as the chip has CUs
  *q = per_end < 1 ? 1 : (per_end > kExMaxQ ? kExMaxQ : per_end);
}
hipError_t launch_den_exp_rows(const DenArgs& a, hipStream_t st) {
  const dim3 grid(2 * a.B * a.ex_q), block(kExNT);
  const int ch = (a.D / 4 + kExNT - 1) / kExNT;
  if (a.x_half) {
    switch (ch) {
      case 1: hipLaunchKernelGGL((den_exp_rows_kernel<1, true>), grid, block, 0, st, a); break;
      case 2: hipLaunchKernelGGL((den_exp_rows_kernel<2, true>), grid, block, 0, st, a); break;
      case 3: hipLaunchKernelGGL((den_exp_rows_kernel<3, true>), grid, block, 0, st, a); break;
      case 4: hipLaunchKernelGGL((den_exp_rows_kernel<4, true>), grid, block, 0, st, a); break;
      case 5: hipLaunchKernelGGL((den_exp_rows_kernel<5, true>), grid, block, 0, st, a); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (ch) {
    case 1: hipLaunchKernelGGL(den_exp_rows_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(den_exp_rows_kernel<2>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(den_exp_rows_kernel<3>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(den_exp_rows_kernel<4>, grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL(den_exp_rows_kernel<5>, grid, block, 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
namespace {
// ---- time segments (DenArgs::tseg): every speculated row next to a segment against the TRUE row its neighbour stored.
// alpha segment k > 0 speculated row s_k - 1 (true: segment k-1's last row); beta segment k < S-1 speculated row e_k + 1 (true:
// segment k+1's last row).  Rows are in per-frame scales of their own: compared as distributions, max |p - q| / max p <= 4e-6 -
// a filter that agrees that well one frame outside a segment agrees better inside it (it contracts).  A miss sets redo[0] (the
// fallback recursion launch behind this kernel then runs) and counts into redo[1].
// (two fp32 recursions that started differently agree to 4e-7 ... 5e-7 at best - their own rounding, measured on C3 / C4 at
// burn-ins of 192 ... 256 frames -, a gradient inherits about the mismatch, and the parity bar is 1e-4)
constexpr float kSpliceTol = 4e-6f;
__global__ __launch_bounds__(256) void den_splice_check_kernel(const DenArgs a) {
  __shared__ float red[3][4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Lb = seq_len(a.lengths, b, a.T);
  if (den_tseg_off(a)) {                                   // cooling down (DenArgs::tstate): nothing was speculated - the uncut launch runs
    if (b == 0 && tid == 0) atomicAdd(a.redo, 1);
    return;
  }
  const int tburn = den_tburn(a);
  const int nseg = (a.tseg > 1 && Lb >= 2 * tburn) ? a.tseg : 1;
  int miss = 0;
  float worst = 0.f;
  for (int dir = 0; dir < 2; dir++)
    for (int k = dir == 0 ? 1 : 0; k < (dir == 0 ? nseg : nseg - 1); k++) {
      const int s = (int)(((long)k * Lb) / nseg), e = k + 1 == nseg ? Lb : (int)(((long)(k + 1) * Lb) / nseg);
      if (dir == 0 && s - tburn <= 0) continue;          // (the segment started at the true start: nothing speculated)
      if (dir == 1 && e + tburn >= Lb) continue;
      const float* truth = dir == 0 ? a.alpha_store + ((size_t)b * a.T + (s - 1)) * a.Hp : a.beta_store + ((size_t)b * (a.T + 1) + (e + 1)) * a.Hp;
      const float* spec = a.splice + ((size_t)(b * 2 + dir) * kMaxTimeSegs + k) * 2 * a.Hp;
      float sp = 0.f, sq = 0.f;
      for (int i = tid; i < a.H; i += 256) { sp += truth[i]; sq += spec[i]; }
      sp = wave_sum(sp); sq = wave_sum(sq);
      __syncthreads();
      if ((tid & 63) == 0) { red[0][tid >> 6] = sp; red[1][tid >> 6] = sq; }
      __syncthreads();
      sp = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]); sq = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
      const float ip = 1.f / sp, iq = 1.f / sq;
      float dmax = 0.f, pmax = 0.f;
      for (int i = tid; i < a.H; i += 256) { const float p = truth[i] * ip; dmax = fmaxf(dmax, fabsf(p - spec[i] * iq)); pmax = fmaxf(pmax, p); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64)); pmax = fmaxf(pmax, __shfl_xor(pmax, o, 64)); }
      __syncthreads();
      if ((tid & 63) == 0) { red[0][tid >> 6] = dmax; red[1][tid >> 6] = pmax; }
      __syncthreads();
      dmax = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])); pmax = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
      if (!(dmax <= kSpliceTol * pmax)) miss++;           // (also: NaN, a dead row)
      worst = fmaxf(worst, dmax / pmax);
    }
  if (tid == 0 && miss) { atomicAdd(a.redo, 1); atomicAdd(a.redo + 1, miss); }
  // the call's worst mismatch (a positive float orders like its bits): reported in totals[7], how much margin the burn-in has
  if (tid == 0 && worst > 0.f) atomicMax(reinterpret_cast<unsigned int*>(a.redo + 2), __float_as_uint(worst));
}
}  // namespace
hipError_t launch_den_splice_check(const DenArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(den_splice_check_kernel, dim3(a.B), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_den_finish(const DenArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(den_finish_kernel, dim3(a.B), dim3(kFinNT), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_den_gate(const int32_t* progress, int target, int32_t* bad, hipStream_t st) {
  hipLaunchKernelGGL(den_gate_kernel, dim3(1), dim3(64), 0, st, progress, target, bad);
  return hipGetLastError();
}

bool den_stream_eligible(const DenArgs& a, int gamma_max_groups, int resident_slot_rows) {
  return (a.lazy || a.pair) && gamma_stream_shape_ok(a, resident_slot_rows, gamma_max_groups);
}
const char* den_recursion_kernel_name(const DenArgs& a, int resident_slot_rows) {
  if (a.pair) return "den_recursion_pair_kernel";
  if (a.lazy && a.shape == kShapeDma && !a.sg && den_q_eligible(a, resident_slot_rows)) return "den_recursion_lazy_kernel<dma; one-word states>";
  if (a.lazy) return a.shape == kShapeSmall ? "den_recursion_lazy_kernel<small>" : (a.shape == kShapeDma ? (a.sg ? (a.xf ? "den_recursion_lazy_kernel<dma; one gather per arc; crossing>" : "den_recursion_lazy_kernel<dma; one gather per arc>") : "den_recursion_lazy_kernel<dma>") : "den_recursion_lazy_kernel");
  return "den_recursion_kernel";
}
const char* den_occupancy_kernel_name(const DenArgs& a, int gamma_max_groups, int resident_slot_rows) {
  return gamma2_eligible(a, (resident_slot_rows >> 20) & 127, gamma_max_groups) ? "den_gamma2_kernel" : "den_gamma_kernel";
}
int den_recursion_blocks(const DenArgs& a) { return a.pair ? 2 * ((a.B + 1) / 2) : 2 * a.B; }

bool den_uses_gamma2(const DenArgs& a, int gamma_max_groups, int resident_slot_rows) {
  return gamma2_eligible(a, (resident_slot_rows >> 20) & 127, gamma_max_groups);
}
// 2-byte network output and gradient (DenArgs::x_half): the two-frame kernel, or the one-frame kernel in its float4-chunk forms
bool den_occupancy_half_ok(const DenArgs& a, int gamma_max_groups, int resident_slot_rows) {
  if (gamma2_eligible(a, (resident_slot_rows >> 20) & 127, gamma_max_groups)) return true;
  return a.D % 4 == 0 && a.D <= 4 * 4 * kNT && pick_r(a, (resident_slot_rows >> 10) & 511, 2 * a.Hp) > 0;
}

hipError_t launch_den(const DenArgs& a, int gamma_max_groups, int resident_slot_rows, hipStream_t st,
                      const char** why) {
  const int Dp = (a.D + 3) & ~3;
  const bool db = a.D % 4 == 0 && a.D <= 4 * kNT;        // the <4, 1> instantiation: two 16 KiB nnet-output buffers
  const size_t lds_rec = sizeof(float) * (3 * (size_t)a.Hp + (db ? 2 * (size_t)(kXOff / 4) : (size_t)Dp) + 32);
  const size_t lds_gam = sizeof(float) * (2 * (size_t)a.Hp + 2 * (size_t)Dp + (size_t)gamma_max_groups * 64 + 16);
  if (lds_rec > 160 * 1024 || lds_gam + kStaticLds > 160 * 1024) {
    *why = "state vector + nnet-output row do not fit the 160 KiB LDS of one CU";
    return hipErrorInvalidValue;
  }
  // rows are addressed as 32-bit byte offsets into one sequence's slab (buffer loads / stores)
  if ((size_t)a.T * a.D * 4 >= (size_t)1 << 31 || ((size_t)a.T + 1) * a.Hp * 4 >= (size_t)1 << 31) {
    *why = "one sequence's nnet-output slab or state trajectory reaches 2 GiB";
    return hipErrorInvalidValue;
  }
  const int gx = den_compact_grid_x(a);
  const int D = a.D, r = resident_slot_rows;
  if (D % 4 == 0) {
    if (D <= 4 * 1 * kNT) return launch_r<4, 1>(a, r, lds_rec, lds_gam, gx, st, gamma_max_groups);
    if (D <= 4 * 2 * kNT) return launch_r<4, 2>(a, r, lds_rec, lds_gam, gx, st, gamma_max_groups);
    if (D <= 4 * 3 * kNT) return launch_r<4, 3>(a, r, lds_rec, lds_gam, gx, st, gamma_max_groups);
    if (D <= 4 * 4 * kNT) return launch_r<4, 4>(a, r, lds_rec, lds_gam, gx, st, gamma_max_groups);
  } else if (D <= 4 * kNT) {
    return launch_r<1, 4>(a, r, lds_rec, lds_gam, gx, st, gamma_max_groups);
  }
  return launch_r<1, 0>(a, r, lds_rec, lds_gam, gx, st, gamma_max_groups);
}

}  // namespace pychain_hip
