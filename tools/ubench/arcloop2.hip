// arcloop2.hip - whole-frame model of den_recursion_kernel on one CU (gfx950), second pass: the arc phase
// WITH what surrounds it in the real kernel (nnet-output row prefetch + exp + LDS store, group ends,
// HBM row stores, reductions, barriers), for three frame structures:
//
//   S_SHIPPED  raw sums to LDS at the group ends; wave sums; barrier; totals; normalise pass over the row
//              (LDS -> fma -> LDS + HBM); barrier                                  [round-1 kernel]
//   S_REGS     raw sums stay in the lane that produced them; after the barrier every lane normalises its
//              own values and writes them to LDS + HBM; barrier
//   S_LAZY     "lazy normalisation": the state vector is kept un-normalised as float2 {raw, coef*leaky}
//              (double-buffered, ONE ds_read_b64 per arc), two accumulators {sum w*raw, sum w*leaky} as one
//              packed fma, the previous frame's normaliser applied at the group ends, ONE barrier per frame
//
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench/arcloop2.hip -o tools/ubench/arcloop2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const float lds_cf;
typedef __attribute__((address_space(3))) const v2f lds_cv2;
typedef __attribute__((address_space(3))) char lds_ch;
typedef __amdgpu_buffer_rsrc_t XBuf;
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
__device__ __forceinline__ float ld1(uint32_t a) { return *(lds_cf*)(a); }
__device__ __forceinline__ v2f ld2(uint32_t a) { return *(lds_cv2*)(a); }
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_ch*)(p); }
#define WAIT_LGKM(n) __builtin_amdgcn_s_waitcnt(0xC07F | ((n) << 8))
#define DPP_ADD(v, ctrl) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true)))
__device__ __forceinline__ float row_sum(float v) {
  v = DPP_ADD(v, 0xB1); v = DPP_ADD(v, 0x4E); v = DPP_ADD(v, 0x141); v = DPP_ADD(v, 0x140);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row_sum(v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ XBuf make_xbuf(const float* p, size_t bytes) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float exp_bounded(float c) {
  const float kL2E = 1.44269502162933349609375f;
  const float t = c * kL2E;
  const float r = fmaf(c, kL2E, -t);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.693147182464599609375f, e);
}
__device__ __forceinline__ float clamp_exp(float v) { return exp_bounded(__builtin_amdgcn_fmed3f(v, -30.f, 30.f)); }
__device__ __forceinline__ float clamp_exp_cheap(float v) { return __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(v, -30.f, 30.f) * 1.44269502162933349609375f); }

enum { S_SHIPPED = 0, S_REGS = 1, S_LAZY = 2 };
// ablations (timing only)
enum { A_NOHBM = 1, A_NOLDSW = 2, A_NOX = 4, A_NORED = 8, A_NOXLOAD = 16,
       // options
       O_CHEAPEXP = 32,    // exp as v_exp_f32(c * log2 e), no compensation term
       O_XINTER = 64,      // nnet-output row as 4 dword loads per thread, element k exp'd + stored inside chunk NC-4+k
       O_LATESTORE = 128,  // lazy: group-end values kept in registers, HBM row stores issued after the arc phase
       O_ROWSUM4 = 256 };  // lazy: 4 row sums per wave before the barrier, the rest of the reduction after it
constexpr int kCh = 4;
constexpr int kD = 3456, kHp = 3008, kXOff = 16384;

// NG group ends per frame and wave, after chunks gend(0..NG-1)
template <int NC, int NG>
__host__ __device__ constexpr bool is_gend(int c) {
  for (int g = 0; g < NG; g++) if (c == (g + 1) * NC / NG - 1) return true;
  return false;
}
template <int NC, int NG>
__host__ __device__ constexpr int gidx(int c) {
  int n = 0;
  for (int g = 0; g < NG; g++) if ((g + 1) * NC / NG - 1 < c) n++;
  return n;
}

template <int NW, int R, int NG, int ST, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void frame_kernel(float* out, unsigned long long* cyc, const uint32_t* __restrict__ idx,
                                                         const float* __restrict__ xg, float* __restrict__ store, int iters, int T) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NW * 64, NC = R / kCh;
  constexpr int USZ = ST == S_LAZY ? 8 : 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // LDS: xr[2] (16 KiB each), U (lazy: [2][Hp] float2; else [Hp] float), raw [Hp], red [64]
  float* xr = reinterpret_cast<float*>(smem);
  float* U = xr + 2 * (kXOff / 4);
  constexpr int kUBuf = kHp * (USZ / 4);                 // floats per state buffer
  float* raw = U + 2 * kUBuf;
  float* red = raw + kHp;
  for (int i = tid; i < 2 * (kXOff / 4) + 2 * kUBuf + kHp + 128; i += NT) xr[i] = 1.0f / 3008.f;
  const uint32_t ub = lds_addr(U), vb = lds_addr(xr);
  uint32_t pk[R];
  float p[R];
#pragma unroll
  for (int s = 0; s < R; s++) {
    const uint32_t w = idx[(wave * R + s) * 64 + lane];
    pk[s] = (ub + (w & 0xfff) * USZ) | ((vb + ((w >> 12) & 0xfff) * 4) << 16);
    p[s] = 0.05f + 1e-4f * (float)(w >> 24);
  }
  // this lane's output positions (group g of this wave) and its coef*leaky there
  int pos[NG];
  float cl[NG];
#pragma unroll
  for (int g = 0; g < NG; g++) { pos[g] = ((wave * NG + g) * 64 + lane) % kHp; cl[g] = 1e-9f * (float)(lane + 1); }
  const XBuf xbuf = make_xbuf(xg, (size_t)T * kD * 4);
  const XBuf sbuf = make_xbuf(store, (size_t)T * kHp * 4);
  __syncthreads();
  float inv = 1.0f, keep = 0.f;
  double logsum = 0.0;
  unsigned long long t0 = 0;
  u32x4 xq = {0, 0, 0, 0};

#define FRAME(IT, PAR)                                                                                                    \
  do {                                                                                                                    \
    const int it = (IT);                                                                                                  \
    constexpr uint32_t VOFF = (PAR) ? kXOff : 0;                /* nnet-output buffer this frame gathers from */          \
    constexpr uint32_t UOFF = (ST == S_LAZY && (PAR)) ? kUBuf * 4 : 0;  /* lazy: state buffer this frame gathers from */  \
    constexpr uint32_t UNEXT = (ST == S_LAZY && !(PAR)) ? kUBuf : 0;    /* lazy: ... and the one it writes (floats) */    \
    const int trow = (it + 1) % T;                                                                                        \
    const int soff = __builtin_amdgcn_readfirstlane(trow * kD * 4);                                                       \
    if (!(ABL & A_NOXLOAD)) {                                                                                             \
      if (ABL & O_XINTER) {                                                                                               \
        xq.x = __builtin_amdgcn_raw_buffer_load_b32(xbuf, tid * 4, soff, 0);                                              \
        xq.y = __builtin_amdgcn_raw_buffer_load_b32(xbuf, (tid + NT) * 4, soff, 0);                                       \
        xq.z = __builtin_amdgcn_raw_buffer_load_b32(xbuf, (tid + 2 * NT) * 4, soff, 0);                                   \
        if (tid + 3 * NT < kD) xq.w = __builtin_amdgcn_raw_buffer_load_b32(xbuf, (tid + 3 * NT) * 4, soff, 0);           \
      } else if (tid * 4 < kD) xq = __builtin_amdgcn_raw_buffer_load_b128(xbuf, tid * 16, soff, 0);                       \
    }                                                                                                                     \
    float valreg[NG];                                                                                                     \
    const int row_off = __builtin_amdgcn_readfirstlane((it % T) * kHp * 4);                                               \
    float u1[2][kCh], v1[2][kCh];                                                                                         \
    v2f u2[2][kCh];                                                                                                       \
    v2f acc = {0.f, 0.f};                                                                                                 \
    float s0 = 0.f;                                                                                                       \
    float rawreg[NG];                                                                                                     \
    _Pragma("unroll") for (int k = 0; k < kCh; k++) {                                                                     \
      asm volatile("" : "+v"(pk[k]));                                                                                     \
      if (ST == S_LAZY) u2[0][k] = ld2((pk[k] & 0xffffu) + UOFF); else u1[0][k] = ld1(pk[k] & 0xffffu);                    \
      v1[0][k] = ld1((pk[k] >> 16) + VOFF);                                                                               \
    }                                                                                                                     \
    _Pragma("unroll") for (int c = 0; c < NC; c++) {                                                                      \
      const int cb = c & 1;                                                                                               \
      if (c + 1 < NC) {                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < kCh; k++) {                                                                 \
          const int s = (c + 1) * kCh + k;                                                                                \
          asm volatile("" : "+v"(pk[s]));                                                                                 \
          if (ST == S_LAZY) u2[cb ^ 1][k] = ld2((pk[s] & 0xffffu) + UOFF); else u1[cb ^ 1][k] = ld1(pk[s] & 0xffffu);      \
          v1[cb ^ 1][k] = ld1((pk[s] >> 16) + VOFF);                                                                      \
        }                                                                                                                 \
      }                                                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                                  \
      if (c + 1 < NC) WAIT_LGKM(2 * kCh); else WAIT_LGKM(0);                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                                  \
      _Pragma("unroll") for (int k = 0; k < kCh; k++) {                                                                   \
        if (ST == S_LAZY) {                                                                                               \
          const float w = p[c * kCh + k] * v1[cb][k];                                                                     \
          acc = __builtin_elementwise_fma(v2f{w, w}, u2[cb][k], acc);                                                     \
        } else {                                                                                                          \
          acc.x = fmaf(p[c * kCh + k] * u1[cb][k], v1[cb][k], acc.x);                                                     \
        }                                                                                                                 \
      }                                                                                                                   \
      if ((ABL & O_XINTER) && !(ABL & A_NOX) && c >= NC - 4) {                                                            \
        const int k = c - (NC - 4);                                                                                       \
        const float xv = __uint_as_float(k == 0 ? xq.x : k == 1 ? xq.y : k == 2 ? xq.z : xq.w);                            \
        const float e = (ABL & O_CHEAPEXP) ? clamp_exp_cheap(xv) : clamp_exp(xv);                                         \
        if (tid + k * NT < kD) xr[(kXOff - VOFF) / 4 + tid + k * NT] = e;                                                 \
      }                                                                                                                   \
      if (is_gend<NC, NG>(c)) {                                                                                           \
        const int g = gidx<NC, NG>(c);                                                                                    \
        if (ST == S_LAZY) {                                                                                               \
          const float val = fmaf(acc.x, inv, acc.y);                                                                      \
          if (!(ABL & A_NOLDSW)) U[UNEXT + 2 * pos[g]] = val;                                                             \
          if (ABL & O_LATESTORE) valreg[g] = val;                                                                        \
          else if (!(ABL & A_NOHBM)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), sbuf, pos[g] * 4, row_off, 16); \
          s0 += val;                                                                                                      \
        } else if (ST == S_REGS) {                                                                                        \
          rawreg[g] = acc.x; s0 += acc.x;                                                                                 \
        } else {                                                                                                          \
          raw[pos[g]] = acc.x; s0 += acc.x;                                                                               \
        }                                                                                                                 \
        acc = v2f{0.f, 0.f};                                                                                              \
      }                                                                                                                   \
    }                                                                                                                     \
    /* the next frame's nnet-output row: exp'd into the other buffer */                                                   \
    if (ABL & O_LATESTORE) {                                                                                              \
      _Pragma("unroll") for (int g = 0; g < NG; g++)                                                                      \
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(valreg[g]), sbuf, pos[g] * 4, row_off, 16);                 \
    }                                                                                                                     \
    if (!(ABL & (A_NOX | O_XINTER)) && tid * 4 < kD) {                                                                    \
      float4 q;                                                                                                           \
      if (ABL & O_CHEAPEXP) {                                                                                             \
        q.x = clamp_exp_cheap(__uint_as_float(xq.x)); q.y = clamp_exp_cheap(__uint_as_float(xq.y));                       \
        q.z = clamp_exp_cheap(__uint_as_float(xq.z)); q.w = clamp_exp_cheap(__uint_as_float(xq.w));                       \
      } else {                                                                                                            \
      q.x = clamp_exp(__uint_as_float(xq.x)); q.y = clamp_exp(__uint_as_float(xq.y));                                     \
      q.z = clamp_exp(__uint_as_float(xq.z)); q.w = clamp_exp(__uint_as_float(xq.w));                                     \
      }                                                                                                                   \
      *reinterpret_cast<float4*>(xr + (kXOff - VOFF) / 4 + tid * 4) = q;                                                  \
    }                                                                                                                     \
    if (ST == S_LAZY && (ABL & O_ROWSUM4)) {                                                                              \
      s0 = row_sum(s0);                                                                                                   \
      red[(PAR) * 64 + wave * 4 + (lane >> 4)] = s0;               /* 16 lanes write the same word */                     \
      __syncthreads();                                                                                                    \
      const float tot = wave_sum(red[(PAR) * 64 + lane]);                                                                 \
      inv = __builtin_amdgcn_rcpf(tot);                                                                                   \
      if (wave == 0) logsum += (double)(__builtin_amdgcn_logf(tot) * 0.693147182464599609375f);                           \
    } else if (ST == S_LAZY) {                                                                                            \
      if (!(ABL & A_NORED)) s0 = wave_sum(s0);                                                                            \
      if (lane == 0) red[(PAR) * 16 + wave] = s0;                                                                         \
      __syncthreads();                                                                                                    \
      const float tot = row_sum(red[(PAR) * 16 + (lane & 15)]);      /* used at the NEXT frame's group ends */            \
      inv = __builtin_amdgcn_rcpf(tot);                                                                                   \
      if (wave == 0) logsum += (double)(__builtin_amdgcn_logf(tot) * 0.693147182464599609375f);                           \
    } else {                                                                                                              \
      if (!(ABL & A_NORED)) s0 = wave_sum(s0);                                                                            \
      if (lane == 0) red[wave] = s0;                                                                                      \
      __syncthreads();                                                                                                    \
      const float tot = row_sum(red[lane & 15]);                                                                          \
      inv = __builtin_amdgcn_rcpf(tot);                                                                                   \
      logsum += (double)(__builtin_amdgcn_logf(tot) * 0.693147182464599609375f);                                          \
      if (ST == S_REGS) {                                                                                                 \
        _Pragma("unroll") for (int g = 0; g < NG; g++) {                                                                  \
          const float v = fmaf(rawreg[g], inv, cl[g]);                                                                    \
          U[pos[g]] = v;                                                                                                  \
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), sbuf, pos[g] * 4, row_off, 16);                       \
        }                                                                                                                 \
      } else {                                                                                                            \
        for (int i = tid * 4; i < kHp; i += NT * 4) {                                                                     \
          const float4 r = *reinterpret_cast<const float4*>(raw + i);                                                     \
          const float4 v = make_float4(fmaf(r.x, inv, cl[0]), fmaf(r.y, inv, cl[0]), fmaf(r.z, inv, cl[0]), fmaf(r.w, inv, cl[0])); \
          *reinterpret_cast<float4*>(U + i) = v;                                                                          \
          u32x4 q; q.x = __float_as_uint(v.x); q.y = __float_as_uint(v.y); q.z = __float_as_uint(v.z); q.w = __float_as_uint(v.w); \
          __builtin_amdgcn_raw_buffer_store_b128(q, sbuf, i * 4, row_off, 16);                                            \
        }                                                                                                                 \
      }                                                                                                                   \
      __syncthreads();                                                                                                    \
    }                                                                                                                     \
  } while (0)

  for (int it = -8; it < iters; it += 2) {
    if (it == 0) t0 = __builtin_readcyclecounter();
    FRAME(it, 0);
    FRAME(it + 1, 1);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * NT + tid] = keep + inv + U[tid] + (float)logsum;
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static std::vector<uint32_t> make_idx(int nw, int r) {
  std::vector<uint32_t> v((size_t)nw * r * 64);
  uint32_t rng = 12345;
  auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  for (int w = 0; w < nw; w++)
    for (int s = 0; s < r; s++) {
      const uint32_t rot0 = next() & 31, rot1 = next() & 31;
      for (int l = 0; l < 64; l++) {
        const uint32_t i0 = ((l + rot0) & 31) + 32 * (next() % 94), i1 = ((l + rot1) & 31) + 32 * (next() % 108);
        v[((size_t)w * r + s) * 64 + l] = i0 | (i1 << 12) | ((next() & 0xff) << 24);
      }
    }
  return v;
}

template <int NW, int R, int NG, int ST, int ABL = 0>
void run(const char* label, float* out, unsigned long long* cyc, uint32_t* idx_dev, const float* xg, float* store, int T) {
  auto h = make_idx(NW, R);
  hipMemcpy(idx_dev, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const size_t lds = 4 * (2 * (kXOff / 4) + 2 * kHp * (ST == S_LAZY ? 2 : 1) + kHp + 64);
  hipFuncSetAttribute(reinterpret_cast<const void*>(frame_kernel<NW, R, NG, ST, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int iters = 3000;
  hipLaunchKernelGGL((frame_kernel<NW, R, NG, ST, ABL>), dim3(1), dim3(NW * 64), lds, 0, out, cyc, idx_dev, xg, store, iters, T);
  hipError_t e = hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per = (double)c / iters;
  printf("%-46s waves %2d rows/wave %2d groups/wave %d: %7.0f cycles/frame  %5.2f per slot-row  %s\n", label, NW, R, NG, per,
         per / (NW * R), e == hipSuccess ? "" : hipGetErrorString(e));
  fflush(stdout);
}

int main() {
  float* out; unsigned long long* cyc; uint32_t* idx; float *xg, *store;
  const int T = 1500;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8); hipMalloc(&idx, 4 << 20);
  hipMalloc(&xg, (size_t)T * kD * 4); hipMalloc(&store, (size_t)T * kHp * 4);
  hipMemset(xg, 0, (size_t)T * kD * 4);
  // waves per workgroup at equal total slot-rows (640): what the per-wave serial chains of a frame cost
  run<16, 40, 3, S_SHIPPED>("shipped frame", out, cyc, idx, xg, store, T);
  run<12, 52, 4, S_SHIPPED>("shipped frame", out, cyc, idx, xg, store, T);
  run<8, 80, 6, S_SHIPPED>("shipped frame", out, cyc, idx, xg, store, T);
  run<16, 40, 3, S_LAZY, O_LATESTORE | O_ROWSUM4>("lazy, late store, rowsum4", out, cyc, idx, xg, store, T);
  run<12, 52, 4, S_LAZY, O_LATESTORE | O_ROWSUM4>("lazy, late store, rowsum4", out, cyc, idx, xg, store, T);
  run<8, 80, 6, S_LAZY, O_LATESTORE | O_ROWSUM4>("lazy, late store, rowsum4", out, cyc, idx, xg, store, T);
  run<8, 72, 6, S_LAZY, O_LATESTORE | O_ROWSUM4>("lazy, late store, rowsum4", out, cyc, idx, xg, store, T);
  run<4, 160, 12, S_LAZY, O_LATESTORE | O_ROWSUM4>("lazy, late store, rowsum4", out, cyc, idx, xg, store, T);
  run<8, 80, 6, S_LAZY, O_LATESTORE | O_ROWSUM4 | A_NOX | A_NOXLOAD>("lazy, no x at all", out, cyc, idx, xg, store, T);
  run<16, 40, 3, S_LAZY, O_LATESTORE | O_ROWSUM4 | A_NOX | A_NOXLOAD>("lazy, no x at all", out, cyc, idx, xg, store, T);
  return 0;
}
