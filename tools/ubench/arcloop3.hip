// arcloop3.hip - frame model of a recursion workgroup with SPECIALISED waves (gfx950): NA waves walk arcs
// and nothing else; 16 - NA service waves stage the next frame's nnet-output row (HBM -> registers one
// frame ahead -> exp -> LDS), stream the previous frame's completed row to HBM (LDS -> fma -> HBM) and
// never touch an arc.  Lazy normalisation (one barrier per frame; an arc wave reduces the previous frame's
// totals once per frame, at its first group end).  Compare with arcloop2's "lazy" (every wave does
// everything): what do the per-wave serial chains at the frame end cost?
//
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench/arcloop3.hip -o tools/ubench/arcloop3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
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

constexpr int kCh = 4;
constexpr int kD = 3456, kHp = 3008, kXOff = 16384;
constexpr int kUBuf = kHp * 2;      // floats per state buffer (float2 elements)

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

// NA arc waves of R slot-rows each, NG group ends per arc wave and frame; SVC: 0 = the service waves idle
// (what the arc waves alone cost), 1 = nnet-output rows, 2 = + row stores
template <int NA, int R, int NG, int SVC>
__global__ __launch_bounds__(1024) void frame_kernel(float* out, unsigned long long* cyc, const uint32_t* __restrict__ idx,
                                                      const float* __restrict__ xg, float* __restrict__ store, int iters, int T) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NC = R / kCh, NS = 16 - NA, NST = NS * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* xr = reinterpret_cast<float*>(smem);          // [2][4096]
  float* U = xr + 2 * (kXOff / 4);                     // [2][kHp] float2
  float* red = U + 2 * kUBuf;                          // [2][64]
  for (int i = tid; i < 2 * (kXOff / 4) + 2 * kUBuf + 128; i += 1024) xr[i] = i >= 2 * (kXOff / 4) + 2 * kUBuf ? 0.01f : 1.0f / 3008.f;
  const uint32_t ub = lds_addr(U), vb = lds_addr(xr);
  const XBuf xbuf = make_xbuf(xg, (size_t)T * kD * 4);
  const XBuf sbuf = make_xbuf(store, (size_t)T * kHp * 4);
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  float keep = 0.f;
  if (wave < NA) {
    // ------------------------------------------------------------ arc waves
    uint32_t pk[R];
    float p[R];
#pragma unroll
    for (int s = 0; s < R; s++) {
      const uint32_t w = idx[(wave * R + s) * 64 + lane];
      pk[s] = (ub + (w & 0xfff) * 8) | ((vb + ((w >> 12) & 0xfff) * 4) << 16);
      p[s] = 0.05f + 1e-4f * (float)(w >> 24);
    }
    int pos[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) pos[g] = ((wave * NG + g) * 64 + lane) % kHp;
#define ARC_FRAME(IT, PAR)                                                                                  \
    do {                                                                                                    \
      constexpr uint32_t VOFF = (PAR) ? kXOff : 0;                                                          \
      constexpr uint32_t UOFF = (PAR) ? kUBuf * 4 : 0;                                                      \
      constexpr uint32_t UNEXT = (PAR) ? 0 : kUBuf;                                                         \
      float u_[1];                                                                                          \
      (void)u_;                                                                                             \
      float v1[2][kCh];                                                                                     \
      v2f u2[2][kCh];                                                                                       \
      v2f acc = {0.f, 0.f};                                                                                 \
      float s0 = 0.f, inv = 0.f;                                                                            \
      _Pragma("unroll") for (int k = 0; k < kCh; k++) {                                                     \
        asm volatile("" : "+v"(pk[k]));                                                                     \
        u2[0][k] = ld2((pk[k] & 0xffffu) + UOFF);                                                           \
        v1[0][k] = ld1((pk[k] >> 16) + VOFF);                                                               \
      }                                                                                                     \
      _Pragma("unroll") for (int c = 0; c < NC; c++) {                                                      \
        const int cb = c & 1;                                                                               \
        if (c + 1 < NC) {                                                                                   \
          _Pragma("unroll") for (int k = 0; k < kCh; k++) {                                                 \
            const int s = (c + 1) * kCh + k;                                                                \
            asm volatile("" : "+v"(pk[s]));                                                                 \
            u2[cb ^ 1][k] = ld2((pk[s] & 0xffffu) + UOFF);                                                  \
            v1[cb ^ 1][k] = ld1((pk[s] >> 16) + VOFF);                                                      \
          }                                                                                                 \
        }                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (c + 1 < NC) WAIT_LGKM(2 * kCh); else WAIT_LGKM(0);                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        _Pragma("unroll") for (int k = 0; k < kCh; k++) {                                                   \
          const float w = p[c * kCh + k] * v1[cb][k];                                                       \
          acc = __builtin_elementwise_fma(v2f{w, w}, u2[cb][k], acc);                                       \
        }                                                                                                   \
        if (is_gend<NC, NG>(c)) {                                                                           \
          const int g = gidx<NC, NG>(c);                                                                    \
          if (g == 0) {                              /* once per frame: the previous frame's total */        \
            const float tot = wave_sum(red[((PAR) ^ 1) * 64 + lane]);                                       \
            inv = __builtin_amdgcn_rcpf(tot);                                                               \
          }                                                                                                 \
          const float val = fmaf(acc.x, inv, acc.y);                                                        \
          U[UNEXT + 2 * pos[g]] = val * 1e-3f + 1.0f / 3008.f;                                              \
          s0 += val;                                                                                        \
          acc = v2f{0.f, 0.f};                                                                              \
        }                                                                                                   \
      }                                                                                                     \
      s0 = row_sum(s0);                                                                                     \
      red[(PAR) * 64 + wave * 4 + (lane >> 4)] = s0 * 1e-3f + 0.01f;                                        \
      keep += inv;                                                                                          \
      __syncthreads();                                                                                      \
    } while (0)
    for (int it = -8; it < iters; it += 2) {
      if (it == 0) t0 = __builtin_readcyclecounter();
      ARC_FRAME(it, 0);
      ARC_FRAME(it + 1, 1);
    }
    t1 = __builtin_readcyclecounter();
  } else {
    // ------------------------------------------------------------ service waves
    const int st = tid - NA * 64;                      // 0 .. NST-1
    constexpr int XF4 = (kD / 4 + NST - 1) / NST;      // float4 of a nnet-output row per service thread
    constexpr int RF4 = (kHp / 2 + NST - 1) / NST;     // ds_read_b128 (two float2 states) of a state row per service thread
    u32x4 xa[XF4], xb[XF4];
#pragma unroll
    for (int k = 0; k < XF4; k++) { xa[k] = u32x4{0, 0, 0, 0}; xb[k] = xa[k]; }
    // SRC: registers holding the row staged this frame; DST: registers the load of the row after it goes to
#define SVC_FRAME(IT, PAR, SRC, DST)                                                                        \
    do {                                                                                                    \
      const int it = (IT);                                                                                  \
      if (SVC >= 1) {                                                                                       \
        const int soff = __builtin_amdgcn_readfirstlane(((it + 2) % T) * kD * 4);                           \
        _Pragma("unroll") for (int k = 0; k < XF4; k++) {                                                   \
          const int e = (k * NST + st) * 4;                                                                 \
          DST[k] = __builtin_amdgcn_raw_buffer_load_b128(xbuf, min(e, kD - 4) * 4, soff, 0);                \
        }                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < XF4; k++) {                                                   \
          const int e = (k * NST + st) * 4;                                                                 \
          if (e < kD) {                                                                                     \
            float4 q;                                                                                       \
            q.x = clamp_exp(__uint_as_float(SRC[k].x)); q.y = clamp_exp(__uint_as_float(SRC[k].y));         \
            q.z = clamp_exp(__uint_as_float(SRC[k].z)); q.w = clamp_exp(__uint_as_float(SRC[k].w));         \
            *reinterpret_cast<float4*>(xr + ((PAR) ? 0 : kXOff / 4) + e) = q;                               \
          }                                                                                                 \
        }                                                                                                   \
      }                                                                                                     \
      if (SVC >= 2) {                                                                                       \
        /* the vector this frame gathers from = the previous frame's finished row: complete it, stream it out */ \
        const float tot = wave_sum(red[((PAR) ^ 1) * 64 + lane]);                                           \
        const int row_off = __builtin_amdgcn_readfirstlane((it % T) * kHp * 4);                             \
        _Pragma("unroll") for (int k = 0; k < RF4; k += 2) {                                                \
          const int e = (k * NST + st) * 2;              /* first of two states */                          \
          const int e2 = ((k + 1) * NST + st) * 2;                                                          \
          if (e2 + 1 < kHp) {                                                                               \
            const float4 a0 = *reinterpret_cast<const float4*>(U + ((PAR) ? kUBuf : 0) + 2 * e);           \
            const float4 a1 = *reinterpret_cast<const float4*>(U + ((PAR) ? kUBuf : 0) + 2 * e2);          \
            u32x4 q;                                                                                        \
            q.x = __float_as_uint(fmaf(tot, a0.y, a0.x)); q.y = __float_as_uint(fmaf(tot, a0.w, a0.z));     \
            q.z = __float_as_uint(fmaf(tot, a1.y, a1.x)); q.w = __float_as_uint(fmaf(tot, a1.w, a1.z));     \
            __builtin_amdgcn_raw_buffer_store_b128(q, sbuf, (k / 2 * NST + st) * 16, row_off, 16);          \
          }                                                                                                 \
        }                                                                                                   \
      }                                                                                                     \
      __syncthreads();                                                                                      \
    } while (0)
    for (int it = -8; it < iters; it += 2) {
      SVC_FRAME(it, 0, xa, xb);
      SVC_FRAME(it + 1, 1, xb, xa);
    }
  }
  out[blockIdx.x * 1024 + tid] = keep + U[tid];
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

template <int NA, int R, int NG, int SVC>
void run(const char* label, float* out, unsigned long long* cyc, uint32_t* idx_dev, const float* xg, float* store, int T) {
  auto h = make_idx(NA, R);
  hipMemcpy(idx_dev, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const size_t lds = 4 * (2 * (kXOff / 4) + 2 * kUBuf + 128);
  hipFuncSetAttribute(reinterpret_cast<const void*>(frame_kernel<NA, R, NG, SVC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int iters = 3000;
  hipLaunchKernelGGL((frame_kernel<NA, R, NG, SVC>), dim3(1), dim3(1024), lds, 0, out, cyc, idx_dev, xg, store, iters, T);
  hipError_t e = hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per = (double)c / iters;
  printf("%-40s arc waves %2d x %2d rows (%3d slot-rows), %d group ends, service level %d: %7.0f cycles/frame  %5.2f per slot-row  %s\n",
         label, NA, R, NA * R, NG, SVC, per, per / (NA * R), e == hipSuccess ? "" : hipGetErrorString(e));
  fflush(stdout);
}

int main() {
  float* out; unsigned long long* cyc; uint32_t* idx; float *xg, *store;
  const int T = 1500;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8); hipMalloc(&idx, 4 << 20);
  hipMalloc(&xg, (size_t)T * kD * 4); hipMalloc(&store, (size_t)T * kHp * 4);
  hipMemset(xg, 0, (size_t)T * kD * 4);
  run<14, 44, 3, 0>("arc waves alone", out, cyc, idx, xg, store, T);
  run<14, 44, 3, 1>("+ nnet-output rows by service waves", out, cyc, idx, xg, store, T);
  run<14, 44, 3, 2>("+ row stores by service waves", out, cyc, idx, xg, store, T);
  run<14, 44, 4, 2>("+ row stores by service waves", out, cyc, idx, xg, store, T);
  run<14, 40, 3, 2>("+ row stores by service waves", out, cyc, idx, xg, store, T);
  run<12, 52, 4, 2>("four service waves", out, cyc, idx, xg, store, T);
  run<15, 40, 3, 2>("one service wave", out, cyc, idx, xg, store, T);
  run<15, 44, 3, 2>("one service wave", out, cyc, idx, xg, store, T);
  return 0;
}
