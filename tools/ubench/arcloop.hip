// arcloop.hip - frame-step model of den_recursion_kernel on one CU (gfx950): what a frame costs as a
// function of waves per workgroup, registers per arc, packed arithmetic, gather width and the number of
// barrier-separated phases.  One persistent workgroup per CU, `iters` frames, wave 0 reports
// cycles per frame (all waves meet at a barrier every frame, so that is the CU's time).
//
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench/arcloop.hip -o tools/ubench/arcloop
//
// Addresses are random but bank-conflict free within each 32-lane half (what the plan compiler
// achieves to within 7 %): index = ((lane + rot_row) & 31) + 32 * random.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const float lds_cf;
typedef __attribute__((address_space(3))) const v2f lds_cv2;
typedef __attribute__((address_space(3))) char lds_ch;
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

// MODE bits
enum {
  M_REG3 = 1,     // separate address registers (3 VGPRs per arc) instead of 16:16 packed (2 VGPRs)
  M_PK = 2,       // packed fp32 arithmetic over two consecutive slot-rows
  M_LAZY = 4,     // lazy normalisation: U is float2 {raw, leaky} (ds_read_b64), two accumulators, ONE barrier per frame
  M_SEQ2 = 8,     // two sequences per workgroup: U and V are float2 {seq A, seq B}
  M_NOLDS = 16,   // operands from registers (VALU floor)
  M_NOVALU = 32,  // gathers only (LDS floor): results or-ed together
  M_TAIL = 64,    // with the serial tail of the shipped kernel: wave sums, barrier, totals, normalise pass, barrier
};

constexpr int kCh = 4;

template <int NW, int R, int MODE>
__global__ __launch_bounds__(NW * 64) void arc_kernel(float* out, unsigned long long* cyc, const uint32_t* __restrict__ idx,
                                                       int iters, int Hp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool REG3 = MODE & M_REG3, PK = MODE & M_PK, LAZY = MODE & M_LAZY, SEQ2 = MODE & M_SEQ2;
  constexpr bool NOLDS = MODE & M_NOLDS, NOVALU = MODE & M_NOVALU, TAIL = MODE & M_TAIL;
  constexpr int USZ = (LAZY || SEQ2) ? 8 : 4;         // bytes per U element
  constexpr int VSZ = SEQ2 ? 8 : 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* U = reinterpret_cast<float*>(smem);                       // [2][4096] elements of USZ bytes (double-buffered for LAZY)
  float* V = U + 2 * 4096 * (USZ / 4);                             // [4096] elements of VSZ
  float* raw = V + 4096 * (VSZ / 4);                               // [4096 * 2]
  float* red = raw + 8192;                                         // [64]
  for (int i = tid; i < 2 * 4096 * (USZ / 4) + 4096 * (VSZ / 4) + 8192 + 64; i += NW * 64) U[i] = 1.0f / 4096.f;
  const uint32_t ub = lds_addr(U), vb = lds_addr(V);
  uint32_t a0[R], a1[REG3 ? R : 1], pk[REG3 ? 1 : R];
  float p[R];
#pragma unroll
  for (int s = 0; s < R; s++) {
    const uint32_t w = idx[(wave * R + s) * 64 + lane];
    const uint32_t i0 = w & 0xfff, i1 = (w >> 12) & 0xfff;
    const uint32_t x0 = ub + i0 * USZ, x1 = vb + i1 * VSZ;
    if (REG3) { a0[s] = x0; a1[s] = x1; asm volatile("" : "+v"(a0[s]), "+v"(a1[s])); }
    else { pk[s] = x0 | (x1 << 16); a0[s] = 0; }
    p[s] = 0.5f + 1e-3f * (float)(w >> 24);
  }
  __syncthreads();
  float inv = 1.0f, keep = 0.f;
  unsigned long long t0 = 0;
  for (int it = -8; it < iters; it++) {
    if (it == 0) t0 = __builtin_readcyclecounter();
    const uint32_t uoff = LAZY ? (uint32_t)(it & 1) * 4096u * USZ : 0u;   // lazy form: double-buffered state vector
    // ---- arc phase: software pipeline over chunks of kCh slot-rows
    float u1[2][kCh], v1[2][kCh];
    v2f u2[2][kCh], v2[2][kCh];
    auto gather = [&](int c, int buf) {
#pragma unroll
      for (int k = 0; k < kCh; k++) {
        const int s = c * kCh + k;
        uint32_t x0, x1;
        if (REG3) { x0 = a0[s]; x1 = a1[s]; }
        else { asm volatile("" : "+v"(pk[s])); x0 = pk[s] & 0xffffu; x1 = pk[s] >> 16; }
        if (NOLDS) {
          u1[buf][k] = __uint_as_float(x0); v1[buf][k] = __uint_as_float(x1);
          u2[buf][k] = v2f{__uint_as_float(x0), __uint_as_float(x1)}; v2[buf][k] = u2[buf][k];
        } else {
          if (LAZY || SEQ2) u2[buf][k] = ld2(x0 + uoff); else u1[buf][k] = ld1(x0);
          if (SEQ2) v2[buf][k] = ld2(x1); else v1[buf][k] = ld1(x1);
        }
      }
    };
    float acc = 0.f, accb = 0.f;          // (accb: second accumulator of the lazy form / odd rows of the packed form)
    v2f acc2 = {0.f, 0.f}, acc2b = {0.f, 0.f};
    gather(0, 0);
    constexpr int NC = R / kCh;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const int cb = c & 1;
      if (c + 1 < NC) gather(c + 1, cb ^ 1);
      if (!NOLDS) {
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NC) WAIT_LGKM(2 * kCh); else WAIT_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (NOVALU) {
#pragma unroll
        for (int k = 0; k < kCh; k++) {
          if (LAZY || SEQ2) { acc = __uint_as_float(__float_as_uint(acc) | __float_as_uint(u2[cb][k].x)); accb = __uint_as_float(__float_as_uint(accb) | __float_as_uint(u2[cb][k].y)); }
          else acc = __uint_as_float(__float_as_uint(acc) | __float_as_uint(u1[cb][k]));
          if (SEQ2) { acc = __uint_as_float(__float_as_uint(acc) | __float_as_uint(v2[cb][k].x)); accb = __uint_as_float(__float_as_uint(accb) | __float_as_uint(v2[cb][k].y)); }
          else accb = __uint_as_float(__float_as_uint(accb) | __float_as_uint(v1[cb][k]));
        }
        continue;
      }
      if constexpr (SEQ2) {
        // two sequences: {uA,uB} * p * {vA,vB}
#pragma unroll
        for (int k = 0; k < kCh; k++) {
          const float pp = p[c * kCh + k];
          if constexpr (LAZY) {       // beta-like lazy: w = p*v; acc1 += w*u; acc2 += w
            const v2f w = v2f{pp, pp} * v2[cb][k];
            acc2 = __builtin_elementwise_fma(w, u2[cb][k], acc2);
            acc2b = acc2b + w;
          } else {
            const v2f m = v2f{u2[cb][k].x * pp, u2[cb][k].y * pp};
            acc2 = __builtin_elementwise_fma(m, v2[cb][k], acc2);
          }
        }
      } else if constexpr (LAZY) {
        // w = p * v; acc += w * raw; accb += w * leaky
        if constexpr (PK) {
#pragma unroll
          for (int k = 0; k < kCh; k += 2) {
            const v2f w = v2f{p[c * kCh + k], p[c * kCh + k + 1]} * v2f{v1[cb][k], v1[cb][k + 1]};
            acc2 = __builtin_elementwise_fma(w, v2f{u2[cb][k].x, u2[cb][k + 1].x}, acc2);
            acc2b = __builtin_elementwise_fma(w, v2f{u2[cb][k].y, u2[cb][k + 1].y}, acc2b);
          }
        } else {
#pragma unroll
          for (int k = 0; k < kCh; k++) {
            const float w = p[c * kCh + k] * v1[cb][k];
            acc = fmaf(w, u2[cb][k].x, acc);
            accb = fmaf(w, u2[cb][k].y, accb);
          }
        }
      } else if constexpr (PK) {
#pragma unroll
        for (int k = 0; k < kCh; k += 2) {
          const v2f m = v2f{p[c * kCh + k], p[c * kCh + k + 1]} * v2f{u1[cb][k], u1[cb][k + 1]};
          acc2 = __builtin_elementwise_fma(m, v2f{v1[cb][k], v1[cb][k + 1]}, acc2);
        }
      } else {
#pragma unroll
        for (int k = 0; k < kCh; k++) acc = fmaf(p[c * kCh + k] * u1[cb][k], v1[cb][k], acc);
      }
    }
    if (PK || SEQ2) { acc += acc2.x + acc2.y; accb += acc2b.x + acc2b.y; }
    // ---- frame end
    if constexpr (LAZY) {
      // group end of the lazy form: value = acc * inv(previous frame) + accb, straight into the OTHER state buffer
      const float val = fmaf(acc, inv, accb * 1e-5f);
      const uint32_t wpos = ((uint32_t)((it + 1) & 1) * 4096u + (uint32_t)(wave * 64 + lane)) * (USZ / 4);
      U[wpos] = val * 1e-3f + 1.0f / 4096.f;
      float s0 = wave_sum(val);
      if (lane == 0) red[(it & 1) * 16 + wave] = s0;
      __syncthreads();
      const float tot = row_sum(red[(it & 1) * 16 + (lane & 15)]);     // needed at the NEXT frame's group ends only
      inv = __builtin_amdgcn_rcpf(tot + 1.0f);
    } else if constexpr (TAIL) {
      raw[wave * 64 + lane] = acc;
      float s0 = wave_sum(acc);
      if (lane == 0) red[wave] = s0;
      __syncthreads();
      const float tot = row_sum(red[lane & 15]);
      inv = __builtin_amdgcn_rcpf(tot + 1.0f);
      for (int i = tid * 4; i < Hp; i += NW * 64 * 4) {     // normalise pass
        const float4 r = *reinterpret_cast<const float4*>(raw + i);
        *reinterpret_cast<float4*>(U + i * (USZ / 4)) = make_float4(r.x * inv * 1e-3f + 2.4e-4f, r.y * inv * 1e-3f + 2.4e-4f, r.z * inv * 1e-3f + 2.4e-4f, r.w * inv * 1e-3f + 2.4e-4f);
      }
      __syncthreads();
    } else {
      keep += acc + accb;
      __syncthreads();
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * NW * 64 + tid] = keep + inv + U[tid];
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static std::vector<uint32_t> make_idx(int nw, int r, bool conflict_free) {
  std::vector<uint32_t> v((size_t)nw * r * 64);
  uint32_t rng = 12345;
  auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  for (int w = 0; w < nw; w++)
    for (int s = 0; s < r; s++) {
      const uint32_t rot0 = next() & 31, rot1 = next() & 31;
      for (int l = 0; l < 64; l++) {
        uint32_t i0, i1;
        if (conflict_free) { i0 = ((l + rot0) & 31) + 32 * (next() % 94); i1 = ((l + rot1) & 31) + 32 * (next() % 108); }
        else { i0 = next() % 3008; i1 = next() % 3456; }
        v[((size_t)w * r + s) * 64 + l] = i0 | (i1 << 12) | ((next() & 0xff) << 24);
      }
    }
  return v;
}

template <int NW, int R, int MODE>
void run(const char* label, float* out, unsigned long long* cyc, uint32_t* idx_dev, bool conflict_free = true) {
  auto h = make_idx(NW, R, conflict_free);
  hipMemcpy(idx_dev, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  constexpr bool LAZY = MODE & M_LAZY, SEQ2 = MODE & M_SEQ2;
  const size_t lds = 4 * (2 * 4096 * ((LAZY || SEQ2) ? 2 : 1) + 4096 * (SEQ2 ? 2 : 1) + 8192 + 64);
  hipFuncSetAttribute(reinterpret_cast<const void*>(arc_kernel<NW, R, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int iters = 2000;
  hipLaunchKernelGGL((arc_kernel<NW, R, MODE>), dim3(1), dim3(NW * 64), lds, 0, out, cyc, idx_dev, iters, 3008);
  hipError_t e = hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per = (double)c / iters;
  printf("%-58s waves %2d rows/wave %2d (%3d slot-rows)%s: %7.0f cycles/frame  %5.2f per slot-row  %s\n", label, NW, R, NW * R,
         conflict_free ? "" : " RANDOM-BANKS", per, per / (NW * R), e == hipSuccess ? "" : hipGetErrorString(e));
  fflush(stdout);
}

int main() {
  float* out; unsigned long long* cyc; uint32_t* idx;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8); hipMalloc(&idx, 4 << 20);
  // floors
  run<16, 40, M_NOVALU>("gathers only, b32+b32", out, cyc, idx);
  run<16, 40, M_NOVALU | M_LAZY>("gathers only, b64+b32", out, cyc, idx);
  run<16, 40, M_NOVALU | M_SEQ2>("gathers only, b64+b64", out, cyc, idx);
  run<8, 72, M_NOVALU | M_REG3>("gathers only, b32+b32, 3-reg", out, cyc, idx);
  run<12, 48, M_NOVALU | M_REG3>("gathers only, b32+b32, 3-reg", out, cyc, idx);
  run<16, 40, M_NOLDS>("VALU only, packed addr (4 VALU/arc)", out, cyc, idx);
  run<8, 72, M_NOLDS | M_REG3>("VALU only, 3-reg (2 VALU/arc)", out, cyc, idx);
  run<8, 72, M_NOLDS | M_REG3 | M_PK>("VALU only, 3-reg packed math (1 VALU/arc)", out, cyc, idx);
  // shipped structure
  run<16, 40, 0>("packed addr, mul+fma [shipped arc loop]", out, cyc, idx);
  run<16, 40, 0>("packed addr, mul+fma [shipped arc loop]", out, cyc, idx, false);
  run<16, 32, 0>("packed addr, mul+fma", out, cyc, idx);
  run<16, 36, 0>("packed addr, mul+fma", out, cyc, idx);
  run<16, 40, M_TAIL>("packed addr, mul+fma + serial tail [shipped frame]", out, cyc, idx);
  run<16, 40, M_PK>("packed addr, packed math", out, cyc, idx);
  run<16, 32, M_PK>("packed addr, packed math", out, cyc, idx);
  // 3 registers per arc
  run<8, 64, M_REG3>("3-reg, mul+fma", out, cyc, idx);
  run<8, 72, M_REG3>("3-reg, mul+fma", out, cyc, idx);
  run<8, 72, M_REG3 | M_TAIL>("3-reg, mul+fma + serial tail", out, cyc, idx);
  run<8, 64, M_REG3 | M_PK>("3-reg, packed math", out, cyc, idx);
  run<8, 72, M_REG3 | M_PK>("3-reg, packed math", out, cyc, idx);
  run<12, 44, M_REG3>("3-reg, mul+fma", out, cyc, idx);
  run<12, 48, M_REG3>("3-reg, mul+fma", out, cyc, idx);
  run<12, 44, M_REG3 | M_PK>("3-reg, packed math", out, cyc, idx);
  run<12, 48, M_REG3 | M_PK>("3-reg, packed math", out, cyc, idx);
  // lazy normalisation (one barrier per frame, no normalise pass)
  run<16, 40, M_LAZY>("lazy, packed addr (5 VALU/arc)", out, cyc, idx);
  run<16, 32, M_LAZY>("lazy, packed addr (5 VALU/arc)", out, cyc, idx);
  run<16, 40, M_LAZY | M_PK>("lazy, packed addr, packed math", out, cyc, idx);
  run<8, 64, M_LAZY | M_REG3>("lazy, 3-reg (3 VALU/arc)", out, cyc, idx);
  run<8, 72, M_LAZY | M_REG3>("lazy, 3-reg (3 VALU/arc)", out, cyc, idx);
  run<8, 64, M_LAZY | M_REG3 | M_PK>("lazy, 3-reg, packed math (1.5 VALU/arc)", out, cyc, idx);
  run<8, 72, M_LAZY | M_REG3 | M_PK>("lazy, 3-reg, packed math (1.5 VALU/arc)", out, cyc, idx);
  run<12, 40, M_LAZY | M_REG3>("lazy, 3-reg (3 VALU/arc)", out, cyc, idx);
  run<12, 44, M_LAZY | M_REG3 | M_PK>("lazy, 3-reg, packed math", out, cyc, idx);
  // two sequences per workgroup
  run<16, 40, M_SEQ2>("2 sequences, packed addr", out, cyc, idx);
  run<16, 32, M_SEQ2>("2 sequences, packed addr", out, cyc, idx);
  run<16, 40, M_SEQ2 | M_TAIL>("2 sequences, packed addr + serial tail (1 row)", out, cyc, idx);
  run<8, 72, M_SEQ2 | M_REG3>("2 sequences, 3-reg", out, cyc, idx);
  run<8, 64, M_SEQ2 | M_REG3>("2 sequences, 3-reg", out, cyc, idx);
  run<8, 56, M_SEQ2 | M_REG3 | M_LAZY>("2 sequences, 3-reg, lazy (beta form)", out, cyc, idx);
  run<16, 28, M_SEQ2 | M_LAZY>("2 sequences, packed addr, lazy (beta form)", out, cyc, idx);
  return 0;
}
