// Micro-benchmarks behind DESIGN.md §4: issue cost of wave64 VALU ops and of LDS gathers on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int ILP>
__global__ void valu_kernel(float* out, unsigned long long* cyc, int iters) {
  float a[ILP];
  for (int i = 0; i < ILP; i++) a[i] = threadIdx.x * 0.001f + i;
  const float b = 1.0001f, c = 0.5f;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) a[i] = fmaf(a[i], b, c);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < ILP; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// each lane gathers from LDS with a precomputed address pattern; NREAD independent reads per iteration
template <int NREAD>
__global__ void lds_kernel(float* out, unsigned long long* cyc, const uint32_t* addr, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  uint32_t off[NREAD];
  for (int i = 0; i < NREAD; i++) off[i] = addr[i * blockDim.x + threadIdx.x];
  __syncthreads();
  float acc = 0.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    float v[NREAD];
#pragma unroll
    for (int i = 0; i < NREAD; i++) { asm volatile("" : "+v"(off[i])); v[i] = lds[off[i]]; }
#pragma unroll
    for (int i = 0; i < NREAD; i++) acc += v[i];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc; uint32_t* addr;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8); hipMalloc(&addr, 4 * 16 * 1024);
  const int iters = 2000;
  unsigned long long h;
  for (int nt : {64, 256, 512, 1024}) {
    valu_kernel<8><<<1, nt>>>(out, cyc, iters); hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("VALU fma ILP8  threads %4d (%2d waves/CU, %.1f per SIMD): %.2f cycles per wave-instruction (wave 0 view), %.2f cycles per instr per SIMD\n",
           nt, nt / 64, nt / 256.0, (double)h / (iters * 8), (double)h / (iters * 8) / (nt / 256.0 > 1 ? nt / 256.0 : 1));
  }
  std::vector<uint32_t> ha(16 * 1024);
  for (int pattern = 0; pattern < 3; pattern++) {
    uint32_t rng = 12345;
    for (int i = 0; i < 16; i++)
      for (int t = 0; t < 1024; t++) {
        rng = rng * 1664525u + 1013904223u;
        uint32_t a = pattern == 0 ? (uint32_t)((t % 64) + 64 * i) : pattern == 1 ? (rng >> 8) % 6000 : (uint32_t)(5 + i);
        ha[i * 1024 + t] = a;
      }
    for (int nt : {64, 256, 512, 1024}) {
      // addr layout must match blockDim: rebuild per nt
      std::vector<uint32_t> hb(16 * nt);
      for (int i = 0; i < 16; i++) for (int t = 0; t < nt; t++) hb[i * nt + t] = ha[i * 1024 + t];
      hipMemcpy(addr, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
      lds_kernel<16><<<1, nt, 32768>>>(out, cyc, addr, iters); hipDeviceSynchronize();
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("LDS ds_read_b32 x16 %-9s threads %4d: %.2f cycles per wave-instruction per CU\n",
             pattern == 0 ? "linear" : pattern == 1 ? "random" : "broadcast", nt, (double)h / (iters * 16) / (nt / 64));
    }
  }
  return 0;
}
