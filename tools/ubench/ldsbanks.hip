// ldsbanks.hip - how gfx950's LDS serves a wave64 gather: number of banks seen by ds_read_b32 / ds_read_b64 and which
// lanes are served together (what plan.cpp's conflict model has to reproduce).  16 waves issue the same access pattern
// back to back, so the LDS pipe is the bottleneck and cycles per instruction = its service time.
//   pattern "stride s": lane l reads element l*s (elements of 4 / 8 bytes)
//   pattern "pair a,b": all lanes conflict-free (lane l -> element l) except lane b, which reads lane a's bank at another address
// Build: hipcc --offload-arch=gfx950 -O3 -o ldsbanks ldsbanks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int BYTES>
__global__ void k(float* out, unsigned long long* cyc, const uint32_t* addr, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
  const uint32_t a = addr[threadIdx.x & 63];            // byte address, the same pattern in every wave
  __syncthreads();
  float acc = 0.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint32_t aa = a; asm volatile("" : "+v"(aa));
      if (BYTES == 4) v[i] = *(const __attribute__((address_space(3))) float*)(aa);
      else { v2f q = *(const __attribute__((address_space(3))) v2f*)(aa); v[i] = q.x + q.y; }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) acc += v[i];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
static float *out; static unsigned long long* cyc; static uint32_t* addr;
template <int BYTES>
double run(const std::vector<uint32_t>& a) {
  const int iters = 400, nt = 1024;
  hipMemcpy(addr, a.data(), 256, hipMemcpyHostToDevice);
  k<BYTES><<<1, nt, 65536>>>(out, cyc, addr, iters); hipDeviceSynchronize();
  k<BYTES><<<1, nt, 65536>>>(out, cyc, addr, iters); hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  return (double)h / (iters * 8 * 16);                  // cycles per wave-instruction at the LDS
}
int main() {
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8); hipMalloc(&addr, 256);
  hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  std::vector<uint32_t> a(64);
  for (int bytes : {4, 8}) {
    printf("ds_read_b%d, lane l -> element l * stride:\n", bytes * 8);
    for (int s : {1, 2, 4, 8, 16, 32, 64, 128}) {
      for (int l = 0; l < 64; l++) a[l] = (uint32_t)((l * s * bytes) % 65536);
      printf("  stride %3d elements: %.2f cycles/instr\n", s, bytes == 4 ? run<4>(a) : run<8>(a));
    }
    printf("ds_read_b%d, conflict-free except lane b reading lane 0's bank at another address (+0 = served in another pass):\n", bytes * 8);
    for (int l = 0; l < 64; l++) a[l] = l * bytes;
    const double base = bytes == 4 ? run<4>(a) : run<8>(a);
    printf("  baseline %.2f\n ", base);
    for (int b = 1; b < 64; b++) {
      for (int l = 0; l < 64; l++) a[l] = l * bytes;
      a[b] = 0 * bytes + 4096;                          // same bank as lane 0 for any bank count dividing 1024 dwords, other address
      const double t = bytes == 4 ? run<4>(a) : run<8>(a);
      printf(" %d:%+.1f", b, t - base);
      if (b % 16 == 15) printf("\n ");
    }
    printf("\n");
    // how many banks: lane 1 reads element (bank of lane 0) + k banks ... find the period at which lane 1 collides with lane 0
    printf("ds_read_b%d, lane 1 reading element e (others conflict-free, lane 0 at element 0); a bump = same bank as lane 0 or its own neighbours:\n ", bytes * 8);
    for (int e : {16, 32, 48, 64, 96, 128, 192, 256, 512, 1024}) {
      for (int l = 0; l < 64; l++) a[l] = l * bytes;
      a[1] = e * bytes;
      const double t = bytes == 4 ? run<4>(a) : run<8>(a);
      printf(" e=%d:%+.2f", e, t - base);
    }
    printf("\n");
  }
  return 0;
}
