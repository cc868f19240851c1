// ldswrap.hip - which address bits does a DS read decode on gfx950?  (Can the upper bits of an LDS address
// VGPR carry other data, e.g. the second operand's index of a packed arc?)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef __attribute__((address_space(3))) const float lds_cf;
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
__device__ __forceinline__ float ld1(uint32_t a) { return *(lds_cf*)(a); }
__global__ void k(float* out, int lds_floats) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  const uint32_t base = 4u * (100u + threadIdx.x);
  uint32_t hi[8] = {0u, 1u << 16, 1u << 17, 1u << 18, 1u << 19, 1u << 20, 1u << 24, 0xABCDu << 18};
  for (int j = 0; j < 8; j++) {
    uint32_t a = base | hi[j];
    asm volatile("" : "+v"(a));
    out[j * 64 + threadIdx.x] = ld1(a);
  }
}
int main() {
  float* out; hipMalloc(&out, 8 * 64 * 4);
  for (int kb : {16, 64, 128, 160}) {
    const int n = kb * 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, n * 4);
    hipMemset(out, 0xff, 8 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), n * 4, 0, out, n);
    hipDeviceSynchronize();
    float h[8 * 64]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("LDS %3d KiB, lane 5 reads address 4*105 | hi:  hi=0 -> %.0f | 1<<16 -> %.0f | 1<<17 -> %.0f | 1<<18 -> %.0f | 1<<19 -> %.0f | 1<<20 -> %.0f | 1<<24 -> %.0f | 0xABCD<<18 -> %.0f\n",
           kb, h[5], h[64 + 5], h[128 + 5], h[192 + 5], h[256 + 5], h[320 + 5], h[384 + 5], h[448 + 5]);
  }
  return 0;
}
