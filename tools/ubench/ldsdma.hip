// ldsdma.hip - what `buffer_load_dwordx4 ... lds` (global -> LDS without VGPRs) does on gfx950: where lane l's 16 bytes
// land relative to the M0 base, what an out-of-range lane writes, and that the wave's own ds_read after
// s_waitcnt vmcnt(0) sees the data.  Build: hipcc --offload-arch=gfx950 -O3 -o ldsdma ldsdma.hip ; run: ./ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const float* g, float* out, int nfloats, int nvalid) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < nfloats; i += blockDim.x) ((float*)smem)[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nvalid * 4, 0x00020000);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // wave w: 1 KiB chunk w of the row -> LDS chunk w; 16 bytes per lane
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(smem + wave * 1024), 16, lane * 16, wave * 1024, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  // the lane's own 16 bytes, straight back
  const float4 mine = *(const float4*)(smem + wave * 1024 + lane * 16);
  out[nfloats + threadIdx.x] = mine.x;
  __syncthreads();
  for (int i = threadIdx.x; i < nfloats; i += blockDim.x) out[i] = ((float*)smem)[i];
}
int main() {
  const int nt = 256, nfloats = nt * 4, nvalid = 1000;      // the last 24 floats are out of the buffer's range
  std::vector<float> h(nfloats);
  for (int i = 0; i < nfloats; i++) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, nfloats * 4); hipMalloc(&o, (nfloats + nt) * 4);
  hipMemcpy(g, h.data(), nfloats * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(nt), nfloats * 4, 0, g, o, nfloats, nvalid);
  std::vector<float> r(nfloats + nt);
  hipMemcpy(r.data(), o, (nfloats + nt) * 4, hipMemcpyDeviceToHost);
  int linear = 1, own = 1;
  for (int i = 0; i < nvalid; i++) linear &= r[i] == (float)i;
  for (int t = 0; t < nt && 4 * t < nvalid; t++) own &= r[nfloats + t] == (float)(4 * t);
  printf("lds[i] == g[i] for the in-range part (lane l -> base + 16 l): %s\n", linear ? "yes" : "NO");
  printf("a lane's own ds_read after vmcnt(0) sees its data: %s\n", own ? "yes" : "NO");
  printf("first floats in LDS:"); for (int i = 0; i < 12; i++) printf(" %g", r[i]); printf("\n");
  printf("floats 250..262:"); for (int i = 250; i < 262; i++) printf(" %g", r[i]); printf("\n");
  printf("out-of-range tail (floats %d..%d):", nvalid - 2, nvalid + 6); for (int i = nvalid - 2; i < nvalid + 6; i++) printf(" %g", r[i]); printf("\n");
  return 0;
}
