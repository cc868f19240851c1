// device_utils.h - small device helpers shared by the den and num kernels.
#ifndef PYCHAIN_HIP_DEVICE_UTILS_H_
#define PYCHAIN_HIP_DEVICE_UTILS_H_

#include <hip/hip_runtime.h>

namespace pychain_hip {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// How a raw nnet-output element enters LDS (pychain/loss.py:30,43):
//   kXExpClamp  exp(clamp(v,-30,30))   denominator fed raw network output (clamp+exp fused)
//   kXIdentity  v                      denominator fed exp'd input (pychain_C.forward_backward contract)
//   kXClamp     clamp(v,-30,30)        numerator (log domain)
enum { kXExpClamp = 0, kXIdentity = 1, kXClamp = 2 };
__device__ __forceinline__ float clamp_exp(float v, int mode) {
  if (mode == kXIdentity) return v;
  const float c = fminf(fmaxf(v, -30.f), 30.f);
  return mode == kXClamp ? c : expf(c);
}

// ---- nnet-output row: global -> registers (early) -> LDS (late) ------------------
template <int NT, int VEC, int XCH>
struct XRow {
  float v[(VEC * XCH) > 0 ? (VEC * XCH) : 1];
  __device__ __forceinline__ void load(const float* __restrict__ row, int D, int tid) {
    if constexpr (XCH > 0) {
#pragma unroll
      for (int c = 0; c < XCH; c++) {
        const int e = (c * NT + tid) * VEC;
        if constexpr (VEC == 4) {
          float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
          if (e < D) q = *reinterpret_cast<const float4*>(row + e);
          v[c * 4 + 0] = q.x; v[c * 4 + 1] = q.y; v[c * 4 + 2] = q.z; v[c * 4 + 3] = q.w;
        } else {
          v[c] = e < D ? row[e] : 0.f;
        }
      }
    }
  }
  __device__ __forceinline__ void store(float* lds, const float* __restrict__ row, int D, int tid, int is_exp) {
    if constexpr (XCH > 0) {
#pragma unroll
      for (int c = 0; c < XCH; c++) {
        const int e = (c * NT + tid) * VEC;
        if (e < D) {
          if constexpr (VEC == 4) {
            float4 q;
            q.x = clamp_exp(v[c * 4 + 0], is_exp); q.y = clamp_exp(v[c * 4 + 1], is_exp);
            q.z = clamp_exp(v[c * 4 + 2], is_exp); q.w = clamp_exp(v[c * 4 + 3], is_exp);
            *reinterpret_cast<float4*>(lds + e) = q;
          } else {
            lds[e] = clamp_exp(v[c], is_exp);
          }
        }
      }
    } else {  // any D: no register staging
      for (int e = tid; e < D; e += NT) lds[e] = clamp_exp(row[e], is_exp);
    }
  }
};

}  // namespace pychain_hip
#endif
