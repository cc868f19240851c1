// num_kernels.h - launch interface of the numerator kernel (num_kernels.hip).
#ifndef PYCHAIN_HIP_NUM_KERNELS_H_
#define PYCHAIN_HIP_NUM_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pychain_hip {

struct NumArgs {
  const int32_t* fwd_trans; const int32_t* fwd_idx; const float* fwd_probs;
  const int32_t* bwd_trans; const int32_t* bwd_idx; const float* bwd_probs;
  const float* initial; const float* final_;
  const float* x;            // [B,T,D] raw
  const int64_t* lengths;    // [B]
  float* objf;               // [B]
  float* grad;               // [B,T,D]
  int32_t* bad;              // [1]
  float* alpha_ws;           // [B,T+1,H]  alpha(t,h) as the reference stores it (log, scaled)
  float* logtot_ws;          // [B,T+1]    alpha-sum(t)
  int graph_stride;          // 1 = per-sequence graphs, 0 = shared
  int B, T, D, H, K;
  int grad_mode;
  float grad_scale;
};

size_t num_lds_bytes(int H, int K, int D);
hipError_t launch_num(const NumArgs& a, hipStream_t st, const char** why);

}  // namespace pychain_hip
#endif
