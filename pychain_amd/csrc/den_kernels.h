// den_kernels.h - launch interface of the denominator kernels (den_kernels.hip).
#ifndef PYCHAIN_HIP_DEN_KERNELS_H_
#define PYCHAIN_HIP_DEN_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pychain_hip {

struct DenArgs {
  const char* plans;         // device plan(s)
  int64_t plan_stride;       // bytes between per-sequence plans, 0 = shared
  const float* x;            // [B,T,D]
  const int64_t* lengths;    // [B]
  float* objf;               // [B]
  float* grad;               // [B,T,D]
  int32_t* bad;              // [1]
  float* alpha_store;        // [B,T,Hp]    alpha'(t,.)/tot(t), alpha numbering
  float* beta_store;         // [B,T+1,Hp]  beta(t,.) (unit sum), beta numbering; row 0 unused
  int B, T, D, H, Hp;
  int input_is_exp;
  int frames_per_block;      // gamma kernel: frames one workgroup handles
  int phase_mask;            // bit0 recursion launch, bit1 gamma launch (bench aid)
  float coef, grad_scale;
  const float* grad_scale_dev;   // optional device scalar multiplied into grad_scale (upstream autograd gradient)
};

// Enqueues the two launches on `st`.  On failure returns the HIP error and, when the
// shape is unsupported, a reason in *why.
hipError_t launch_den(const DenArgs& a, int gamma_max_groups, int resident_slot_rows, hipStream_t st,
                      const char** why);

}  // namespace pychain_hip
#endif
