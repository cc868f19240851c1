// common.h - error reporting and per-call knobs shared by the host-side translation units.
#ifndef PYCHAIN_HIP_COMMON_H_
#define PYCHAIN_HIP_COMMON_H_

#include <cstdarg>
#include <cstdio>

namespace pychain_hip {

char* last_error_buffer();          // thread-local, 512 bytes (api.hip)

// Everything a call takes from the library's settings, read ONCE when the call starts (api.hip:call_knobs):
// the process-wide defaults (pychain_hip_set_option, pychain_hip_set_verbose_level, ...) overlaid with the
// calling thread's overrides (pychain_hip_set_thread_option).  The launch code and the kernels see only this
// snapshot - a second host thread that changes a setting while a call is being enqueued cannot tear it, and a
// validation loop at verbose level 1 in one thread does not slow a training loop in another.
// Nothing on the call path reads the environment.
struct CallKnobs {
  int verbose;                // reference's verbose level (base.h:34-42): >= 1 checks every frame
  int den_phase_mask;         // bit 0 recursion launch, bit 1 occupancy launches (measurement aid)
  int den_lazy;               // 0: never the lazy-normalisation recursions (the two-barrier kernel: second opinion of the tests)
  int den_segments;           // 0 = automatic (the streamed occupancy pass where the shape allows, else gated segments);
                              // n >= 1: the gated schedule with n time segments (1 = no overlap)
  int gamma16;                // 1: the one-frame occupancy kernel also where the two-frame one fits
  int den_pair;               // -1 automatic, 0 never, 1 wherever the shape allows
  int den_dma;                // 0: nnet-output rows of the lazy recursions through registers, else by LDS-direct loads; 2: never exp'd ahead (DenArgs::ex), 3: exp'd ahead wherever the shape allows
  // debug_corrupt_row = "den|num,b,t,scale": one stored alpha row is scaled before the occupancy pass reads it,
  // so that the reference's 5 % invariant (chain-computation.cc:363-390, chain-log-domain-computation.cc:289-303)
  // can be seen to fire
  int num_compat;             // 1: the numerator in the reference's own fp32 arithmetic (num_compat.hip) instead of the exact path
  int den_tseg;               // time segments per (sequence, direction) of the lazy recursions: -1 automatic, 0 / 1 never, 2 or 4 wherever the shape allows
  int den_tburn;              // frames a segment starts outside itself (its burn-in); default 192
  int den_sg;                 // 0: never the one-gather form of the lazy recursions for "pdf by state" plans (den_lazy.inc.h: SG); default 1
  int den_q;                  // 0: never the one-word state vectors of the lazy recursions (den_lazy.inc.h: MAP::kQ; launch hint bit 19); default 0
  int den_cross;              // 0: the recursions of a pdf-by-state plan never emit occupancies themselves (den_lazy.inc.h: XF); default 1
  int chain_slices;           // the fused loss over a batch larger than the chip: -1 automatic, 0 / 1 one call, n >= 2 that many slices (api.hip)
  int plan_split;             // pychain_hip_den_plan_build: 0 = no state on more than one lane (plans of one batch with a stride must have
                              // the same number of positions), -1 = automatic (plan.cpp, "states on several lanes")
  int corrupt_what;           // 0 none, 1 denominator, 2 numerator
  int corrupt_b, corrupt_t;
  float corrupt_scale;
};
CallKnobs call_knobs();

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace pychain_hip
#endif
