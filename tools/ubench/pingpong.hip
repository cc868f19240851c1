// Round-trip time of a flag between two workgroups through device memory (agent-scope relaxed atomics):
// what a per-frame hand-shake between two CUs that share one sequence's recursion would cost.
// Workgroup i runs on XCD i % 8: partner = 8 -> same XCD, partner = 1 -> another XCD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pingpong.hip -o build/ubench_pingpong && build/ubench_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void pingpong(int* flags, int partner, int rounds, long long* cycles) {
  const int me = blockIdx.x;
  if (me != 0 && me != partner) return;
  if (threadIdx.x != 0) return;
  int* mine = flags + (me == 0 ? 0 : 64);       // separate cache lines
  int* theirs = flags + (me == 0 ? 64 : 0);
  const long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; r++) {
    if (me == 0) {
      __hip_atomic_store(theirs, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r) {}
    } else {
      while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r) {}
      __hip_atomic_store(theirs, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (me == 0) *cycles = wall_clock64() - t0;   // 100 MHz ticks
}

int main() {
  int* flags; long long* cyc;
  hipMalloc(&flags, 1024); hipMalloc(&cyc, 8);
  const int rounds = 20000;
  for (int partner : {8, 16, 1, 3}) {
    hipMemset(flags, 0, 1024);
    hipLaunchKernelGGL(pingpong, dim3(32), dim3(64), 0, 0, flags, partner, rounds, cyc);
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("partner workgroup %2d (%s XCD): %.0f ns per round trip\n", partner, partner % 8 == 0 ? "same" : "other",
           (double)c * 10.0 / rounds);
  }
  return 0;
}
