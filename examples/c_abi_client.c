/* c_abi_client.c - the LF-MMI hot path through the C ABI from a plain C host: no Python, no torch.
 *
 * What a maintainer of the reference would write behind its own extension boundary (pytorch_binding/src/pychain.cc:26-129
 * hands the same tensors to chain-computation.cc): compile the denominator graph once, keep the plan on the device, and per
 * minibatch call the denominator (pychain_hip_den_forward_backward) or the fused loss (pychain_hip_chain_loss_forward_backward).
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_abi_client.c \
 *       -L pychain_amd -lpychain_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/pychain_amd -Wl,-rpath,/opt/rocm/lib -o c_abi_client
 *   ./c_abi_client problem.bin result.bin [calls]
 * `calls` > 1 repeats the minibatch call (a training loop's pattern) and prints one line per call with what totals[5..7] report
 * about its time segments; the plan carries a burn-in controller state (pychain_hip_den_tseg_state), so data that forgets
 * slowly costs a C caller one redone call, not every call.
 *
 * problem.bin (little endian; written by tests/test_c_client.py from pychain_amd.synthetic):
 *   int32 B, T, D, H, K, Hn, Kn, fused;  float leaky
 *   denominator graph, reference layout (pychain/graph.py:36-66): int32 ft[K*3], fi[H*2]; float fp[K]; int32 bt[K*3], bi[H*2];
 *     float bp[K]; float leaky_probs[H], initial[H], final[H]
 *   network output float x[B*T*D]; int64 lengths[B]
 *   if fused: numerator graphs [B] of Hn states / Kn arcs (log domain): int32 ft[B*Kn*3], fi[B*Hn*2]; float fp[B*Kn];
 *     int32 bt[B*Kn*3], bi[B*Hn*2]; float bp[B*Kn]; float initial[B*Hn], final[B*Hn]
 * result.bin: float den_objf[B]; (fused: float num_objf[B];) float grad[B*T*D]; int32 bad[2]; float totals[8]
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pychain_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define CHECK_LIB(x) do { int rc_ = (x); if (rc_ != PYCHAIN_HIP_OK) { fprintf(stderr, "%s: %s\n", #x, pychain_hip_last_error()); exit(3); } } while (0)

static void* read_n(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read (%zu bytes)\n", bytes); exit(1); }
  return p;
}
static void* to_device(const void* host, size_t bytes) {
  void* d = NULL;
  CHECK_HIP(hipMalloc(&d, bytes ? bytes : 16));
  CHECK_HIP(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  if (argc != 3 && argc != 4) { fprintf(stderr, "usage: %s problem.bin result.bin [calls]\n", argv[0]); return 1; }
  const int calls = argc == 4 ? atoi(argv[3]) : 1;
  if (pychain_hip_abi_version() != PYCHAIN_HIP_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int32_t dims[8];
  float leaky;
  if (fread(dims, 4, 8, f) != 8 || fread(&leaky, 4, 1, f) != 1) { fprintf(stderr, "bad header\n"); return 1; }
  const int B = dims[0], T = dims[1], D = dims[2], H = dims[3], K = dims[4], Hn = dims[5], Kn = dims[6], fused = dims[7];
  int32_t* ft = read_n(f, (size_t)K * 12); int32_t* fi = read_n(f, (size_t)H * 8); float* fp = read_n(f, (size_t)K * 4);
  int32_t* bt = read_n(f, (size_t)K * 12); int32_t* bi = read_n(f, (size_t)H * 8); float* bp = read_n(f, (size_t)K * 4);
  float* lk = read_n(f, (size_t)H * 4); float* in = read_n(f, (size_t)H * 4); float* fn = read_n(f, (size_t)H * 4);
  const size_t nx = (size_t)B * T * D;
  float* x = read_n(f, nx * 4);
  int64_t* len = read_n(f, (size_t)B * 8);

  /* ---- once per model: the denominator plan (host), then resident on the device */
  const int64_t need = pychain_hip_den_plan_build(ft, fi, fp, bt, bi, bp, lk, in, fn, H, K, D, NULL, 0);
  if (need <= 0) { fprintf(stderr, "plan: %s\n", pychain_hip_last_error()); return 3; }
  void* blob = malloc((size_t)need);
  if (pychain_hip_den_plan_build(ft, fi, fp, bt, bi, bp, lk, in, fn, H, K, D, blob, (size_t)need) != need) { fprintf(stderr, "plan fill\n"); return 3; }
  int32_t info[8];
  CHECK_LIB(pychain_hip_den_plan_info(blob, (size_t)need, info));
  const int positions = info[0], hint = info[4];      /* positions: what the calls take as num_states (>= H: a state may sit on several lanes) */
  void* plan_dev = to_device(blob, (size_t)need);
  /* the plan's burn-in controller: 64 bytes of device memory, zeroed once, attached by plan address (ABI 16) */
  void* tstate = NULL;
  CHECK_HIP(hipMalloc(&tstate, pychain_hip_den_tseg_state_bytes()));
  CHECK_HIP(hipMemset(tstate, 0, pychain_hip_den_tseg_state_bytes()));
  CHECK_LIB(pychain_hip_den_tseg_state(plan_dev, tstate));

  /* ---- per minibatch */
  void* x_dev = to_device(x, nx * 4);
  void* len_dev = to_device(len, (size_t)B * 8);
  float *den_objf, *num_objf, *grad, *totals;
  int32_t* bad;
  CHECK_HIP(hipMalloc((void**)&den_objf, (size_t)B * 4)); CHECK_HIP(hipMalloc((void**)&num_objf, (size_t)B * 4));
  CHECK_HIP(hipMalloc((void**)&grad, nx * 4)); CHECK_HIP(hipMalloc((void**)&totals, PYCHAIN_HIP_TOTALS * 4));
  CHECK_HIP(hipMalloc((void**)&bad, 8));
  CHECK_HIP(hipMemset(bad, 0, 8));
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  if (!fused) {
    const size_t wsb = pychain_hip_den_workspace_bytes(B, T, positions, D);
    void* ws; CHECK_HIP(hipMalloc(&ws, wsb));
    for (int c = 0; c < calls; c++) {
      CHECK_LIB(pychain_hip_den_forward_backward(plan_dev, 0, hint, positions, D, x_dev, PYCHAIN_HIP_F32, 0, len_dev, B, T, leaky, 1.0f,
                                                 den_objf, grad, bad, totals, ws, wsb, st));
      if (calls > 1) {                                  /* (a trainer would not sync here: the state lives on the device) */
        float t8[PYCHAIN_HIP_TOTALS]; int32_t ts[5];
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(t8, totals, sizeof(t8), hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(ts, tstate, sizeof(ts), hipMemcpyDeviceToHost));
        printf("call %d: segments %d rows_missed %d worst_mismatch %.3g next_burn_in %d cooling_down %d\n", c, (int)t8[6], (int)t8[5], t8[7], ts[1], ts[2]);
      }
    }
  } else {
    int32_t* nft = read_n(f, (size_t)B * Kn * 12); int32_t* nfi = read_n(f, (size_t)B * Hn * 8); float* nfp = read_n(f, (size_t)B * Kn * 4);
    int32_t* nbt = read_n(f, (size_t)B * Kn * 12); int32_t* nbi = read_n(f, (size_t)B * Hn * 8); float* nbp = read_n(f, (size_t)B * Kn * 4);
    float* nin = read_n(f, (size_t)B * Hn * 4); float* nfn = read_n(f, (size_t)B * Hn * 4);
    void* d_ft = to_device(nft, (size_t)B * Kn * 12); void* d_fi = to_device(nfi, (size_t)B * Hn * 8); void* d_fp = to_device(nfp, (size_t)B * Kn * 4);
    void* d_bt = to_device(nbt, (size_t)B * Kn * 12); void* d_bi = to_device(nbi, (size_t)B * Hn * 8); void* d_bp = to_device(nbp, (size_t)B * Kn * 4);
    void* d_in = to_device(nin, (size_t)B * Hn * 4); void* d_fn = to_device(nfn, (size_t)B * Hn * 4);
    const size_t dwb = pychain_hip_den_workspace_min_bytes(B, T, positions, D), nwb = pychain_hip_num_workspace_bytes(B, T, Hn, Kn, D);
    void *dws, *nws;
    CHECK_HIP(hipMalloc(&dws, dwb)); CHECK_HIP(hipMalloc(&nws, nwb));
    /* loss = -(num - den), not averaged (loss_scale 1, no normaliser); grad = d loss / d x */
    CHECK_LIB(pychain_hip_chain_loss_forward_backward(plan_dev, 0, hint, positions, leaky, d_ft, d_fi, d_fp, d_bt, d_bi, d_bp, d_in, d_fn, 1, Hn, Kn,
                                                      x_dev, PYCHAIN_HIP_F32, len_dev, B, T, D, 1.0f, den_objf, num_objf, grad, bad,
                                                      1.0f, NULL, totals, dws, dwb, nws, nwb, st));
  }
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_LIB(pychain_hip_den_tseg_state(plan_dev, NULL));
  fclose(f);

  FILE* o = fopen(argv[2], "wb");
  if (!o) { perror(argv[2]); return 1; }
  float* h = malloc(nx * 4 > 64 ? nx * 4 : 64);
  CHECK_HIP(hipMemcpy(h, den_objf, (size_t)B * 4, hipMemcpyDeviceToHost)); fwrite(h, 4, B, o);
  if (fused) { CHECK_HIP(hipMemcpy(h, num_objf, (size_t)B * 4, hipMemcpyDeviceToHost)); fwrite(h, 4, B, o); }
  CHECK_HIP(hipMemcpy(h, grad, nx * 4, hipMemcpyDeviceToHost)); fwrite(h, 4, nx, o);
  int32_t hb[2];
  CHECK_HIP(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost)); fwrite(hb, 4, 2, o);
  CHECK_HIP(hipMemcpy(h, totals, PYCHAIN_HIP_TOTALS * 4, hipMemcpyDeviceToHost)); fwrite(h, 4, PYCHAIN_HIP_TOTALS, o);
  fclose(o);
  printf("c_abi_client: B=%d T=%d D=%d, %d states on %d positions, %s: loss %.6f, bad %d\n", B, T, D, H, positions,
         fused ? "fused loss" : "denominator", h[0], hb[0] + hb[1]);
  return 0;
}
