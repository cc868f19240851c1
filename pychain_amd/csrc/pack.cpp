// pack.cpp - per-utterance numerator graphs -> the batched, padded ChainGraphBatch tensors, natively.
//
// The reference collates a list of ChainGraph objects in Python, one small tensor copy per field and utterance
// (pychain/graph.py:122-175: ~580 copies, 3.3 ms for a 64-utterance batch) and re-indexes every tensor in `reorder`
// (:177-194); a trainer does both on EVERY step, on the thread that launches the loss.  Here the ten tensors of a
// batch live in ONE buffer (64-byte aligned fields, fixed order): one native pack, one H2D copy, and `reorder` is a
// row gather over that buffer - on the host for the CPU-visible tensors, on the device (api.hip) for the staged copy.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#include "../../include/pychain_hip.h"
#include "common.h"

namespace {
struct Field { int64_t row_bytes; int kind; };    // kind 0: zero padding, 1: float padding with `pad`
// forward_transitions, forward_transition_indices, forward_transition_probs, backward_*, final, initial, leaky, start_state
void fields(int K, int H, int log_domain, Field f[PYCHAIN_HIP_BATCH_FIELDS]) {
  const int64_t k = K, h = H;
  f[0] = {k * 12, 0}; f[1] = {h * 8, 0}; f[2] = {k * 4, 0};
  f[3] = {k * 12, 0}; f[4] = {h * 8, 0}; f[5] = {k * 4, 0};
  f[6] = {h * 4, 1}; f[7] = {h * 4, 1}; f[8] = {log_domain ? 0 : h * 4, 0}; f[9] = {8, 0};
}
int64_t align64(int64_t x) { return (x + 63) & ~int64_t(63); }
}  // namespace

extern "C" int64_t pychain_hip_batch_layout(int B, int K, int H, int log_domain, int64_t offsets[PYCHAIN_HIP_BATCH_FIELDS],
                                            int64_t row_bytes[PYCHAIN_HIP_BATCH_FIELDS]) {
  if (B <= 0 || K <= 0 || H <= 0 || !offsets || !row_bytes)
    return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "batch_layout: bad arguments");
  Field f[PYCHAIN_HIP_BATCH_FIELDS];
  fields(K, H, log_domain, f);
  int64_t off = 0;
  for (int i = 0; i < PYCHAIN_HIP_BATCH_FIELDS; i++) { offsets[i] = off; row_bytes[i] = f[i].row_bytes; off = align64(off + f[i].row_bytes * B); }
  return off;
}

extern "C" int pychain_hip_batch_pack(int B, int K, int H, int log_domain, const uint64_t* rec, void* out, size_t out_bytes) {
  int64_t offs[PYCHAIN_HIP_BATCH_FIELDS], rb[PYCHAIN_HIP_BATCH_FIELDS];
  const int64_t total = pychain_hip_batch_layout(B, K, H, log_domain, offs, rb);
  if (total < 0) return (int)total;
  if (!rec || !out || (int64_t)out_bytes < total) return pychain_hip::fail(PYCHAIN_HIP_EWORKSPACE, "batch_pack: buffer too small");
  char* o = (char*)out;
  const float pad = log_domain ? -std::numeric_limits<float>::infinity() : 0.f;     // padded states are unreachable (graph.py:140-145)
  for (int b = 0; b < B; b++) {
    const uint64_t* r = rec + (size_t)b * PYCHAIN_HIP_BATCH_REC_WORDS;
    const int64_t k = (int64_t)r[0], h = (int64_t)r[1];
    if (k < 0 || k > K || h < 0 || h > H)
      return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "batch_pack: graph %d has %lld transitions / %lld states, the batch allows %d / %d",
                               b, (long long)k, (long long)h, K, H);
    const int64_t used[PYCHAIN_HIP_BATCH_FIELDS] = {k * 12, h * 8, k * 4, k * 12, h * 8, k * 4, h * 4, h * 4, log_domain ? 0 : h * 4, 8};
    for (int i = 0; i < PYCHAIN_HIP_BATCH_FIELDS; i++) {
      if (rb[i] == 0) continue;
      char* dst = o + offs[i] + rb[i] * b;
      if (i == 9) { const int64_t s = (int64_t)r[2]; memcpy(dst, &s, 8); continue; }
      const void* src = (const void*)(uintptr_t)r[3 + i];
      if (used[i] > 0) {
        if (!src) return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "batch_pack: graph %d lacks tensor %d", b, i);
        memcpy(dst, src, (size_t)used[i]);
      }
      if (i == 6 || i == 7) { float* p = (float*)(dst + used[i]); for (int64_t j = 0; j < (rb[i] - used[i]) / 4; j++) p[j] = pad; }
      else memset(dst + used[i], 0, (size_t)(rb[i] - used[i]));
    }
  }
  return PYCHAIN_HIP_OK;
}

extern "C" int pychain_hip_batch_reorder(int B_in, int B_out, int K, int H, int log_domain, const void* in, void* out,
                                         const int64_t* order) {
  int64_t oi[PYCHAIN_HIP_BATCH_FIELDS], oo[PYCHAIN_HIP_BATCH_FIELDS], rb[PYCHAIN_HIP_BATCH_FIELDS];
  if (pychain_hip_batch_layout(B_in, K, H, log_domain, oi, rb) < 0 || pychain_hip_batch_layout(B_out, K, H, log_domain, oo, rb) < 0) return PYCHAIN_HIP_EINVAL;
  if (!in || !out || !order) return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "batch_reorder: null pointer");
  for (int b = 0; b < B_out; b++) {
    if (order[b] < 0 || order[b] >= B_in) return pychain_hip::fail(PYCHAIN_HIP_EINVAL, "batch_reorder: index %lld out of range", (long long)order[b]);
    for (int i = 0; i < PYCHAIN_HIP_BATCH_FIELDS; i++)
      if (rb[i]) memcpy((char*)out + oo[i] + rb[i] * b, (const char*)in + oi[i] + rb[i] * order[b], (size_t)rb[i]);
  }
  return PYCHAIN_HIP_OK;
}
