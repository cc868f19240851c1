// cpu.cpp - the HOST twins of the two forward-backward entry points (include/pychain_hip.h: pychain_hip_cpu_*; SURVEY.md
// §8(b) lists them in the boundary).  The reference runs on whatever device its input tensor lives on
// (chain-computation.cc:40,102,183,318; chain-log-domain-computation.cc:123-159,231-271) and code written against it
// unit-tests its criterion on CPU tensors; `pychain_amd` therefore serves CPU tensors from HERE and device tensors from the
// HIP kernels - never one for the other: a device tensor that cannot reach the kernels raises, it does not come here
// (pychain_amd/native.py), and the test suite's CPU checker is no part of it.
//
// Own design, not the reference's loop nest: the sequences of a minibatch are independent (chain-computation.h:33-35), so they
// are dealt to host threads; clamp(-30, 30) and exp (pychain/loss.py:30,43) are applied to a frame's row once, into a scratch
// row, instead of to the whole [B,T,D] tensor in two passes; the per-frame totals, the log-probability and every
// normaliser are accumulated in fp64 (the state vectors and the gradient are fp32, as the reference's); the numerator keeps
// fp64 log-probabilities and an exact max-subtracted log-sum-exp like the device path (num_kernels.hip) - the reference's
// fp32 LogAdd chain with its cut-off is the device's option num_compat, not rebuilt here.  Equations: chain-computation.h:109-156.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/pychain_hip.h"
#include "common.h"

namespace pychain_hip {
namespace {
std::atomic<long> g_cpu_calls{0};

struct Csr {                       // one graph in the reference layout: arcs {src, dst, pdf} in runs per state
  const int32_t* trans;
  const int32_t* idx;
  const float* prob;
};
inline float clamp30(float v) { return v != v ? v : (v < -30.f ? -30.f : (v > 30.f ? 30.f : v)); }

template <class F>
void for_each_sequence(int B, int num_threads, F&& body) {
  int nt = num_threads > 0 ? num_threads : (int)std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (nt > B) nt = B;
  if (nt == 1) { for (int b = 0; b < B; b++) body(b); return; }
  std::atomic<int> next{0};
  std::vector<std::thread> pool;
  for (int i = 0; i < nt; i++)
    pool.emplace_back([&]() { for (int b = next.fetch_add(1); b < B; b = next.fetch_add(1)) body(b); });
  for (auto& th : pool) th.join();
}

// ---- denominator: probability domain, leaky-HMM, per-frame renormalisation (chain-computation.h:124-153) ------------------
// alpha'(t) = alpha(t) + tot(t) coef leaky; alpha(t+1,j) = sum_{k into j} alpha'(t,src) p x(t,pdf) / tot(t);
// beta'(t,i) = sum_{k out of i} p x(t,pdf) beta(t+1,dst) / tot(t); beta(t) = beta'(t) + coef sum_i leaky_i beta'(t,i);
// gamma(t,pdf) += alpha'(t,src) p x(t,pdf) beta(t+1,dst) / tot(t).
bool den_one(const Csr& fwd, const Csr& bwd, const float* leaky, const float* init, const float* fin, const float* x, int input_is_exp,
             int L, int T, int D, int H, float coef, float gscale, float* objf, float* grad) {
  std::vector<float> alpha((size_t)(L + 1) * H), beta(2 * (size_t)H), row((size_t)D);
  std::vector<double> tot((size_t)L + 1);
  bool ok = true;
  auto stage = [&](int t) {
    const float* xr = x + (size_t)t * D;
    if (input_is_exp) { for (int n = 0; n < D; n++) row[n] = xr[n]; }
    else { for (int n = 0; n < D; n++) row[n] = std::exp(clamp30(xr[n])); }
  };
  auto dash = [&](int t) {                     // total, then the leaky term: the stored row is alpha'
    float* a = alpha.data() + (size_t)t * H;
    double s = 0.0;
    for (int h = 0; h < H; h++) s += (double)a[h];
    tot[t] = s;
    const float add = (float)(s * (double)coef);
    for (int h = 0; h < H; h++) a[h] += add * leaky[h];
    if (!(s > 0.0) || !std::isfinite(s)) ok = false;
  };
  for (int h = 0; h < H; h++) alpha[h] = init[h];
  dash(0);
  for (int t = 1; t <= L; t++) {
    stage(t - 1);
    const float* pa = alpha.data() + (size_t)(t - 1) * H;
    float* a = alpha.data() + (size_t)t * H;
    const float inv = (float)(1.0 / tot[t - 1]);
    for (int h = 0; h < H; h++) {
      float acc = 0.f;
      for (int k = bwd.idx[2 * h]; k < bwd.idx[2 * h + 1]; k++)
        acc += pa[bwd.trans[3 * k]] * bwd.prob[k] * row[bwd.trans[3 * k + 2]];
      a[h] = acc * inv;
    }
    dash(t);
  }
  // log-probability: log sum_i alpha'(L,i) final(i) + sum_{t<L} log tot(t)   (chain-computation.cc:209-230)
  const float* aL = alpha.data() + (size_t)L * H;
  double last = 0.0;
  for (int h = 0; h < H; h++) last += (double)aL[h] * (double)fin[h];
  double lp = std::log(last);
  for (int t = 0; t < L; t++) lp += std::log(tot[t]);
  *objf = (float)lp;
  if (!std::isfinite(lp)) ok = false;
  // beta(L) = final / last, + its leaky sum (:232-245, :313-330)
  auto leak = [&](float* b) {
    double s = 0.0;
    for (int h = 0; h < H; h++) s += (double)b[h] * (double)leaky[h];
    const float add = (float)(s * (double)coef);
    for (int h = 0; h < H; h++) b[h] += add;
  };
  float* bn = beta.data() + (size_t)(L & 1) * H;
  for (int h = 0; h < H; h++) bn[h] = (float)((double)fin[h] / last);
  leak(bn);
  for (size_t i = 0; i < (size_t)T * D; i++) grad[i] = 0.f;
  for (int t = L - 1; t >= 0; t--) {
    stage(t);
    const float* a = alpha.data() + (size_t)t * H;
    const float* nb = beta.data() + (size_t)((t + 1) & 1) * H;
    float* b = beta.data() + (size_t)(t & 1) * H;
    float* g = grad + (size_t)t * D;
    const float inv = (float)(1.0 / tot[t]);
    for (int h = 0; h < H; h++) {
      const float occ = a[h] * inv;
      float acc = 0.f;
      for (int k = fwd.idx[2 * h]; k < fwd.idx[2 * h + 1]; k++) {
        const int pdf = fwd.trans[3 * k + 2];
        const float v = fwd.prob[k] * nb[fwd.trans[3 * k + 1]] * row[pdf];
        acc += v;
        g[pdf] += v * occ;
      }
      b[h] = acc * inv;
    }
    if (t == 0) {                              // the reference's check: a frame's occupancies sum to one within 5 % (:345-391)
      double s = 0.0;
      for (int n = 0; n < D; n++) s += (double)g[n];
      if (!(std::fabs(s - 1.0) <= 0.05)) ok = false;
    }
    leak(b);
    if (gscale != 1.f) for (int n = 0; n < D; n++) g[n] *= gscale;
  }
  return ok;
}

// ---- numerator: log domain, no leaky-HMM (chain-log-domain-computation.cc), fp64 log-probabilities ------------------------
struct Lse64 {
  double m = -std::numeric_limits<double>::infinity(), s = 0.0;
  void push(double e) {
    if (e != e) { m = e; return; }
    if (e == -std::numeric_limits<double>::infinity()) return;
    if (e > m) { s = (m == -std::numeric_limits<double>::infinity()) ? 1.0 : s * std::exp(m - e) + 1.0; m = e; }
    else s += std::exp(e - m);
  }
  double value() const { return (m != m || m == -std::numeric_limits<double>::infinity()) ? m : m + std::log(s); }
};
bool num_one(const Csr& fwd, const Csr& bwd, const float* init, const float* fin, const float* x, int L, int T, int D, int H,
             int grad_mode_flags, float gscale, float* objf, float* grad) {
  // (PYCHAIN_HIP_CPU_NO_CLAMP: the network output as it is - the contract of pychain_C.forward_backward_log_domain, whose C++
  // does not clamp (chain-log-domain-computation.cc:137-145); ChainFunction's clamp(-30, 30), pychain/loss.py:30, otherwise)
  const bool clamp = (grad_mode_flags & PYCHAIN_HIP_CPU_NO_CLAMP) == 0;
  const int grad_mode = grad_mode_flags & 0xff;
  auto xv = [clamp](float v) { return clamp ? clamp30(v) : v; };
  const double ninf = -std::numeric_limits<double>::infinity();
  std::vector<double> alpha((size_t)(L + 1) * H), beta(2 * (size_t)H), occ((size_t)D);
  bool ok = true;
  for (int h = 0; h < H; h++) alpha[h] = (double)init[h];
  for (int t = 1; t <= L; t++) {
    const float* xr = x + (size_t)(t - 1) * D;
    const double* pa = alpha.data() + (size_t)(t - 1) * H;
    double* a = alpha.data() + (size_t)t * H;
    for (int h = 0; h < H; h++) {
      Lse64 acc;
      for (int k = bwd.idx[2 * h]; k < bwd.idx[2 * h + 1]; k++)
        acc.push(pa[bwd.trans[3 * k]] + ((double)bwd.prob[k] + (double)xv(xr[bwd.trans[3 * k + 2]])));
      a[h] = acc.value();
    }
  }
  Lse64 tl;
  for (int h = 0; h < H; h++) tl.push(alpha[(size_t)L * H + h] + (double)fin[h]);
  const double logp = tl.value();
  *objf = (float)logp;
  if (!std::isfinite(logp)) ok = false;
  const float fill = grad_mode == PYCHAIN_HIP_GRAD_LOG ? -std::numeric_limits<float>::infinity() : 0.f;
  if (grad_mode != PYCHAIN_HIP_GRAD_ACCUM) for (size_t i = 0; i < (size_t)T * D; i++) grad[i] = fill;
  double* bn = beta.data() + (size_t)(L & 1) * H;
  for (int h = 0; h < H; h++) bn[h] = (double)fin[h];
  std::vector<int> touched;
  for (int t = L - 1; t >= 0; t--) {
    const float* xr = x + (size_t)t * D;
    const double* a = alpha.data() + (size_t)t * H;
    const double* nb = beta.data() + (size_t)((t + 1) & 1) * H;
    double* b = beta.data() + (size_t)(t & 1) * H;
    float* g = grad + (size_t)t * D;
    touched.clear();
    double fsum = 0.0;
    for (int h = 0; h < H; h++) {
      Lse64 acc;
      for (int k = fwd.idx[2 * h]; k < fwd.idx[2 * h + 1]; k++) {
        const int pdf = fwd.trans[3 * k + 2];
        const double term = (double)fwd.prob[k] + nb[fwd.trans[3 * k + 1]] + (double)xv(xr[pdf]);
        acc.push(term);
        const double o = std::exp(a[h] + term - logp);          // occupancy of the arc (0 where a state cannot be reached)
        if (o > 0.0) { if (occ[pdf] == 0.0) touched.push_back(pdf); occ[pdf] += o; fsum += o; }
        else if (o != o) ok = false;
      }
      b[h] = acc.value();
    }
    if (t == 0 && !(std::fabs(fsum - 1.0) <= 0.05)) ok = false;    // chain-log-domain-computation.cc:283-304
    for (int pdf : touched) {
      const double o = occ[pdf];
      occ[pdf] = 0.0;
      if (grad_mode == PYCHAIN_HIP_GRAD_LOG) g[pdf] = (float)std::log(o);
      else if (grad_mode == PYCHAIN_HIP_GRAD_LINEAR) g[pdf] = gscale * (float)o;
      else g[pdf] += gscale * (float)o;
    }
  }
  (void)ninf;
  return ok;
}

int check_common(const char* who, const void* ft, const void* fi, const void* fp, const void* bt, const void* bi, const void* bp,
                 const void* initial, const void* final_, const void* x, const int64_t* lengths, const void* objf, const void* grad,
                 const void* bad, int B, int T, int D, int H, int K) {
  if (!ft || !fi || !fp || !bt || !bi || !bp || !initial || !final_ || !x || !lengths || !objf || !grad || !bad)
    return fail(PYCHAIN_HIP_EINVAL, "%s: null pointer argument", who);
  if (B <= 0 || T <= 0 || D <= 0 || H <= 0 || K <= 0) return fail(PYCHAIN_HIP_EINVAL, "%s: bad sizes B=%d T=%d H=%d K=%d D=%d", who, B, T, H, K, D);
  for (int b = 0; b < B; b++)
    if (lengths[b] < 1 || lengths[b] > T) return fail(PYCHAIN_HIP_EINVAL, "%s: sequence lengths must be in [1, %d]", who, T);
  return PYCHAIN_HIP_OK;
}
}  // namespace
}  // namespace pychain_hip

using namespace pychain_hip;

extern "C" long pychain_hip_cpu_calls(void) { return g_cpu_calls.load(); }

extern "C" int pychain_hip_cpu_den_forward_backward(
    const int32_t* ft, const int32_t* fi, const float* fp, const int32_t* bt, const int32_t* bi, const float* bp,
    const float* leaky, const float* initial, const float* final_, int graph_batch_stride,
    const float* nnet_output, int input_is_exp, const int64_t* seq_lengths, int B, int T, int D, int H, int K,
    float leaky_hmm_coefficient, float grad_scale, float* objf_per_seq, float* grad, int32_t* bad_count, int num_threads) {
  const char* who = "cpu_den_forward_backward";
  int rc = check_common(who, ft, fi, fp, bt, bi, bp, initial, final_, nnet_output, seq_lengths, objf_per_seq, grad, bad_count, B, T, D, H, K);
  if (rc != PYCHAIN_HIP_OK) return rc;
  if (!leaky) return fail(PYCHAIN_HIP_EINVAL, "%s: null leaky_probs", who);
  if (graph_batch_stride != 0 && graph_batch_stride != 1) return fail(PYCHAIN_HIP_EINVAL, "%s: graph_batch_stride must be 0 or 1", who);
  if (!(leaky_hmm_coefficient > 0.f && leaky_hmm_coefficient < 1.f))
    return fail(PYCHAIN_HIP_EINVAL, "%s: leaky_hmm_coefficient must be in (0,1), got %g", who, (double)leaky_hmm_coefficient);
  g_cpu_calls++;
  std::atomic<int> bad{0};
  for_each_sequence(B, num_threads, [&](int b) {
    const size_t g = (size_t)b * graph_batch_stride;
    const Csr fwd{ft + g * K * 3, fi + g * H * 2, fp + g * K}, bwd{bt + g * K * 3, bi + g * H * 2, bp + g * K};
    const bool ok = den_one(fwd, bwd, leaky + g * H, initial + g * H, final_ + g * H, nnet_output + (size_t)b * T * D, input_is_exp,
                            (int)seq_lengths[b], T, D, H, leaky_hmm_coefficient, grad_scale, objf_per_seq + b, grad + (size_t)b * T * D);
    if (!ok) bad++;
  });
  *bad_count = bad.load();
  return PYCHAIN_HIP_OK;
}

extern "C" int pychain_hip_cpu_num_forward_backward(
    const int32_t* ft, const int32_t* fi, const float* fp, const int32_t* bt, const int32_t* bi, const float* bp,
    const float* initial, const float* final_, int graph_batch_stride,
    const float* nnet_output, const int64_t* seq_lengths, int B, int T, int D, int H, int K, int grad_mode, float grad_scale,
    float* objf_per_seq, float* grad, int32_t* bad_count, int num_threads) {
  const char* who = "cpu_num_forward_backward";
  int rc = check_common(who, ft, fi, fp, bt, bi, bp, initial, final_, nnet_output, seq_lengths, objf_per_seq, grad, bad_count, B, T, D, H, K);
  if (rc != PYCHAIN_HIP_OK) return rc;
  if (graph_batch_stride != 0 && graph_batch_stride != 1) return fail(PYCHAIN_HIP_EINVAL, "%s: graph_batch_stride must be 0 or 1", who);
  if ((grad_mode & ~PYCHAIN_HIP_CPU_NO_CLAMP) < PYCHAIN_HIP_GRAD_LOG || (grad_mode & ~PYCHAIN_HIP_CPU_NO_CLAMP) > PYCHAIN_HIP_GRAD_ACCUM)
    return fail(PYCHAIN_HIP_EINVAL, "%s: unknown grad_mode %d", who, grad_mode);
  g_cpu_calls++;
  std::atomic<int> bad{0};
  for_each_sequence(B, num_threads, [&](int b) {
    const size_t g = (size_t)b * graph_batch_stride;
    const Csr fwd{ft + g * K * 3, fi + g * H * 2, fp + g * K}, bwd{bt + g * K * 3, bi + g * H * 2, bp + g * K};
    const bool ok = num_one(fwd, bwd, initial + g * H, final_ + g * H, nnet_output + (size_t)b * T * D, (int)seq_lengths[b], T, D, H,
                            grad_mode, grad_scale, objf_per_seq + b, grad + (size_t)b * T * D);
    if (!ok) bad++;
  });
  *bad_count = bad.load();
  return PYCHAIN_HIP_OK;
}
