/*
 * chain_oracle.c - CPU restatement of the reference LF-MMI forward-backward.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build or call it.  The shipped path (pychain_amd/) does not link it and has
 * no CPU fallback.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every entry point below
 * against tests/golden/*.npz, which were produced by running the reference's
 * own, unmodified CPU code (oracle/build_ref.py + tests/golden/make_golden.py).
 * (The reference repo itself ships no tests or golden vectors.)
 *
 * The loops keep the reference's serial structure (sequence -> state -> arc)
 * and float32 arithmetic so that (a) results agree to rounding and (b) its
 * single-thread run time is a fair stand-in for the reference CPU path when
 * oracle/_ref is unavailable (cpu_baseline.kind = "port").
 *
 * Each function cites the reference lines it follows; paths are relative to
 * /root/reference/pytorch_binding/src/.
 *
 * Build:  gcc -O2 -shared -fPIC -DREAL=float  -DSUF=_f32 chain_oracle.c -lm
 *         gcc -O2 -shared -fPIC -DREAL=double -DSUF=_f64 chain_oracle.c -lm
 * (the _f64 flavour evaluates the same equations in double as a second opinion).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
#ifndef SUF
#define SUF _f32
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

typedef REAL real;

/* base.h:12  kMinLogDiffFloat = log(FLT_EPSILON): LogAdd drops a term that far below the running sum.  The double flavour
 * is the second opinion "the same equations in exact arithmetic": its cut-off is log(DBL_EPSILON), below anything a sum of
 * doubles can register (with the float constant a state of 70 000 in-arcs loses 0.4 % of its mass to the cut-off). */
static const double kMinLogDiff = sizeof(real) == sizeof(float) ? -15.9423847198486328125 : -36.04365338911715;

/* base.h:14-32 LogAdd: returns the larger operand when the difference is below
 * log(FLT_EPSILON); (-inf,-inf) -> NaN diff -> comparison false -> returns x. */
static inline real log_add(real x, real y) {
  real diff;
  if (x < y) { diff = x - y; x = y; } else { diff = y - x; }
  if (diff >= (real)kMinLogDiff) {
    /* base.h:26 `x + std::log1p(std::exp(diff))` on floats: the float overloads, one rounding per operation */
    if (sizeof(real) == sizeof(float)) return x + (real)log1pf(expf((float)diff));
    return x + (real)log1p(exp((double)diff));
  }
  return x;
}

/* torch logsumexp over n values (chain-log-domain-computation.cc:158,177,198) */
static real logsumexp(const real *v, int n) {
  real m = -INFINITY;
  for (int i = 0; i < n; i++) if (v[i] > m) m = v[i];
  if (m == -INFINITY) return -INFINITY;
  /* (torch sums the exponentials in float with a vectorised partial-sum order of its own; an eight-lane float sum here
   * was tried against G6 and is no closer to the binary than this double sum: 8.0e-5 vs 9.2e-5 at T = 1500, 1.5e-5 vs
   * 6.9e-6 at T = 300 - rounding noise either way) */
  double s = 0.0;
  for (int i = 0; i < n; i++) s += exp((double)(v[i] - m));
  return m + (real)log(s);
}

/*
 * Denominator: probability domain, per-frame renormalisation, leaky-HMM.
 * Spec: chain-computation.h:109-156.  Code: chain-computation.cc.
 *
 * graph tensors: [G,K,3] [G,H,2] [G,K] ... with G = B (per-sequence graphs,
 * `graph_stride` = 1) or G = 1 (shared graph, `graph_stride` = 0).
 * exp_x [B,T,D] is the exponentiated, clamped network output (loss.py:30,43).
 * Sequences are independent (chain-computation.h:33-35); a sequence of length
 * L uses frames 0..L-1 (batch_sizes bookkeeping of :120,:254 restated per
 * sequence).  grad rows t >= L stay 0 (zeros_like at :58).
 * Returns 1 if the reference's t==0 invariants hold (ok_, :363-390).
 */
int FN(chain_oracle_den)(
    const int32_t *fwd_trans, const int32_t *fwd_idx, const float *fwd_probs,
    const int32_t *bwd_trans, const int32_t *bwd_idx, const float *bwd_probs,
    const float *leaky, const float *initial, const float *final_, int graph_stride,
    const float *exp_x, const int64_t *lengths, int B, int T, int D, int H, int K,
    float leaky_coef, real *objf_per_seq, real *grad /* [B,T,D] */) {
  int ok = 1;
  real *alpha = (real *)malloc(sizeof(real) * (size_t)(T + 1) * (H + 1));
  real *beta = (real *)malloc(sizeof(real) * 2 * (size_t)H);
  memset(grad, 0, sizeof(real) * (size_t)B * T * D);
  const real coef = (real)leaky_coef;
  for (int s = 0; s < B; s++) {
    const size_t g = (size_t)s * graph_stride;
    const int32_t *ft = fwd_trans + g * K * 3, *fi = fwd_idx + g * H * 2;
    const int32_t *bt = bwd_trans + g * K * 3, *bi = bwd_idx + g * H * 2;
    const float *fp = fwd_probs + g * K, *bp = bwd_probs + g * K;
    const float *lk = leaky + g * H, *in = initial + g * H, *fn = final_ + g * H;
    const float *x = exp_x + (size_t)s * T * D;
    real *gr = grad + (size_t)s * T * D;
    const int L = (int)lengths[s];
    const int W = H + 1;
    /* AlphaFirstFrame :92-95, AlphaSum(0) :97-110, AlphaDash(0) :178-194 */
    for (int t = 0; t <= L; t++) {
      real *a = alpha + (size_t)t * W;
      if (t == 0) {
        for (int h = 0; h < H; h++) a[h] = (real)in[h];
      } else {
        /* AlphaGeneralFrame CPU loop :150-174 */
        const real *pa = alpha + (size_t)(t - 1) * W;
        const float *probs = x + (size_t)(t - 1) * D;
        const real scale = (real)(1.0 / pa[H]);
        for (int h = 0; h < H; h++) {
          real tot = 0;
          for (int k = bi[2 * h]; k != bi[2 * h + 1]; k++)
            tot += pa[bt[3 * k]] * (real)bp[k] * (real)probs[bt[3 * k + 2]];
          a[h] = tot * scale;
        }
      }
      double sum = 0.0;                       /* AlphaSum: sum BEFORE the leaky term */
      for (int h = 0; h < H; h++) sum += a[h];
      a[H] = (real)sum;
      for (int h = 0; h < H; h++)             /* AlphaDash: addcmul_(tot, leaky, coef) */
        a[h] += coef * a[H] * (real)lk[h];
    }
    /* ComputeTotLogLike :209-230 */
    real *aL = alpha + (size_t)L * W;
    double fsum = 0.0;
    for (int h = 0; h < H; h++) fsum += (double)(aL[h] * (real)fn[h]);
    double logtot = log((double)(real)fsum);
    for (int t = 0; t < L; t++) logtot += log((double)alpha[(size_t)t * W + H]);
    objf_per_seq[s] = (real)logtot;
    /* BetaDashLastFrame :232-245 and Beta(L) :313-330 */
    real *bcur = beta + (size_t)(L % 2) * H;
    const real inv_tot = (real)1.0 / (real)fsum;
    for (int h = 0; h < H; h++) bcur[h] = inv_tot * (real)fn[h];
    for (int t = L; t >= 0; t--) {
      real *bd = beta + (size_t)(t % 2) * H;
      if (t < L) {
        /* BetaDashGeneralFrame CPU loop :289-309 */
        const real *nb = beta + (size_t)((t + 1) % 2) * H;
        const real *ad = alpha + (size_t)t * W;
        const float *probs = x + (size_t)t * D;
        real *drow = gr + (size_t)t * D;
        const real scale = (real)(1.0 / ad[H]);
        for (int h = 0; h < H; h++) {
          real totvf = 0;
          const real occ = ad[h] * scale;
          for (int k = fi[2 * h]; k != fi[2 * h + 1]; k++) {
            const int pdf = ft[3 * k + 2];
            const real vf = (real)fp[k] * nb[ft[3 * k + 1]] * (real)probs[pdf];
            totvf += vf;
            drow[pdf] += vf * occ;
          }
          bd[h] = totvf * scale;
        }
        if (t == 0) {                          /* BetaGeneralFrameDebug :345-391 */
          double ab = 0.0, ds = 0.0;
          for (int h = 0; h < H; h++) ab += (double)ad[h] * bd[h];
          for (int n = 0; n < D; n++) ds += drow[n];
          if (!(fabs(ab - 1.0) <= 0.05) || !(fabs(ds - 1.0) <= 0.05)) ok = 0;
        }
      }
      /* Beta(t): beta = beta' + coef * sum_i leaky_i beta'_i */
      double bs = 0.0;
      for (int h = 0; h < H; h++) bs += (double)(bd[h] * (real)lk[h]);
      const real add = coef * (real)bs;
      for (int h = 0; h < H; h++) bd[h] += add;
    }
  }
  free(alpha);
  free(beta);
  return ok;
}

/*
 * Numerator: log domain, no leaky-HMM.  chain-log-domain-computation.cc.
 * x [B,T,D] is the clamped (not exponentiated) network output (loss.py:30,72).
 * log_grad [B,T,D] is initialised to -inf (:57) and receives log occupancies;
 * the Python layer exponentiates it (loss.py:77).
 */
int FN(chain_oracle_num)(
    const int32_t *fwd_trans, const int32_t *fwd_idx, const float *fwd_probs,
    const int32_t *bwd_trans, const int32_t *bwd_idx, const float *bwd_probs,
    const float *initial, const float *final_, int graph_stride,
    const float *x_all, const int64_t *lengths, int B, int T, int D, int H, int K,
    real *objf_per_seq, real *log_grad /* [B,T,D] */) {
  int ok = 1;
  const int W = H + 1;
  real *alpha = (real *)malloc(sizeof(real) * (size_t)(T + 1) * W);
  real *beta = (real *)malloc(sizeof(real) * 2 * (size_t)H);
  real *tmp = (real *)malloc(sizeof(real) * (size_t)H);
  for (size_t i = 0; i < (size_t)B * T * D; i++) log_grad[i] = -INFINITY;
  for (int s = 0; s < B; s++) {
    const size_t g = (size_t)s * graph_stride;
    const int32_t *ft = fwd_trans + g * K * 3, *fi = fwd_idx + g * H * 2;
    const int32_t *bt = bwd_trans + g * K * 3, *bi = bwd_idx + g * H * 2;
    const float *fp = fwd_probs + g * K, *bp = bwd_probs + g * K;
    const float *in = initial + g * H, *fn = final_ + g * H;
    const float *x = x_all + (size_t)s * T * D;
    real *gr = log_grad + (size_t)s * T * D;
    const int L = (int)lengths[s];
    /* AlphaFirstFrame :84-90 (alpha-sum of frame 0 is 0 by fiat) */
    for (int h = 0; h < H; h++) alpha[h] = (real)in[h];
    alpha[H] = 0;
    /* AlphaGeneralFrame :123-158 */
    for (int t = 1; t <= L; t++) {
      real *a = alpha + (size_t)t * W;
      const real *pa = alpha + (size_t)(t - 1) * W;
      const float *probs = x + (size_t)(t - 1) * D;
      for (int h = 0; h < H; h++) {
        real v = -INFINITY;
        for (int k = bi[2 * h]; k != bi[2 * h + 1]; k++)
          v = log_add(v, pa[bt[3 * k]] + (real)bp[k] + (real)probs[bt[3 * k + 2]]);
        a[h] = v - pa[H];
      }
      a[H] = logsumexp(a, H);
    }
    /* ComputeTotLogLike :170-190 */
    real *aL = alpha + (size_t)L * W;
    for (int h = 0; h < H; h++) tmp[h] = aL[h] + (real)fn[h];
    const real last = logsumexp(tmp, H);
    double tot = (double)last;
    for (int t = 0; t < L; t++) {
      const real v = alpha[(size_t)t * W + H];
      if (v != -INFINITY) tot += (double)v;
    }
    objf_per_seq[s] = (real)tot;
    /* BetaLastFrame :192-202 */
    real *bl = beta + (size_t)(L % 2) * H;
    for (int h = 0; h < H; h++) bl[h] = (real)fn[h] - last;
    /* BetaGeneralFrame CPU loop :231-271 */
    for (int t = L - 1; t >= 0; t--) {
      const real *ta = alpha + (size_t)t * W;
      const real *nb = beta + (size_t)((t + 1) % 2) * H;
      real *tb = beta + (size_t)(t % 2) * H;
      const float *probs = x + (size_t)t * D;
      real *drow = gr + (size_t)t * D;
      const real inv_scale = ta[H];
      for (int h = 0; h < H; h++) {
        real totvf = -INFINITY;
        for (int k = fi[2 * h]; k != fi[2 * h + 1]; k++) {
          const int pdf = ft[3 * k + 2];
          const real vf = (real)fp[k] + nb[ft[3 * k + 1]] + (real)probs[pdf] - inv_scale;
          totvf = log_add(totvf, vf);
          drow[pdf] = log_add(drow[pdf], vf + ta[h]);
        }
        tb[h] = totvf;
      }
      if (t == 0) {                            /* BetaGeneralFrameDebug :283-304 */
        double ds = 0.0;
        for (int n = 0; n < D; n++) ds += exp((double)drow[n]);
        if (!(fabs(ds - 1.0) <= 0.05)) ok = 0;
      }
    }
  }
  free(alpha);
  free(beta);
  free(tmp);
  return ok;
}
