/* mu_oracle_c.c -- scalar C restatement of the dense NMF MU iteration and of beta_div.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/mu_oracle.py for the policy).  It restates, with plain loops and a
 * summation order independent of any BLAS:
 *   torchnmf/nmf.py:52-92    _double_backward_update  (all beta branches, l1/l2, gamma)
 *   torchnmf/nmf.py:122-131  beta == 1 closed-form denominators
 *   torchnmf/nmf.py:366-391  W half-step then H half-step (H uses the updated W)
 *   torchnmf/metrics.py:60-96 beta_div
 * Layout as the reference: V (N x C), W (C x R), H (N x R), row-major float32.  Accumulation is float32 for the
 * contractions (like the reference) and double for the loss (a scalar; only used at 1e-5 tolerance).
 * Pinned by tests/test_oracle_golden.py::test_c_oracle_* against the golden vectors the reference produced.
 *
 *   gcc -O2 -shared -fPIC -o libmu_oracle_c.so mu_oracle_c.c -lm
 */
#include <math.h>
#include <stdlib.h>

static const float EPSF = 1.1920928955078125e-07f; /* constants.py:3 */

static void grad_terms(float v, float s, float beta, float* gn, float* gp) { /* nmf.py:61-74 */
  if (beta == 2.f) { *gn = v; *gp = s; }
  else if (beta == 1.f) { *gn = v / (s + EPSF); *gp = 0.f; }
  else if (beta == 0.f) { float r = 1.f / (s + EPSF); *gp = r; *gn = r * r * v; }
  else { float se = s + EPSF; *gn = powf(se, beta - 2.f) * v; *gp = powf(se, beta - 1.f); }
}

static float mu_gamma(float beta) { /* nmf.py:341-346 */
  if (beta < 1.f) return 1.f / (2.f - beta);
  if (beta > 2.f) return 1.f / (beta - 1.f);
  return 1.f;
}

static float apply1(float th, float neg, float pos, int closed, float gamma, float l1, float l2) { /* nmf.py:78-92 */
  neg = (neg > 0.f ? neg : 0.f) + EPSF;
  if (!closed) pos = (pos > 0.f ? pos : 0.f) + EPSF;
  if (l1 > 0.f) pos += l1;
  if (l2 > 0.f) pos += l2 * th;
  float m = neg / pos;
  if (gamma != 1.f) m = powf(m, gamma);
  return th * m;
}

/* one half-step; owner (M x R) is updated in place from panel (K x R) and X addressed as x[m*xs_m + k*xs_k] */
static void half_step(const float* x, long xs_m, long xs_k, float* owner, int M, const float* panel, int K, int R,
                      float beta, float l1, float l2) {
  const float gamma = mu_gamma(beta);
  float* num = (float*)calloc((size_t)R, sizeof(float));
  float* den = (float*)calloc((size_t)R, sizeof(float));
  float* csum = (float*)calloc((size_t)R, sizeof(float));
  for (int k = 0; k < K; ++k)
    for (int r = 0; r < R; ++r) csum[r] += panel[(size_t)k * R + r]; /* nmf.py:122-131 */
  for (int m = 0; m < M; ++m) {
    for (int r = 0; r < R; ++r) num[r] = den[r] = 0.f;
    for (int k = 0; k < K; ++k) {
      float s = 0.f;
      for (int r = 0; r < R; ++r) s += owner[(size_t)m * R + r] * panel[(size_t)k * R + r];
      float gn, gp;
      grad_terms(x[m * xs_m + k * xs_k], s, beta, &gn, &gp);
      for (int r = 0; r < R; ++r) {
        num[r] += gn * panel[(size_t)k * R + r];
        den[r] += gp * panel[(size_t)k * R + r];
      }
    }
    for (int r = 0; r < R; ++r) {
      const int closed = beta == 1.f;
      owner[(size_t)m * R + r] =
          apply1(owner[(size_t)m * R + r], num[r], closed ? csum[r] : den[r], closed, gamma, l1, l2);
    }
  }
  free(num); free(den); free(csum);
}

/* n_iter MU iterations of NMF.fit's loop body (no loss evaluation / stopping) */
void mu_oracle_c_iterate(const float* V, float* W, float* H, int N, int C, int R, float beta, float l1, float l2,
                         int n_iter, int update_w, int update_h) {
  for (int it = 0; it < n_iter; ++it) {
    if (update_w) half_step(V, 1, C, W, C, H, N, R, beta, l1, l2);  /* X = V^T: x[c][n] = V[n*C + c] */
    if (update_h) half_step(V, C, 1, H, N, W, C, R, beta, l1, l2);
  }
}

/* metrics.py:60-96 on the reconstruction H W^T */
double mu_oracle_c_beta_div(const float* V, const float* W, const float* H, int N, int C, int R, float beta) {
  double t_kl = 0, sx = 0, sy = 0, acc = 0;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      for (int r = 0; r < R; ++r) s += H[(size_t)n * R + r] * W[(size_t)c * R + r];
      const double x = s, y = V[(size_t)n * C + c];
      if (beta == 2.f) acc += 0.5 * (x - y) * (x - y);
      else if (beta == 1.f) { t_kl += y * (log(y + EPSF) - log(x + EPSF)); sx += x; sy += y; }
      else if (beta == 0.f) acc += (y + EPSF) / (x + EPSF) - log(y + EPSF) + log(x + EPSF) - 1.0;
      else {
        const double xe = x + EPSF, yb = beta < 0.f ? y + EPSF : y, bm = beta - 1.0;
        acc += (pow(yb, beta) + bm * pow(xe, beta) - beta * yb * pow(xe, bm)) / (beta * bm);
      }
    }
  return beta == 1.f ? t_kl - sy + sx : acc;
}
