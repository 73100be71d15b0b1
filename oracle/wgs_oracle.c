/*
 * oracle/wgs_oracle.c -- CPU restatement of WGSassign's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (wgsassign_amd/) never does.  Parity status: PINNED -- every function here is
 * checked bit-for-bit against outputs of the real reference (built from /root/reference in the
 * build container) through the fixtures under tests/golden/ (tests/test_oracle_golden.py).
 *
 * The arithmetic follows the C that Cython 3.2.9 emits for the cited .pyx lines: bare integer
 * and float literals become C doubles, so every product/divide/log is evaluated in double and
 * rounded to float32 only on assignment to the `cdef float` locals.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fPIC -shared (see oracle/Makefile).  No
 * -ffast-math, no FMA contraction: the reference's x86-64 build has neither.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* emMAF_cy.pyx:10-23  emMAF_update(L, f, t): one EM step of every SNP's frequency, in place.
 * L is (m, 2n) float32 C-contiguous, row s = g0_0 g1_0 g0_1 g1_1 ...  */
void orc_emmaf_update(const float *L, int64_t m, int64_t n, float *f, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t s = 0; s < m; ++s) {
        const float *row = L + s * 2 * n;
        const float fs = f[s];                        /* f[s] is read n times, never written in the loop */
        float tmp = 0.0f;                             /* emMAF_cy.pyx:17 */
        for (int64_t i = 0; i < n; ++i) {             /* emMAF_cy.pyx:18-22 */
            float p0 = (float)(((double)row[2 * i] * (1.0 - (double)fs)) * (1.0 - (double)fs));
            float p1 = (float)((((double)row[2 * i + 1] * 2.0) * (double)fs) * (1.0 - (double)fs));
            float p2 = (float)((((1.0 - (double)row[2 * i]) - (double)row[2 * i + 1]) * (double)fs) * (double)fs);
            tmp = (float)((double)tmp + ((double)p1 + 2.0 * (double)p2) / (2.0 * (double)((p0 + p1) + p2)));
        }
        f[s] = tmp / (float)n;                        /* emMAF_cy.pyx:23 */
    }
}

/* emMAF_cy.pyx:26-33  rmse1d(v1, v2): serial float32 accumulation, sqrt in double. */
double orc_rmse1d(const float *v1, const float *v2, int64_t m)
{
    float res = 0.0f;
    for (int64_t i = 0; i < m; ++i)
        res = res + (v1[i] - v2[i]) * (v1[i] - v2[i]);
    res = res / (float)m;
    return sqrt((double)res);
}

/* glassy_cy.pyx:12-21  loglike(L, A, loglike_vec, t, i, k): per-site log-likelihood of
 * individual i under population k's allele frequencies, ACCUMULATED into vec. */
void orc_loglike(const float *L, int64_t m, int64_t n, const float *A, int64_t K,
                 float *vec, int64_t i, int64_t k, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t s = 0; s < m; ++s) {
        const float g0 = L[s * 2 * n + 2 * i], g1 = L[s * 2 * n + 2 * i + 1];
        const float a = A[s * K + k];
        float like0 = (float)(((double)g0 * (1.0 - (double)a)) * (1.0 - (double)a));
        float like1 = (float)((((double)g1 * 2.0) * (1.0 - (double)a)) * (double)a);
        float like2 = (float)((((1.0 - (double)g0) - (double)g1) * (double)a) * (double)a);
        vec[s] = (float)((double)vec[s] + log((double)((like0 + like1) + like2)));
    }
}

/* Column gather of WGSassign.py:227-233 / glassy.py:69-77: keep individuals idx[0..n_sub) in
 * the order given (the reference sorts the column indices, i.e. file order). */
void orc_gather_columns(const float *L, int64_t m, int64_t n, const int64_t *idx, int64_t n_sub,
                        float *out, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t s = 0; s < m; ++s) {
        const float *row = L + s * 2 * n;
        float *o = out + s * 2 * n_sub;
        for (int64_t j = 0; j < n_sub; ++j) {
            o[2 * j] = row[2 * idx[j]];
            o[2 * j + 1] = row[2 * idx[j] + 1];
        }
    }
}

/* emMAF.py:15-27  the EM driver around the two kernels above.  Returns the 1-based iteration at
 * which `diff < tole` fired, or 0 when `iter` was exhausted (the reference prints nothing then). */
int orc_emmaf(const float *L, int64_t m, int64_t n, int iter, double tole, float *f, float *f_prev,
              int threads)
{
    for (int64_t s = 0; s < m; ++s) f[s] = 0.25f, f_prev[s] = 0.25f;   /* emMAF.py:17-19 */
    for (int it = 0; it < iter; ++it) {
        orc_emmaf_update(L, m, n, f, threads);                           /* emMAF.py:21 */
        double diff = orc_rmse1d(f, f_prev, m);                          /* emMAF.py:22 */
        if (diff < tole) return it + 1;                                  /* emMAF.py:23-25 */
        memcpy(f_prev, f, (size_t)m * sizeof(float));                    /* emMAF.py:26 */
    }
    return 0;
}

/* glassy_cy.pyx:21 for a zero-initialised vector: (float)log((double)x), libm's double log. */
void orc_log_f32(const float *x, float *out, int64_t n, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = (float)log((double)x[i]);
}

/* fisher_cy.pyx:12-30 / 41-56: the observed-information term of one (SNP, individual) under
 * frequency th.  All locals of the reference are `cdef float`; bare literals are doubles. */
static inline float fisher_term(float g0, float g1, float th)
{
    const float g2 = (float)((1.0 - (double)g0) - (double)g1);
    const float u = (float)(((((double)g0 * (1.0 - (double)th)) * (1.0 - (double)th)) +
                             ((((double)g1 * 2.0) * (double)th) * (1.0 - (double)th))) +
                            (double)((g2 * th) * th));      /* g2 and th are both float: float products */
    const float n1 = (float)(2.0 * ((double)(g0 + g2) - (2.0 * (double)g1)));
    const float n2 = (float)((double)(th * n1) + (2.0 * (double)(g1 - g0)));
    return (float)(-1.0 * (double)((n1 / u) - ((n2 / u) * (n2 / u))));
}

/* fisher_cy.pyx:12-30 fisher_obs(L_pop, A, t, i, n, f_pop) then :32-39 ne_obs(...): per SNP the
 * serial float32 sum of the terms over the n individuals of L_pop, accumulated into f_pop; and
 * n_tilde = 0.5 * f_pop * a * (1 - a) accumulated into ne_pop. */
void orc_fisher_obs(const float *L_pop, int64_t m, int64_t n, const float *A, int64_t K, int64_t col,
                    float *f_pop, float *ne_pop, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t s = 0; s < m; ++s) {
        const float th = A[s * K + col];
        float term_sum = 0.0f;
        for (int64_t r = 0; r < n; ++r) term_sum = term_sum + fisher_term(L_pop[s * 2 * n + 2 * r], L_pop[s * 2 * n + 2 * r + 1], th);
        f_pop[s] = f_pop[s] + term_sum;
        const float n_tilde = (float)(((0.5 * (double)f_pop[s]) * (double)th) * (1.0 - (double)th));
        ne_pop[s] = ne_pop[s] + n_tilde;
    }
}

/* fisher_cy.pyx:41-56 fisher_obs_ind then :58-65 ne_obs_ind for individual i under column col. */
void orc_fisher_obs_ind(const float *L, int64_t m, int64_t n, const float *A, int64_t K, int64_t i, int64_t col,
                        float *f_ind, float *ne_ind, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t s = 0; s < m; ++s) {
        const float th = A[s * K + col];
        f_ind[s] = f_ind[s] + fisher_term(L[s * 2 * n + 2 * i], L[s * 2 * n + 2 * i + 1], th);
        const float n_tilde = (float)(((0.5 * (double)f_ind[s]) * (double)th) * (1.0 - (double)th));
        ne_ind[s] = ne_ind[s] + n_tilde;
    }
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
