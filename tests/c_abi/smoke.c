/* A plain-C caller of libwgsassign_hip.so: the boundary is a C ABI, usable without Python.
 * Runs emMAF_update (thin mirror), a device-resident two-population EM batch and an assignment
 * sweep on a tiny synthetic matrix and checks a few invariants.  Exit code 0 = OK.
 *   gcc -std=c99 -I include tests/c_abi/smoke.c -o smoke -L wgsassign_amd -lwgsassign_hip -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "wgsassign_hip.h"

#define CHECK(call)                                                          \
    do {                                                                     \
        if ((call) != 0) {                                                   \
            fprintf(stderr, "%s failed: %s\n", #call, wgs_last_error());     \
            return 1;                                                        \
        }                                                                    \
    } while (0)

/* a one-rank "transport" for wgs_comm_create_host: the sum over one rank is the buffer itself */
static int calls = 0;
static int one_rank_sum(double *buf, int64_t n, void *user)
{
    (void)buf;
    (void)n;
    (void)user;
    ++calls;
    return 0;
}

int main(void)
{
    enum { M = 1000, N = 6, K = 2 };
    static float L[M][2 * N], f[M], A[M][K];
    unsigned s = 12345u;
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            s = s * 1664525u + 1013904223u;
            const float g0 = (float)((s >> 8) % 1000) / 1000.0f;
            s = s * 1664525u + 1013904223u;
            const float g1 = (1.0f - g0) * (float)((s >> 8) % 1000) / 1000.0f;
            L[i][2 * j] = g0;
            L[i][2 * j + 1] = g1;
        }
    wgs_ctx *ctx = NULL;
    CHECK(wgs_ctx_create(0, &ctx));

    /* thin mirror: one EM update of all individuals as one population */
    for (int i = 0; i < M; ++i) f[i] = 0.25f;
    CHECK(wgs_emmaf_update(ctx, &L[0][0], M, N, f, WGS_MODE_EXACT));
    for (int i = 0; i < M; ++i)
        if (!(f[i] >= 0.0f && f[i] <= 1.0f)) {
            fprintf(stderr, "frequency %d out of range: %g\n", i, f[i]);
            return 1;
        }

    /* device-resident: two populations of three individuals, ten EM updates */
    const int32_t group_of[N] = {0, 0, 0, 1, 1, 1}, fit_group[K] = {0, 1};
    wgs_beagle *b = NULL;
    wgs_em *em = NULL;
    CHECK(wgs_beagle_create(ctx, M, N, group_of, K, 0, &b));
    CHECK(wgs_beagle_upload_rows(b, &L[0][0], 0, M));
    CHECK(wgs_em_create(b, K, fit_group, NULL, WGS_MODE_EXACT, &em));
    double ssq[K], first[K] = {0, 0};
    for (int it = 0; it < 10; ++it) {
        CHECK(wgs_em_step(em, ssq));
        if (it == 0) first[0] = ssq[0], first[1] = ssq[1];
    }
    if (!(ssq[0] < first[0] && ssq[1] < first[1])) {
        fprintf(stderr, "EM did not contract: %g -> %g, %g -> %g\n", first[0], ssq[0], first[1], ssq[1]);
        return 1;
    }
    float carry = 0.0f;
    CHECK(wgs_em_rmse_chain(em, 0, 0.0f, &carry));
    if (fabs((double)carry - ssq[0]) > 1e-3 * ssq[0]) {
        fprintf(stderr, "serial chain %g vs float64 sum %g\n", carry, ssq[0]);
        return 1;
    }

    /* assignment of all individuals to both populations */
    wgs_afset *af = NULL;
    CHECK(wgs_afset_create(ctx, M, K, &af));
    for (int k = 0; k < K; ++k) {
        CHECK(wgs_em_clamp(em, k, 1.0f / 8.0f, 7.0f / 8.0f));
        CHECK(wgs_afset_set_column_from_em(af, k, em, k));
    }
    CHECK(wgs_ctx_sync(ctx));
    CHECK(wgs_afset_download(af, &A[0][0]));
    double out[N * K] = {0};
    CHECK(wgs_assign(b, af, NULL, WGS_MODE_EXACT, out));
    for (int i = 0; i < N * K; ++i)
        if (!(out[i] < 0.0) || !isfinite(out[i])) {
            fprintf(stderr, "log-likelihood %d not negative/finite: %g\n", i, out[i]);
            return 1;
        }
    /* the whole of emMAF.emMAF for both populations in one call: fresh batch, run to convergence */
    wgs_em *em2 = NULL;
    int32_t iters[K] = {0, 0};
    CHECK(wgs_em_create(b, K, fit_group, NULL, WGS_MODE_EXACT, &em2));
    CHECK(wgs_em_fit(em2, 200, 1e-4, M, NULL, 0.0, iters));
    if (iters[0] < 2 || iters[0] > 200 || iters[1] < 2 || iters[1] > 200) {
        fprintf(stderr, "wgs_em_fit: implausible iteration counts %d %d\n", iters[0], iters[1]);
        return 1;
    }
    static float fa[M], fb[M];
    CHECK(wgs_em_get_f(em2, 0, fa));
    CHECK(wgs_em_get_f_range(em2, 0, 1, 0, M, fb));
    double d2 = 0.0, rm = 0.0;
    for (int i = 0; i < M; ++i) d2 += (double)(fa[i] - fb[i]) * (double)(fa[i] - fb[i]);
    CHECK(wgs_rmse1d(ctx, fa, fb, M, &rm));
    if (!(rm < 1e-4) || fabs(rm - sqrt(d2 / M)) > 1e-3 * rm) {
        fprintf(stderr, "wgs_em_fit stopped at rmse %g (float64 estimate %g)\n", rm, sqrt(d2 / M));
        return 1;
    }
    /* the same fit through a communicator over the caller's own all-reduce function: same iteration counts */
    wgs_comm *hc = NULL;
    wgs_em *em3 = NULL;
    int32_t iters3[K] = {0, 0};
    CHECK(wgs_comm_create_host(ctx, 0, 1, one_rank_sum, NULL, &hc));
    CHECK(wgs_em_create(b, K, fit_group, NULL, WGS_MODE_EXACT, &em3));
    CHECK(wgs_em_fit(em3, 200, 1e-4, M, hc, 0.0, iters3));
    if (iters3[0] != iters[0] || iters3[1] != iters[1] || calls < iters[0]) {
        fprintf(stderr, "wgs_em_fit over a host communicator: iterations %d %d (expected %d %d), %d all-reduce calls\n", iters3[0], iters3[1],
                iters[0], iters[1], calls);
        return 1;
    }
    wgs_em_destroy(em3);
    wgs_comm_destroy(hc);
    /* glassy.loo in one call: N re-fits, sticky columns, scores and exact partition sums */
    double loo[N * K];
    float parts[N * 2 * K];
    int32_t loo_iters[N];
    for (int k = 0; k < K; ++k) {
        CHECK(wgs_em_clamp(em2, k, 1.0f / 8.0f, 7.0f / 8.0f));
        CHECK(wgs_afset_set_column_from_em(af, k, em2, k));
    }
    CHECK(wgs_loo(b, NULL, af, 200, 1e-4, M, NULL, 2, 0, WGS_MODE_EXACT, WGS_MODE_EXACT, loo, parts, loo_iters));
    for (int i = 0; i < N; ++i)
        for (int k = 0; k < K; ++k) {
            const double tot = (double)parts[(i * 2 + 0) * K + k] + (double)parts[(i * 2 + 1) * K + k];
            if (!(loo[i * K + k] < 0.0) || fabs(tot - loo[i * K + k]) > 1e-4 * fabs(tot) || loo_iters[i] < 1) {
                fprintf(stderr, "wgs_loo: individual %d population %d: sum %g, partitions %g, iterations %d\n", i, k,
                        loo[i * K + k], tot, loo_iters[i]);
                return 1;
            }
        }
    printf("C ABI smoke OK: f[0]=%.6f ssq=%.6g logl[0][0]=%.4f fit iterations %d/%d loo[0][0]=%.4f\n", f[0], ssq[0], out[0],
           iters[0], iters[1], loo[0]);
    wgs_em_destroy(em2);
    wgs_afset_destroy(af);
    wgs_em_destroy(em);
    wgs_beagle_destroy(b);
    wgs_ctx_destroy(ctx);
    return 0;
}
