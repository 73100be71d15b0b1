// The EM fits of the C ABI (include/wgsassign_hip.h: wgs_em_*): emMAF.py:15-27 as ONE call per batch of fits (wgs_em_fit: iterations
// enqueued ahead of the host, decisions on the device), the step-by-step twin, the exact convergence chain, and the policy that decides
// which sweep kernel a batch takes (direct, grouped leave-one-out, class-coded -- and when the codes are worth building).
// Host-side orchestration only; the arithmetic is in em_kernels.hip.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>

#include "common.h"
#include "em_state.h"

extern "C" {

/* ------------------------------------------------------------------ EM */

// (struct wgs_em: em_state.h)


void wgs_em_destroy(wgs_em *em)
{
    if (!em || !wgs_live_remove(em)) return;          // (destroyed already, e.g. together with its matrix)
    (void)hipSetDevice(em->b->ctx->device);
    for (int i = 0; i < 3; ++i)
        if (em->fbuf[i]) (void)hipFree(em->fbuf[i]);
    if (em->d_part_b) (void)hipFree(em->d_part_b);
    for (int i = 0; i < 2; ++i)
        if (em->h_ssq[i]) (void)hipHostFree(em->h_ssq[i]);
    if (em->d_descs) (void)hipFree(em->d_descs);
    if (em->h_descs) (void)hipHostFree(em->h_descs);
    if (em->d_groups) (void)hipFree(em->d_groups);
    if (em->h_groups) (void)hipHostFree(em->h_groups);
    for (int i = 0; i < 2; ++i) {
        if (em->d_groups2[i]) (void)hipFree(em->d_groups2[i]);
        if (em->h_groups2[i]) (void)hipHostFree(em->h_groups2[i]);
    }
    if (em->d_ssq) (void)hipFree(em->d_ssq);
    if (em->d_part) (void)hipFree(em->d_part);
    if (em->d_part2) (void)hipFree(em->d_part2);
    if (em->d_carry) (void)hipFree(em->d_carry);
    if (em->d_chain_work) (void)hipFree(em->d_chain_work);
    if (em->ev0) (void)hipEventDestroy(em->ev0);
    if (em->ev1) (void)hipEventDestroy(em->ev1);
    for (hipEvent_t e : em->ev_sw) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) {
        if (em->d_descs2[i]) (void)hipFree(em->d_descs2[i]);
        if (em->h_descs2[i]) (void)hipHostFree(em->h_descs2[i]);
        if (em->h_state[i]) (void)hipHostFree(em->h_state[i]);
        if (em->ev_it[i]) (void)hipEventDestroy(em->ev_it[i]);
    }
    for (void *p : {(void *)em->d_state, (void *)em->d_ssq2, (void *)em->d_jobs, (void *)em->d_chain_out, em->d_chain_batch})
        if (p) (void)hipFree(p);
    for (void *p : {(void *)em->h_jobs, (void *)em->h_chain_out, (void *)em->h_setstate})
        if (p) (void)hipHostFree(p);
    delete em;
}

int wgs_em_create(wgs_beagle *b, int32_t n_fits, const int32_t *fit_group, const int32_t *fit_skip, int mode, wgs_em **out)
{
    WGS_REQUIRE(b && fit_group && out, "null argument");
    WGS_REQUIRE(n_fits > 0, "n_fits must be positive");
    WGS_REQUIRE(mode == WGS_MODE_EXACT || mode == WGS_MODE_FAST, "unknown mode %d", mode);
    HIP_TRY(hipSetDevice(b->ctx->device));
    wgs_em *em = new wgs_em();
    wgs_live_add(em, WGS_LIVE_EM, b);
    auto guard = on_failure([&] { wgs_em_destroy(em); });
    em->b = b;
    em->n_fits = n_fits;
    em->mode = mode;
    em->group.resize(n_fits);
    em->skip_local.resize(n_fits);
    em->n_eff.resize(n_fits);
    em->cur.assign(n_fits, 0);
    em->prev.assign(n_fits, 1);
    em->fuse_used.assign(n_fits, 1);
    em->pend_cur.assign(n_fits, 0);
    em->pend_prev.assign(n_fits, 1);
    em->active.assign(n_fits, 1);
    for (int j = 0; j < n_fits; ++j) {
        const int g = fit_group[j];
        if (g < 0 || g >= b->n_groups || b->slabs[g].ncols == 0) {
            wgs_set_error("fit %d: group %d is out of range or empty", j, g);
            return 2;
        }
        int skip = -1;
        if (fit_skip && fit_skip[j] >= 0) {
            const int i = fit_skip[j];
            if (i >= b->n || b->group_of[i] != g) {
                wgs_set_error("fit %d: left-out individual %d does not belong to group %d", j, i, g);
                return 2;
            }
            skip = b->col_of[i];
        }
        em->group[j] = g;
        em->skip_local[j] = skip;
        em->n_eff[j] = b->slabs[g].ncols - (skip >= 0 ? 1 : 0);
    }
    const size_t fbytes = (size_t)n_fits * b->m * sizeof(float);
    for (int i = 0; i < 2; ++i) {
        if (wgs_malloc(&em->fbuf[i], fbytes) != hipSuccess) {
            wgs_set_error("hipMalloc of %zu bytes for EM frequencies failed", fbytes);
            return 1;
        }
    }
    HIP_TRY(wgs_malloc(&em->d_descs, sizeof(FitDesc) * n_fits));
    HIP_TRY(wgs_malloc(&em->d_ssq, sizeof(double) * n_fits));
    HIP_TRY(wgs_malloc(&em->d_part, sizeof(double) * (size_t)n_fits * wgs_ntiles(b->m)));
    HIP_TRY(wgs_malloc(&em->d_part2, sizeof(double) * (size_t)n_fits * ssq_reduce_chunks()));
    HIP_TRY(wgs_malloc(&em->d_carry, 2 * sizeof(float)));
    HIP_TRY(wgs_malloc(&em->d_chain_work, rmse_chain_workspace_bytes(b->m)));
    HIP_TRY(hipEventCreate(&em->ev0));
    HIP_TRY(hipEventCreate(&em->ev1));
    HIP_TRY(hipHostMalloc(&em->h_descs, sizeof(FitDesc) * n_fits, hipHostMallocDefault));
    HIP_TRY(wgs_malloc(&em->d_groups, sizeof(int32_t) * 2 * n_fits));
    HIP_TRY(hipHostMalloc(&em->h_groups, sizeof(int32_t) * 2 * n_fits, hipHostMallocDefault));
    if (launch_fill(b->ctx, em->fbuf[0], (int64_t)n_fits * b->m, 0.25f)) return 1;   // emMAF.py:17-18
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    guard.dismiss();
    *out = em;
    return 0;
}


/* Whether building the class codes pays for the EM sweeps still to come (codes.hip builds them in one pass over the matrix):
 *   the encode pass costs wgs_codes_build_ms_estimate (the matrix's bytes at ~1.8 TB/s, more with larger hash tables);
 *   a coded sweep saves a share of the direct sweep (the slabs' bytes at ~6 TB/s) that depends on how many of a population's
 *   individuals share a class: min(0.6, 1.0 - 2.2 x classes / individuals) (em_codes_model has the measurements; round 4's
 *   0.92 - 2.72 x was fitted before a coded sweep ran two iterations);
 *   sweeps to come: what the caller knows -- wgs_em_fit its iteration limit, of which a fit rarely uses more than ~14 (the
 *   reference's default tolerance: 11-17 iterations on every data set here); a step-by-step caller nothing, so there a matrix
 *   that has been swept directly three times is taken to be in a long run.
 * A first estimate assumes fixed-error low-depth data (no sample pass: small matrices are turned away for free); when that says yes
 * the sample pass (~0.4 ms, once per matrix) supplies the matrix's own classes per slab and table size.
 * WGSASSIGN_EM_CODES_SWEEPS=k replaces the model by "k or more sweeps ahead" (0: always; tests). */
// Share of a leave-one-out sweep's time the codes save (<= 0: none).  Such batches (many fits per slab) are bound by instruction
// issue, not by memory: em_sweep_group_kernel spends 25.9 vector instructions per (fit, SNP, individual) term; through the codes
// (em_coded_group_kernel) a (fit, SNP) costs one quotient (29 instructions) per table row of its tile -- the richest SNP of the
// tile, ~1.7 x the mean classes per (slab, SNP) -- 4 per individual, and the equivalent of ~250 more in issue slots it leaves open
// (measured at 2M x 500, K=8, 62 individuals per slab and 12.7 classes: 458 ms of sweeps against 655).
static double loo_codes_saving(const wgs_codes_plan *P, double cols)
{
    if (!P || P->state <= 0 || P->lrows == 0) return 0.0;
    return 1.0 - (1.7 * P->mean_l * 29.0 + 4.0 * cols + 250.0) / (25.9 * std::max(1.0, cols));
}

// The model's own numbers for a batch of fits that sweep `swept` bytes of slabs with `cols` individuals per slab on average: what a
// sweep over the float32 slabs takes, the share of it a coded sweep saves, what the encode pass costs.  em_codes_pay decides with
// them and wgs_codes_model hands them out, so that what is reported beside a measurement is what decided.
struct EmCodesModel {
    double direct_ms = 0.0, saves = 0.0, build_ms = 0.0;
    bool sampled = false;          // saves / build_ms come from the matrix's own sample pass (else from the fixed-error typical)
    bool codable = false;          // the sample pass found the matrix worth coding, with the slabs' own numbering
};
static EmCodesModel em_codes_model(wgs_beagle *b, double swept, double cols, bool shared, bool sample)
{
    EmCodesModel M;
    auto saves = [&](double classes_per_slab, int lrows) { return wgs_em_codes_saving(classes_per_slab, cols, lrows); };
    if (shared) {
        constexpr double LOO_MS_PER_TERM = 7.6e-10;          // em_sweep_group_kernel, per (fit, SNP, individual)
        M.direct_ms = swept / 8.0 * LOO_MS_PER_TERM;
        const wgs_codes_plan *P = wgs_beagle_codes_plan(b);
        M.sampled = true;
        M.codable = P && P->state > 0 && P->lrows > 0;
        if (!M.codable) return M;
        M.saves = std::max(0.0, loo_codes_saving(P, cols));
        M.build_ms = wgs_codes_build_ms_estimate(b, P->slots);
        return M;
    }
    M.direct_ms = swept / 6.0e9;
    // fixed-error 2x data shows ~4.6 * cols^0.25 classes per (slab, SNP): 14.7 at 100, 12.7 at 62, 10.5 at 40
    M.saves = saves(4.6 * pow(std::max(1.0, cols), 0.25), 24);
    M.build_ms = wgs_codes_build_ms_estimate(b, 64);
    M.codable = true;
    if (!sample) return M;
    const wgs_codes_plan *P = wgs_beagle_codes_plan(b);
    M.sampled = true;
    M.codable = P && P->state > 0 && P->lrows > 0;
    if (!M.codable) {
        M.saves = 0.0;
        return M;
    }
    M.saves = saves(P->mean_l, P->lrows);
    M.build_ms = wgs_codes_build_ms_estimate(b, P->slots);
    return M;
}

static bool em_codes_pay(const wgs_em *em, const std::vector<int32_t> &order, int fewest_cols, int sweeps_ahead, bool shared)
{
    wgs_beagle *b = em->b;
    if (const char *sw = getenv("WGSASSIGN_EM_CODES_SWEEPS")) return sweeps_ahead >= atoi(sw) || b->direct_sweeps >= 3;
    // a fit uses ~14 iterations: what this one has done already (the codes' memory may arrive in the middle of it) no longer counts --
    // but a fit that has gone past 14 is taken to need a few more
    double ahead = std::min<double>(sweeps_ahead, std::max(3, 14 - em->fit_iterations));
    if (sweeps_ahead <= 0 && b->direct_sweeps >= 3) ahead = 12;
    double swept = 0.0, cols = 0.0;
    for (int j : order) {
        swept += 8.0 * (double)b->slabs[em->group[j]].ncols * (double)b->m;
        cols += (double)b->slabs[em->group[j]].ncols;
    }
    cols /= (double)std::max<size_t>(1, order.size());
    (void)fewest_cols;
    if (!shared) {      // a first estimate with the classes typical of fixed-error data turns small matrices away without a sample pass
        const EmCodesModel T = em_codes_model(b, swept, cols, false, false);
        if (ahead * T.saves * T.direct_ms <= T.build_ms) return false;
    }
    const EmCodesModel M = em_codes_model(b, swept, cols, shared, true);
    return M.codable && ahead * M.saves * M.direct_ms > M.build_ms;
}

/* The cost models' own predictions for this matrix (so that a caller can print them beside what it measures: bench.py,
 * tests/test_gpu_codes.py): the fits of --get_reference_af (one per population slab) and a --get_pop_like sweep over K populations.
 * out[0..11]: EM sweep over the float32 slabs, ms | share of it a coded sweep saves | encode pass incl. the slabs' numbering, ms |
 * sweeps the decision counts (14) | 1 = the model builds the codes for such a fit | scoring sweep over the float32 slabs, ms | share
 * of it the coded sweep costs | encode pass for scoring alone, ms | 1 = the model builds them for scoring | 1 = predictions from the
 * matrix's own sample pass | classes per (slab, SNP) in the sample | classes per SNP in the sample.  Runs the sample pass (~0.4 ms)
 * unless the first estimate already turns the matrix away. */
int wgs_codes_model(wgs_beagle *b, int32_t K_score, double *out)
{
    WGS_REQUIRE(b && out, "null argument");
    HIP_TRY(hipSetDevice(b->ctx->device));
    double swept = 0.0, cols = 0.0;
    int groups = 0;
    for (int g = 0; g < b->n_groups; ++g) {
        if (b->slabs[g].ncols == 0) continue;
        swept += 8.0 * (double)b->slabs[g].ncols * (double)b->m;
        cols += (double)b->slabs[g].ncols;
        ++groups;
    }
    cols /= std::max(1, groups);
    const EmCodesModel T = em_codes_model(b, swept, cols, false, false);
    const bool first_ok = 14.0 * T.saves * T.direct_ms > T.build_ms;
    const EmCodesModel M = first_ok ? em_codes_model(b, swept, cols, false, true) : T;
    for (int i = 0; i < 12; ++i) out[i] = 0.0;
    out[0] = M.direct_ms;
    out[1] = M.saves;
    out[2] = M.build_ms;
    out[3] = 14.0;
    out[4] = (first_ok && M.codable && 14.0 * M.saves * M.direct_ms > M.build_ms) ? 1.0 : 0.0;
    out[9] = M.sampled ? 1.0 : 0.0;
    if (K_score > 0) {
        if (wgs_codes_scoring_model(b, K_score, &out[5], &out[6], &out[7])) out[8] = out[5] * (1.0 - out[6]) > out[7] ? 1.0 : 0.0;
        out[9] = 1.0;
    }
    if (b->plan.state != 0) {
        out[10] = b->plan.mean_l;
        out[11] = b->plan.mean_g;
    }
    return 0;
}

// The buffers of two iterations per sweep: a third frequency buffer and a second set of partial sums.  false: no memory for them.
static bool em_fuse_buffers(wgs_em *em)
{
    if (em->fbuf[2]) return true;
    const size_t fbytes = (size_t)em->n_fits * em->b->m * sizeof(float);
    const int64_t ntiles = wgs_ntiles(em->b->m);
    if (wgs_malloc(&em->fbuf[2], fbytes) != hipSuccess || wgs_malloc(&em->d_part_b, sizeof(double) * (size_t)em->n_fits * ntiles) != hipSuccess) {
        (void)hipGetLastError();
        if (em->fbuf[2]) (void)hipFree(em->fbuf[2]);
        em->fbuf[2] = nullptr;
        return false;
    }
    return true;
}

/* Enqueue one sweep (+ the fixed-order reduction of its sums) for the fits in `list`: descriptors into the pinned
 * array H and from there to D.  Fits of different populations stream their slabs once (nontemporal loads); when
 * several fits share a slab (leave-one-out batches) they are ordered by slab and swept in groups of up to
 * em_fits_per_group() per wavefront (group table Hg -> Dg), which share the tile's loads and conversions.
 * ssq_base[j] receives fit j's sum; state_base (device, may be NULL) holds the fit states a sweep honours. */
static int em_enqueue_sweep(wgs_em *em, const std::vector<int32_t> &list, FitDesc *H, FitDesc *D, int32_t *Hg, int32_t *Dg,
                            double *ssq_base, int32_t *state_base, hipEvent_t ev0, hipEvent_t ev1, int sweeps_ahead,
                            const std::vector<int32_t> *may_fuse = nullptr, double *ssq_base_b = nullptr, bool fuse_agreed = true,
                            int *can_fuse_out = nullptr)
{
    wgs_ctx *ctx = em->b->ctx;
    const int64_t ntiles = wgs_ntiles(em->b->m);
    std::vector<int32_t> order(list);
    std::vector<char> seen(em->b->n_groups, 0);
    bool shared = false;
    for (int j : order) {
        shared = shared || seen[em->group[j]];
        seen[em->group[j]] = 1;
    }
    if (shared) std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return em->group[x] < em->group[y]; });
    // exact mode on a coded matrix: the sweep through the class codes (same frequencies, bit for bit)
    // -- leave-one-out batches (several fits per slab) where the table saves instructions (loo_codes_saving), else they stay with
    // em_sweep_group_kernel and its shared loads and conversions
    // -- and small populations stay with em_sweep_kernel too: below ~28 individuals the table costs more than it saves
    // (measured: 20 individuals 0.98x, 30 1.16x, 36 1.26x, 62 1.64x, 100 2.1x)
    // -- and the codes are BUILT for it only when the sweeps still to come repay the encode pass (em_codes_pay below).
    // Codes that exist already (a scoring sweep built them, or wgs_beagle_codes_prepare) are used at once.
    const bool loo_codes = !(getenv("WGSASSIGN_LOO_CODES") && atoi(getenv("WGSASSIGN_LOO_CODES")) == 0);
    const char *codes_env = getenv("WGSASSIGN_CODES");      // ("0": no codes at all -- and no sample pass to decide about them; codes.hip reads it the same way)
    bool worth = em->mode == WGS_MODE_EXACT && (!shared || loo_codes) && !(codes_env && codes_env[0] == '0');
    const char *min_env = getenv("WGSASSIGN_EM_CODES_MIN");    // tests lower it to run small populations through the codes
    const int min_cols = min_env ? atoi(min_env) : 28;
    int fewest = INT32_MAX;
    for (int j : order) fewest = std::min(fewest, (int)em->b->slabs[em->group[j]].ncols);
    worth = worth && fewest >= min_cols;
    if (worth && shared && !getenv("WGSASSIGN_EM_CODES_SWEEPS")) {      // (also when the codes exist already: a scoring sweep may have built them)
        double cols = 0.0;
        for (int j : order) cols += (double)em->b->slabs[em->group[j]].ncols;
        worth = loo_codes_saving(wgs_beagle_codes_plan(em->b), cols / (double)order.size()) > 0.03;
    }
    const bool build = worth && em_codes_pay(em, order, fewest, sweeps_ahead, shared);
    wgs_codes *codes = nullptr;
    {
        WGS_STALL_SCOPE("wgs_beagle_codes from the sweep");
        codes = worth ? wgs_beagle_codes(em->b, build, false) : nullptr;     // (memory not there yet: this sweep goes direct)
        if (codes && codes->lrows == 0 && codes->local_skipped && build) {
            // built by a scoring sweep, without the slabs' own numbering: this fit repays a full build
            const wgs_codes_plan keep = em->b->plan;
            wgs_beagle_drop_codes(em->b);
            em->b->plan = keep;
            codes = wgs_beagle_codes(em->b, true, false);
        }
    }
    WGS_STALL_SCOPE("the sweep's launches");
    if (codes && (codes->lrows == 0 || !em_coded_usable(ctx))) codes = nullptr;
    if (worth && !codes) ++em->b->direct_sweeps;          // (a sweep the codes could have served)
    // two iterations per sweep (em_kernels.hip: fused iterations): the coded sweep only, for the fits the caller allows (iterations
    // left, a place for the second sums); needs a third frequency buffer and a second set of partial sums, allocated on first use.
    // `can_fuse` is what THIS rank could do -- every input of it is rank-local (the codes' arrival, a cost model, environment
    // switches, free memory) -- and is what the rank tells the others (wgs_em_fit puts it into the sweep's collective); what the
    // sweep DOES additionally needs `fuse_agreed`: every rank has said it can.
    const bool fuse_on = !(getenv("WGSASSIGN_EM_FUSE") && atoi(getenv("WGSASSIGN_EM_FUSE")) < 2);   // (read at every sweep: tests compare both)
    bool fusing = false, can_fuse = false;
    if (codes && may_fuse && ssq_base_b && fuse_on && !shared) {    // (leave-one-out batches are bound by arithmetic: a second iteration that turns out unneeded is not free there)
        bool wanted = false;
        for (int j : order) wanted = wanted || (*may_fuse)[j] >= 2;
        can_fuse = (em->fbuf[2] || wanted) && em_fuse_buffers(em);         // (no memory for the buffers: one iteration per sweep)
        fusing = can_fuse && wanted && (fuse_agreed || wgs_hook("em_fuse_without_agreement") != 0);
    }
    if (can_fuse_out) *can_fuse_out = can_fuse ? 1 : 0;
    const int nb = em->fbuf[2] ? 3 : 2;
    int coded_rows_max = 0;
    for (size_t i = 0; i < order.size(); ++i) {
        const int j = order[i];
        const Slab &s = em->b->slabs[em->group[j]];
        FitDesc &d = H[i];
        d.lcodes = codes ? codes->slabs[em->group[j]].lcodes : nullptr;
        d.ldict = codes ? codes->slabs[em->group[j]].ldict : nullptr;
        d.lrows = codes ? codes->lrows : 0;
        d.tile_rows = codes ? codes->slabs[em->group[j]].tile_rows : nullptr;
        d.nquads = codes ? codes->slabs[em->group[j]].nquads : 0;
        coded_rows_max = std::max(coded_rows_max, (int)d.lrows);
        d.slab = s.base;
        const int c = em->cur[j], n1 = (c + 1) % nb, n2 = (c + 2) % nb;
        const int fuse = fusing && (*may_fuse)[j] >= 2 ? 2 : 1;
        d.f_old = em_f(em, j, c);
        d.f_new = em_f(em, j, n1);
        d.f_new2 = fuse == 2 ? em_f(em, j, n2) : d.f_new;
        d.fuse = fuse;
        d.ssq = ssq_base + j;
        d.ssq2 = ssq_base_b ? ssq_base_b + j : d.ssq;
        d.ssq_part = em->d_part + (size_t)j * ntiles;
        d.ssq_part2 = fuse == 2 ? em->d_part_b + (size_t)j * ntiles : d.ssq_part;
        em->fuse_used[j] = (uint8_t)fuse;
        em->pend_cur[j] = (uint8_t)(fuse == 2 ? n2 : n1);
        em->pend_prev[j] = (uint8_t)(fuse == 2 ? n1 : c);
        d.npairs = s.npairs;
        d.ncols = s.ncols;
        d.skip = em->skip_local[j];
        d.n_eff = em->n_eff[j];
        d.state = state_base ? state_base + j : nullptr;
    }
    // H (pinned) stays untouched until the caller has waited for this sweep
    HIP_TRY(hipMemcpyAsync(D, H, sizeof(FitDesc) * order.size(), hipMemcpyHostToDevice, ctx->stream));
    int32_t n_groups = 0;
    if (shared) {
        // groups of fits of one slab: four per wavefront for the float32 kernel (shared loads and conversions); through the codes a
        // wavefront walks up to 16 fits one after the other (the dictionary rows stay in registers)
        const int fg = codes ? 16 : em_fits_per_group();
        for (size_t i = 0; i < order.size();) {
            size_t k = i + 1;
            while (k < order.size() && (int)(k - i) < fg && em->group[order[k]] == em->group[order[i]]) ++k;
            Hg[2 * n_groups] = (int32_t)i;
            Hg[2 * n_groups + 1] = (int32_t)(k - i);
            ++n_groups;
            i = k;
        }
        HIP_TRY(hipMemcpyAsync(Dg, Hg, sizeof(int32_t) * 2 * n_groups, hipMemcpyHostToDevice, ctx->stream));
    }
    if (ev0) HIP_TRY(hipEventRecord(ev0, ctx->stream));
    const int64_t per_unit = ((ntiles + 3) / 4 + 7) / 8 * 8 + 8;       // workgroups per fit / per group: slices stay below 2^31
    const size_t max_units = (size_t)std::max<int64_t>(1, ((1ll << 31) - 1) / per_unit);
    if (codes && shared) {
        const int64_t per_group = (ntiles + 7) / 8 * 8 + 8;
        const size_t max_groups = (size_t)std::max<int64_t>(1, ((1ll << 31) - 1) / per_group);
        for (size_t off = 0; off < (size_t)n_groups; off += max_groups) {
            const int cnt = (int)std::min<size_t>(max_groups, (size_t)n_groups - off);
            if (launch_em_coded_groups(ctx, D, Dg + 2 * off, cnt, em->b->m, coded_rows_max)) return 1;
        }
    } else if (codes) {
        const int64_t per_fit = (ntiles + 7) / 8 * 8 + 8;            // at least one tile per workgroup
        const size_t max_fits = (size_t)std::max<int64_t>(1, ((1ll << 31) - 1) / per_fit);
        for (size_t off = 0; off < order.size(); off += max_fits) {
            const int cnt = (int)std::min<size_t>(max_fits, order.size() - off);
            if (launch_em_coded(ctx, D + off, cnt, em->b->m, coded_rows_max)) return 1;
        }
    } else if (shared) {
        for (size_t off = 0; off < (size_t)n_groups; off += max_units) {
            const int cnt = (int)std::min<size_t>(max_units, (size_t)n_groups - off);
            if (launch_em_sweep_groups(ctx, D, Dg + 2 * off, cnt, em->b->m, em->mode)) return 1;
        }
    } else {
        for (size_t off = 0; off < order.size(); off += max_units) {
            const int cnt = (int)std::min<size_t>(max_units, order.size() - off);
            if (launch_em_sweep(ctx, D + off, cnt, em->b->m, em->mode)) return 1;
        }
    }
    if (ev1) HIP_TRY(hipEventRecord(ev1, ctx->stream));
    for (size_t off = 0; off < order.size(); off += 65535) {
        const int cnt = (int)std::min<size_t>(65535, order.size() - off);
        if (launch_ssq_reduce(ctx, D + off, cnt, em->b->m, em->d_part2 + off * ssq_reduce_chunks())) return 1;
        if (fusing && launch_ssq_reduce(ctx, D + off, cnt, em->b->m, em->d_part2 + off * ssq_reduce_chunks(), 1)) return 1;
    }
    return 0;
}

int wgs_em_step_dev(wgs_em *em, double *ssq_dev)
{
    WGS_REQUIRE(em && ssq_dev, "null argument");
    wgs_ctx *ctx = em->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    em->last.clear();
    for (int j = 0; j < em->n_fits; ++j)
        if (em->active[j]) em->last.push_back(j);
    HIP_TRY(hipMemsetAsync(ssq_dev, 0, sizeof(double) * em->n_fits, ctx->stream));
    if (em->last.empty()) return 0;
    // h_descs / h_groups (pinned) stay untouched until the next step, which the caller only starts after
    // consuming this step's sums
    if (em_enqueue_sweep(em, em->last, em->h_descs, em->d_descs, em->h_groups, em->d_groups, ssq_dev, nullptr, em->ev0, em->ev1, 0)) return 1;
    for (int j : em->last) {                  // the new frequencies are now current; prev holds the ones before
        em->cur[j] = em->pend_cur[j];
        em->prev[j] = em->pend_prev[j];
    }
    return 0;
}

int wgs_em_step(wgs_em *em, double *ssq_host)
{
    WGS_REQUIRE(em, "null argument");
    if (wgs_em_step_dev(em, em->d_ssq)) return 1;
    wgs_ctx *ctx = em->b->ctx;
    if (ssq_host) {
        HIP_TRY(hipMemcpyAsync(ssq_host, em->d_ssq, sizeof(double) * em->n_fits, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int wgs_em_rmse_chain(wgs_em *em, int32_t fit, float carry_in, float *carry_out)
{
    WGS_REQUIRE(em && carry_out, "null argument");
    WGS_REQUIRE(fit >= 0 && fit < em->n_fits, "fit index out of range");
    wgs_ctx *ctx = em->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    if (launch_rmse_chain(ctx, em_f(em, fit, em->cur[fit]), em_f(em, fit, em->prev[fit]), em->b->m, carry_in, em->d_carry,
                          em->d_chain_work, reinterpret_cast<int *>(em->d_carry + 1)))
        return 1;
    float host[2];
    HIP_TRY(hipMemcpyAsync(host, em->d_carry, 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *carry_out = host[0];
    memcpy(&em->last_chain_serial_blocks, &host[1], sizeof(int));
    return 0;
}

/* ---- emMAF.py:15-27 for every fit of the batch in ONE call ------------------------------------------
 * The host enqueues iteration t (sweep, sum reduction, [RCCL all-reduce], decision kernel, state readback)
 * BEFORE it reads the decisions of iteration t-1, so the GPU never waits for the host:
 *   - the decision kernel settles the clear cases on the device (EM_CONVERGED / EM_ACTIVE) and parks the
 *     fits whose float64 sum lies in the guard band (EM_UNDECIDED);
 *   - a sweep skips every fit that is not EM_ACTIVE, so a fit that converged at t-1 keeps the frequencies of
 *     update t-1 (emMAF.py:23-25 breaks after the update) and a parked fit keeps both vectors its exact
 *     chain needs;
 *   - the host, one iteration behind, resolves parked fits with the exact serial float32 chain (all of them
 *     in one batched launch; across SNP shards the float32 carries travel in rank order) and either
 *     finishes them or re-activates them -- such a fit simply runs its next sweep one iteration later.
 * Decisions use only all-reduced sums, so every rank takes the same path. */
// d_chain_out: [n_fits] float32 carries (broadcast from shard to shard, the root's tag row behind them) | room for that row |
// [n_fits] serial-block counts
static size_t em_chain_serial_off(size_t n) { return (n + 1) / 2 * 2 + 2 * wgs_comm_tail_doubles(); }
static size_t em_chain_out_floats(size_t n) { return em_chain_serial_off(n) + n; }

static int em_fit_alloc(wgs_em *em)
{
    if (em->d_state) return 0;
    const size_t n = (size_t)em->n_fits;
    HIP_TRY(wgs_malloc(&em->d_state, sizeof(int32_t) * n));
    // (the sums of a sweep | its second iteration's | room for the rows the communicator attaches to their all-reduce)
    HIP_TRY(wgs_malloc(&em->d_ssq2, sizeof(double) * (2 * n + wgs_comm_tail_doubles())));
    HIP_TRY(hipMemset(em->d_ssq2, 0, sizeof(double) * (2 * n + wgs_comm_tail_doubles())));
    HIP_TRY(wgs_malloc(&em->d_jobs, sizeof(ChainJob) * n));
    HIP_TRY(wgs_malloc(&em->d_chain_out, sizeof(float) * em_chain_out_floats(n)));
    // workspace of the exact chains for all fits at once (60 bytes per fit and block of 4096 SNPs): no allocation
    // inside the convergence loop
    HIP_TRY(wgs_malloc(&em->d_chain_batch, rmse_chain_workspace_bytes(em->b->m) * n));
    em->chain_batch_jobs = n;
    HIP_TRY(hipHostMalloc(&em->h_jobs, sizeof(ChainJob) * n, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&em->h_chain_out, sizeof(float) * 2 * n, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&em->h_setstate, sizeof(int32_t) * n, hipHostMallocDefault));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(wgs_malloc(&em->d_descs2[i], sizeof(FitDesc) * n));
        HIP_TRY(hipHostMalloc(&em->h_descs2[i], sizeof(FitDesc) * n, hipHostMallocDefault));
        HIP_TRY(wgs_malloc(&em->d_groups2[i], sizeof(int32_t) * 2 * n));
        HIP_TRY(hipHostMalloc(&em->h_groups2[i], sizeof(int32_t) * 2 * n, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(&em->h_state[i], sizeof(int32_t) * n, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(&em->h_ssq[i], sizeof(double) * (2 * n + wgs_comm_tail_doubles()), hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&em->ev_it[i], hipEventDisableTiming));
    }
    return 0;
}

/* Exact chains of `fits` (all at once): converged[i] = the reference's `diff < tole` for fits[i]. */
static int em_resolve_chains(wgs_em *em, const std::vector<int32_t> &fits, double tole, int64_t m_total, wgs_comm *comm,
                             std::vector<char> &converged, int32_t generation, int32_t iteration)
{
    wgs_ctx *ctx = em->b->ctx;
    const int nj = (int)fits.size();
    converged.assign(nj, 0);
    if (nj == 0) return 0;
    int world = 1, rank = 0;
    if (comm) wgs_comm_rank(comm, &rank, &world);
    // The serial float32 chain crosses the SNP shards in rank order ON THE STREAM: rank r walks its blocks from the
    // running values it received and broadcasts the result (`world` broadcasts of nj float32, one readback at the end).
    for (int i = 0; i < nj; ++i) {
        const int j = fits[i];
        em->h_jobs[i] = ChainJob{em_f(em, j, em->cur[j]), em_f(em, j, em->prev[j]), 0.0f};
    }
    HIP_TRY(hipMemcpyAsync(em->d_jobs, em->h_jobs, sizeof(ChainJob) * nj, hipMemcpyHostToDevice, ctx->stream));
    for (int r = 0; r < world; ++r) {
        if (r == rank) {
            if (r > 0 && launch_chain_set_carry(ctx, em->d_jobs, em->d_chain_out, nj)) return 1;
            if (launch_rmse_chain_batch(ctx, em->d_jobs, nj, em->b->m, em->d_chain_out, em->d_chain_batch,
                                        reinterpret_cast<int *>(em->d_chain_out + em_chain_serial_off(em->n_fits))))
                return 1;
        }
        const wgs_coll_tag tag = {WGS_OP_EM_CHAIN, generation, iteration, nj, r, 0};
        if (world > 1 && wgs_comm_bcast_tagged(comm, em->d_chain_out, (int64_t)sizeof(float) * nj, r, &tag)) return 1;
    }
    HIP_TRY(hipMemcpyAsync(em->h_chain_out, em->d_chain_out, sizeof(float) * nj, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));       // also: h_jobs has been consumed
    if (wgs_comm_check(comm)) return 1;               // (a receiver's view of the senders' rows)
    const float *carry = em->h_chain_out;
    ++em->fit_chain_batches;
    for (int i = 0; i < nj; ++i) {
        const float res = carry[i] / (float)m_total;         // emMAF_cy.pyx:32
        converged[i] = sqrt((double)res) < tole;             // emMAF_cy.pyx:33, emMAF.py:23
    }
    return 0;
}

int wgs_em_fit(wgs_em *em, int32_t max_iter, double tole, int64_t m_total, wgs_comm *comm, double guard_floor, int32_t *iters_out)
{
    WGS_REQUIRE(em && iters_out, "null argument");
    WGS_REQUIRE(m_total >= em->b->m, "m_total (%lld) is smaller than this shard (%lld SNPs)", (long long)m_total, (long long)em->b->m);
    wgs_ctx *ctx = em->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    if (em_fit_alloc(em)) return 1;
    const int n = em->n_fits;
    // the band of device.py: guard_band / decide_converged
    double lo = -1.0, hi = -INFINITY;                        // tole <= 0 or NaN: `diff < tole` never holds
    if (tole > 0) {
        const double thresh = tole * tole * (double)m_total;
        const double g = std::max(guard_floor, (double)m_total * 0x1p-24) + 1e-6;
        lo = g < 1.0 ? thresh * (1.0 - g) : -1.0;
        hi = thresh * (1.0 + g);
    }
    // Across SNP shards every rank must run the same number of iterations per sweep (the sums of a sweep are all-reduced, and the
    // bookkeeping below counts iterations): two per sweep only once EVERY rank can -- its codes built with the slabs' own numbering
    // and the buffers at hand.  Whether a rank can is rank-local (its cost model, when its helper thread's allocation arrives, its
    // environment), so it is never acted upon directly: each sweep's all-reduce carries every rank's "I could" (the free word of its
    // tag row), the host reads the rows with the sweep's decisions -- one sweep behind, like everything else here -- and from then on
    // every rank fuses, at the same sweep.  A fit whose codes arrive during it on some rank therefore starts with one iteration per
    // sweep everywhere and switches to two everywhere.  The tag of the all-reduce also says how many fits the sweep lists and how
    // many EM iterations it runs; a rank that got this wrong is found at that very collective (rccl_comm.hip), not by its numbers.
    int world = 1, rank = 0;
    if (comm) wgs_comm_rank(comm, &rank, &world);
    const int32_t generation = comm ? wgs_comm_next_generation(comm) : 0;
    bool fuse_agreed = comm == nullptr;                      // one shard: nothing to agree on
    std::vector<char> fin(n, 0), skipped(n, 0);
    std::vector<int32_t> sweeps(n, 0), init(n), may_fuse(n, 1), ran, parked, parked_a, lists[2];
    for (int j = 0; j < n; ++j) {
        iters_out[j] = 0;
        fin[j] = !em->active[j];
        init[j] = em->active[j] ? EM_ACTIVE : EM_CONVERGED;
    }
    HIP_TRY(hipMemcpyAsync(em->d_state, init.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    em->fit_iterations = em->fit_chain_batches = 0;
    em->fit_sweep_ms = 0.0;
    em->fit_timed = 0;
    em->fit_sweep_pending = false;
    const auto t_begin = std::chrono::steady_clock::now();
    // the reference's `diff < tole` from a float64 sum, as em_decide_kernel classifies it
    auto classify = [&](double v) { return (v != v || v >= hi) ? EM_ACTIVE : (v < lo ? EM_CONVERGED : EM_UNDECIDED); };
    // the fit ends with the frequencies in `cur`, `it` iterations after its start
    auto finish = [&](int j, int it) {
        fin[j] = 1;
        iters_out[j] = it;
    };
    // stream-ordered behind the iteration in flight (whose sweep must see the fit parked throughout)
    auto set_state = [&](int j, int32_t st) -> int {
        em->h_setstate[j] = st;
        HIP_TRY(hipMemcpyAsync(em->d_state + j, em->h_setstate + j, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        return 0;
    };
    bool launched_prev = false;
    for (int t = 1;; ++t) {
        const int slot = t & 1;
        // Who ran at t-1 is known now: its list minus the fits found finished or parked when the decisions
        // of t-2 were read (those sweeps returned at once).
        ran.clear();
        for (int j : lists[slot ^ 1]) {
            if (skipped[j]) continue;
            sweeps[j] += em->fuse_used[j];                   // one EM iteration, or the two of a fused sweep
            em->cur[j] = em->pend_cur[j];                    // the new frequencies are current; prev holds the ones before
            em->prev[j] = em->pend_prev[j];
            ran.push_back(j);
        }
        std::fill(skipped.begin(), skipped.end(), 0);
        // ---- enqueue iteration t (fits that turn out to have converged at t-1 return at once)
        std::vector<int32_t> &L = lists[slot];
        L.clear();
        for (int j = 0; j < n; ++j) {
            if (!fin[j] && sweeps[j] < max_iter) L.push_back(j);
            may_fuse[j] = max_iter - sweeps[j] >= 2 ? 2 : 1;
        }
        if (!L.empty()) {
            hipEvent_t sw0 = nullptr, sw1 = nullptr;         // this iteration's pair (the first EM_TIMED_SWEEPS iterations of a fit are timed)
            if (em->fit_timed < EM_TIMED_SWEEPS) {
                while ((int)em->ev_sw.size() < 2 * em->fit_timed + 2) {
                    hipEvent_t e = nullptr;
                    HIP_TRY(hipEventCreate(&e));
                    em->ev_sw.push_back(e);
                }
                sw0 = em->ev_sw[2 * em->fit_timed];
                sw1 = em->ev_sw[2 * em->fit_timed + 1];
                ++em->fit_timed;
                em->fit_sweep_pending = true;
            }
            int can_fuse = 0;
            if (em_enqueue_sweep(em, L, em->h_descs2[slot], em->d_descs2[slot], em->h_groups2[slot], em->d_groups2[slot], em->d_ssq2,
                                 em->d_state, sw0, sw1, max_iter - t + 1, &may_fuse, em->d_ssq2 + n, fuse_agreed, &can_fuse))
                return 1;
            // Fits that skipped this sweep have stale sums; the decision kernel ignores them, and they are stale
            // in the same way on every rank (all ranks take the same decisions).
            if (comm) {
                int32_t iterations_run = 0;
                for (int j : L) iterations_run += em->fuse_used[j];
                const wgs_coll_tag tag = {WGS_OP_EM_SUMS, generation, t, (int32_t)L.size(), iterations_run, can_fuse};
                if (wgs_comm_allreduce_tagged(comm, em->d_ssq2, 2 * n, &tag)) return 1;
            }
            if (launch_em_decide(ctx, em->d_descs2[slot], (int)L.size(), lo, hi)) return 1;
            HIP_TRY(hipMemcpyAsync(em->h_state[slot], em->d_state, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipMemcpyAsync(em->h_ssq[slot], em->d_ssq2, sizeof(double) * (2 * n + (comm ? world * WGS_TAG_WORDS : 0)), hipMemcpyDeviceToHost,
                                   ctx->stream));
            HIP_TRY(hipEventRecord(em->ev_it[slot], ctx->stream));
            ++em->fit_iterations;
        }
        // ---- read the decisions of iteration t-1 while the GPU works on iteration t
        if (launched_prev) {
            const int ps = slot ^ 1;
            HIP_TRY(hipEventSynchronize(em->ev_it[ps]));     // also: the pinned descriptors of t-1 have been consumed
            if (comm) {
                if (wgs_comm_check(comm)) return 1;          // some rank's sweep t-1 was not this rank's sweep t-1
                bool all = true;                             // every rank's "I could run two iterations per sweep" as of sweep t-1
                for (int r = 0; r < world; ++r) all = all && em->h_ssq[ps][2 * n + r * WGS_TAG_WORDS + WGS_TAG_AUX] == 1.0;
                fuse_agreed = all;                           // acted upon from sweep t+1 on, by every rank alike
            }
            parked.clear();
            parked_a.clear();
            for (int j : ran) {
                const int st = em->h_state[ps][j];
                if (st == EM_CONVERGED) {
                    skipped[j] = 1;                          // its sweep t (if enqueued) returned at once
                    finish(j, sweeps[j]);
                } else if (st == EM_CONVERGED_A) {
                    // the FIRST of the sweep's two iterations converged: its frequencies are the result, the second is dropped
                    skipped[j] = 1;
                    const int third = 3 - em->cur[j] - em->prev[j];
                    em->cur[j] = em->prev[j];
                    em->prev[j] = (uint8_t)third;
                    sweeps[j] -= 1;
                    finish(j, sweeps[j]);
                } else if (st == EM_UNDECIDED) {
                    parked.push_back(j);
                    skipped[j] = 1;
                } else if (st == EM_UNDECIDED_A) {
                    parked_a.push_back(j);
                    skipped[j] = 1;
                } else if (sweeps[j] >= max_iter) {
                    fin[j] = 1;                              // exhausted: the reference prints nothing, iters stays 0
                }
            }
            // parked after the first of two iterations: the exact chain speaks about (f_a, f_in) -- looked at through cur / prev
            // for the call; when it says "not converged" the second iteration counts and its sum is classified here as the
            // device would have (the same thresholds on the same all-reduced float64), possibly parking the fit again
            if (!parked_a.empty()) {
                std::vector<uint8_t> fb(parked_a.size()), fa(parked_a.size());
                for (size_t i = 0; i < parked_a.size(); ++i) {
                    const int j = parked_a[i];
                    fb[i] = em->cur[j];
                    fa[i] = em->prev[j];
                    em->cur[j] = fa[i];
                    em->prev[j] = (uint8_t)(3 - fa[i] - fb[i]);
                }
                std::vector<char> conv;
                if (em_resolve_chains(em, parked_a, tole, m_total, comm, conv, generation, 2 * t)) return 1;
                for (size_t i = 0; i < parked_a.size(); ++i) {
                    const int j = parked_a[i];
                    if (conv[i]) {
                        sweeps[j] -= 1;
                        finish(j, sweeps[j]);
                        if (set_state(j, EM_CONVERGED)) return 1;
                        continue;
                    }
                    em->cur[j] = fb[i];                      // the first iteration goes on: the second one's result stands
                    em->prev[j] = fa[i];
                    const int cls = classify(em->h_ssq[ps][n + j]);
                    if (cls == EM_CONVERGED) {
                        finish(j, sweeps[j]);
                        if (set_state(j, EM_CONVERGED)) return 1;
                    } else if (cls == EM_UNDECIDED) {
                        parked.push_back(j);
                    } else if (sweeps[j] >= max_iter) {
                        fin[j] = 1;
                        if (set_state(j, EM_CONVERGED)) return 1;
                    } else if (set_state(j, EM_ACTIVE)) {
                        return 1;
                    }
                }
            }
            if (!parked.empty()) {
                std::vector<char> conv;
                if (em_resolve_chains(em, parked, tole, m_total, comm, conv, generation, 2 * t + 1)) return 1;
                for (size_t i = 0; i < parked.size(); ++i) {
                    const int j = parked[i];
                    if (conv[i]) finish(j, sweeps[j]);
                    else if (sweeps[j] >= max_iter) fin[j] = 1;
                    if (set_state(j, conv[i] ? EM_CONVERGED : EM_ACTIVE)) return 1;
                }
            }
        }
        launched_prev = !L.empty();
        if (!launched_prev) break;                           // nothing in flight: every fit finished or exhausted
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (comm) {
        // the ranks close the fit together: what each of them found (iteration counts, sweeps enqueued, chain resolutions) travels
        // as the tag of one last collective, so ranks that ended with different results fail here instead of returning them
        int64_t sum = 0, mix = 0;
        for (int j = 0; j < n; ++j) {
            sum += iters_out[j];
            mix = (mix * 31 + iters_out[j] + 7 * (j + 1)) % 16777213;
        }
        const wgs_coll_tag tag = {WGS_OP_EM_FIT_END, generation, em->fit_iterations, (int32_t)(sum % 16777213), (int32_t)mix, em->fit_chain_batches};
        if (wgs_comm_allreduce_host_tagged(comm, nullptr, 0, &tag, nullptr)) return 1;
    }
    for (int j = 0; j < n; ++j)
        if (iters_out[j] > 0) em->active[j] = 0;             // frozen, as wgs_em_set_active(j, 0) would
    em->fit_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return 0;
}

/* Diagnostics of the last wgs_em_fit: iterations enqueued, batched exact-chain resolutions, wall seconds, and the
 * summed duration of its sweep kernels (HIP events on the context's stream around each iteration's sweep; read here, not
 * inside the fit; -1 while the codes' memory is being allocated on the helper thread -- the query would wait for it). */
int wgs_em_fit_stats(wgs_em *em, int32_t *iterations, int32_t *chain_batches, double *seconds, double *sweep_ms)
{
    WGS_REQUIRE(em, "null argument");
    if (iterations) *iterations = em->fit_iterations;
    if (chain_batches) *chain_batches = em->fit_chain_batches;
    if (seconds) *seconds = em->fit_seconds;
    if (sweep_ms) {
        if (em->fit_sweep_pending && em->b->ctx->allocs_in_flight.load() > 0) {
            *sweep_ms = -1.0;
            return 0;
        }
        if (em->fit_sweep_pending) {
            HIP_TRY(hipSetDevice(em->b->ctx->device));
            em->fit_sweep_ms = 0.0;
            for (int i = 0; i < em->fit_timed; ++i) {
                float ms = 0.0f;
                if (hipEventElapsedTime(&ms, em->ev_sw[2 * i], em->ev_sw[2 * i + 1]) == hipSuccess) em->fit_sweep_ms += ms;
            }
            (void)hipGetLastError();
            em->fit_sweep_pending = false;
        }
        *sweep_ms = em->fit_sweep_ms;
    }
    return 0;
}

int wgs_em_last_chain_serial_blocks(wgs_em *em) { return em ? em->last_chain_serial_blocks : -1; }

int wgs_em_last_sweep_ms(wgs_em *em, float *ms)
{
    WGS_REQUIRE(em && ms, "null argument");
    HIP_TRY(hipSetDevice(em->b->ctx->device));
    HIP_TRY(hipEventSynchronize(em->ev1));
    HIP_TRY(hipEventElapsedTime(ms, em->ev0, em->ev1));
    return 0;
}

int wgs_em_set_active(wgs_em *em, int32_t fit, int active)
{
    WGS_REQUIRE(em && fit >= 0 && fit < em->n_fits, "fit index out of range");
    em->active[fit] = active ? 1 : 0;
    return 0;
}

int wgs_em_n_active(wgs_em *em)
{
    int c = 0;
    for (int j = 0; j < em->n_fits; ++j) c += em->active[j];
    return c;
}

int wgs_em_clamp(wgs_em *em, int32_t fit, float lo, float hi)
{
    WGS_REQUIRE(em && fit >= 0 && fit < em->n_fits, "fit index out of range");
    HIP_TRY(hipSetDevice(em->b->ctx->device));
    return launch_clamp(em->b->ctx, em_f(em, fit, em->cur[fit]), em->b->m, lo, hi);
}

int wgs_em_get_f(wgs_em *em, int32_t fit, float *f_host)
{
    WGS_REQUIRE(em && f_host && fit >= 0 && fit < em->n_fits, "bad argument");
    HIP_TRY(hipSetDevice(em->b->ctx->device));
    HIP_TRY(hipMemcpyAsync(f_host, em_f(em, fit, em->cur[fit]), sizeof(float) * em->b->m, hipMemcpyDeviceToHost, em->b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(em->b->ctx->stream));
    return 0;
}

int wgs_em_get_f_range(wgs_em *em, int32_t fit, int previous, int64_t row0, int64_t nrows, float *f_host)
{
    WGS_REQUIRE(em && f_host && fit >= 0 && fit < em->n_fits, "bad argument");
    WGS_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= em->b->m, "row range [%lld, %lld) outside 0..%lld", (long long)row0,
                (long long)(row0 + nrows), (long long)em->b->m);
    if (nrows == 0) return 0;
    HIP_TRY(hipSetDevice(em->b->ctx->device));
    const float *src = em_f(em, fit, previous ? em->prev[fit] : em->cur[fit]) + row0;
    HIP_TRY(hipMemcpyAsync(f_host, src, sizeof(float) * nrows, hipMemcpyDeviceToHost, em->b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(em->b->ctx->stream));
    return 0;
}

int wgs_em_set_f(wgs_em *em, int32_t fit, const float *f_host)
{
    WGS_REQUIRE(em && f_host && fit >= 0 && fit < em->n_fits, "bad argument");
    HIP_TRY(hipSetDevice(em->b->ctx->device));
    HIP_TRY(hipMemcpyAsync(em_f(em, fit, em->cur[fit]), f_host, sizeof(float) * em->b->m, hipMemcpyHostToDevice, em->b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(em->b->ctx->stream));
    return 0;
}

const float *wgs_em_f_dev(wgs_em *em, int32_t fit)
{
    if (!em || fit < 0 || fit >= em->n_fits) return nullptr;
    return em_f(em, fit, em->cur[fit]);
}

}   // extern "C"
