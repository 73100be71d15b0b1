// State of a batch of EM fits (wgs_em of include/wgsassign_hip.h): shared by em_api.hip (the fits) and api.hip (frequency sets
// that copy a fit's column).
#pragma once
#include "common.h"

constexpr int EM_TIMED_SWEEPS = 4096;     // iterations of one wgs_em_fit whose sweeps are bracketed by HIP events

struct wgs_em {
    wgs_beagle *b = nullptr;
    int32_t n_fits = 0;
    int mode = WGS_MODE_EXACT;
    std::vector<int32_t> group, skip_local, n_eff;
    std::vector<uint8_t> cur, prev, active;   // per fit: the buffer holding the current / the previous frequencies
    float *fbuf[3] = {nullptr, nullptr, nullptr};  // n_fits x m each; the third only once a sweep runs two iterations at a time (em_api.hip)
    // per fit, of the sweep last enqueued for it: iterations it runs (1 or 2) and where cur / prev point once it has run
    std::vector<uint8_t> fuse_used, pend_cur, pend_prev;
    double *d_part_b = nullptr;           // n_fits x ntiles partial sums of a fused sweep's second iteration
    FitDesc *d_descs = nullptr;
    FitDesc *h_descs = nullptr;           // pinned
    int32_t *d_groups = nullptr, *h_groups = nullptr;         // (first, count) pairs of the fit-group sweep, step path
    int32_t *d_groups2[2] = {nullptr, nullptr}, *h_groups2[2] = {nullptr, nullptr};   // ... wgs_em_fit slots
    double *d_ssq = nullptr;
    double *d_part = nullptr;             // n_fits x ntiles per-tile partial sums
    double *d_part2 = nullptr;            // n_fits x ssq_reduce_chunks() slice sums
    float *d_carry = nullptr;             // [0] carry out, [1] (as int) serial-block count
    void *d_chain_work = nullptr;
    std::vector<int32_t> last;            // fits swept by the last step
    int last_chain_serial_blocks = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // bracket the sweep kernel(s) of the last step
    // wgs_em_fit: device-side fit states, two-slot pinned rings for the one-iteration lookahead
    int32_t *d_state = nullptr;           // [n_fits] EM_ACTIVE / EM_CONVERGED / EM_UNDECIDED
    FitDesc *d_descs2[2] = {nullptr, nullptr}, *h_descs2[2] = {nullptr, nullptr};
    int32_t *h_state[2] = {nullptr, nullptr}, *h_setstate = nullptr;
    double *d_ssq2 = nullptr;             // [2 n_fits] sums of the iteration(s) in flight: first | second of a fused sweep
    double *h_ssq[2] = {nullptr, nullptr};  // pinned read-backs of them, one per slot
    hipEvent_t ev_it[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev_sw;        // pairs bracketing the sweep kernel(s) of the iterations of the last wgs_em_fit (grown as needed, EM_TIMED_SWEEPS at most);
    int fit_timed = 0;                    //   read when wgs_em_fit_stats is asked (hipEventElapsedTime waits for a hipMalloc in flight: common.h)
    bool fit_sweep_pending = false;
    double fit_sweep_ms = 0.0;            // summed sweep-kernel time of the last wgs_em_fit (HIP events)
    ChainJob *d_jobs = nullptr, *h_jobs = nullptr;
    float *d_chain_out = nullptr, *h_chain_out = nullptr;     // [n_fits] carries | [n_fits] serial-block counts
    void *d_chain_batch = nullptr;
    size_t chain_batch_jobs = 0;
    double fit_seconds = 0.0;
    int fit_iterations = 0, fit_chain_batches = 0;
};

static inline float *em_f(wgs_em *em, int fit, int which) { return em->fbuf[which] + (size_t)fit * em->b->m; }
