// EM allele-frequency update (emMAF_cy.pyx:10-23) and convergence metric (emMAF_cy.pyx:26-33)
// as CDNA4 (gfx950) kernels.  Compiled with -ffp-contract=off: in WGS_MODE_EXACT every
// floating-point operation below is one rounding of the reference's C expression, so no
// contraction or reassociation is allowed; fused operations are written explicitly.
//
// Mapping (exact mode must reproduce a SERIAL float32 accumulation over individuals):
//   lane <-> SNP           (64 SNPs per wavefront; the serial chain lives in one lane)
//   wave <-> tile of 64 SNPs x all individuals of one population slab
//   The slab is stored tile-interleaved (common.h: Slab), so the wave's k-th load instruction
//   is ONE aligned contiguous 1 KiB: lane l receives (g0,g1) of individuals 2k and 2k+1 of ITS
//   SNP.  No LDS, no barriers, no cross-lane traffic; HBM bytes == algorithmic bytes.  All
//   lanes process the same individual at the same time, so the leave-one-out skip and the
//   column bound are wave-uniform branches.  Loads for the next U pairs are issued before the
//   current U pairs are consumed (register double buffer).
#include <type_traits>

#include "common.h"

namespace {

constexpr int WAVES = 4;
typedef float f4 __attribute__((ext_vector_type(4)));   // plain vector type: stays in VGPRs
// Pointers read out of the descriptor table are generic to the compiler (-> flat_load, which
// cannot be pipelined with counted waits); they are device-global by construction.
typedef const f4 __attribute__((address_space(1))) *gf4_ptr;
typedef const float __attribute__((address_space(1))) *gf32_ptr;
typedef float __attribute__((address_space(1))) *gf32_wptr;
typedef const uint32_t __attribute__((address_space(1))) *gu32_ptr;
typedef const double __attribute__((address_space(1))) *gf64_ptr;

template <bool NT>
__device__ __forceinline__ f4 ldg(gf4_ptr p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

struct SnpState {
    double fd, omf, fd2;              // f, 1-f, 2f in double (per SNP, hoisted)
    float ff, omff, ff2;              // fast-mode float copies
};

// Correctly rounded double quotient num/den for den = a float32 value widened to double (or 0) and num >= 0.
// v_rcp_f64 seed (measured 2^-24.4 relative), ONE Newton refinement (r within 2^-48.7 of 1/den: checked for every
// float32 mantissa of den, tests/test_gpu_log.py), q0 = RN(num * r), residual fma, one correction, v_div_fixup for
// the 0/0, x/0 and NaN results of `/`.  Why one refinement suffices where a general IEEE divide needs two:
//   * before its rounding the corrected quotient fma(rem, r, q0) is within q * 2^-97 of num/den (the residual is known
//     to 2^-53 relative, the reciprocal to 2^-48.7, and the correction itself is only q * 2^-48.6 large);
//   * den has 24 significant bits and num at most 53, so num/den is never closer than q * 2^-78 to the midpoint of two
//     doubles (and never on one): with num = A * 2^a, den = B * 2^b, midpoint M * 2^u (M odd, 54 bits),
//     |num - den * M 2^u| is a non-zero multiple of 2^(b+u), i.e. >= 2^(a-25), and dividing by den < 2^(b+24)
//     leaves >= 2^(a-b-49) = q * 2^-78;
// so the final rounding lands on the correctly rounded quotient -- the same bits as the compiler's IEEE divide
// (wgs_debug_div_mismatch: 0 of 8.6e9 EM-shaped operand pairs), with two fewer FP64 instructions per term and
// without the v_div_scale range scaling these operands never need.
__device__ __forceinline__ double refined_rcp(double den)
{
    double r = __builtin_amdgcn_rcp(den);
    const double e = __builtin_fma(-den, r, 1.0);
    return __builtin_fma(r, e, r);
}

// FIXUP = false leaves out v_div_fixup: correct whenever den is a positive finite float32 (subnormals included: as
// doubles they are normal and num/den stays far inside the double range) and num is finite -- the hot loops check
// exactly that per buffer (one v_cmp_class per term, ANDed in scalar registers) and redo the rare buffer that holds
// a sum of 0, NaN or inf with FIXUP = true, which is what produces the reference's 0/0 -> NaN.
template <bool FIXUP = true>
__device__ __forceinline__ double div_exact(double num, double den)
{
    const double r = refined_rcp(den);
    double q = num * r;
    const double rem = __builtin_fma(-den, q, num);
    q = __builtin_fma(rem, r, q);
    return FIXUP ? __builtin_amdgcn_div_fixup(q, den, num) : q;
}

constexpr unsigned FP_POS_FINITE_NONZERO = 0x0100 | 0x0080;     // +normal | +subnormal

// One (SNP, individual) term of emMAF_cy.pyx:19-22, exact rounding sequence:
//   p0 = (float)(((double)g0*(1.0-f))*(1.0-f))
//   p1 = (float)((((double)g1*2.0)*f)*(1.0-f))          (g1*2.0)*f == g1*(2f): exact scaling
//   p2 = (float)((((1.0-(double)g0)-(double)g1)*f)*f)
//   tmp = (float)((double)tmp + ((double)p1 + 2.0*(double)p2) / (2.0*(double)((p0+p1)+p2)))
// Scalings by 2 are exact, so 2.0*p2 + p1 is one fma and x/(2s) == 0.5*(x/s) folds into the
// accumulation's fma: every remaining operation is one rounding of the reference's expression.
// (g0d, g1d, g2d = g0, g1 widened to double and (1-g0)-g1: they do not depend on the fit, so leave-one-out fits of one
// population that walk a tile together share them.)  FIXUP = false: see div_exact; `ok` collects whether every sum
// was a positive finite number.
template <bool FIXUP>
__device__ __forceinline__ void term_exact_shared(double g0d, double g1d, double g2d, const SnpState &st, float &tmp, bool &ok)
{
    const float p0 = (float)((g0d * st.omf) * st.omf);
    const float p1 = (float)((g1d * st.fd2) * st.omf);
    const float p2 = (float)((g2d * st.fd) * st.fd);
    const float s = (p0 + p1) + p2;
    if (!FIXUP) ok = ok && __builtin_isfpclass(s, FP_POS_FINITE_NONZERO);
    const double num = __builtin_fma(2.0, (double)p2, (double)p1);
    const double q = div_exact<FIXUP>(num, (double)s);
    tmp = (float)__builtin_fma(0.5, q, (double)tmp);
}

template <bool FIXUP>
__device__ __forceinline__ void term_exact(float g0, float g1, const SnpState &st, float &tmp, bool &ok)
{
    const double g0d = (double)g0, g1d = (double)g1;
    term_exact_shared<FIXUP>(g0d, g1d, (1.0 - g0d) - g1d, st, tmp, ok);
}

// Fast mode: the same expression evaluated in float32 (one reciprocal), same accumulation order.
__device__ __forceinline__ void term_fast(float g0, float g1, const SnpState &st, float &tmp)
{
    const float p0 = g0 * st.omff * st.omff;
    const float p1 = g1 * st.ff2 * st.omff;
    const float p2 = ((1.0f - g0) - g1) * st.ff * st.ff;
    const float s = (p0 + p1) + p2;
    const float num = p1 + 2.0f * p2;
    tmp = tmp + num * __builtin_amdgcn_rcpf(2.0f * s);
}

// One fit per wavefront, nontemporal slab loads (each byte is used once: fits of different populations; fits that
// share a slab -- leave-one-out -- run em_sweep_group_kernel below).
template <int MODE, int U>
__global__ __launch_bounds__(WAVES * 64) void em_sweep_kernel(const FitDesc *__restrict__ fits, int n_fits, int64_t m)
{
    constexpr bool NT = true;
    // Fit index varies fastest across workgroups: the K fits interleave K streams over the slabs.
    const int fit = (int)(blockIdx.x % (unsigned)n_fits);
    const int64_t tgroup = blockIdx.x / (unsigned)n_fits;
    const FitDesc fd = fits[fit];
    // a fit whose convergence was decided (or is being decided) on the device by the previous iteration's
    // em_decide_kernel is skipped: the host enqueues sweeps one iteration ahead of what it has read back
    if (fd.state && *fd.state != EM_ACTIVE) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = tgroup * WAVES + wave;
    const int64_t row0 = tile * 64;
    if (row0 >= m) return;                       // wave-uniform; there are no barriers below

    const int64_t my_row = row0 + lane;
    const int64_t my_row_c = my_row < m ? my_row : m - 1;
    const float f_old = ((gf32_ptr)fd.f_old)[my_row_c];
    SnpState st;
    st.fd = (double)f_old;
    st.omf = 1.0 - st.fd;
    st.fd2 = 2.0 * st.fd;
    st.ff = f_old;
    st.omff = 1.0f - f_old;
    st.ff2 = 2.0f * f_old;

    const int npairs = fd.npairs;
    gf4_ptr src = (gf4_ptr)fd.slab + tile * npairs * 64 + lane;
    const int last = npairs - 1;
    f4 cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = ldg<NT>(src + (u < last ? u : last) * 64);   // clamped: tail re-reads hit cache
    float tmp = 0.0f;
    bool dummy_ok = true;            // the checked (tail / left-out) buffers always use the fixup
    for (int p0 = 0; p0 < npairs; p0 += U) {
        if (p0 + U < npairs) {                   // one wave-uniform branch per buffer, loads unconditional
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pp = p0 + U + u;
                nxt[u] = ldg<NT>(src + (pp < last ? pp : last) * 64);
            }
        }
        // buffers whose 2U individuals are all present take the branch-free path; the buffer holding
        // the left-out individual (LOO) or the odd tail checks each individual (wave-uniform)
        const bool plain = 2 * (p0 + U) <= fd.ncols && (fd.skip < 2 * p0 || fd.skip >= 2 * (p0 + U));
        if (plain) {
            if (MODE == WGS_MODE_EXACT) {
                const float tmp0 = tmp;
                bool ok = true;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const f4 v = cur[u];
                    term_exact<false>(v.x, v.y, st, tmp, ok);
                    term_exact<false>(v.z, v.w, st, tmp, ok);
                }
                if (!__all(ok)) {                // a sum of 0, NaN or inf somewhere in this buffer: redo it with the fixup
                    tmp = tmp0;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const f4 v = cur[u];
                        term_exact<true>(v.x, v.y, st, tmp, ok);
                        term_exact<true>(v.z, v.w, st, tmp, ok);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const f4 v = cur[u];
                    term_fast(v.x, v.y, st, tmp);
                    term_fast(v.z, v.w, st, tmp);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f4 v = cur[u];
                const int ia = 2 * (p0 + u), ib = ia + 1;
                if (ia < fd.ncols && ia != fd.skip) {
                    if (MODE == WGS_MODE_EXACT) term_exact<true>(v.x, v.y, st, tmp, dummy_ok); else term_fast(v.x, v.y, st, tmp);
                }
                if (ib < fd.ncols && ib != fd.skip) {
                    if (MODE == WGS_MODE_EXACT) term_exact<true>(v.z, v.w, st, tmp, dummy_ok); else term_fast(v.z, v.w, st, tmp);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
    const float f_new = tmp / (float)fd.n_eff;           // emMAF_cy.pyx:23 (float32 divide)
    double sq = 0.0;
    if (my_row < m) {
        ((gf32_wptr)fd.f_new)[my_row] = f_new;
        const float d = f_new - f_old;                    // emMAF_cy.pyx:31, float32
        sq = (double)(d * d);
    }
    // Per-tile partial sum, stored (not atomically added: a million same-address float64 atomics
    // serialise at ~12 ns each and would dominate the sweep); ssq_reduce_kernel adds them in a
    // fixed order, so the sums are reproducible run to run.
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
    if (lane == 0) fd.ssq_part[tile] = sq;
}

// Leave-one-out batches: up to FG fits of ONE population slab per wavefront.  They read the same tile, so its GL
// loads, the float->double conversions and g2 = (1-g0)-g1 are done once for all of them (4 of the 29 FP64-rate
// instructions of a term; the re-fits are bound by FP64 issue, not by memory), each fit keeping its own serial
// float32 accumulation.  groups[g] = (first descriptor, count) into `fits`, all of one slab.  The loads are
// cacheable and the workgroup order is XCD-aware: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8
// labels the XCD), so tile group = 8 * (j / n_groups) + blockIdx % 8 keeps ALL groups of a tile group on one XCD
// and its tiles enter one L2 once, not eight times.  A wrong guess about placement costs speed only.
constexpr int FG = 4;

template <int MODE, int U>
__global__ __launch_bounds__(WAVES * 64) void em_sweep_group_kernel(const FitDesc *__restrict__ fits, const int2 *__restrict__ groups,
                                                                     int n_groups, int64_t m)
{
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const int grp = (int)(j % (unsigned)n_groups);
    const int64_t tgroup = (int64_t)(j / (unsigned)n_groups) * 8 + xcd;
    const int2 gd = groups[grp];
    FitDesc fd[FG];
    bool on[FG];
    bool any = false;
#pragma unroll
    for (int f = 0; f < FG; ++f) {
        on[f] = f < gd.y;
        fd[f] = fits[gd.x + (on[f] ? f : 0)];
        if (on[f] && fd[f].state && *fd[f].state != EM_ACTIVE) on[f] = false;     // decided on the device: skip (see em_sweep_kernel)
        any = any || on[f];
    }
    if (!any) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = tgroup * WAVES + wave;
    const int64_t row0 = tile * 64;
    if (row0 >= m) return;
    const int64_t my_row = row0 + lane;
    const int64_t my_row_c = my_row < m ? my_row : m - 1;
    SnpState st[FG];
    float f_old[FG], tmp[FG];
#pragma unroll
    for (int f = 0; f < FG; ++f) {
        f_old[f] = ((gf32_ptr)fd[f].f_old)[my_row_c];
        st[f].fd = (double)f_old[f];
        st[f].omf = 1.0 - st[f].fd;
        st[f].fd2 = 2.0 * st[f].fd;
        st[f].ff = f_old[f];
        st[f].omff = 1.0f - f_old[f];
        st[f].ff2 = 2.0f * f_old[f];
        tmp[f] = 0.0f;
    }
    const int npairs = fd[0].npairs, ncols = fd[0].ncols;
    gf4_ptr src = (gf4_ptr)fd[0].slab + tile * npairs * 64 + lane;
    const int last = npairs - 1;
    f4 cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = ldg<false>(src + (u < last ? u : last) * 64);
    // one individual (g0, g1) for every fit that is on and does not leave it out.  (The fixup-free divide of the
    // single-fit kernel does not pay here: with four accumulators to save and a second copy of the body the kernel
    // measured 8 % slower.)
    bool ok = true;
    auto individual = [&](float g0, float g1, int col, bool checked) {
        if (MODE == WGS_MODE_EXACT) {
            const double g0d = (double)g0, g1d = (double)g1, g2d = (1.0 - g0d) - g1d;
#pragma unroll
            for (int f = 0; f < FG; ++f)
                if (on[f] && (!checked || col != fd[f].skip)) term_exact_shared<true>(g0d, g1d, g2d, st[f], tmp[f], ok);
        } else {
#pragma unroll
            for (int f = 0; f < FG; ++f)
                if (on[f] && (!checked || col != fd[f].skip)) term_fast(g0, g1, st[f], tmp[f]);
        }
    };
    for (int p0 = 0; p0 < npairs; p0 += U) {
        if (p0 + U < npairs) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pp = p0 + U + u;
                nxt[u] = ldg<false>(src + (pp < last ? pp : last) * 64);
            }
        }
        bool plain = 2 * (p0 + U) <= ncols;
#pragma unroll
        for (int f = 0; f < FG; ++f) plain = plain && (!on[f] || fd[f].skip < 2 * p0 || fd[f].skip >= 2 * (p0 + U));
        if (plain) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f4 v = cur[u];
                individual(v.x, v.y, 0, false);
                individual(v.z, v.w, 0, false);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f4 v = cur[u];
                const int ia = 2 * (p0 + u), ib = ia + 1;
                if (ia < ncols) individual(v.x, v.y, ia, true);
                if (ib < ncols) individual(v.z, v.w, ib, true);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
#pragma unroll
    for (int f = 0; f < FG; ++f) {
        if (!on[f]) continue;                                // wave-uniform
        const float f_new = tmp[f] / (float)fd[f].n_eff;     // emMAF_cy.pyx:23 (float32 divide)
        double sq = 0.0;
        if (my_row < m) {
            ((gf32_wptr)fd[f].f_new)[my_row] = f_new;
            const float d = f_new - f_old[f];                // emMAF_cy.pyx:31, float32
            sq = (double)(d * d);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
        if (lane == 0) fd[f].ssq_part[tile] = sq;
    }
}

// ---- the sweep through the class codes (common.h: wgs_codes, SlabLocal) -----------------------------------------
// The quotient q = (p1 + 2 p2) / ((p0 + p1) + p2) of a term depends on the individual only through its (g0, g1); a SNP has
// few distinct (g0, g1) and a population slab fewer still.  Per (fit, tile), lane <-> SNP as before, one wavefront per
// workgroup (the table is private to it: no barriers):
//   phase 1  the slab's own dictionary rows of the tile -- rank r = the r-th class PRESENT in the slab at that SNP -- are
//            requested all at once (coalesced, only as many rows as the tile's richest SNP has) and every lane turns each
//            (g0, g1) into the term's quotient: the rounding sequence of term_exact_shared up to the divide, evaluated once
//            per class instead of once per individual, ILP classes at a time (the divide is a chain of dependent FP64
//            operations); the quotients go to the table q[rank][lane] in LDS;
//   phase 2  the individuals are walked in order: tmp = (float)fma(0.5, q[rank of its class], (double)tmp) -- the serial
//            float32 accumulation of emMAF_cy.pyx:22 on the very same addends, with the leave-one-out skip and the column
//            bound wave-uniform as in the direct kernels.
// The table has one row per class present in the SLAB (<= 24 in 99 % of the tiles of 100 low-depth individuals), not per class
// of the SNP (40): 12 KiB instead of 20 per wavefront, and the sweep is bound by how many wavefronts share a CU (measured with
// padded LDS: 24.3 / 13.2 / 9.4 / 7.5 ms at 2 / 4 / 6 / 8 wavefronts per CU for the 10M x 1000 x K=10 sweep).  The rows
// (wgs_codes::lrows) are chosen per matrix so that ~1 % of the tiles at most have a richer SNP; those take the direct path.

// FUSED ITERATIONS (round 4).  The EM update of a SNP depends on that SNP's frequency and data only (emMAF_cy.pyx:16-23); what
// couples the SNPs is the convergence test between iterations, and that needs only the sums.  So a sweep may run iteration t + 2
// right behind t + 1 while the tile's dictionary rows are still in registers: fd.fuse = 2 writes both frequency vectors (f_new,
// f_new2) and both per-tile partial sums (ssq_part, ssq_part2); the decision kernel looks at the first sum, then at the second
// (em_decide_kernel), and the host takes the vector that belongs to the iteration that stopped (wgs_em_fit).  The dictionary --
// two thirds of the sweep's traffic -- is read once per two iterations; the arithmetic of each iteration is unchanged.
// The pieces of one EM iteration of one (fit, tile) through the slab's class table.  q = this wavefront's quotient table in LDS (slot of
// rank k of the SNP of lane l: q[k * 64 + l], `q` already offset by the lane), src = the tile's code words (offset by the lane).

// The quotient (p1 + 2 p2) / ((p0 + p1) + p2) of the class whose (g0, g1) bit patterns are the halves of `raw`: term_exact_shared's
// rounding sequence up to the divide (emMAF_cy.pyx:19-22).
__device__ __forceinline__ double class_quotient(double raw, const SnpState &st)
{
    const float g0 = __uint_as_float((uint32_t)((unsigned long long)__double_as_longlong(raw))),
                g1 = __uint_as_float((uint32_t)((unsigned long long)__double_as_longlong(raw) >> 32));
    const double g0d = (double)g0, g1d = (double)g1, g2d = (1.0 - g0d) - g1d;
    const float p0 = (float)((g0d * st.omf) * st.omf);
    const float p1 = (float)((g1d * st.fd2) * st.omf);
    const float p2 = (float)((g2d * st.fd) * st.fd);
    const float ssum = (p0 + p1) + p2;
    const double num = __builtin_fma(2.0, (double)p2, (double)p1);
    return div_exact<true>(num, (double)ssum);
}

// The quotient table of a wavefront in LDS, ROWS x 64 doubles, kept as two planes of 32-bit halves: low words of rank k at dword
// k * 64 + lane, high words ROWS * 64 dwords further.  A look-up's address is then (rank << 8) | (lane << 2) -- ONE v_perm_b32 takes
// the code byte out of the code word and puts it above the lane's byte (a double-wide table needs a bit-field extract and a shifted
// add per look-up: 5.5 instead of 4 vector instructions per individual) -- and both halves come back from one ds_read2st64_b32.
// The table is the kernel's only LDS and the dynamic allocation of a kernel without static LDS starts at address 0, so the permuted
// word IS the address.  That assumption is checked on the HOST, once per context, by a probe kernel of the same shape
// (em_coded_usable below): where it does not hold -- an instrumented build, a runtime that places something of its own first -- the
// coded EM sweeps are not used and the fits run over the float32 slabs (no device-side trap).
typedef __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
struct QTable {
    uint32_t lane4;      // lane * 4: one byte
};
__device__ __forceinline__ QTable qtable_of(double *lds, int lane)
{
    (void)lds;
    QTable t;
    t.lane4 = (uint32_t)lane * 4u;
    return t;
}
template <int ROWS>
__device__ __forceinline__ void qtable_store(const QTable &t, int rank, double v)
{
    lds_u32_ptr p = (lds_u32_ptr)(uintptr_t)t.lane4;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    p[rank * 64] = (uint32_t)bits;
    p[(ROWS + rank) * 64] = (uint32_t)(bits >> 32);
}
// the quotient of the class in byte H of code word w
template <int ROWS, int H>
__device__ __forceinline__ double qtable_lookup(const QTable &t, uint32_t w)
{
    // result bytes, low to high: lane4's byte 0, w's byte H, zero, zero   (selector values 0-3: second operand, 4-7: first, 0x0c: zero)
    const uint32_t a = __builtin_amdgcn_perm(w, t.lane4, 0x0c0c0000u | ((4u + H) << 8));
    lds_u32_ptr p = (lds_u32_ptr)(uintptr_t)a;
    const uint32_t lo = p[0], hi = p[ROWS * 64];
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// phase 1, lane <-> SNP: r = the tile's dictionary rows (registers); every lane fills the rows of its own SNP, ILP at a time
template <int ILP, int ROWS>
__device__ __forceinline__ void coded_phase1_lane(const double (&r)[ROWS], const QTable &q, int nrows, float f_old)
{
    SnpState st;
    st.fd = (double)f_old;
    st.omf = 1.0 - st.fd;
    st.fd2 = 2.0 * st.fd;
    // (lanes whose SNP has fewer classes compute on whatever the unwritten rows hold: never looked up)
#pragma unroll
    for (int r0 = 0; r0 < ROWS; r0 += ILP) {
        if (r0 < nrows) {                                // wave-uniform
            double qv[ILP];
#pragma unroll
            for (int x = 0; x < ILP; ++x) {
                // (opaque to the compiler: else it hoists the three float -> double forms of every row out of the iteration
                // loop -- six registers per row instead of two, 202 VGPRs and two wavefronts per SIMD instead of four)
                double raw = r[r0 + x];
                asm volatile("" : "+v"(raw));
                qv[x] = class_quotient(raw, st);
            }
#pragma unroll
            for (int x = 0; x < ILP; ++x) qtable_store<ROWS>(q, r0 + x, qv[x]);
        }
    }
}

// phase 2: the serial float32 sum over the slab's individuals (all but `skip`), emMAF_cy.pyx:19-22; the quotients of a buffer of U
// quads are read from the table before the chain of that buffer starts.  cur = the first U code words (requested before phase 1).
template <int U, int ROWS>
__device__ __forceinline__ float coded_phase2(const QTable &q, gu32_ptr src, int nquads, int ncols, int skip, uint32_t (&cur)[U])
{
    const int last = nquads - 1;
    uint32_t nxt[U];
    float tmp = 0.0f;
    for (int q0 = 0; q0 < nquads; q0 += U) {
        if (q0 + U < nquads) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int qq = q0 + U + u;
                nxt[u] = src[(qq < last ? qq : last) * 64];
            }
        }
        double qv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            qv[u][0] = qtable_lookup<ROWS, 0>(q, cur[u]);
            qv[u][1] = qtable_lookup<ROWS, 1>(q, cur[u]);
            qv[u][2] = qtable_lookup<ROWS, 2>(q, cur[u]);
            qv[u][3] = qtable_lookup<ROWS, 3>(q, cur[u]);
        }
        const bool plain = 4 * (q0 + U) <= ncols && (skip < 4 * q0 || skip >= 4 * (q0 + U));
        if (plain) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int h = 0; h < 4; ++h) tmp = (float)__builtin_fma(0.5, qv[u][h], (double)tmp);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int col = 4 * (q0 + u) + h;
                    if (col < ncols && col != skip) tmp = (float)__builtin_fma(0.5, qv[u][h], (double)tmp);
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
    return tmp;
}

#ifdef WGS_EM_STATS
// (experiments: -DWGS_EM_STATS adds up over the wavefronts of em_coded_kernel the clock cycles [0] from the wavefront's start until the
// tile's row count is known, [1] until its dictionary rows have arrived, [2] in phase 1, [3] in phase 2, [4] finishing (divide, store,
// the tile's share of the sum of squares); [5] iterations, [6] wavefronts; printed and reset by wgs_debug_em_stats.)
__device__ unsigned long long g_em_stats[8];
#define EM_CLOCK(i) do { const unsigned long long now_ = clock64(); stat_[i] += now_ - mark_; mark_ = now_; } while (0)
#else
#define EM_CLOCK(i) do { } while (0)
#endif

template <int U, int ILP, int ROWS>
__device__ __forceinline__ float coded_iteration(const double (&r)[ROWS], const QTable &q, int nrows, gu32_ptr src, int nquads, int ncols, int skip, float f_old
#ifdef WGS_EM_STATS
                                                 , unsigned long long (&stat_)[8], unsigned long long &mark_
#endif
)
{
    const int last = nquads - 1;
    uint32_t cur[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = src[(u < last ? u : last) * 64];      // in flight during phase 1
    coded_phase1_lane<ILP, ROWS>(r, q, nrows, f_old);
#ifdef WGS_EM_STATS
    {
        const unsigned long long now_ = clock64();
        stat_[2] += now_ - mark_;
        mark_ = now_;
    }
#endif
    return coded_phase2<U, ROWS>(q, src, nquads, ncols, skip, cur);
}

template <int U, int ILP, int ROWS>
__global__ __launch_bounds__(64) void em_coded_kernel(const FitDesc *__restrict__ fits, int n_fits, int64_t m)
{
    extern __shared__ __align__(16) double qtab_all[];
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const int fit = (int)(j % (unsigned)n_fits);
    const int64_t tile = (int64_t)(j / (unsigned)n_fits) * 8 + xcd;
    const FitDesc fd = fits[fit];
    if (fd.state && *fd.state != EM_ACTIVE) return;
    const int lane = threadIdx.x;
    const int64_t row0 = tile * 64;
    if (row0 >= m) return;                       // wave-uniform; no barriers below
#ifdef WGS_EM_STATS
    unsigned long long stat_[8] = {0}, mark_ = clock64();
#endif
    const QTable q = qtable_of(qtab_all, lane);

    const int64_t my_row = row0 + lane;
    const int64_t my_row_c = my_row < m ? my_row : m - 1;
    float f_old = ((gf32_ptr)fd.f_old)[my_row_c];
    const int n_iter = fd.fuse == 2 ? 2 : 1;

    // one iteration's result: the new frequencies and this tile's share of sum (f_new - f_old)^2
    auto finish = [&](float tmp, int it) -> float {
        const float f_new = tmp / (float)fd.n_eff;           // emMAF_cy.pyx:23 (float32 divide)
        double sq = 0.0;
        if (my_row < m) {
            ((gf32_wptr)(it ? fd.f_new2 : fd.f_new))[my_row] = f_new;
            const float d = f_new - f_old;                    // emMAF_cy.pyx:31, float32
            sq = (double)(d * d);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
        if (lane == 0) (it ? fd.ssq_part2 : fd.ssq_part)[tile] = sq;
        return f_new;
    };

    // the first code words: in flight during phase 1
    const int nquads = fd.nquads;
    gu32_ptr src = (gu32_ptr)fd.lcodes + tile * nquads * 64 + lane;      // (device-global by construction: global_load, not flat_load)
    // rows of this tile: the most classes one of its 64 SNPs has in this slab (255: a SNP the encoder gave up on)
    int nrows;
    {
        unsigned mx = 0;                                       // one byte per WGS_ENC_MIN_SNPS SNPs of the tile
#pragma unroll
        for (int x = 0; x < WGS_TILE_ROWS_BYTES / 8; ++x) {
            const unsigned long long w = reinterpret_cast<const unsigned long long *>(fd.tile_rows + tile * WGS_TILE_ROWS_BYTES)[x];
#pragma unroll
            for (int k = 0; k < 8; ++k) mx = max(mx, (unsigned)((w >> (8 * k)) & 255u));
        }
        nrows = (int)__builtin_amdgcn_readfirstlane((int)mx);
    }
    EM_CLOCK(0);
    if (nrows > ROWS || nrows > fd.lrows) {
        // a SNP of this tile shows more classes in this slab than the table has rows (~1 % of the tiles, codes.hip):
        // the tile is swept from the float32 slab, term by term as em_sweep_kernel does
        const int npairs = fd.npairs;
        gf4_ptr gl = (gf4_ptr)fd.slab + tile * npairs * 64 + lane;
        for (int it = 0; it < n_iter; ++it) {
            SnpState st;
            st.fd = (double)f_old;
            st.omf = 1.0 - st.fd;
            st.fd2 = 2.0 * st.fd;
            float tmp = 0.0f;
            bool ok = true;
            for (int p = 0; p < npairs; ++p) {
                const f4 v = gl[(int64_t)p * 64];
                if (2 * p < fd.ncols && 2 * p != fd.skip) term_exact<true>(v.x, v.y, st, tmp, ok);
                if (2 * p + 1 < fd.ncols && 2 * p + 1 != fd.skip) term_exact<true>(v.z, v.w, st, tmp, ok);
            }
            f_old = finish(tmp, it);
        }
        return;
    }
    // the tile's rows of the slab's dictionary: requested once, kept in registers for every iteration of this sweep
    double r[ROWS];
    {
        gf64_ptr drow = (gf64_ptr) reinterpret_cast<const double *>(fd.ldict) + tile * fd.lrows * 64 + lane;
#pragma unroll
        for (int g4 = 0; g4 < ROWS / 4; ++g4) {              // (four rows at a time: a tile's richest SNP has ~19 classes in its slab,
            if (4 * g4 < nrows) {                            //  so eight at a time read a fifth more dictionary than needed; wave-uniform)
#pragma unroll
                for (int u = 0; u < 4; ++u) r[4 * g4 + u] = drow[(int64_t)(4 * g4 + u) * 64];
            }
        }
    }
#ifdef WGS_EM_STATS
    {
        double any = 0.0;                                    // (the rows' arrival, made visible to the clock)
#pragma unroll
        for (int x = 0; x < ROWS; x += 4)
            if (x < nrows) any += r[x];
        asm volatile("" : "+v"(any));
        EM_CLOCK(1);
    }
#endif
    for (int it = 0; it < n_iter; ++it) {
#ifdef WGS_EM_STATS
        const float tmp = coded_iteration<U, ILP, ROWS>(r, q, nrows, src, nquads, fd.ncols, fd.skip, f_old, stat_, mark_);
        EM_CLOCK(3);
        f_old = finish(tmp, it);
        EM_CLOCK(4);
#else
        const float tmp = coded_iteration<U, ILP, ROWS>(r, q, nrows, src, nquads, fd.ncols, fd.skip, f_old);     // (the second iteration finds the code words in cache)
        f_old = finish(tmp, it);
#endif
        // (the table of the next iteration is written by the same lanes that read this one's: program order suffices)
    }
#ifdef WGS_EM_STATS
    if (lane == 0 && (blockIdx.x & 127u) < 8u) {                // (a sample: atomics from 1.5 M wavefronts to seven addresses would BE the kernel's time)
#pragma unroll
        for (int i = 0; i < 5; ++i) atomicAdd(&g_em_stats[i], stat_[i]);
        atomicAdd(&g_em_stats[5], (unsigned long long)n_iter);
        atomicAdd(&g_em_stats[6], 1ull);
    }
#endif
}

// Leave-one-out batches through the codes: groups[g] = (first descriptor, count) into `fits`, all of one slab (as for
// em_sweep_group_kernel).  One wavefront takes a tile and walks the group's fits one after the other: the tile's dictionary rows
// are read once and stay in registers, the code words come from cache, and the chain of dependent requests a (fit, tile) pair
// starts with (descriptor -> state, table rows, frequencies, dictionary) is paid once per group instead of once per fit -- the
// re-fits are bound by instruction issue, and with one pair per wavefront (em_coded_kernel) a quarter of the time went to those
// waits (measured at 2M x 500, K=8: 526 ms of sweeps against 657 ms for the float32 group kernel; this kernel 469 ms: 938 vector
// instructions per (fit, tile) against 1596, ~80 % of the issue slots used; four table rows at a time instead of two: 480 ms).
// The next fit's frequencies are requested before the current fit's work.  One iteration per sweep (no fused second iteration: an
// iteration that turns out unneeded costs its full arithmetic here).
template <int U, int ILP, int ROWS>
__global__ __launch_bounds__(64) void em_coded_group_kernel(const FitDesc *__restrict__ fits, const int2 *__restrict__ groups, int n_groups, int64_t m)
{
    extern __shared__ __align__(16) double qtab_all[];
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const int grp = (int)(j % (unsigned)n_groups);
    const int64_t tile = (int64_t)(j / (unsigned)n_groups) * 8 + xcd;
    const int lane = threadIdx.x;
    const int64_t row0 = tile * 64;
    if (row0 >= m) return;                       // wave-uniform; no barriers below
    const int2 gd = groups[grp];
    const FitDesc *gf = fits + gd.x;
    const QTable q = qtable_of(qtab_all, lane);
    const int64_t my_row = row0 + lane;
    const int64_t my_row_c = my_row < m ? my_row : m - 1;
    // what the fits of a group share: the slab
    const int nquads = gf[0].nquads, ncols = gf[0].ncols, lrows = gf[0].lrows;
    gu32_ptr src = (gu32_ptr)gf[0].lcodes + tile * nquads * 64 + lane;
    int nrows;
    {
        unsigned mx = 0;
#pragma unroll
        for (int x = 0; x < WGS_TILE_ROWS_BYTES / 8; ++x) {
            const unsigned long long w = reinterpret_cast<const unsigned long long *>(gf[0].tile_rows + tile * WGS_TILE_ROWS_BYTES)[x];
#pragma unroll
            for (int k = 0; k < 8; ++k) mx = max(mx, (unsigned)((w >> (8 * k)) & 255u));
        }
        nrows = (int)__builtin_amdgcn_readfirstlane((int)mx);
    }
    auto finish = [&](const FitDesc *fd, float tmp, float f_old) {
        const float f_new = tmp / (float)fd->n_eff;          // emMAF_cy.pyx:23 (float32 divide)
        double sq = 0.0;
        if (my_row < m) {
            ((gf32_wptr)fd->f_new)[my_row] = f_new;
            const float d = f_new - f_old;                    // emMAF_cy.pyx:31, float32
            sq = (double)(d * d);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
        if (lane == 0) fd->ssq_part[tile] = sq;
    };
    if (nrows > ROWS || nrows > lrows) {
        // a tile richer than the table: every fit of the group from the float32 slab, term by term
        const int npairs = gf[0].npairs;
        gf4_ptr gl = (gf4_ptr)gf[0].slab + tile * npairs * 64 + lane;
        for (int f = 0; f < gd.y; ++f) {
            const FitDesc *fd = gf + f;
            if (fd->state && *fd->state != EM_ACTIVE) continue;
            const float f_old = ((gf32_ptr)fd->f_old)[my_row_c];
            const int skip = fd->skip;
            SnpState st;
            st.fd = (double)f_old;
            st.omf = 1.0 - st.fd;
            st.fd2 = 2.0 * st.fd;
            float tmp = 0.0f;
            bool ok = true;
            for (int p = 0; p < npairs; ++p) {
                const f4 v = gl[(int64_t)p * 64];
                if (2 * p < ncols && 2 * p != skip) term_exact<true>(v.x, v.y, st, tmp, ok);
                if (2 * p + 1 < ncols && 2 * p + 1 != skip) term_exact<true>(v.z, v.w, st, tmp, ok);
            }
            finish(fd, tmp, f_old);
        }
        return;
    }
    // the tile's rows of the slab's dictionary: requested once for all fits of the group
    // (Tried: dealing the tile's (SNP, class) pairs evenly over the lanes instead of every lane evaluating the tile's richest SNP's row
    // count -- 14 quotients per lane instead of 20 at 62 individuals per slab.  10 % fewer vector instructions, 2 % less time: fetching
    // the pair's frequency through LDS and 162 registers took back what the rows saved.  Not kept.)
    double r[ROWS];
    {
        gf64_ptr drow = (gf64_ptr) reinterpret_cast<const double *>(gf[0].ldict) + tile * lrows * 64 + lane;
#pragma unroll
        for (int g4 = 0; g4 < ROWS / 4; ++g4) {
            if (4 * g4 < nrows) {
#pragma unroll
                for (int u = 0; u < 4; ++u) r[4 * g4 + u] = drow[(int64_t)(4 * g4 + u) * 64];
            }
        }
    }
    float f_next = ((gf32_ptr)gf[0].f_old)[my_row_c];
    for (int f = 0; f < gd.y; ++f) {
        const FitDesc *fd = gf + f;
        const float f_old = f_next;
        if (f + 1 < gd.y) f_next = ((gf32_ptr)fd[1].f_old)[my_row_c];       // in flight during this fit's arithmetic
        if (fd->state && *fd->state != EM_ACTIVE) continue;                  // decided on the device: skip (wave-uniform)
#ifdef WGS_EM_STATS
        unsigned long long stat_[8] = {0}, mark_ = 0;              // (this kernel is not the one being timed)
        const float tmp = coded_iteration<U, ILP, ROWS>(r, q, nrows, src, nquads, ncols, fd->skip, f_old, stat_, mark_);
#else
        const float tmp = coded_iteration<U, ILP, ROWS>(r, q, nrows, src, nquads, ncols, fd->skip, f_old);
#endif
        finish(fd, tmp, f_old);
    }
}

// ssq[fit] = sum over tiles of the per-tile partials, in a fixed summation order (reproducible):
// stage 1, RED_CHUNKS workgroups per fit each reduce a contiguous slice of the partials into
// part2[fit][chunk]; stage 2, one wavefront per fit adds the RED_CHUNKS slice sums.
constexpr int RED_CHUNKS = 64;

__global__ __launch_bounds__(256) void ssq_reduce1_kernel(const FitDesc *__restrict__ fits, int64_t ntiles, double *__restrict__ part2, int second)
{
    __shared__ double red[256];
    const int fit = blockIdx.x / RED_CHUNKS, chunk = blockIdx.x % RED_CHUNKS;
    const FitDesc fd = fits[fit];
    const int64_t per = (ntiles + RED_CHUNKS - 1) / RED_CHUNKS;
    const int64_t t0 = chunk * per;
    int64_t t1 = t0 + per;
    if (t1 > ntiles) t1 = ntiles;
    double acc = 0.0;
    const double *part = second ? fd.ssq_part2 : fd.ssq_part;
    for (int64_t t = t0 + threadIdx.x; t < t1; t += 256) acc += part[t];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part2[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(64) void ssq_reduce2_kernel(const FitDesc *__restrict__ fits, const double *__restrict__ part2, int second)
{
    double v = part2[blockIdx.x * RED_CHUNKS + threadIdx.x];       // RED_CHUNKS == 64 lanes
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (threadIdx.x == 0) *(second ? fits[blockIdx.x].ssq2 : fits[blockIdx.x].ssq) = v;
}

// emMAF.py:22-23 on the device for the clear cases: the float64 sum S decides `diff < tole` unless it lies in
// the band [lo, hi) around tole^2 * m in which the reference's serial float32 sum may fall on either side
// (device.py: guard_band); those fits are parked as EM_UNDECIDED for the exact chain.  Fits that did not
// sweep (state != EM_ACTIVE) keep their state.  NaN never converges (NaN < tole is False).
__global__ void em_decide_kernel(const FitDesc *__restrict__ fits, int n_fits, double lo, double hi)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_fits) return;
    const FitDesc fd = fits[j];
    int32_t *st = const_cast<int32_t *>(fd.state);
    if (!st || *st != EM_ACTIVE) return;
    auto classify = [&](double s) { return (s != s || s >= hi) ? EM_ACTIVE : (s < lo ? EM_CONVERGED : EM_UNDECIDED); };
    const int a = classify(*fd.ssq);
    if (fd.fuse != 2) {
        *st = a;
        return;
    }
    // two iterations ran: the first decides first; only when it goes on does the second count
    if (a == EM_CONVERGED) *st = EM_CONVERGED_A;
    else if (a == EM_UNDECIDED) *st = EM_UNDECIDED_A;
    else *st = classify(*fd.ssq2);
}

__global__ void fill_kernel(float *p, int64_t count, float v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) p[i] = v;
}

// WGSassign.py:236-240: af[af < lo] = lo; af[af > hi] = hi  (float32 compares; NaN untouched)
__global__ void clamp_kernel(float *p, int64_t count, float lo, float hi)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        float v = p[i];
        if (v < lo) v = lo;
        if (v > hi) v = hi;
        p[i] = v;
    }
}

// emMAF_cy.pyx:30-31 continued over [0, m): res = res + (a-b)*(a-b), float32, index order.
// One wavefront: 64 lanes form the squares of 1024 elements in parallel (each square is an
// independent float32 sub+mul, identical to the serial code), lane 0 adds them in order.
__global__ __launch_bounds__(64) void rmse_chain_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                         int64_t m, float carry, float *out)
{
    __shared__ float sq[1024];
    const int lane = threadIdx.x;
    float res = carry;
    for (int64_t base = 0; base < m; base += 1024) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t i = base + k * 64 + lane;
            float v = 0.0f;
            if (i < m) {
                const float d = a[i] - b[i];
                v = d * d;
            }
            sq[k * 64 + lane] = v;
        }
        __syncthreads();
        if (lane == 0) {
            const int cnt = (m - base) < 1024 ? (int)(m - base) : 1024;
            for (int t = 0; t < cnt; ++t) res = res + sq[t];
        }
        __syncthreads();
    }
    if (lane == 0) *out = res;
}


// ------------------------------------------------------------------------------------------
// Block-parallel EXACT emulation of the serial float32 chain  res = res + d_i  (emMAF_cy.pyx:31).
//
// While the running sum stays in one binade [2^e, 2^(e+1)) it is M * u with u = 2^(e-23) and
// M an integer in [2^23, 2^24); adding d_i >= 0 and rounding to nearest-even adds the INTEGER
// r_i = RN(d_i / u) to M, and r_i does not depend on M except at exact ties (d_i/u = q + 1/2),
// where the parity of M + q decides.  So a block of elements acts on the chain as a function of
// the incoming parity only: p -> Delta_p, and such functions compose associatively
// ((A then B)(p) = A_p + B_{(p + A_p) & 1}) -- a parallel reduction.  The incoming binade is
// predicted from float64 prefix sums (the serial sum deviates from the exact one by a few per
// cent at most, so candidates e-1, e, e+1 cover it); a single wave then walks the blocks,
// applying Delta when the prediction holds and M + Delta stays below 2^24, and otherwise redoing
// that one block with the literal serial loop (binade crossings, the first blocks, NaN/Inf).
constexpr int RB = 4096;                       // elements per block
constexpr long long RSAT = 1ll << 40;          // "does not fit": forces the serial fallback

__device__ __forceinline__ float sqdiff(const float *a, const float *b, int64_t i)
{
    const float d = a[i] - b[i];               // emMAF_cy.pyx:31, float32 sub and mul
    return d * d;
}

// All chain kernels take a device array of jobs; blockIdx.y selects the job (one fit's pair of vectors), and
// every per-block array (S, expo, cand) is laid out [job][block].
__global__ __launch_bounds__(256) void rmse_block_sum_kernel(const ChainJob *__restrict__ jobs, int64_t m, double *__restrict__ S_all)
{
    __shared__ double red[256];
    const float *__restrict__ a = jobs[blockIdx.y].a, *__restrict__ b = jobs[blockIdx.y].b;
    double *__restrict__ S = S_all + (size_t)blockIdx.y * gridDim.x;
    const int64_t base = (int64_t)blockIdx.x * RB;
    double acc = 0.0;
    for (int k = 0; k < RB / 256; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i < m) acc += (double)sqdiff(a, b, i);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) S[blockIdx.x] = red[0];
}

// expo[b] = biased float32 exponent predicted for the running sum at the start of block b
// (0: unknown -> serial; -1: the block adds only zeros -> the sum passes through unchanged).
__global__ __launch_bounds__(1024) void rmse_predict_kernel(const double *__restrict__ S_all, int nblocks, const ChainJob *__restrict__ jobs,
                                                            int *__restrict__ expo_all)
{
    __shared__ double part[1024];
    const double *__restrict__ S = S_all + (size_t)blockIdx.x * nblocks;
    int *__restrict__ expo = expo_all + (size_t)blockIdx.x * nblocks;
    const float carry = jobs[blockIdx.x].carry_in;
    const int per = (nblocks + 1023) / 1024;
    const int b0 = threadIdx.x * per;
    double acc = 0.0;
    for (int j = 0; j < per; ++j) {
        const int b = b0 + j;
        if (b < nblocks) acc += S[b];
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {                    // 1024-entry exclusive scan, serial: ~1024 adds
        double run = (double)carry;
        for (int t = 0; t < 1024; ++t) {
            const double v = part[t];
            part[t] = run;
            run += v;
        }
    }
    __syncthreads();
    double run = part[threadIdx.x];
    for (int j = 0; j < per; ++j) {
        const int b = b0 + j;
        if (b >= nblocks) break;
        const double sb = S[b];
        int e = 0;
        if (sb == 0.0) {
            e = -1;
        } else if (run == run && sb == sb) {
            const unsigned int bits = __float_as_uint((float)run);
            e = (int)((bits >> 23) & 0xFF);
            if (e == 255) e = 0;
        }
        expo[b] = e;
        run += sb;
    }
}

// r(p) for one element d (float32 bits, d >= 0) on the grid of biased exponent be.
__device__ __forceinline__ void chain_step(unsigned int dbits, int be, long long &D0, long long &D1)
{
    const int bd_raw = (int)((dbits >> 23) & 0xFF);
    if (bd_raw == 255 || (dbits >> 31)) { D0 = RSAT; D1 = RSAT; return; }      // NaN / Inf (never negative)
    if ((dbits & 0x7FFFFFFF) == 0) return;                                      // + 0.0
    const int bd = bd_raw ? bd_raw : 1;
    const unsigned int md = (dbits & 0x7FFFFF) | (bd_raw ? 0x800000u : 0u);     // d = md * 2^(bd-150)
    const int sh = be - bd;                                                     // d / u = md * 2^-sh
    if (sh <= 0) {
        const long long r = sh < -16 ? RSAT : ((long long)md << (-sh));
        D0 += r; D1 += r;
    } else if (sh <= 24) {
        const unsigned int q = md >> sh, rem = md & ((1u << sh) - 1u), half = 1u << (sh - 1);
        if (rem > half) { D0 += q + 1; D1 += q + 1; }
        else if (rem < half) { D0 += q; D1 += q; }
        else {                                                                  // tie: to even of (M + q)
            D0 += q + (unsigned int)((D0 + q) & 1);                             // chain that entered with even M
            D1 += q + (unsigned int)((1 + D1 + q) & 1);                         // chain that entered with odd M
        }
    }                                                                           // sh >= 25: d < u/2, r = 0
    if (D0 > RSAT) D0 = RSAT;
    if (D1 > RSAT) D1 = RSAT;
}

// cand[b][c][p], c = 0,1,2 for biased exponents expo[b]-1, expo[b], expo[b]+1.
__global__ __launch_bounds__(256) void rmse_candidates_kernel(const ChainJob *__restrict__ jobs, int64_t m, const int *__restrict__ expo_all,
                                                             long long *__restrict__ cand_all)
{
    __shared__ long long sh[256][6];
    const float *__restrict__ a = jobs[blockIdx.y].a, *__restrict__ b = jobs[blockIdx.y].b;
    const int *__restrict__ expo = expo_all + (size_t)blockIdx.y * gridDim.x;
    long long *__restrict__ cand = cand_all + (size_t)blockIdx.y * gridDim.x * 6;
    const int e = expo[blockIdx.x];
    long long D[6] = {0, 0, 0, 0, 0, 0};
    if (e > 0) {
        const int64_t base = (int64_t)blockIdx.x * RB + (int64_t)threadIdx.x * (RB / 256);
        // the thread's sixteen pairs of values first, all requests in flight together (one pair at a time, each waited for before the
        // next was requested, the kernel read its 0.8 GB at 0.76 TB/s), then the chain steps
        float av[RB / 256], bv[RB / 256];
#pragma unroll
        for (int k = 0; k < RB / 256; ++k) {
            const int64_t at = base + k < m ? base + k : m - 1;
            av[k] = a[at], bv[k] = b[at];
        }
#pragma unroll
        for (int k = 0; k < RB / 256; ++k) {
            if (base + k < m) {
                const float d = av[k] - bv[k];                  // emMAF_cy.pyx:31, float32 sub and mul (sqdiff)
                const unsigned int dbits = __float_as_uint(d * d);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int be = e - 1 + c;
                    if (be >= 1 && be <= 254) chain_step(dbits, be, D[2 * c], D[2 * c + 1]);
                    else D[2 * c] = D[2 * c + 1] = RSAT;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) sh[threadIdx.x][j] = D[j];
    __syncthreads();
    // ordered composition: (A then B)(p) = A_p + B_{(p + A_p) & 1}
    for (int s = 1; s < 256; s <<= 1) {
        if ((threadIdx.x & (2 * s - 1)) == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const long long A0 = sh[threadIdx.x][2 * c], A1 = sh[threadIdx.x][2 * c + 1];
                const long long B0 = sh[threadIdx.x + s][2 * c], B1 = sh[threadIdx.x + s][2 * c + 1];
                long long R0 = A0 + ((A0 & 1) ? B1 : B0);
                long long R1 = A1 + (((1 + A1) & 1) ? B1 : B0);
                if (R0 > RSAT) R0 = RSAT;
                if (R1 > RSAT) R1 = RSAT;
                sh[threadIdx.x][2 * c] = R0;
                sh[threadIdx.x][2 * c + 1] = R1;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) cand[(int64_t)blockIdx.x * 6 + threadIdx.x] = sh[0][threadIdx.x];
}

// One wavefront walks the blocks in order.  All 64 lanes hold the same running value (uniform
// control flow), so the serial fallback of a block can stage its 4096 squares in LDS cooperatively.
__global__ __launch_bounds__(64) void rmse_walk_kernel(const ChainJob *__restrict__ jobs, int64_t m, const int *__restrict__ expo_all,
                                                       const long long *__restrict__ cand_all, int nblocks, float *out_all, int *n_serial_all)
{
    __shared__ __attribute__((aligned(16))) float sq[RB];
    const int lane = threadIdx.x;
    const float *__restrict__ a = jobs[blockIdx.x].a, *__restrict__ b = jobs[blockIdx.x].b;
    const int *__restrict__ expo = expo_all + (size_t)blockIdx.x * nblocks;
    const long long *__restrict__ cand = cand_all + (size_t)blockIdx.x * nblocks * 6;
    float *out = out_all + blockIdx.x;
    int *n_serial = n_serial_all ? n_serial_all + blockIdx.x : nullptr;
    // (the running value as BITS in a scalar register -- every lane holds the same one, and a step is integer arithmetic on it: said
    // this way the compiler keeps the steps on the scalar unit instead of hopping between the units for every comparison)
    unsigned rbits = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(jobs[blockIdx.x].carry_in));
    int serial = 0;
#ifdef WGS_WALK_STATS
    unsigned long long t_serial = 0, t_all0 = clock64();
#endif
    // Lane l fetches block l's exponent and six candidates for sixty-four blocks at once -- nothing of that depends on the running
    // value -- and a step takes its block's from that lane's registers (v_readlane).  Read where they are needed, they were two
    // dependent loads per step: 1070 cycles per step, two thirds of the walk (-DWGS_WALK_STATS).  (The values are made opaque to
    // the compiler, which otherwise re-issues the loads of this read-only memory inside the steps instead of keeping registers.)
    for (int blk0 = 0; blk0 < nblocks; blk0 += 64) {
      const int nb = nblocks - blk0 < 64 ? nblocks - blk0 : 64;
      const int mine = blk0 + lane < nblocks ? blk0 + lane : nblocks - 1;
      int my_e = expo[mine];
      unsigned my_lo[6], my_hi[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
          const unsigned long long v = (unsigned long long)cand[(int64_t)mine * 6 + c];
          my_lo[c] = (unsigned)v, my_hi[c] = (unsigned)(v >> 32);
      }
      asm volatile("" : "+v"(my_e));
#pragma unroll
      for (int c = 0; c < 6; ++c) asm volatile("" : "+v"(my_lo[c]), "+v"(my_hi[c]));
      for (int i = 0; i < nb; ++i) {
        const int blk = blk0 + i;
        const int e = __builtin_amdgcn_readlane(my_e, i);
        if (e == -1) continue;                                   // res + 0.0f == res
        const unsigned int bits = rbits;
        const int be = (int)((bits >> 23) & 0xFF);
        bool done = false;
        if (e > 0 && be >= 1 && be <= 254 && !(bits >> 31) && be >= e - 1 && be <= e + 1) {
            const unsigned int M = (bits & 0x7FFFFF) | 0x800000u;
            const int which = 2 * (be - e + 1) + (int)(M & 1);
            unsigned lo = 0, hi = 0;
            switch (which) {
                case 0: lo = (unsigned)__builtin_amdgcn_readlane((int)my_lo[0], i), hi = (unsigned)__builtin_amdgcn_readlane((int)my_hi[0], i); break;
                case 1: lo = (unsigned)__builtin_amdgcn_readlane((int)my_lo[1], i), hi = (unsigned)__builtin_amdgcn_readlane((int)my_hi[1], i); break;
                case 2: lo = (unsigned)__builtin_amdgcn_readlane((int)my_lo[2], i), hi = (unsigned)__builtin_amdgcn_readlane((int)my_hi[2], i); break;
                case 3: lo = (unsigned)__builtin_amdgcn_readlane((int)my_lo[3], i), hi = (unsigned)__builtin_amdgcn_readlane((int)my_hi[3], i); break;
                case 4: lo = (unsigned)__builtin_amdgcn_readlane((int)my_lo[4], i), hi = (unsigned)__builtin_amdgcn_readlane((int)my_hi[4], i); break;
                default: lo = (unsigned)__builtin_amdgcn_readlane((int)my_lo[5], i), hi = (unsigned)__builtin_amdgcn_readlane((int)my_hi[5], i); break;
            }
            const long long D = (long long)(((unsigned long long)hi << 32) | lo);
            if ((long long)M + D < (1ll << 24)) {
                const unsigned int M2 = (unsigned int)((long long)M + D);
                rbits = ((unsigned int)be << 23) | (M2 & 0x7FFFFF);
                done = true;
            }
        }
        if (!done) {                                             // literal serial loop for this block
            float res = __uint_as_float(rbits);
            ++serial;
#ifdef WGS_WALK_STATS
            const unsigned long long ts0 = clock64();
#endif
            const int64_t base = (int64_t)blk * RB;
            // the block's 4096 squares into LDS, sixteen pairs of loads in flight per lane (one pair at a time -- each waited for
            // before the next was issued -- this staging was 64 trips to memory per block, a third of the walk: -DWGS_WALK_STATS)
            for (int k0 = 0; k0 < RB / 64; k0 += 16) {
                float av[16], bv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int64_t idx = base + (int64_t)(k0 + u) * 64 + lane;
                    const int64_t at = idx < m ? idx : m - 1;
                    av[u] = a[at], bv[u] = b[at];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int64_t idx = base + (int64_t)(k0 + u) * 64 + lane;
                    const float d = av[u] - bv[u];              // emMAF_cy.pyx:31, float32 sub and mul (sqdiff)
                    sq[(k0 + u) * 64 + lane] = idx < m ? d * d : 0.0f;
                }
            }
            __syncthreads();
            const int cnt = (m - base) < RB ? (int)(m - base) : RB;
            // float32 += float32 in element order.  The adds are one dependent chain; the LDS reads are not: sixteen values per
            // batch of four 16-byte reads, requested while the sixteen before them are added (as chain_walk_kernel does it).
            int t = 0;
            if (cnt >= 16) {                                      // (the next sixteen are requested before the current sixteen are added)
                const float4 *q = reinterpret_cast<const float4 *>(sq);
                float4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
                for (; t + 32 <= cnt; t += 16) {
                    const float4 w0 = q[t / 4 + 4], w1 = q[t / 4 + 5], w2 = q[t / 4 + 6], w3 = q[t / 4 + 7];
                    res = res + v0.x; res = res + v0.y; res = res + v0.z; res = res + v0.w;
                    res = res + v1.x; res = res + v1.y; res = res + v1.z; res = res + v1.w;
                    res = res + v2.x; res = res + v2.y; res = res + v2.z; res = res + v2.w;
                    res = res + v3.x; res = res + v3.y; res = res + v3.z; res = res + v3.w;
                    v0 = w0, v1 = w1, v2 = w2, v3 = w3;
                }
                res = res + v0.x; res = res + v0.y; res = res + v0.z; res = res + v0.w;
                res = res + v1.x; res = res + v1.y; res = res + v1.z; res = res + v1.w;
                res = res + v2.x; res = res + v2.y; res = res + v2.z; res = res + v2.w;
                res = res + v3.x; res = res + v3.y; res = res + v3.z; res = res + v3.w;
                t += 16;
            }
            for (; t < cnt; ++t) res = res + sq[t];
            rbits = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(res));
            __syncthreads();
#ifdef WGS_WALK_STATS
            t_serial += clock64() - ts0;
#endif
        }
      }
    }
#ifdef WGS_WALK_STATS
    if (lane == 0 && blockIdx.x == 0) printf("[walk stats] %d blocks, %d serial; cycles: all %llu, serial blocks %llu\n", nblocks, serial, clock64() - t_all0, t_serial);
#endif
    if (lane == 0) {
        *out = __uint_as_float(rbits);
        if (n_serial) *n_serial = serial;
    }
}


// ------------------------------------------------------------------------------------------
// Observed Fisher information / effective sample size (--ne_obs; fisher_cy.pyx:12-65).  Same
// shape as the EM update: per SNP a serial float32 sum over the individuals of a population,
// so the same lane<->SNP sweep over the tile-interleaved slab applies.
//
// fisher_cy.pyx:21-27, all locals float32, literals double (one rounding per operation):
//   g2 = (float)((1.0 - g0) - g1)
//   u  = (float)((g0*(1.0-th))*(1.0-th) + ((g1*2.0)*th)*(1.0-th) + (double)((g2*th)*th))
//   n1 = (float)(2.0*((double)(g0+g2) - 2.0*g1));  n2 = (float)((double)(th*n1) + 2.0*(double)(g1-g0))
//   term = -((n1/u) - (n2/u)*(n2/u))                                   (float32 divides)
__device__ __forceinline__ float fisher_term(float g0, float g1, float th, double thd, double omt)
{
    const double g0d = (double)g0, g1d = (double)g1;
    const float g2 = (float)((1.0 - g0d) - g1d);
    const float u = (float)((((g0d * omt) * omt) + (((g1d * 2.0) * thd) * omt)) + (double)((g2 * th) * th));
    const float n1 = (float)(2.0 * ((double)(g0 + g2) - (2.0 * g1d)));
    const float n2 = (float)((double)(th * n1) + (2.0 * (double)(g1 - g0)));
    const float q = n2 / u;
    return -((n1 / u) - (q * q));
}

__global__ __launch_bounds__(WAVES * 64) void fisher_pop_kernel(const FisherDesc *__restrict__ descs, int n_desc, int64_t m)
{
    const FisherDesc fd = descs[blockIdx.x % (unsigned)n_desc];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)(blockIdx.x / (unsigned)n_desc) * WAVES + wave;
    const int64_t row0 = tile * 64;
    if (row0 >= m) return;
    const int64_t my_row = row0 + lane;
    const int64_t my_row_c = my_row < m ? my_row : m - 1;
    const float th = ((gf32_ptr)fd.th)[my_row_c];
    const double thd = (double)th, omt = 1.0 - thd;
    gf4_ptr src = (gf4_ptr)fd.slab + tile * fd.npairs * 64 + lane;
    float term_sum = 0.0f;
    for (int p = 0; p < fd.npairs; ++p) {
        const f4 v = __builtin_nontemporal_load(src + p * 64);
        term_sum = term_sum + fisher_term(v.x, v.y, th, thd, omt);                        // fisher_cy.pyx:28
        if (2 * p + 1 < fd.ncols) term_sum = term_sum + fisher_term(v.z, v.w, th, thd, omt);
    }
    if (my_row < m) {
        ((gf32_wptr)fd.f_out)[my_row] = term_sum;                                         // 0.0f + term_sum (fisher_cy.pyx:29)
        ((gf32_wptr)fd.ne_out)[my_row] = (float)(((0.5 * (double)term_sum) * thd) * omt);  // fisher_cy.pyx:37-38
    }
}

// Per-site effective-sample-size terms of `count` individuals (fisher_cy.pyx:41-65: f_ind[s] = term,
// ne_ind[s] = 0.5 * f_ind[s] * a * (1 - a)) written as float32 rows out[j][s], so that the host can
// take np.mean of each row exactly as fisher.py:59 does.  Thread <-> (SNP, individual): slab reads
// and row writes are coalesced along SNPs.
__global__ __launch_bounds__(256) void fisher_ind_sites_kernel(const float2 *__restrict__ slab2, const int32_t *__restrict__ cols,
                                                              const float *__restrict__ th_vec, float *__restrict__ out, int64_t m,
                                                              int npairs, int count)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (s >= m || j >= count) return;
    const int c = cols[j];
    const float th = th_vec[s];
    const double thd = (double)th, omt = 1.0 - thd;
    const float2 g = slab2[((((s >> 6) * npairs + (c >> 1)) << 6) + (s & 63)) * 2 + (c & 1)];
    const float term = fisher_term(g.x, g.y, th, thd, omt);
    out[(int64_t)j * m + s] = (float)(((0.5 * (double)term) * thd) * omt);
}

// ---- np.mean of a float32 row, the way NumPy forms it (fisher.py:59) ---------------------------------------------------
// NumPy's add.reduce over a contiguous float32 vector adds, chunk after chunk of 8192 elements (api.hip: PairwisePlan), the
// chunk's pairwise sum (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum, restated) to the running total: more
// than 128 elements split into halves, the left one n/2 rounded down to a multiple of 8; a piece of 8..128 elements is summed with EIGHT accumulators r[i % 8], combined as ((r0+r1)+(r2+r3)) +
// ((r4+r5)+(r6+r7)), the n % 8 leftovers added one by one; fewer than 8 elements are added in order.  The pieces
// ("leaves") and the order in which their sums are added depend only on the length, so the host lays them out once
// (api.hip: PairwisePlan); here 8 lanes form a leaf -- lane r IS accumulator r, so the loads are coalesced -- and one
// thread per row then adds the leaf sums in the plan's order (a stack program: op >= 0 pushes leaf op, -1 adds the two
// on top).  np.mean divides the float32 sum by the count in float64 and stores float32.
__global__ __launch_bounds__(256) void pairwise_leaf_kernel(const float *__restrict__ rows, int64_t m, const int64_t *__restrict__ leaf_lo,
                                                           const int32_t *__restrict__ leaf_len, int nleaf, float *__restrict__ leaf_sums)
{
    const int64_t leaf = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int r = threadIdx.x & 7, j = blockIdx.y;
    if (leaf >= nleaf) return;                               // the 8 lanes of a leaf leave together
    const float *a = rows + (int64_t)j * m + leaf_lo[leaf];
    const int n = leaf_len[leaf];
    float res;
    if (n < 8) {
        res = -0.0f;
        if (r == 0)
            for (int i = 0; i < n; ++i) res = res + a[i];
    } else {
        const int n8 = n - (n % 8);
        float acc = a[r];
        for (int i = 8; i < n8; i += 8) acc = acc + a[i + r];
        acc = acc + __shfl_xor(acc, 1);                      // r0+r1 | r2+r3 | r4+r5 | r6+r7   (addition commutes: both lanes of a
        acc = acc + __shfl_xor(acc, 2);                      // (r0+r1)+(r2+r3) | (r4+r5)+(r6+r7)  pair hold the same value)
        res = acc + __shfl_xor(acc, 4);
        if (r == 0)
            for (int i = n8; i < n; ++i) res = res + a[i];
    }
    if (r == 0) leaf_sums[(int64_t)j * nleaf + leaf] = res;
}

// carry (may be NULL): the running totals after the SNP shards before this one -- the program then adds every chunk to it.
// m > 0: out = np.mean's float32(float64(total) / m); m == 0: out = the running total itself (a later shard continues).
__global__ __launch_bounds__(64) void pairwise_combine_kernel(const float *__restrict__ leaf_sums, int nleaf, const int32_t *__restrict__ prog,
                                                             int nprog, int count, int64_t m, const float *__restrict__ carry,
                                                             float *__restrict__ means)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const float *ls = leaf_sums + (int64_t)j * nleaf;
    float stack[64];                                         // depth <= log2(8192 / 64) + 3
    int sp = 0;
    if (carry) stack[sp++] = carry[j];
    for (int p = 0; p < nprog; ++p) {
        const int op = prog[p];
        if (op >= 0) {
            stack[sp++] = ls[op];
        } else {
            --sp;
            stack[sp - 1] = stack[sp - 1] + stack[sp];
        }
    }
    means[j] = m > 0 ? (float)((double)stack[0] / (double)m) : stack[0];
}

// Test hook: div_exact against the compiler's IEEE divide on operands shaped like the EM term's
// (den = a float32 sum widened to double, num = p1 + 2*p2 with float32 p1, p2 >= 0, num <~ 2 den,
// plus den = 0 and tiny/huge magnitudes).  Counts bitwise mismatches.
__device__ __forceinline__ unsigned int mix32(unsigned int x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void div_check_kernel(unsigned long long seed, unsigned long long per_thread, unsigned long long *mismatch)
{
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0;
    unsigned int st = mix32((unsigned int)(tid ^ seed) + 0x9e3779b9u * (unsigned int)(seed >> 32));
    for (unsigned long long it = 0; it < per_thread; ++it) {
        st = mix32(st + 0x6d2b79f5u);
        const unsigned int a = st;
        st = mix32(st + 0x6d2b79f5u);
        const unsigned int b = st;
        st = mix32(st + 0x6d2b79f5u);
        const unsigned int c = st;
        // float32 values with exponents spread over [2^-100, 2^1]: sign 0, exponent 27..128
        const float p0 = __uint_as_float((((a >> 23) % 102u + 27u) << 23) | (a & 0x7FFFFFu));
        const float p1 = (c & 7u) == 0 ? 0.0f : __uint_as_float((((b >> 23) % 102u + 27u) << 23) | (b & 0x7FFFFFu));
        const float p2 = (c & 56u) == 0 ? 0.0f : __uint_as_float((((c >> 23) % 102u + 27u) << 23) | (c & 0x7FFFFFu));
        const float s = (c & 0x3FFu) == 1 ? 0.0f : (p0 + p1) + p2;
        const double num = __builtin_fma(2.0, (double)p2, (double)p1), den = (double)s;
        const double q1 = div_exact(num, den), q2 = num / den;
        const bool same_bits = __double_as_longlong(q1) == __double_as_longlong(q2);
        const bool both_nan = q1 != q1 && q2 != q2;
        bad += !(same_bits || both_nan);
    }
    if (bad) atomicAdd(mismatch, bad);
}

// Test hook: the largest relative error of refined_rcp over EVERY float32 mantissa (2^23 denominators) at the given
// binary exponent, against the correctly rounded 1/den (IEEE divide), as the ordered integer image of the double.
__global__ void rcp_error_kernel(int exponent, unsigned long long *max_bits)
{
    const unsigned int mant = blockIdx.x * blockDim.x + threadIdx.x;      // 2^23 threads
    if (mant >= (1u << 23)) return;
    const double den = __builtin_ldexp((double)__uint_as_float(0x3F800000u | mant), exponent);
    const double exact = 1.0 / den;
    const double err = __builtin_fabs(refined_rcp(den) - exact) / exact;
    atomicMax(max_bits, (unsigned long long)__double_as_longlong(err));
}

}  // namespace

int launch_rcp_error(wgs_ctx *ctx, int exponent, unsigned long long *d_max_bits)
{
    hipLaunchKernelGGL(rcp_error_kernel, dim3((1u << 23) / 256), dim3(256), 0, ctx->stream, exponent, d_max_bits);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Fits of different populations: every slab byte is used once -> nontemporal loads, fit index fastest.
// (U = 4 pairs per register buffer measured best of {2, 4, 8}; fits that SHARE a slab go through
// launch_em_sweep_groups.)
int launch_em_sweep(wgs_ctx *ctx, const FitDesc *d_descs, int32_t n_fits, int64_t m, int mode)
{
    if (n_fits <= 0 || m <= 0) return 0;
    const int64_t tiles = (m + 63) / 64;
    const int64_t blocks = (tiles + WAVES - 1) / WAVES * n_fits;
    WGS_REQUIRE(blocks < (1ll << 31), "em sweep: %lld workgroups exceed one launch; split the fit batch", (long long)blocks);
    dim3 grid((unsigned)blocks);
    if (mode == WGS_MODE_EXACT)
        hipLaunchKernelGGL((em_sweep_kernel<WGS_MODE_EXACT, 4>), grid, dim3(WAVES * 64), 0, ctx->stream, d_descs, n_fits, m);
    else
        hipLaunchKernelGGL((em_sweep_kernel<WGS_MODE_FAST, 4>), grid, dim3(WAVES * 64), 0, ctx->stream, d_descs, n_fits, m);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Where the dynamic LDS of a kernel without static LDS begins (the address the coded EM kernels take to be 0).
__global__ void dyn_lds_base_probe_kernel(uint32_t *out)
{
    extern __shared__ __align__(16) double probe_dyn[];
    if (threadIdx.x == 0) {
        probe_dyn[0] = 1.0;                  // (the allocation must be real)
        out[0] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)probe_dyn;
    }
}

// true when the coded EM sweeps may run on this context: their quotient table addresses LDS from 0 (qtable_of).  Probed once.
bool em_coded_usable(wgs_ctx *ctx)
{
    if (ctx->dyn_lds_base < 0) {
        uint32_t *d = nullptr, h = 0xffffffffu;
        if (hipMalloc(&d, sizeof(uint32_t)) == hipSuccess) {
            hipLaunchKernelGGL(dyn_lds_base_probe_kernel, dim3(1), dim3(64), 24 * 512, ctx->stream, d);
            if (hipMemcpyAsync(&h, d, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                h = 0xffffffffu;
            (void)hipFree(d);
        }
        (void)hipGetLastError();
        ctx->dyn_lds_base = (int64_t)h;
    }
    return ctx->dyn_lds_base == 0;
}

int launch_em_coded(wgs_ctx *ctx, const FitDesc *d_descs, int32_t n_fits, int64_t m, int rows)
{
    if (n_fits <= 0 || m <= 0) return 0;
    WGS_REQUIRE(rows >= 8 && rows <= 64 && rows % 8 == 0, "em sweep: %d table rows", rows);
    const int64_t tiles = (m + 63) / 64;
    const int64_t tgroups = (tiles + 7) / 8 * 8;                    // the XCD-aware order covers whole groups of 8
    const int64_t blocks = tgroups * n_fits;
    WGS_REQUIRE(blocks < (1ll << 31), "em sweep: %lld workgroups exceed one launch; split the fit batch", (long long)blocks);
    const size_t lds = (size_t)rows * 512 + (size_t)wgs_hook("em_coded_extra_lds");   // (test hook: how the sweep depends on wavefronts per CU)
#define WGS_EMC(RW) hipLaunchKernelGGL((em_coded_kernel<4, 2, RW>), dim3((unsigned)blocks), dim3(64), lds, ctx->stream, d_descs, n_fits, m)
    switch (rows) {
        case 8: WGS_EMC(8); break;
        case 16: WGS_EMC(16); break;
        case 24: WGS_EMC(24); break;
        case 32: WGS_EMC(32); break;
        case 40: WGS_EMC(40); break;
        case 48: WGS_EMC(48); break;
        case 56: WGS_EMC(56); break;
        default: WGS_EMC(64); break;
    }
#undef WGS_EMC
    HIP_TRY(hipGetLastError());
#ifdef WGS_EM_STATS
    {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        unsigned long long st[8] = {0}, zero[8] = {0};
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_em_stats), sizeof st);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_em_stats), zero, sizeof zero);
        const double w = st[6] ? (double)st[6] : 1.0;
        fprintf(stderr, "[em stats] %llu wavefronts, %.2f iterations each; cycles per wavefront: until the row count %.0f, until the dictionary rows %.0f, phase 1 %.0f, "
                "phase 2 %.0f, finishing %.0f\n", st[6], st[5] / w, st[0] / w, st[1] / w, st[2] / w, st[3] / w, st[4] / w);
    }
#endif
    return 0;
}

// d_groups: n_groups (first, count) pairs into d_descs, every group's fits on one slab (leave-one-out batches), through the codes
int launch_em_coded_groups(wgs_ctx *ctx, const FitDesc *d_descs, const int32_t *d_groups, int32_t n_groups, int64_t m, int rows)
{
    if (n_groups <= 0 || m <= 0) return 0;
    WGS_REQUIRE(rows >= 8 && rows <= 64 && rows % 8 == 0, "em sweep: %d table rows", rows);
    const int64_t tiles = (m + 63) / 64;
    const int64_t tgroups = (tiles + 7) / 8 * 8;
    const int64_t blocks = tgroups * n_groups;
    WGS_REQUIRE(blocks < (1ll << 31), "em sweep: %lld workgroups exceed one launch; split the fit batch", (long long)blocks);
    const int2 *g = reinterpret_cast<const int2 *>(d_groups);
    const size_t lds = (size_t)rows * 512;
#define WGS_EMG(RW) hipLaunchKernelGGL((em_coded_group_kernel<4, 2, RW>), dim3((unsigned)blocks), dim3(64), lds, ctx->stream, d_descs, g, n_groups, m)
    switch (rows) {
        case 8: WGS_EMG(8); break;
        case 16: WGS_EMG(16); break;
        case 24: WGS_EMG(24); break;
        case 32: WGS_EMG(32); break;
        case 40: WGS_EMG(40); break;
        case 48: WGS_EMG(48); break;
        case 56: WGS_EMG(56); break;
        default: WGS_EMG(64); break;
    }
#undef WGS_EMG
    HIP_TRY(hipGetLastError());
    return 0;
}

int em_fits_per_group(void) { return FG; }

// d_groups: n_groups (first, count) pairs into d_descs, every group's fits on one slab (leave-one-out batches)
int launch_em_sweep_groups(wgs_ctx *ctx, const FitDesc *d_descs, const int32_t *d_groups, int32_t n_groups, int64_t m, int mode)
{
    if (n_groups <= 0 || m <= 0) return 0;
    const int64_t tiles = (m + 63) / 64;
    const int64_t tgroups = ((tiles + WAVES - 1) / WAVES + 7) / 8 * 8;          // the XCD-aware order covers whole groups of 8
    const int64_t blocks = tgroups * n_groups;
    WGS_REQUIRE(blocks < (1ll << 31), "em sweep: %lld workgroups exceed one launch; split the fit batch", (long long)blocks);
    const int2 *g = reinterpret_cast<const int2 *>(d_groups);
    if (mode == WGS_MODE_EXACT)
        hipLaunchKernelGGL((em_sweep_group_kernel<WGS_MODE_EXACT, 4>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, ctx->stream, d_descs, g, n_groups, m);
    else
        hipLaunchKernelGGL((em_sweep_group_kernel<WGS_MODE_FAST, 4>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, ctx->stream, d_descs, g, n_groups, m);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ssq_reduce_chunks(void) { return RED_CHUNKS; }

// part2: n_fits * ssq_reduce_chunks() doubles of device scratch
int launch_div_check(wgs_ctx *ctx, unsigned long long seed, unsigned long long per_thread, unsigned long long *d_mismatch)
{
    hipLaunchKernelGGL(div_check_kernel, dim3(4096), dim3(256), 0, ctx->stream, seed, per_thread, d_mismatch);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_fisher_pop(wgs_ctx *ctx, const FisherDesc *d_descs, int32_t n_desc, int64_t m)
{
    if (n_desc <= 0 || m <= 0) return 0;
    const int64_t blocks = ((wgs_ntiles(m) + WAVES - 1) / WAVES) * n_desc;
    WGS_REQUIRE(blocks < (1ll << 31), "fisher sweep: too many workgroups");
    hipLaunchKernelGGL(fisher_pop_kernel, dim3((unsigned)blocks), dim3(WAVES * 64), 0, ctx->stream, d_descs, n_desc, m);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_fisher_ind_sites(wgs_ctx *ctx, const float4 *slab, const int32_t *d_cols, const float *th, float *d_out, int64_t m,
                            int npairs, int count)
{
    if (m <= 0 || count <= 0) return 0;
    dim3 grid((unsigned)((m + 255) / 256), (unsigned)count);
    hipLaunchKernelGGL(fisher_ind_sites_kernel, grid, dim3(256), 0, ctx->stream, reinterpret_cast<const float2 *>(slab), d_cols, th,
                       d_out, m, npairs, count);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_pairwise_mean(wgs_ctx *ctx, const float *d_rows, int count, int64_t m, int64_t divide_by, const int64_t *d_leaf_lo,
                         const int32_t *d_leaf_len, int nleaf, const int32_t *d_prog, int nprog, float *d_leaf_sums, const float *d_carry,
                         float *d_means)
{
    if (count <= 0 || m <= 0) return 0;
    dim3 grid((unsigned)(((int64_t)nleaf * 8 + 255) / 256), (unsigned)count);
    hipLaunchKernelGGL(pairwise_leaf_kernel, grid, dim3(256), 0, ctx->stream, d_rows, m, d_leaf_lo, d_leaf_len, nleaf, d_leaf_sums);
    hipLaunchKernelGGL(pairwise_combine_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, ctx->stream, d_leaf_sums, nleaf, d_prog, nprog,
                       count, divide_by, d_carry, d_means);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_ssq_reduce(wgs_ctx *ctx, const FitDesc *d_descs, int32_t n_fits, int64_t m, double *part2, int second)
{
    if (n_fits <= 0 || m <= 0) return 0;
    hipLaunchKernelGGL(ssq_reduce1_kernel, dim3((unsigned)n_fits * RED_CHUNKS), dim3(256), 0, ctx->stream, d_descs, wgs_ntiles(m), part2, second);
    hipLaunchKernelGGL(ssq_reduce2_kernel, dim3((unsigned)n_fits), dim3(64), 0, ctx->stream, d_descs, part2, second);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_em_decide(wgs_ctx *ctx, const FitDesc *d_descs, int32_t n_fits, double lo, double hi)
{
    if (n_fits <= 0) return 0;
    hipLaunchKernelGGL(em_decide_kernel, dim3((unsigned)((n_fits + 255) / 256)), dim3(256), 0, ctx->stream, d_descs, n_fits, lo, hi);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_fill(wgs_ctx *ctx, float *p, int64_t count, float v)
{
    if (count <= 0) return 0;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p, count, v);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_clamp(wgs_ctx *ctx, float *p, int64_t count, float lo, float hi)
{
    if (count <= 0) return 0;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p, count, lo, hi);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_rmse_chain_serial(wgs_ctx *ctx, const float *a, const float *b, int64_t m, float carry_in, float *d_out)
{
    hipLaunchKernelGGL(rmse_chain_kernel, dim3(1), dim3(64), 0, ctx->stream, a, b, m, carry_in, d_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t rmse_chain_workspace_bytes(int64_t m)
{
    const size_t nb = (size_t)((m + RB - 1) / RB);
    return ((nb * (sizeof(double) + sizeof(int) + 6 * sizeof(long long)) + 7) & ~(size_t)7) + 64;   // + one ChainJob
}

// The running values after the preceding SNP shard (broadcast into `carry`) become the jobs' starting values.
__global__ void chain_set_carry_kernel(ChainJob *__restrict__ jobs, const float *__restrict__ carry, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) jobs[j].carry_in = carry[j];
}

int launch_chain_set_carry(wgs_ctx *ctx, ChainJob *d_jobs, const float *d_carry, int n_jobs)
{
    if (n_jobs <= 0) return 0;
    hipLaunchKernelGGL(chain_set_carry_kernel, dim3((unsigned)((n_jobs + 255) / 256)), dim3(256), 0, ctx->stream, d_jobs, d_carry, n_jobs);
    HIP_TRY(hipGetLastError());
    return 0;
}

// work: n_jobs * rmse_chain_workspace_bytes(m) bytes of device memory (the first sizeof(ChainJob) * n_jobs bytes
// of each call's job table live at its end); d_serial (may be null) receives per job the number of blocks that
// took the literal serial loop.
int launch_rmse_chain_batch(wgs_ctx *ctx, const ChainJob *d_jobs, int n_jobs, int64_t m, float *d_out, void *work, int *d_serial)
{
    if (n_jobs <= 0) return 0;
    WGS_REQUIRE(m > 0, "empty chain");
    WGS_REQUIRE(n_jobs <= 65535, "too many convergence chains in one batch");
    const int nb = (int)((m + RB - 1) / RB);
    long long *cand = reinterpret_cast<long long *>(work);
    double *S = reinterpret_cast<double *>(cand + (size_t)n_jobs * nb * 6);
    int *expo = reinterpret_cast<int *>(S + (size_t)n_jobs * nb);
    hipLaunchKernelGGL(rmse_block_sum_kernel, dim3(nb, n_jobs), dim3(256), 0, ctx->stream, d_jobs, m, S);
    hipLaunchKernelGGL(rmse_predict_kernel, dim3(n_jobs), dim3(1024), 0, ctx->stream, S, nb, d_jobs, expo);
    hipLaunchKernelGGL(rmse_candidates_kernel, dim3(nb, n_jobs), dim3(256), 0, ctx->stream, d_jobs, m, expo, cand);
    hipLaunchKernelGGL(rmse_walk_kernel, dim3(n_jobs), dim3(64), 0, ctx->stream, d_jobs, m, expo, cand, nb, d_out, d_serial);
    HIP_TRY(hipGetLastError());
    return 0;
}

// One chain; the job record is kept in the 64 spare bytes at the end of `work` (rmse_chain_workspace_bytes).
int launch_rmse_chain(wgs_ctx *ctx, const float *a, const float *b, int64_t m, float carry_in, float *d_out, void *work,
                      int *d_serial)
{
    if (m <= 0) {
        hipLaunchKernelGGL(rmse_chain_kernel, dim3(1), dim3(64), 0, ctx->stream, a, b, m, carry_in, d_out);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    ChainJob job{a, b, carry_in};
    ChainJob *d_job = reinterpret_cast<ChainJob *>(reinterpret_cast<char *>(work) + rmse_chain_workspace_bytes(m) - 64);
    HIP_TRY(hipMemcpyAsync(d_job, &job, sizeof job, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));        // `job` is a local
    return launch_rmse_chain_batch(ctx, d_job, 1, m, d_out, work, d_serial);
}
