// Native RCCL communicator: the one collective of the SNP-sharded path (sum all-reduce of a few
// float64 over xGMI) without any tensor framework.  librccl is dlopen'ed on first use, so the
// library has no link-time dependency on it and single-GPU use never loads it.
//
// Bootstrap: rank 0 obtains the 128-byte ncclUniqueId (wgs_comm_unique_id), the host side hands it
// to the other ranks (wgsassign_amd/comm.py does that over a TCP socket on MASTER_ADDR), every
// rank calls wgs_comm_init.  The all-reduce is enqueued on the context's HIP stream, i.e. behind
// the EM sweep that produced the sums.
//
// Every collective is SELF-CHECKING (round 5).  The ranks of the sharded path must issue the same sequence of collectives with
// the same meaning; what they issue is decided by host code that -- in principle -- could be steered by rank-local state (a cost
// model, the arrival of a helper thread's allocation).  A rank that falls out of step used to produce wrong numbers silently (its
// sums were added to the sums of a different iteration).  Now every collective carries, in the SAME RCCL call as its payload, a
// row of eight float64 per rank -- {sequence number of the collective on this communicator, opcode, generation, iteration, two
// shape words, payload size, one free word that is not compared} -- written by the rank into its own row of a world x 8 table
// behind the payload (zeros elsewhere), so that the sum all-reduce all-gathers the rows; a one-wavefront kernel behind the
// collective compares every rank's row with this rank's and, on a difference, records both rows in page-locked host memory.  The
// host looks at that record wherever it synchronises anyway (wgs_comm_check) and fails with a message that names both tuples.  A
// broadcast carries the root's row behind its payload and every receiver compares.  Over a host-backed communicator the same rows
// pass through the caller's all-reduce function and are compared on the host.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "common.h"

namespace {

typedef struct { char internal[128]; } rcclUniqueId;
typedef void *rcclComm_t;
// values from rccl.h (ncclDataType_t / ncclRedOp_t)
constexpr int kFloat64 = 8, kUint8 = 1, kSum = 0;

struct Api {
    void *handle = nullptr;
    int (*GetUniqueId)(rcclUniqueId *) = nullptr;
    int (*CommInitRank)(rcclComm_t *, int, rcclUniqueId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    // what RCCL itself says about a communicator (optional symbols: a librccl without them only loses the cross-check)
    int (*CommCount)(rcclComm_t, int *) = nullptr;
    int (*CommUserRank)(rcclComm_t, int *) = nullptr;
    int (*CommCuDevice)(rcclComm_t, int *) = nullptr;
};

Api *api()
{
    static Api a;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle) break;
        }
        if (a.handle) {
            a.GetUniqueId = (int (*)(rcclUniqueId *))dlsym(a.handle, "ncclGetUniqueId");
            a.CommInitRank = (int (*)(rcclComm_t *, int, rcclUniqueId, int))dlsym(a.handle, "ncclCommInitRank");
            a.AllReduce = (int (*)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t))dlsym(a.handle, "ncclAllReduce");
            a.Broadcast = (int (*)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t))dlsym(a.handle, "ncclBroadcast");
            a.CommDestroy = (int (*)(rcclComm_t))dlsym(a.handle, "ncclCommDestroy");
            a.GetErrorString = (const char *(*)(int))dlsym(a.handle, "ncclGetErrorString");
            a.CommCount = (int (*)(rcclComm_t, int *))dlsym(a.handle, "ncclCommCount");
            a.CommUserRank = (int (*)(rcclComm_t, int *))dlsym(a.handle, "ncclCommUserRank");
            a.CommCuDevice = (int (*)(rcclComm_t, int *))dlsym(a.handle, "ncclCommCuDevice");
        }
    }
    if (!a.handle || !a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.Broadcast || !a.CommDestroy) return nullptr;
    return &a;
}

}  // namespace

struct wgs_comm {
    wgs_ctx *ctx = nullptr;
    rcclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double *buf = nullptr;     // device bounce buffer for host-side reductions
    size_t buf_elems = 0;
    // host-backed variant (wgs_comm_create_host): the sums are formed by the caller's function (e.g. over TCP)
    wgs_allreduce_fn host_fn = nullptr;
    void *host_user = nullptr;
    double *host_stage = nullptr;      // pinned
    size_t host_elems = 0;
    // what crossed ranks so far (wgs_comm_stats): collectives, their payload, host round trips they cost
    int64_t n_allreduce = 0, n_bcast = 0, bytes_moved = 0, n_syncs = 0;
    // what RCCL reports about the communicator it built (-1: not asked / not available)
    int rccl_count = -1, rccl_rank = -1, rccl_device = -1;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // wgs_comm_time_collectives
    // self-checking collectives: what this rank has issued so far, and where a kernel reports a rank that issued something else
    int64_t seq = 0;                   // collectives issued on this communicator (same on every rank, or the ranks are out of step)
    int32_t generation = 0;            // wgs_comm_next_generation: which fit / scoring call / leave-one-out batch
    CommFault *fault = nullptr;        // page-locked, written by comm_tag_check_kernel
    CommFault *fault_dev = nullptr;    // the device's address of it
    bool failed = false;               // a mismatch has been reported: every later collective of this communicator fails at once
    std::string failure;
};

static const char *op_name(int op)
{
    switch (op) {
    case WGS_OP_GENERIC: return "untagged collective";
    case WGS_OP_EM_SUMS: return "EM convergence sums";
    case WGS_OP_EM_CHAIN: return "EM exact-chain carry";
    case WGS_OP_EM_FIT_END: return "end of an EM fit";
    case WGS_OP_SCORE_TOTALS: return "running log-likelihood totals";
    case WGS_OP_PART_CHAINS: return "partition-chain carries";
    case WGS_OP_LOO_BATCH: return "leave-one-out batch size";
    case WGS_OP_TIMING: return "collective timing";
    case WGS_OP_HOST: return "host all-reduce";
    default: return "collective";
    }
}

static std::string describe_row(int rank, const double *w)
{
    char buf[320];
    snprintf(buf, sizeof buf, "rank %d issued collective #%lld: %s (op %d), generation %d, iteration %d, shape %d / %d, %lld payload elements",
             rank, (long long)w[WGS_TAG_SEQ], op_name((int)w[WGS_TAG_OP]), (int)w[WGS_TAG_OP], (int)w[WGS_TAG_GEN], (int)w[WGS_TAG_ITER],
             (int)w[WGS_TAG_SHAPE_A], (int)w[WGS_TAG_SHAPE_B], (long long)w[WGS_TAG_COUNT]);
    return buf;
}

// One wavefront: this rank's row into its place of the world x 8 table behind the payload, zeros into the other ranks' rows.
__global__ void comm_tag_write_kernel(double *__restrict__ table, int world, int rank, CommRow row)
{
    for (int i = threadIdx.x; i < world * WGS_TAG_WORDS; i += blockDim.x)
        table[i] = (i / WGS_TAG_WORDS == rank) ? row.w[i % WGS_TAG_WORDS] : 0.0;
}

// One wavefront behind the collective: rows [0, nrows) of the table (all ranks' after an all-reduce; the root's after a broadcast:
// first_rank = root) against this rank's.  The free word (WGS_TAG_AUX) is not compared.  The first difference found is recorded.
__global__ void comm_tag_check_kernel(const double *__restrict__ table, int nrows, int first_rank, int rank, CommRow row, CommFault *fault)
{
    for (int r = threadIdx.x; r < nrows; r += blockDim.x) {
        bool same = true;
        for (int w = 0; w < WGS_TAG_AUX; ++w) same = same && table[r * WGS_TAG_WORDS + w] == row.w[w];
        if (!same && atomicCAS(&fault->claimed, 0, 1) == 0) {
            fault->rank = rank;
            fault->other = first_rank + r;
            for (int w = 0; w < WGS_TAG_WORDS; ++w) {
                fault->mine[w] = row.w[w];
                fault->theirs[w] = table[r * WGS_TAG_WORDS + w];
            }
            __threadfence_system();
            atomicExch(&fault->flag, 1);
            __threadfence_system();
        }
    }
}

static CommRow make_row(wgs_comm *c, const wgs_coll_tag *tag, int64_t count)
{
    CommRow row;
    row.w[WGS_TAG_SEQ] = (double)(++c->seq);
    row.w[WGS_TAG_OP] = tag ? (double)tag->op : (double)WGS_OP_GENERIC;
    row.w[WGS_TAG_GEN] = tag ? (double)tag->generation : 0.0;
    row.w[WGS_TAG_ITER] = tag ? (double)tag->iteration : 0.0;
    row.w[WGS_TAG_SHAPE_A] = tag ? (double)tag->shape_a : 0.0;
    row.w[WGS_TAG_SHAPE_B] = tag ? (double)tag->shape_b : 0.0;
    row.w[WGS_TAG_COUNT] = (double)count;
    row.w[WGS_TAG_AUX] = tag ? (double)tag->aux : 0.0;
    return row;
}

static int comm_fail(wgs_comm *c, int mine_rank, const double *mine, int other_rank, const double *theirs)
{
    c->failed = true;
    c->failure = "collective mismatch: the ranks have stopped issuing the same sequence of collectives -- " +
                 describe_row(mine_rank, mine) + "; " + describe_row(other_rank, theirs);
    wgs_set_error("%s", c->failure.c_str());
    return 1;
}

// Rows of a table that has come back to the HOST (host-backed communicators; the host variants of the RCCL path).
static int comm_check_rows_host(wgs_comm *c, const double *table, int nrows, int first_rank, const CommRow &row)
{
    for (int r = 0; r < nrows; ++r)
        for (int w = 0; w < WGS_TAG_AUX; ++w)
            if (table[(size_t)r * WGS_TAG_WORDS + w] != row.w[w]) return comm_fail(c, c->rank, row.w, first_rank + r, table + (size_t)r * WGS_TAG_WORDS);
    return 0;
}

static int comm_fault_alloc(wgs_comm *c)
{
    if (c->fault) return 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->fault), sizeof(CommFault), hipHostMallocMapped));
    memset(c->fault, 0, sizeof(CommFault));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->fault_dev), c->fault, 0));
    return 0;
}

#define RCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        int r_ = (expr);                                                                            \
        if (r_ != 0) {                                                                              \
            wgs_set_error("%s failed: %s", #expr, A->GetErrorString ? A->GetErrorString(r_) : "RCCL error"); \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)

extern "C" {

int wgs_comm_unique_id(uint8_t *id128)
{
    WGS_REQUIRE(id128, "null argument");
    Api *A = api();
    WGS_REQUIRE(A, "librccl could not be loaded (dlopen librccl.so.1)");
    rcclUniqueId id;
    RCCL_TRY(A->GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return 0;
}

int wgs_comm_init(wgs_ctx *ctx, const uint8_t *id128, int rank, int world, wgs_comm **out)
{
    WGS_REQUIRE(ctx && id128 && out && world >= 1 && rank >= 0 && rank < world, "bad argument");
    WGS_REQUIRE(world <= WGS_COMM_MAX_WORLD, "at most %d ranks per communicator (asked for %d)", WGS_COMM_MAX_WORLD, world);
    Api *A = api();
    WGS_REQUIRE(A, "librccl could not be loaded (dlopen librccl.so.1)");
    HIP_TRY(hipSetDevice(ctx->device));
    wgs_comm *c = new wgs_comm();
    auto guard = on_failure([&] { wgs_comm_destroy(c); });
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    if (comm_fault_alloc(c)) return 1;
    rcclUniqueId id;
    memcpy(id.internal, id128, 128);
    RCCL_TRY(A->CommInitRank(&c->comm, world, id, rank));
    // the communicator RCCL built must be the one asked for: `world` ranks, this one at `rank`, on this context's device -- a
    // launcher that reports N ranks over a communicator of fewer (or over a fallback transport) cannot pass for an N-GPU run
    if (A->CommCount) RCCL_TRY(A->CommCount(c->comm, &c->rccl_count));
    if (A->CommUserRank) RCCL_TRY(A->CommUserRank(c->comm, &c->rccl_rank));
    if (A->CommCuDevice) RCCL_TRY(A->CommCuDevice(c->comm, &c->rccl_device));
    if ((c->rccl_count >= 0 && c->rccl_count != world) || (c->rccl_rank >= 0 && c->rccl_rank != rank) ||
        (c->rccl_device >= 0 && c->rccl_device != ctx->device)) {
        wgs_set_error("RCCL built a communicator of %d ranks with this one at %d on device %d; asked for %d ranks, rank %d, device %d",
                      c->rccl_count, c->rccl_rank, c->rccl_device, world, rank, ctx->device);
        return 1;
    }
    guard.dismiss();
    *out = c;
    return 0;
}

/* A communicator whose sum all-reduce is the caller's function (in place over n host doubles, 0 = success): the loops
 * that run inside the library (wgs_em_fit, wgs_loo) then work over any transport -- the TCP all-reduce of
 * wgsassign_amd/comm.py uses it, which is also how those loops are exercised with several ranks on one GPU.  Device
 * buffers are staged through pinned host memory (a stream synchronisation per all-reduce: slower than RCCL, same sums). */
int wgs_comm_create_host(wgs_ctx *ctx, int rank, int world, wgs_allreduce_fn fn, void *user, wgs_comm **out)
{
    WGS_REQUIRE(ctx && fn && out && world >= 1 && rank >= 0 && rank < world, "bad argument");
    WGS_REQUIRE(world <= WGS_COMM_MAX_WORLD, "at most %d ranks per communicator (asked for %d)", WGS_COMM_MAX_WORLD, world);
    wgs_comm *c = new wgs_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->host_fn = fn;
    c->host_user = user;
    *out = c;
    return 0;
}

int wgs_comm_rank(wgs_comm *c, int *rank, int *world)
{
    WGS_REQUIRE(c, "null argument");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return 0;
}

void wgs_comm_destroy(wgs_comm *c)
{
    if (!c) return;
    Api *A = c->comm ? api() : nullptr;
    (void)hipSetDevice(c->ctx->device);
    if (A && c->comm) (void)A->CommDestroy(c->comm);
    if (c->buf) (void)hipFree(c->buf);
    if (c->host_stage) (void)hipHostFree(c->host_stage);
    if (c->fault) (void)hipHostFree(c->fault);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
}

/* The next generation number of this communicator: callers that run a whole phase over it (an EM fit, a scoring call, a
 * leave-one-out batch) take one and put it into the tags of the phase's collectives.  The ranks take them in the same order
 * or their tags differ. */
int32_t wgs_comm_next_generation(wgs_comm *c) { return c ? ++c->generation : 0; }

/* Whether a collective enqueued earlier on this communicator has found another rank out of step (the check kernel behind every
 * collective reports into page-locked memory; this reads it -- no synchronisation, so call it after one).  1 + the error text
 * (both ranks' tuples) when so; every later collective of the communicator fails with the same text. */
int wgs_comm_check(wgs_comm *c)
{
    if (!c) return 0;
    if (c->failed) {
        wgs_set_error("%s", c->failure.c_str());
        return 1;
    }
    if (c->fault && *reinterpret_cast<volatile int *>(&c->fault->flag)) return comm_fail(c, c->fault->rank, c->fault->mine, c->fault->other, c->fault->theirs);
    return 0;
}

static int host_stage_for(wgs_comm *c, size_t doubles)
{
    if (doubles <= c->host_elems) return 0;
    if (c->host_stage) (void)hipHostFree(c->host_stage);
    c->host_stage = nullptr;
    c->host_elems = 0;
    const size_t want = doubles < 1024 ? 1024 : doubles;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->host_stage), sizeof(double) * want, hipHostMallocDefault));
    c->host_elems = want;
    return 0;
}

/* In-place sum all-reduce of n float64 in DEVICE memory, enqueued on the context's stream, with this rank's tag row all-gathered
 * in the same call: dev_buf must have room for n + wgs_comm_tail_doubles() float64 (the table of rows lives behind the payload;
 * after the call rows_of(dev_buf + n)[r * WGS_TAG_WORDS + WGS_TAG_AUX] is rank r's free word). */
int wgs_comm_allreduce_tagged(wgs_comm *c, double *dev_buf, int64_t n, const wgs_coll_tag *tag)
{
    WGS_REQUIRE(c && dev_buf && n >= 0, "bad argument");
    if (wgs_comm_check(c)) return 1;
    HIP_TRY(hipSetDevice(c->ctx->device));
    c->n_allreduce += 1;
    c->bytes_moved += n * (int64_t)sizeof(double);
    const CommRow row = make_row(c, tag, n);
    const int64_t wire = n + (int64_t)c->world * WGS_TAG_WORDS;
    if (c->host_fn) {
        c->n_syncs += 2;
        if (host_stage_for(c, (size_t)wire)) return 1;
        HIP_TRY(hipMemcpyAsync(c->host_stage, dev_buf, sizeof(double) * n, hipMemcpyDeviceToHost, c->ctx->stream));
        HIP_TRY(hipStreamSynchronize(c->ctx->stream));
        double *table = c->host_stage + n;
        for (int i = 0; i < c->world * WGS_TAG_WORDS; ++i) table[i] = (i / WGS_TAG_WORDS == c->rank) ? row.w[i % WGS_TAG_WORDS] : 0.0;
        if (c->host_fn(c->host_stage, wire, c->host_user) != 0) {
            wgs_set_error("the communicator's all-reduce function failed");
            return 1;
        }
        if (comm_check_rows_host(c, table, c->world, 0, row)) return 1;
        HIP_TRY(hipMemcpyAsync(dev_buf, c->host_stage, sizeof(double) * wire, hipMemcpyHostToDevice, c->ctx->stream));
        HIP_TRY(hipStreamSynchronize(c->ctx->stream));       // the staging buffer is reused by the next call
        return 0;
    }
    Api *A = api();
    WGS_REQUIRE(A, "librccl not loaded");
    hipLaunchKernelGGL(comm_tag_write_kernel, dim3(1), dim3(64), 0, c->ctx->stream, dev_buf + n, c->world, c->rank, row);
    RCCL_TRY(A->AllReduce(dev_buf, dev_buf, (size_t)wire, kFloat64, kSum, c->comm, c->ctx->stream));
    hipLaunchKernelGGL(comm_tag_check_kernel, dim3(1), dim3(64), 0, c->ctx->stream, dev_buf + n, c->world, 0, c->rank, row, c->fault_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* Broadcast of `bytes` bytes (a multiple of 4) of DEVICE memory from rank `root`, enqueued on the context's stream: how a running
 * value (np.sum's float64 total, a float32 chain carry) is handed from SNP shard to SNP shard without a host round trip.  The
 * root's tag row travels behind the payload (dev_buf needs room for round_up(bytes, 8) + 8 * WGS_TAG_WORDS more bytes) and every
 * receiver compares it with its own.  Over a host-backed communicator the payload goes as 32-bit words widened to float64 through
 * the caller's sum all-reduce with zeros from everybody else -- exact for every bit pattern, NaN payloads included -- and the
 * rows of ALL ranks are compared. */
int wgs_comm_bcast_tagged(wgs_comm *c, void *dev_buf, int64_t bytes, int root, const wgs_coll_tag *tag)
{
    WGS_REQUIRE(c && dev_buf && bytes >= 0 && root >= 0 && root < c->world, "bad argument");
    if (bytes == 0 || c->world == 1) return 0;
    if (wgs_comm_check(c)) return 1;
    HIP_TRY(hipSetDevice(c->ctx->device));
    c->n_bcast += 1;
    c->bytes_moved += bytes;
    wgs_coll_tag t = tag ? *tag : wgs_coll_tag{WGS_OP_GENERIC, 0, 0, 0, 0, 0};
    t.shape_b = root;                                       // (who sends is part of what the collective is)
    const CommRow row = make_row(c, &t, bytes);
    if (c->host_fn) {
        WGS_REQUIRE(bytes % 4 == 0, "broadcast payload must be a multiple of 4 bytes");
        const size_t words = (size_t)bytes / 4;
        const size_t wire = words + (size_t)c->world * WGS_TAG_WORDS;
        c->n_syncs += 2;
        if (host_stage_for(c, wire)) return 1;
        char *raw = reinterpret_cast<char *>(c->host_stage);               // the words first, widened in place from the back
        if (c->rank == root) {
            HIP_TRY(hipMemcpyAsync(raw, dev_buf, (size_t)bytes, hipMemcpyDeviceToHost, c->ctx->stream));
            HIP_TRY(hipStreamSynchronize(c->ctx->stream));
            for (size_t i = words; i-- > 0;) {
                uint32_t w;
                memcpy(&w, raw + 4 * i, 4);
                const double d = (double)w;
                memcpy(raw + 8 * i, &d, 8);
            }
        } else {
            HIP_TRY(hipStreamSynchronize(c->ctx->stream));                   // the staging buffer may still be in flight
            for (size_t i = 0; i < words; ++i) c->host_stage[i] = 0.0;
        }
        double *table = c->host_stage + words;
        for (int i = 0; i < c->world * WGS_TAG_WORDS; ++i) table[i] = (i / WGS_TAG_WORDS == c->rank) ? row.w[i % WGS_TAG_WORDS] : 0.0;
        if (c->host_fn(c->host_stage, (int64_t)wire, c->host_user) != 0) {
            wgs_set_error("the communicator's all-reduce function failed");
            return 1;
        }
        if (comm_check_rows_host(c, table, c->world, 0, row)) return 1;
        for (size_t i = 0; i < words; ++i) {
            double d;
            memcpy(&d, raw + 8 * i, 8);
            const uint32_t w = (uint32_t)d;
            memcpy(raw + 4 * i, &w, 4);
        }
        HIP_TRY(hipMemcpyAsync(dev_buf, raw, (size_t)bytes, hipMemcpyHostToDevice, c->ctx->stream));
        HIP_TRY(hipStreamSynchronize(c->ctx->stream));
        return 0;
    }
    Api *A = api();
    WGS_REQUIRE(A, "librccl not loaded");
    const size_t padded = ((size_t)bytes + 7) & ~(size_t)7;
    double *table = reinterpret_cast<double *>(reinterpret_cast<char *>(dev_buf) + padded);
    if (c->rank == root) hipLaunchKernelGGL(comm_tag_write_kernel, dim3(1), dim3(64), 0, c->ctx->stream, table, 1, 0, row);
    RCCL_TRY(A->Broadcast(dev_buf, dev_buf, padded + sizeof(double) * WGS_TAG_WORDS, kUint8, root, c->comm, c->ctx->stream));
    if (c->rank != root)
        hipLaunchKernelGGL(comm_tag_check_kernel, dim3(1), dim3(64), 0, c->ctx->stream, table, 1, root, c->rank, row, c->fault_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* The same two collectives for buffers WITHOUT room behind the payload (any caller of the public header): through the
 * communicator's bounce buffer, tagged as WGS_OP_GENERIC -- sequence number and payload size are still compared. */
int wgs_comm_allreduce_f64_dev(wgs_comm *c, double *dev_buf, int64_t n)
{
    WGS_REQUIRE(c && dev_buf && n >= 0, "bad argument");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->ctx->device));
    double *buf = wgs_comm_buffer(c, n);
    if (!buf) return 1;
    if (buf != dev_buf) HIP_TRY(hipMemcpyAsync(buf, dev_buf, sizeof(double) * n, hipMemcpyDeviceToDevice, c->ctx->stream));
    if (wgs_comm_allreduce_tagged(c, buf, n, nullptr)) return 1;
    if (buf != dev_buf) HIP_TRY(hipMemcpyAsync(dev_buf, buf, sizeof(double) * n, hipMemcpyDeviceToDevice, c->ctx->stream));
    return 0;
}

int wgs_comm_bcast_dev(wgs_comm *c, void *dev_buf, int64_t bytes, int root)
{
    WGS_REQUIRE(c && dev_buf && bytes >= 0 && root >= 0 && root < c->world, "bad argument");
    if (bytes == 0 || c->world == 1) return 0;
    HIP_TRY(hipSetDevice(c->ctx->device));
    double *buf = wgs_comm_buffer(c, (bytes + 7) / 8);
    if (!buf) return 1;
    if (c->rank == root) HIP_TRY(hipMemcpyAsync(buf, dev_buf, (size_t)bytes, hipMemcpyDeviceToDevice, c->ctx->stream));
    if (wgs_comm_bcast_tagged(c, buf, bytes, root, nullptr)) return 1;
    HIP_TRY(hipMemcpyAsync(dev_buf, buf, (size_t)bytes, hipMemcpyDeviceToDevice, c->ctx->stream));
    return 0;
}

/* stats[0..3]: all-reduces, broadcasts, payload bytes, host round trips (stream synchronisations) the collectives of
 * this communicator have cost so far. */
int wgs_comm_stats(wgs_comm *c, int64_t *stats)
{
    WGS_REQUIRE(c && stats, "null argument");
    stats[0] = c->n_allreduce;
    stats[1] = c->n_bcast;
    stats[2] = c->bytes_moved;
    stats[3] = c->n_syncs;
    return 0;
}

/* info[0..7]: 1 = RCCL communicator / 0 = host-backed; ranks, this rank and the device as RCCL ITSELF reports them
 * (ncclCommCount, ncclCommUserRank, ncclCommCuDevice; -1 where not available or host-backed); world and rank as given at
 * creation; [6] collectives issued so far (the sequence number the next one is checked with is this + 1); [7] 1 once a rank
 * was found out of step.  bench.py and the command line put these into what they report for N > 1. */
int wgs_comm_info(wgs_comm *c, int64_t *info)
{
    WGS_REQUIRE(c && info, "null argument");
    info[0] = c->comm ? 1 : 0;
    info[1] = c->rccl_count;
    info[2] = c->rccl_rank;
    info[3] = c->rccl_device;
    info[4] = c->world;
    info[5] = c->rank;
    info[6] = c->seq;
    info[7] = c->failed ? 1 : 0;
    return 0;
}

/* Device time of the two collectives the sharded path uses, measured with HIP events on the context's stream: `reps`
 * back-to-back sum all-reduces of n float64 (the per-iteration exchange of the EM fit) and `reps` broadcasts of n float64 from
 * rank 0 (the hand-over of a running total), both WITH their tag rows and check kernels -- what the path really issues;
 * us_out[0..1] = mean microseconds of each.  Collective: every rank calls it. */
int wgs_comm_time_collectives(wgs_comm *c, int32_t reps, int64_t n, double *us_out)
{
    WGS_REQUIRE(c && us_out && reps > 0 && n > 0, "bad argument");
    HIP_TRY(hipSetDevice(c->ctx->device));
    if (!c->ev0) HIP_TRY(hipEventCreate(&c->ev0));
    if (!c->ev1) HIP_TRY(hipEventCreate(&c->ev1));
    double *buf = wgs_comm_buffer(c, n);
    if (!buf) return 1;
    HIP_TRY(hipMemsetAsync(buf, 0, sizeof(double) * (size_t)n, c->ctx->stream));
    wgs_coll_tag tag = {WGS_OP_TIMING, wgs_comm_next_generation(c), 0, reps, 0, 0};
    for (int which = 0; which < 2; ++which) {
        // one untimed call first (connection set-up of the first collective of a kind)
        tag.iteration = which;
        if (which == 0 ? wgs_comm_allreduce_tagged(c, buf, n, &tag) : wgs_comm_bcast_tagged(c, buf, n * 8, 0, &tag)) return 1;
        HIP_TRY(hipEventRecord(c->ev0, c->ctx->stream));
        for (int r = 0; r < reps; ++r)
            if (which == 0 ? wgs_comm_allreduce_tagged(c, buf, n, &tag) : wgs_comm_bcast_tagged(c, buf, n * 8, 0, &tag)) return 1;
        HIP_TRY(hipEventRecord(c->ev1, c->ctx->stream));
        HIP_TRY(hipEventSynchronize(c->ev1));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        us_out[which] = (double)ms * 1e3 / reps;
    }
    return wgs_comm_check(c);
}

/* The communicator's device bounce buffer, grown to hold n float64 (+ the tag rows behind them; contents not preserved when it
 * grows). */
double *wgs_comm_buffer(wgs_comm *c, int64_t n)
{
    if (!c || n < 0) return nullptr;
    if (hipSetDevice(c->ctx->device) != hipSuccess) return nullptr;
    const size_t need = (size_t)n + wgs_comm_tail_doubles();
    if (need > c->buf_elems) {
        (void)hipStreamSynchronize(c->ctx->stream);
        if (c->buf) (void)hipFree(c->buf);
        c->buf = nullptr;
        c->buf_elems = 0;
        const size_t want = need < 2048 ? 2048 : need;
        if (wgs_malloc(&c->buf, sizeof(double) * want) != hipSuccess) {
            wgs_set_error("hipMalloc of the communicator bounce buffer failed");
            return nullptr;
        }
        c->buf_elems = want;
    }
    return c->buf;
}

/* All-reduce the first n float64 of the bounce buffer in place (behind the work already on the
 * stream, e.g. wgs_em_step_dev into wgs_comm_buffer) and copy them to host_out; synchronises. */
int wgs_comm_allreduce_buffer(wgs_comm *c, int64_t n, double *host_out)
{
    WGS_REQUIRE(c && host_out && n >= 0 && (size_t)n + wgs_comm_tail_doubles() <= c->buf_elems, "bad argument");
    if (n == 0) return 0;
    if (wgs_comm_allreduce_tagged(c, c->buf, n, nullptr)) return 1;
    HIP_TRY(hipMemcpyAsync(host_out, c->buf, sizeof(double) * n, hipMemcpyDeviceToHost, c->ctx->stream));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    return wgs_comm_check(c);
}

/* Same for a HOST buffer: staged through the bounce buffer; returns after the result is back and its rows have been compared.
 * rows_out (may be NULL; world * WGS_TAG_WORDS float64) receives the table of rows -- every rank's free word included. */
int wgs_comm_allreduce_host_tagged(wgs_comm *c, double *host_buf, int64_t n, const wgs_coll_tag *tag, double *rows_out)
{
    WGS_REQUIRE(c && (host_buf || n == 0) && n >= 0, "bad argument");      // (n == 0: the rows alone -- a checkpoint of the ranks)
    if (wgs_comm_check(c)) return 1;
    const int64_t wire = n + (int64_t)c->world * WGS_TAG_WORDS;
    if (c->host_fn) {
        c->n_allreduce += 1;
        c->bytes_moved += n * (int64_t)sizeof(double);
        const CommRow row = make_row(c, tag, n);
        std::vector<double> stage((size_t)wire, 0.0);
        if (n) memcpy(stage.data(), host_buf, sizeof(double) * n);
        memcpy(stage.data() + n + (size_t)c->rank * WGS_TAG_WORDS, row.w, sizeof(double) * WGS_TAG_WORDS);
        if (c->host_fn(stage.data(), wire, c->host_user) != 0) {
            wgs_set_error("the communicator's all-reduce function failed");
            return 1;
        }
        if (comm_check_rows_host(c, stage.data() + n, c->world, 0, row)) return 1;
        if (n) memcpy(host_buf, stage.data(), sizeof(double) * n);
        if (rows_out) memcpy(rows_out, stage.data() + n, sizeof(double) * c->world * WGS_TAG_WORDS);
        return 0;
    }
    HIP_TRY(hipSetDevice(c->ctx->device));
    c->n_syncs += 1;
    if (!wgs_comm_buffer(c, n)) return 1;
    if (n) HIP_TRY(hipMemcpyAsync(c->buf, host_buf, sizeof(double) * n, hipMemcpyHostToDevice, c->ctx->stream));
    if (wgs_comm_allreduce_tagged(c, c->buf, n, tag)) return 1;
    if (n) HIP_TRY(hipMemcpyAsync(host_buf, c->buf, sizeof(double) * n, hipMemcpyDeviceToHost, c->ctx->stream));
    std::vector<double> rows((size_t)c->world * WGS_TAG_WORDS);
    HIP_TRY(hipMemcpyAsync(rows.data(), c->buf + n, sizeof(double) * rows.size(), hipMemcpyDeviceToHost, c->ctx->stream));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    if (wgs_comm_check(c)) return 1;
    if (rows_out) memcpy(rows_out, rows.data(), sizeof(double) * rows.size());
    return 0;
}

int wgs_comm_allreduce_f64(wgs_comm *c, double *host_buf, int64_t n)
{
    if (n == 0) return 0;
    return wgs_comm_allreduce_host_tagged(c, host_buf, n, nullptr, nullptr);
}

/* Test hook (include/wgsassign_hip_debug.h): the two kernels around a tagged RCCL collective, on rows made up by the caller -- what
 * every rank of a `world`-rank job would run, with the sum all-reduce in between done here by adding the ranks' tables.  rows:
 * world x 8 float64 (what each rank would issue); as_rank: whose check runs.  fault_out[0] = 1 if the check kernel reported a
 * difference, then fault_out[1] = the other rank, fault_out[2..9] = this rank's row, fault_out[10..17] = the other's. */
int wgs_debug_comm_tag_kernels(wgs_ctx *ctx, int32_t world, const double *rows, int32_t as_rank, double *fault_out)
{
    WGS_REQUIRE(ctx && rows && fault_out && world >= 1 && world <= WGS_COMM_MAX_WORLD && as_rank >= 0 && as_rank < world, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t words = (size_t)world * WGS_TAG_WORDS;
    double *d_tab = nullptr, *d_sum = nullptr;
    CommFault *fault = nullptr;
    HIP_TRY(wgs_malloc(reinterpret_cast<void **>(&d_tab), sizeof(double) * words));
    HIP_TRY(wgs_malloc(reinterpret_cast<void **>(&d_sum), sizeof(double) * words));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&fault), sizeof(CommFault), hipHostMallocDefault));
    memset(fault, 0, sizeof(CommFault));
    std::vector<double> sum(words, 0.0), one(words);
    for (int r = 0; r < world; ++r) {                                  // every rank's table, as comm_tag_write_kernel lays it out
        CommRow row;
        for (int w = 0; w < WGS_TAG_WORDS; ++w) row.w[w] = rows[(size_t)r * WGS_TAG_WORDS + w];
        hipLaunchKernelGGL(comm_tag_write_kernel, dim3(1), dim3(64), 0, ctx->stream, d_tab, world, r, row);
        HIP_TRY(hipMemcpyAsync(one.data(), d_tab, sizeof(double) * words, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < words; ++i) sum[i] += one[i];           // (the sum all-reduce)
    }
    HIP_TRY(hipMemcpyAsync(d_sum, sum.data(), sizeof(double) * words, hipMemcpyHostToDevice, ctx->stream));
    CommRow mine;
    for (int w = 0; w < WGS_TAG_WORDS; ++w) mine.w[w] = rows[(size_t)as_rank * WGS_TAG_WORDS + w];
    CommFault *fault_dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&fault_dev), fault, 0));
    hipLaunchKernelGGL(comm_tag_check_kernel, dim3(1), dim3(64), 0, ctx->stream, d_sum, world, 0, as_rank, mine, fault_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    fault_out[0] = (double)fault->flag;
    fault_out[1] = (double)fault->other;
    for (int w = 0; w < WGS_TAG_WORDS; ++w) {
        fault_out[2 + w] = fault->mine[w];
        fault_out[2 + WGS_TAG_WORDS + w] = fault->theirs[w];
    }
    (void)hipFree(d_tab);
    (void)hipFree(d_sum);
    (void)hipHostFree(fault);
    return 0;
}

}  // extern "C"
