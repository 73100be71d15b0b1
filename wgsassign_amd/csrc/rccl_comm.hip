// Native RCCL communicator: the one collective of the SNP-sharded path (sum all-reduce of a few
// float64 over xGMI) without any tensor framework.  librccl is dlopen'ed on first use, so the
// library has no link-time dependency on it and single-GPU use never loads it.
//
// Bootstrap: rank 0 obtains the 128-byte ncclUniqueId (wgs_comm_unique_id), the host side hands it
// to the other ranks (wgsassign_amd/comm.py does that over a TCP socket on MASTER_ADDR), every
// rank calls wgs_comm_init.  The all-reduce is enqueued on the context's HIP stream, i.e. behind
// the EM sweep that produced the sums.
#include <dlfcn.h>
#include <string.h>

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
};

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
    Api *A = api();
    WGS_REQUIRE(A, "librccl could not be loaded (dlopen librccl.so.1)");
    HIP_TRY(hipSetDevice(ctx->device));
    wgs_comm *c = new wgs_comm();
    auto guard = on_failure([&] { delete c; });
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
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
        (void)A->CommDestroy(c->comm);
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
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
}

/* In-place sum all-reduce of n float64 in DEVICE memory, enqueued on the context's stream. */
int wgs_comm_allreduce_f64_dev(wgs_comm *c, double *dev_buf, int64_t n)
{
    WGS_REQUIRE(c && dev_buf && n >= 0, "bad argument");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->ctx->device));
    c->n_allreduce += 1;
    c->bytes_moved += n * (int64_t)sizeof(double);
    if (c->host_fn) {
        c->n_syncs += 2;
        if ((size_t)n > c->host_elems) {
            if (c->host_stage) (void)hipHostFree(c->host_stage);
            c->host_stage = nullptr;
            c->host_elems = 0;
            const size_t want = (size_t)n < 1024 ? 1024 : (size_t)n;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->host_stage), sizeof(double) * want, hipHostMallocDefault));
            c->host_elems = want;
        }
        HIP_TRY(hipMemcpyAsync(c->host_stage, dev_buf, sizeof(double) * n, hipMemcpyDeviceToHost, c->ctx->stream));
        HIP_TRY(hipStreamSynchronize(c->ctx->stream));
        if (c->host_fn(c->host_stage, n, c->host_user) != 0) {
            wgs_set_error("the communicator's all-reduce function failed");
            return 1;
        }
        HIP_TRY(hipMemcpyAsync(dev_buf, c->host_stage, sizeof(double) * n, hipMemcpyHostToDevice, c->ctx->stream));
        HIP_TRY(hipStreamSynchronize(c->ctx->stream));       // the staging buffer is reused by the next call
        return 0;
    }
    Api *A = api();
    WGS_REQUIRE(A, "librccl not loaded");
    RCCL_TRY(A->AllReduce(dev_buf, dev_buf, (size_t)n, kFloat64, kSum, c->comm, c->ctx->stream));
    return 0;
}

/* Broadcast of `bytes` bytes of DEVICE memory from rank `root`, enqueued on the context's stream: how a running value
 * (np.sum's float64 total, a float32 chain carry) is handed from SNP shard to SNP shard without a host round trip.
 * Over a host-backed communicator the payload goes as 32-bit words widened to float64 through the caller's sum
 * all-reduce with zeros from everybody else -- exact for every bit pattern, NaN payloads included. */
int wgs_comm_bcast_dev(wgs_comm *c, void *dev_buf, int64_t bytes, int root)
{
    WGS_REQUIRE(c && dev_buf && bytes >= 0 && root >= 0 && root < c->world, "bad argument");
    if (bytes == 0 || c->world == 1) return 0;
    HIP_TRY(hipSetDevice(c->ctx->device));
    c->n_bcast += 1;
    c->bytes_moved += bytes;
    if (c->host_fn) {
        WGS_REQUIRE(bytes % 4 == 0, "broadcast payload must be a multiple of 4 bytes");
        const size_t words = (size_t)bytes / 4;
        c->n_syncs += 2;
        if (words > c->host_elems) {
            if (c->host_stage) (void)hipHostFree(c->host_stage);
            c->host_stage = nullptr;
            c->host_elems = 0;
            const size_t want = words < 1024 ? 1024 : words;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->host_stage), sizeof(double) * want, hipHostMallocDefault));
            c->host_elems = want;
        }
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
        if (c->host_fn(c->host_stage, (int64_t)words, c->host_user) != 0) {
            wgs_set_error("the communicator's all-reduce function failed");
            return 1;
        }
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
    RCCL_TRY(A->Broadcast(dev_buf, dev_buf, (size_t)bytes, kUint8, root, c->comm, c->ctx->stream));
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
 * creation; [6..7] reserved.  bench.py and the command line put these into what they report for N > 1. */
int wgs_comm_info(wgs_comm *c, int64_t *info)
{
    WGS_REQUIRE(c && info, "null argument");
    info[0] = c->comm ? 1 : 0;
    info[1] = c->rccl_count;
    info[2] = c->rccl_rank;
    info[3] = c->rccl_device;
    info[4] = c->world;
    info[5] = c->rank;
    info[6] = info[7] = 0;
    return 0;
}

/* Device time of the two collectives the sharded path uses, measured with HIP events on the context's stream: `reps`
 * back-to-back sum all-reduces of n float64 (the per-iteration exchange of the EM fit) and `reps` broadcasts of n float64 from
 * rank 0 (the hand-over of a running total); us_out[0..1] = mean microseconds of each.  Collective: every rank calls it. */
int wgs_comm_time_collectives(wgs_comm *c, int32_t reps, int64_t n, double *us_out)
{
    WGS_REQUIRE(c && us_out && reps > 0 && n > 0, "bad argument");
    HIP_TRY(hipSetDevice(c->ctx->device));
    if (!c->ev0) HIP_TRY(hipEventCreate(&c->ev0));
    if (!c->ev1) HIP_TRY(hipEventCreate(&c->ev1));
    double *buf = wgs_comm_buffer(c, n);
    if (!buf) return 1;
    HIP_TRY(hipMemsetAsync(buf, 0, sizeof(double) * (size_t)n, c->ctx->stream));
    for (int which = 0; which < 2; ++which) {
        // one untimed call first (connection set-up of the first collective of a kind)
        if (which == 0 ? wgs_comm_allreduce_f64_dev(c, buf, n) : wgs_comm_bcast_dev(c, buf, n * 8, 0)) return 1;
        HIP_TRY(hipEventRecord(c->ev0, c->ctx->stream));
        for (int r = 0; r < reps; ++r)
            if (which == 0 ? wgs_comm_allreduce_f64_dev(c, buf, n) : wgs_comm_bcast_dev(c, buf, n * 8, 0)) return 1;
        HIP_TRY(hipEventRecord(c->ev1, c->ctx->stream));
        HIP_TRY(hipEventSynchronize(c->ev1));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        us_out[which] = (double)ms * 1e3 / reps;
    }
    return 0;
}

/* The communicator's device bounce buffer, grown to hold n float64 (contents not preserved when it grows). */
double *wgs_comm_buffer(wgs_comm *c, int64_t n)
{
    if (!c || n < 0) return nullptr;
    if (hipSetDevice(c->ctx->device) != hipSuccess) return nullptr;
    if ((size_t)n > c->buf_elems) {
        (void)hipStreamSynchronize(c->ctx->stream);
        if (c->buf) (void)hipFree(c->buf);
        c->buf = nullptr;
        c->buf_elems = 0;
        const size_t want = (size_t)n < 1024 ? 1024 : (size_t)n;
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
    WGS_REQUIRE(c && host_out && n >= 0 && (size_t)n <= c->buf_elems, "bad argument");
    if (n == 0) return 0;
    if (wgs_comm_allreduce_f64_dev(c, c->buf, n)) return 1;
    HIP_TRY(hipMemcpyAsync(host_out, c->buf, sizeof(double) * n, hipMemcpyDeviceToHost, c->ctx->stream));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    return 0;
}

/* Same for a HOST buffer: staged through the bounce buffer; returns after the result is back. */
int wgs_comm_allreduce_f64(wgs_comm *c, double *host_buf, int64_t n)
{
    WGS_REQUIRE(c && host_buf && n >= 0, "bad argument");
    if (n == 0) return 0;
    if (c->host_fn) {
        c->n_allreduce += 1;
        c->bytes_moved += n * (int64_t)sizeof(double);
        if (c->host_fn(host_buf, n, c->host_user) != 0) {
            wgs_set_error("the communicator's all-reduce function failed");
            return 1;
        }
        return 0;
    }
    HIP_TRY(hipSetDevice(c->ctx->device));
    c->n_syncs += 1;
    if (!wgs_comm_buffer(c, n)) return 1;
    HIP_TRY(hipMemcpyAsync(c->buf, host_buf, sizeof(double) * n, hipMemcpyHostToDevice, c->ctx->stream));
    if (wgs_comm_allreduce_f64_dev(c, c->buf, n)) return 1;
    HIP_TRY(hipMemcpyAsync(host_buf, c->buf, sizeof(double) * n, hipMemcpyDeviceToHost, c->ctx->stream));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    return 0;
}

}  // extern "C"
