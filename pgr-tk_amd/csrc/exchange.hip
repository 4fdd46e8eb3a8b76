// exchange.hip -- the multi-GPU exchange step behind the C ABI.
//
// Contigs shard across ranks (one process per GPU); every rank computes the shimmers of its own contigs
// (CompactSeqDB::get_shmmrs_from_seqs maps over contigs, pgr-db/src/seq_db.rs:460-467) and the per-rank lists are
// all-gathered over RCCL / xGMI so that the rank that owns the frag_map (seq_db.rs:605-612) -- or every rank, for a
// replicated query index -- holds the full set.  What travels is the final MM128 list (16 B per shimmer, rid = global
// sequence id); pair records are adjacent shimmers and are derived by the receiver (pgr_index_add_shmmrs).
//
// One collective per step on the exchange's OWN stream: a padded ncclAllGather of `cap_per_rank` elements per rank (the
// capacity is fixed at start-up, so no rank has to learn another rank's size before posting the collective) next to an
// all-gather of the element counts, which stay on the device for the consumer and reach the host only at wait().
// Nothing in start() blocks the host, the compute stream keeps running the next step's kernels.
//
// RCCL is loaded with dlopen at the first pgr_exchange_* call (librccl.so.1, the soname both the ROCm install and the
// PyTorch wheel carry): a process that never exchanges does not load it, and a process that already has PyTorch's
// copy shares it.  No CUDA/NCCL dual path: "nccl*" are RCCL's own symbol names.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pgr_ctx.h"
#include "pgr_index.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.err = std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p && r.err.empty()) r.err = std::string("RCCL symbol missing: ") + n;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}

}  // namespace

struct pgr_exchange {
    pgr_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;          // the collective's own stream: overlaps the compute stream
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    unsigned long long *d_cnt = nullptr;   // [1 + world]: this rank's count, then everybody's
    unsigned long long *h_cnt = nullptr;   // pinned [1 + world]
    bool in_flight = false;
};

static_assert(PGR_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "pgr_hip.h and rccl.h disagree on the unique id size");

#define PGR_NCCL(ctx, expr)                                                                          \
    do {                                                                                             \
        ncclResult_t _r = (expr);                                                                    \
        if (_r != ncclSuccess)                                                                       \
            return (ctx)->fail(PGR_ERR_DEVICE, std::string(#expr) + ": " + rccl().GetErrorString(_r)); \
    } while (0)

extern "C" int pgr_exchange_unique_id(pgr_ctx *ctx, uint8_t *id) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!id) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    Rccl &R = rccl();
    if (!R.err.empty()) return ctx->fail(PGR_ERR_DEVICE, R.err);
    PGR_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    PGR_NCCL(ctx, R.GetUniqueId(&u));
    memcpy(id, u.internal, PGR_UNIQUE_ID_BYTES);
    return PGR_OK;
}

extern "C" int pgr_exchange_create(pgr_ctx *ctx, const uint8_t *id, int rank, int world, pgr_exchange **out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return ctx->fail(PGR_ERR_INVALID_ARG, "bad exchange arguments");
    *out = nullptr;
    Rccl &R = rccl();
    if (!R.err.empty()) return ctx->fail(PGR_ERR_DEVICE, R.err);
    PGR_HIP(ctx, hipSetDevice(ctx->device));
    pgr_exchange *x = new pgr_exchange();
    x->ctx = ctx;
    x->rank = rank;
    x->world = world;
    auto bail = [&](int code) {
        pgr_exchange_destroy(x);
        return code;
    };
    hipError_t e = hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&x->d_cnt, (size_t)(1 + world) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&x->h_cnt, (size_t)(1 + world) * sizeof(unsigned long long), hipHostMallocDefault);
    if (e != hipSuccess) return bail(ctx->fail(PGR_ERR_DEVICE, std::string("exchange setup: ") + hipGetErrorString(e)));
    ncclUniqueId u;
    memcpy(u.internal, id, PGR_UNIQUE_ID_BYTES);
    ncclResult_t r = R.CommInitRank(&x->comm, world, u, rank);
    if (r != ncclSuccess) {
        x->comm = nullptr;
        return bail(ctx->fail(PGR_ERR_DEVICE, std::string("ncclCommInitRank: ") + R.GetErrorString(r)));
    }
    *out = x;
    return PGR_OK;
}

extern "C" void pgr_exchange_destroy(pgr_exchange *x) {
    if (!x) return;
    (void)hipSetDevice(x->ctx->device);
    if (x->stream) (void)hipStreamSynchronize(x->stream);
    if (x->comm) (void)rccl().CommDestroy(x->comm);
    if (x->ev_ready) (void)hipEventDestroy(x->ev_ready);
    if (x->ev_done) (void)hipEventDestroy(x->ev_done);
    if (x->d_cnt) (void)hipFree(x->d_cnt);
    if (x->h_cnt) (void)hipHostFree(x->h_cnt);
    if (x->stream) (void)hipStreamDestroy(x->stream);
    delete x;
}

extern "C" int pgr_exchange_rank(const pgr_exchange *x) { return x ? x->rank : -1; }
extern "C" int pgr_exchange_world(const pgr_exchange *x) { return x ? x->world : 0; }

extern "C" int pgr_exchange_allgather_shmmrs_start(pgr_exchange *x, const pgr_mm128 *d_local, uint64_t n_local,
                                                   pgr_mm128 *d_out, uint64_t cap_per_rank) {
    if (!x) return PGR_ERR_INVALID_ARG;
    pgr_ctx *ctx = x->ctx;
    if (x->in_flight) return ctx->fail(PGR_ERR_STATE, "an all-gather is already in flight on this exchange (call pgr_exchange_wait)");
    if (!d_local || !d_out || cap_per_rank == 0) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (n_local > cap_per_rank) return ctx->fail(PGR_ERR_INVALID_ARG, "this rank's shimmer list exceeds cap_per_rank");
    Rccl &R = rccl();
    PGR_HIP(ctx, hipSetDevice(ctx->device));
    // the list was produced on the context's stream: the collective starts behind it, the host does not wait
    PGR_HIP(ctx, hipEventRecord(x->ev_ready, ctx->stream));
    PGR_HIP(ctx, hipStreamWaitEvent(x->stream, x->ev_ready, 0));
    x->h_cnt[0] = n_local;
    PGR_HIP(ctx, hipMemcpyAsync(x->d_cnt, x->h_cnt, sizeof(unsigned long long), hipMemcpyHostToDevice, x->stream));
    PGR_NCCL(ctx, R.AllGather(x->d_cnt, x->d_cnt + 1, 1, ncclUint64, x->comm, x->stream));
    // padded: every rank contributes cap_per_rank elements (2 x u64 each); the tail beyond its count is never read
    PGR_NCCL(ctx, R.AllGather(d_local, d_out, (size_t)cap_per_rank * 2, ncclUint64, x->comm, x->stream));
    PGR_HIP(ctx, hipMemcpyAsync(x->h_cnt + 1, x->d_cnt + 1, (size_t)x->world * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                x->stream));
    PGR_HIP(ctx, hipEventRecord(x->ev_done, x->stream));
    x->in_flight = true;
    return PGR_OK;
}

extern "C" int pgr_exchange_wait(pgr_exchange *x, uint64_t *counts) {
    if (!x) return PGR_ERR_INVALID_ARG;
    pgr_ctx *ctx = x->ctx;
    if (!x->in_flight) return ctx->fail(PGR_ERR_STATE, "no all-gather in flight");
    PGR_HIP(ctx, hipSetDevice(ctx->device));
    PGR_HIP(ctx, hipEventSynchronize(x->ev_done));
    // later work on the compute stream (index build from the gathered lists, reuse of the buffers) is ordered behind it
    PGR_HIP(ctx, hipStreamWaitEvent(ctx->stream, x->ev_done, 0));
    x->in_flight = false;
    if (counts)
        for (int r = 0; r < x->world; ++r) counts[r] = x->h_cnt[1 + r];
    return PGR_OK;
}

// device pointer to the world counts of the last all-gather (for consumers that stay on the device)
extern "C" const uint64_t *pgr_exchange_device_counts(const pgr_exchange *x) {
    return x ? (const uint64_t *)(x->d_cnt + 1) : nullptr;
}

// Blocking convenience for host programs (host/pgr_mdb.cpp --ranks N): all-gather this rank's lists and add every
// rank's lists to `ix` in rank order.  Two phases, because a host program has no capacity agreed in advance: the counts
// first (their maximum becomes the padded size), then the payload.  s == NULL: this rank contributes nothing this round.
extern "C" int pgr_exchange_gather_into_index(pgr_exchange *x, const pgr_shmmrs *s, const uint32_t *rids, pgr_index *ix,
                                              uint64_t *n_gathered) {
    if (!x) return PGR_ERR_INVALID_ARG;
    pgr_ctx *ctx = x->ctx;
    if (x->in_flight) return ctx->fail(PGR_ERR_STATE, "an all-gather is already in flight on this exchange");
    if (s && s->n && !rids) return ctx->fail(PGR_ERR_INVALID_ARG, "null rid list");
    Rccl &R = rccl();
    PGR_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n_local = s ? s->count : 0;
    // phase 1: counts (the compute stream's work is done: pgr_shmmrs_compute synchronizes)
    x->h_cnt[0] = n_local;
    PGR_HIP(ctx, hipMemcpyAsync(x->d_cnt, x->h_cnt, sizeof(unsigned long long), hipMemcpyHostToDevice, x->stream));
    PGR_NCCL(ctx, R.AllGather(x->d_cnt, x->d_cnt + 1, 1, ncclUint64, x->comm, x->stream));
    PGR_HIP(ctx, hipMemcpyAsync(x->h_cnt + 1, x->d_cnt + 1, (size_t)x->world * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                x->stream));
    PGR_HIP(ctx, hipStreamSynchronize(x->stream));
    uint64_t cap = 0, total = 0;
    for (int r = 0; r < x->world; ++r) {
        cap = std::max<uint64_t>(cap, x->h_cnt[1 + r]);
        total += x->h_cnt[1 + r];
    }
    if (n_gathered) *n_gathered = total;
    if (cap == 0) return PGR_OK;
    // phase 2: padded payload
    pgr::Tmp local(ctx), out(ctx);
    int rc;
    if ((rc = local.alloc(cap * sizeof(pgr_mm128))) || (rc = out.alloc((size_t)x->world * cap * sizeof(pgr_mm128)))) return rc;
    if (n_local && (rc = pgr_shmmrs_copy_to_device_rids(ctx, s, local.as<pgr_mm128>(), cap, rids))) return rc;  // synchronizes
    std::vector<uint64_t> counts((size_t)x->world);
    if ((rc = pgr_exchange_allgather_shmmrs_start(x, local.as<pgr_mm128>(), n_local, out.as<pgr_mm128>(), cap)) ||
        (rc = pgr_exchange_wait(x, counts.data())))
        return rc;
    if (ix)
        for (int r = 0; r < x->world; ++r)
            if (counts[(size_t)r] && (rc = pgr_index_add_shmmrs(ctx, ix, out.as<pgr_mm128>() + (size_t)r * cap, counts[(size_t)r], 1)))
                return rc;
    PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the temporaries go back to the allocator
    return PGR_OK;
}
