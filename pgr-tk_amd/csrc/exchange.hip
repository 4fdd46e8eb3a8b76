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
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pgr_ctx.h"
#include "pgr_index.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

std::atomic<bool> g_rccl_loaded{false};  // set once rccl() has run: lets error paths name an RCCL error without loading the library

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        struct Mark {
            ~Mark() { g_rccl_loaded.store(true); }
        } mark;
        // a copy that is already in the process (PyTorch loads its own librccl.so) is taken first: two RCCL instances in one
        // process would each keep their own communicators, topology state and IPC handles on the same GPUs
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (r.handle) break;
        }
        if (!r.handle && dlsym(RTLD_DEFAULT, "ncclCommInitRank")) r.handle = dlopen(nullptr, RTLD_NOW);  // linked into the host program
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.handle) break;
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
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
        r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}

}  // namespace

struct pgr_exchange {
    pgr_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    // world == 1: every collective of this file is a device-to-device copy on the exchange's stream (an all-gather of one rank, a
    // broadcast from oneself; send/recv never happen).  RCCL is then neither loaded nor initialised (librccl.so.1 is a 570 MB
    // image whose first load and ncclCommInitRank cost seconds -- minutes from a cold disk -- for nothing); the context option
    // exchange_rccl_world1 = 1 asks for the real communicator all the same (plumbing tests on a one-GPU box).
    bool local = false;
    hipStream_t stream = nullptr;          // the collective's own stream: overlaps the compute stream
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    unsigned long long *d_cnt = nullptr;   // [1 + world]: this rank's count, then everybody's
    unsigned long long *h_cnt = nullptr;   // pinned [1 + world]
    bool in_flight = false;
    std::vector<uint64_t> splitters;  // world - 1 key-range boundaries of the last pgr_exchange_shard_records that sampled
    bool have_splitters = false;
    bool broken = false;  // a collective timed out and the communicator was aborted: every later call fails at once
};

// Watchdog (the reference has no failure detection of its own; a multi-process build needs one).
// A peer that died, a fabric that never comes up or a rank that took another code path leaves ncclCommInitRank or a
// collective waiting for ever; the context option exchange_timeout_s (PGR_EXCHANGE_TIMEOUT_S when the context is created;
// default 300 s, 0 = wait for ever) bounds both.  On a timeout the
// communicator is aborted (ncclCommAbort), the call returns PGR_ERR_DEVICE and the exchange refuses further work, so that a
// host program can fall back to another transport or fail with a message instead of hanging.
static double exchange_timeout_s(const pgr_ctx *ctx) { return ctx->opt.exchange_timeout_s < 0 ? 0.0 : (double)ctx->opt.exchange_timeout_s; }
// (a wait for a collective is also a wait for the slowest peer to ARRIVE at it: its own, longer bound -- never shorter than the
// one for the rendezvous, unless that one was set to "for ever")
static double exchange_collective_timeout_s(const pgr_ctx *ctx) {
    const double c = ctx->opt.exchange_collective_timeout_s < 0 ? 0.0 : (double)ctx->opt.exchange_collective_timeout_s;
    return c;
}

// wait for the exchange's stream, bounded
static int exchange_sync(pgr_exchange *x, const char *what) {
    pgr_ctx *ctx = x->ctx;
    if (x->broken) return ctx->fail(PGR_ERR_STATE, "this exchange was aborted after a timeout");
    const double limit = exchange_collective_timeout_s(ctx);
    if (limit <= 0) {
        PGR_HIP(ctx, hipStreamSynchronize(x->stream));
        return PGR_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(x->stream);
        if (q == hipSuccess) return PGR_OK;
        if (q != hipErrorNotReady) return ctx->fail(PGR_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(q));
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > limit) break;
        if (el > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(el > 0.1 ? 1000 : 50));  // spin first: steps are ms
    }
    x->broken = true;
    if (x->comm && g_rccl_loaded.load() && rccl().CommAbort) (void)rccl().CommAbort(x->comm);
    x->comm = nullptr;
    char msg[200];
    snprintf(msg, sizeof msg, "%s did not complete within %.0f s (option exchange_collective_timeout_s): communicator aborted, rank %d of %d", what,
             limit, x->rank, x->world);
    return ctx->fail(PGR_ERR_DEVICE, msg);
}

// Blocks from the context's caching allocator belong to the context's stream (work queued there may still read a recycled block, and
// debug_poison fills it there): before the exchange's stream touches blocks that were just taken it is put behind the context's stream.
// (The other direction is the host's: every function here waits for the exchange's stream -- exchange_sync -- before its blocks go back.)
static int order_behind_compute(pgr_exchange *x) {
    pgr_ctx *ctx = x->ctx;
    PGR_HIP(ctx, hipEventRecord(x->ev_ready, ctx->stream));
    PGR_HIP(ctx, hipStreamWaitEvent(x->stream, x->ev_ready, 0));
    return PGR_OK;
}

static_assert(PGR_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "pgr_hip.h and rccl.h disagree on the unique id size");

static std::string nccl_error_string(ncclResult_t r) {
    if (g_rccl_loaded.load() && rccl().GetErrorString) return rccl().GetErrorString(r);
    return r == ncclUnhandledCudaError ? "device-to-device copy failed (one-rank exchange without RCCL)" : "error " + std::to_string((int)r);
}

#define PGR_NCCL(ctx, expr)                                                                          \
    do {                                                                                             \
        ncclResult_t _r = (expr);                                                                    \
        if (_r != ncclSuccess)                                                                       \
            return (ctx)->fail(PGR_ERR_DEVICE, std::string(#expr) + ": " + nccl_error_string(_r)); \
    } while (0)

namespace {
// what the functions below call collectives through: RCCL, or -- for an exchange of one rank -- copies on the exchange's stream
struct Xport {
    const pgr_exchange *x;
    static ncclResult_t copy(const void *src, void *dst, size_t words, hipStream_t st) {
        if (src == dst || words == 0) return ncclSuccess;
        return hipMemcpyAsync(dst, src, words * 8, hipMemcpyDeviceToDevice, st) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
    }
    ncclResult_t AllGather(const void *s, void *r, size_t n, ncclDataType_t t, ncclComm_t c, hipStream_t st) const {
        return x->local ? copy(s, r, n, st) : rccl().AllGather(s, r, n, t, c, st);  // (every payload of this file is ncclUint64 words)
    }
    ncclResult_t Broadcast(const void *s, void *r, size_t n, ncclDataType_t t, int root, ncclComm_t c, hipStream_t st) const {
        return x->local ? copy(s, r, n, st) : rccl().Broadcast(s, r, n, t, root, c, st);
    }
    ncclResult_t Send(const void *s, size_t n, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) const {
        return x->local ? ncclInvalidUsage : rccl().Send(s, n, t, peer, c, st);  // (one rank has no peer)
    }
    ncclResult_t Recv(void *r, size_t n, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) const {
        return x->local ? ncclInvalidUsage : rccl().Recv(r, n, t, peer, c, st);
    }
    ncclResult_t GroupStart() const { return x->local ? ncclSuccess : rccl().GroupStart(); }
    ncclResult_t GroupEnd() const { return x->local ? ncclSuccess : rccl().GroupEnd(); }
    std::string GetErrorString(ncclResult_t r) const { return nccl_error_string(r); }
};
}  // namespace

namespace {
// A blocking RCCL entry (loading the library itself, ncclGetUniqueId, ncclCommInitRank) runs on a thread of its own beside a clock
// (option exchange_timeout_s).  The thread says which step it is in, so that a time-out NAMES the call that did not return; a thread
// that is still inside RCCL when the clock runs out is detached and owns everything it touches through the shared state.
struct Guarded {
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
    std::atomic<const char *> step{"starting"};
    std::string load_error;
    ncclResult_t r = ncclSuccess;
    ncclComm_t comm = nullptr;
    ncclUniqueId id;
    std::chrono::steady_clock::time_point t_step = std::chrono::steady_clock::now();
};

template <class F>
int run_beside_a_clock(pgr_ctx *ctx, const char *what, int rank, int world, std::shared_ptr<Guarded> st, F body) {
    const int device = ctx->device;
    const bool debug = ctx->opt.debug != 0;
    std::thread worker([st, device, body, debug] {
        (void)hipSetDevice(device);
        const auto t0 = std::chrono::steady_clock::now();
        st->step = "loading librccl.so.1 (dlopen)";
        Rccl &R = rccl();
        const double t_load = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (!R.err.empty()) st->load_error = R.err;
        else body(R, *st);
        if (debug)
            fprintf(stderr, "[pgr] exchange: RCCL ready after %.3f s, %s returned after %.3f s\n", t_load, st->step.load(),
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        std::lock_guard<std::mutex> g(st->m);
        st->done = true;
        st->cv.notify_all();
    });
    const double limit = exchange_timeout_s(ctx);
    std::unique_lock<std::mutex> g(st->m);
    if (limit <= 0) st->cv.wait(g, [&] { return st->done; });
    else st->cv.wait_for(g, std::chrono::duration<double>(limit), [&] { return st->done; });
    if (!st->done) {
        g.unlock();
        worker.detach();
        char msg[320];
        snprintf(msg, sizeof msg, "%s: %s did not return within %.0f s (option exchange_timeout_s): rank %d of %d", what, st->step.load(),
                 limit, rank, world);
        return ctx->fail(PGR_ERR_DEVICE, msg);
    }
    g.unlock();
    worker.join();
    if (!st->load_error.empty()) return ctx->fail(PGR_ERR_DEVICE, st->load_error);
    if (st->r != ncclSuccess) return ctx->fail(PGR_ERR_DEVICE, std::string(what) + ": " + st->step.load() + ": " + nccl_error_string(st->r));
    return PGR_OK;
}
}  // namespace

extern "C" int pgr_exchange_unique_id(pgr_ctx *ctx, uint8_t *id) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!id) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    PGR_ENTER(ctx);
    auto st = std::make_shared<Guarded>();
    const int rc = run_beside_a_clock(ctx, "pgr_exchange_unique_id", 0, 0, st, [](Rccl &R, Guarded &g) {
        g.step = "ncclGetUniqueId";
        g.r = R.GetUniqueId(&g.id);
    });
    if (rc) return rc;
    memcpy(id, st->id.internal, PGR_UNIQUE_ID_BYTES);
    return PGR_OK;
}

extern "C" int pgr_exchange_create(pgr_ctx *ctx, const uint8_t *id, int rank, int world, pgr_exchange **out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!out || world < 1 || rank < 0 || rank >= world) return ctx->fail(PGR_ERR_INVALID_ARG, "bad exchange arguments");
    *out = nullptr;
    const bool local = world == 1 && !ctx->opt.exchange_rccl_world1;
    if (!local && !id) return ctx->fail(PGR_ERR_INVALID_ARG, "bad exchange arguments: no unique id");
    PGR_ENTER(ctx);
    pgr_exchange *x = new pgr_exchange();
    x->ctx = ctx;
    x->rank = rank;
    x->world = world;
    x->local = local;
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
    if (local) {
        if (ctx->opt.debug) fprintf(stderr, "[pgr] exchange of one rank: copies on its own stream, RCCL not loaded\n");
        *out = x;
        return PGR_OK;
    }
    // ncclCommInitRank blocks until all `world` ranks have called it (see exchange_timeout_s)
    auto st = std::make_shared<Guarded>();
    memcpy(st->id.internal, id, PGR_UNIQUE_ID_BYTES);
    const int rc = run_beside_a_clock(ctx, "pgr_exchange_create", rank, world, st, [world, rank](Rccl &R, Guarded &g) {
        g.step = "ncclCommInitRank";
        g.r = R.CommInitRank(&g.comm, world, g.id, rank);
    });
    if (rc) return bail(rc);
    x->comm = st->comm;
    *out = x;
    return PGR_OK;
}

extern "C" void pgr_exchange_destroy(pgr_exchange *x) {
    if (!x) return;
    (void)hipSetDevice(x->ctx->device);
    if (x->stream && !x->broken) (void)hipStreamSynchronize(x->stream);
    if (x->comm && g_rccl_loaded.load()) (void)rccl().CommDestroy(x->comm);
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
    if (x->broken) return ctx->fail(PGR_ERR_STATE, "this exchange was aborted after a timeout");
    if (x->in_flight) return ctx->fail(PGR_ERR_STATE, "an all-gather is already in flight on this exchange (call pgr_exchange_wait)");
    if (!d_local || !d_out || cap_per_rank == 0) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (n_local > cap_per_rank) return ctx->fail(PGR_ERR_INVALID_ARG, "this rank's shimmer list exceeds cap_per_rank");
    const Xport R{x};
    PGR_ENTER(ctx);
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
    PGR_ENTER(ctx);
    int rc = exchange_sync(x, "shimmer all-gather");  // (ev_done is the last thing on the exchange's stream)
    if (rc) {
        x->in_flight = false;
        return rc;
    }
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
    if (x->broken) return ctx->fail(PGR_ERR_STATE, "this exchange was aborted after a timeout");
    if (x->in_flight) return ctx->fail(PGR_ERR_STATE, "an all-gather is already in flight on this exchange");
    if (s && s->n && !rids) return ctx->fail(PGR_ERR_INVALID_ARG, "null rid list");
    const Xport R{x};
    PGR_ENTER(ctx);
    const uint64_t n_local = s ? s->count : 0;
    // phase 1: counts (the compute stream's work is done: pgr_shmmrs_compute synchronizes)
    x->h_cnt[0] = n_local;
    PGR_HIP(ctx, hipMemcpyAsync(x->d_cnt, x->h_cnt, sizeof(unsigned long long), hipMemcpyHostToDevice, x->stream));
    PGR_NCCL(ctx, R.AllGather(x->d_cnt, x->d_cnt + 1, 1, ncclUint64, x->comm, x->stream));
    PGR_HIP(ctx, hipMemcpyAsync(x->h_cnt + 1, x->d_cnt + 1, (size_t)x->world * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                x->stream));
    int rc;
    if ((rc = exchange_sync(x, "count all-gather"))) return rc;
    uint64_t cap = 0, total = 0;
    for (int r = 0; r < x->world; ++r) {
        cap = std::max<uint64_t>(cap, x->h_cnt[1 + r]);
        total += x->h_cnt[1 + r];
    }
    if (n_gathered) *n_gathered = total;
    if (cap == 0) return PGR_OK;
    // phase 2: padded payload
    pgr::Tmp local(ctx), out(ctx);
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

// ------------------------------------------------------------------------------------------------
// Key-range sharded index ("key-range partitioned"; the pieces without a collective are in
// csrc/shard.hip).  One variable all-to-all: every rank cuts ITS pair records into `world` key ranges (splitters = quantiles
// of a pooled sample of first hashes, identical on every rank) and sends range r to rank r with grouped ncclSend/ncclRecv,
// straight into the receiver's index.  Each rank then sorts total/world records, whatever the number of ranks.
namespace {
constexpr uint32_t SHARD_SAMPLES = 4096;  // per rank; the imbalance of the ranges is ~1/sqrt(world * samples)
constexpr size_t REC_WORDS = sizeof(pgr_frag_rec) / 8;
static_assert(sizeof(pgr_frag_rec) % 8 == 0, "records travel as u64 words");
}  // namespace

extern "C" int pgr_exchange_shard_records(pgr_exchange *x, const pgr_frag_rec *d_recs, uint64_t n, pgr_index *ix,
                                          int reuse_splitters, uint64_t *splitters_out, uint64_t *n_received) {
    if (!x) return PGR_ERR_INVALID_ARG;
    pgr_ctx *ctx = x->ctx;
    if (x->broken) return ctx->fail(PGR_ERR_STATE, "this exchange was aborted after a timeout");
    if (x->in_flight) return ctx->fail(PGR_ERR_STATE, "an all-gather is in flight on this exchange (call pgr_exchange_wait)");
    if (!ix || (n && !d_recs)) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (ix->ctx != ctx) return ctx->fail(PGR_ERR_STATE, "index belongs to another context");
    const Xport R{x};
    PGR_ENTER(ctx);
    const int world = x->world, me = x->rank;
    int rc;
    if (reuse_splitters && !x->have_splitters) return ctx->fail(PGR_ERR_STATE, "no splitters yet: the first call must sample");
    // A failure that only THIS rank sees (out of memory, the 2^32 record limit, a partition error) must not leave the other
    // ranks blocked in the next collective: every collective of this call carries an error word, and all ranks leave together.
    constexpr uint64_t FAILED = ~0ull;
    int local_rc = PGR_OK;
    auto peers_failed = [&](const char *where) {
        return local_rc ? local_rc
                        : ctx->fail(PGR_ERR_STATE, std::string("pgr_exchange_shard_records: another rank failed ") + where +
                                                       " (its own error message says why); nothing was exchanged");
    };
    // Every device block of this call is taken BEFORE the first collective, and a failure to get one is this rank's error word
    // like any other local failure.  (The two blocks of the sample all-gather themselves -- 33 KB and world x 33 KB -- are the
    // exception: without them this rank cannot say anything; that, a failing enqueue of a copy or a collective -- a broken device
    // or communicator -- and nothing else is left to the watchdog.  pgr_shard_splitters runs on the same pooled sample on every
    // rank: it fails everywhere or nowhere.)
    const size_t row = (size_t)world + 1;
    pgr::Tmp d_part(ctx), d_cnt(ctx), d_mat(ctx);
    int alloc_rc = d_cnt.alloc(row * 8);  // (the small ones first: with them this rank can at least say that it failed)
    if (!alloc_rc) alloc_rc = d_mat.alloc((size_t)world * row * 8);
    if (!alloc_rc) alloc_rc = d_part.alloc(std::max<uint64_t>(n, 1) * sizeof(pgr_frag_rec));
    std::vector<uint64_t> splitters((size_t)std::max(world - 1, 1));
    if (reuse_splitters) {
        splitters = x->splitters;
    } else {
        // ---- 1. pooled sample of first hashes -> splitters (the same on every rank)
        std::vector<uint64_t> mine(1 + SHARD_SAMPLES, 0);
        uint32_t n_s = 0;
        local_rc = alloc_rc ? alloc_rc : pgr_shard_sample_keys(ctx, d_recs, n, SHARD_SAMPLES, mine.data() + 1, &n_s);  // synchronizes
        mine[0] = local_rc ? FAILED : n_s;
        pgr::Tmp d_smp(ctx), d_all(ctx);
        if ((rc = d_smp.alloc(mine.size() * 8)) || (rc = d_all.alloc((size_t)world * mine.size() * 8)) || (rc = order_behind_compute(x))) return rc;
        std::vector<uint64_t> all((size_t)world * mine.size());
        PGR_HIP(ctx, hipMemcpyAsync(d_smp.p, mine.data(), mine.size() * 8, hipMemcpyHostToDevice, x->stream));
        PGR_NCCL(ctx, R.AllGather(d_smp.p, d_all.p, mine.size(), ncclUint64, x->comm, x->stream));
        PGR_HIP(ctx, hipMemcpyAsync(all.data(), d_all.p, all.size() * 8, hipMemcpyDeviceToHost, x->stream));
        if ((rc = exchange_sync(x, "sample all-gather"))) return rc;
        std::vector<uint64_t> pool;
        bool any_failed = false;
        for (int r = 0; r < world; ++r) {
            const uint64_t *blk = all.data() + (size_t)r * mine.size();
            any_failed = any_failed || blk[0] == FAILED;
            if (blk[0] != FAILED) pool.insert(pool.end(), blk + 1, blk + 1 + std::min<uint64_t>(blk[0], SHARD_SAMPLES));
        }
        if (any_failed) return peers_failed("while sampling its keys");
        if (pgr_shard_splitters(pool.data(), pool.size(), world, splitters.data())) return ctx->fail(PGR_ERR_INTERNAL, "splitters");
        x->splitters = splitters;
        x->have_splitters = true;
    }
    if (splitters_out)
        for (int j = 0; j + 1 < world; ++j) splitters_out[j] = splitters[(size_t)j];
    // ---- 2. stable partition of this rank's records by destination
    std::vector<uint64_t> send_cnt((size_t)world + 1, 0);  // [world] = error word
    if (alloc_rc && (!d_cnt.p || !d_mat.p)) return alloc_rc;  // (not even the count blocks: this rank cannot take part; see above)
    local_rc = alloc_rc ? alloc_rc
                        : pgr_shard_partition(ctx, d_recs, n, splitters.data(), world, d_part.as<pgr_frag_rec>(), send_cnt.data());
    send_cnt[(size_t)world] = local_rc ? FAILED : 0;
    // ---- 3. everybody's counts: M[src][dst] (+ the error word of src)
    std::vector<uint64_t> mat((size_t)world * row);
    if ((rc = order_behind_compute(x))) return rc;
    PGR_HIP(ctx, hipMemcpyAsync(d_cnt.p, send_cnt.data(), row * 8, hipMemcpyHostToDevice, x->stream));
    PGR_NCCL(ctx, R.AllGather(d_cnt.p, d_mat.p, row, ncclUint64, x->comm, x->stream));
    PGR_HIP(ctx, hipMemcpyAsync(mat.data(), d_mat.p, mat.size() * 8, hipMemcpyDeviceToHost, x->stream));
    if ((rc = exchange_sync(x, "count all-gather"))) return rc;
    for (int s = 0; s < world; ++s)
        if (mat[(size_t)s * row + (size_t)world] == FAILED) return peers_failed("while partitioning its records");
    uint64_t recv_total = 0;
    for (int s = 0; s < world; ++s) recv_total += mat[(size_t)s * row + me];
    // ---- 3b. room for what arrives: a rank that cannot take its range says so before anybody sends
    if (ix->n_raw + recv_total >= (1ull << 32))
        local_rc = ctx->fail(PGR_ERR_INVALID_ARG, "index shard would hold 2^32 or more records");
    else
        local_rc = pgr::index_grow_raw(ctx, ix, ix->n_raw + recv_total);
    x->h_cnt[0] = local_rc ? FAILED : 0;
    if ((rc = order_behind_compute(x))) return rc;  // (the index's append block may have moved: its copy ran on the context's stream)
    PGR_HIP(ctx, hipMemcpyAsync(x->d_cnt, x->h_cnt, sizeof(unsigned long long), hipMemcpyHostToDevice, x->stream));
    PGR_NCCL(ctx, R.AllGather(x->d_cnt, x->d_cnt + 1, 1, ncclUint64, x->comm, x->stream));
    PGR_HIP(ctx, hipMemcpyAsync(x->h_cnt + 1, x->d_cnt + 1, (size_t)world * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                x->stream));
    if ((rc = exchange_sync(x, "ready all-gather"))) return rc;
    for (int s = 0; s < world; ++s)
        if (x->h_cnt[1 + s] == FAILED) return peers_failed("while making room for its key range");
    // ---- 4. the payload: blocks arrive in source-rank order, each in its sender's (sid, frg_id) order
    PGR_NCCL(ctx, R.GroupStart());
    uint64_t s_off = 0, r_off = 0;
    ncclResult_t nr = ncclSuccess;
    hipError_t he = hipSuccess;
    for (int p = 0; p < world; ++p) {
        const uint64_t sc = send_cnt[(size_t)p], rcv = mat[(size_t)p * row + me];
        if (p == me) {
            if (sc && he == hipSuccess)
                he = hipMemcpyAsync(ix->raw + ix->n_raw + r_off, d_part.as<pgr_frag_rec>() + s_off, sc * sizeof(pgr_frag_rec),
                                    hipMemcpyDeviceToDevice, x->stream);
        } else {
            if (sc && nr == ncclSuccess)
                nr = R.Send(d_part.as<pgr_frag_rec>() + s_off, sc * REC_WORDS, ncclUint64, p, x->comm, x->stream);
            if (rcv && nr == ncclSuccess)
                nr = R.Recv(ix->raw + ix->n_raw + r_off, rcv * REC_WORDS, ncclUint64, p, x->comm, x->stream);
        }
        s_off += sc;
        r_off += rcv;
    }
    const ncclResult_t ge = R.GroupEnd();
    if (nr != ncclSuccess || ge != ncclSuccess)
        return ctx->fail(PGR_ERR_DEVICE, std::string("ncclSend/ncclRecv: ") + R.GetErrorString(nr != ncclSuccess ? nr : ge));
    if (he != hipSuccess) return ctx->fail(PGR_ERR_DEVICE, std::string("local block copy: ") + hipGetErrorString(he));
    if ((rc = exchange_sync(x, "record all-to-all"))) return rc;
    PGR_HIP(ctx, hipGetLastError());
    ix->n_raw += recv_total;
    ix->finalized = false;
    if (n_received) *n_received = recv_total;
    return PGR_OK;
}

// The replicated query index from the finalized shards: every rank receives every shard in rank order (ncclBroadcast per
// rank, grouped).  The ranges are disjoint and ascending, so the concatenation is already the sorted record array of the
// whole index; pgr_index_finalize notices and skips the sort.
extern "C" int pgr_exchange_allgather_index(pgr_exchange *x, const pgr_index *shard, pgr_index **out) {
    if (!x) return PGR_ERR_INVALID_ARG;
    pgr_ctx *ctx = x->ctx;
    if (!shard || !out) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (!shard->finalized) return ctx->fail(PGR_ERR_STATE, "shard index not finalized");
    if (x->broken) return ctx->fail(PGR_ERR_STATE, "this exchange was aborted after a timeout");
    if (x->in_flight) return ctx->fail(PGR_ERR_STATE, "an all-gather is in flight on this exchange");
    *out = nullptr;
    const Xport R{x};
    PGR_ENTER(ctx);
    const int world = x->world, me = x->rank;
    int rc;
    x->h_cnt[0] = shard->n;
    PGR_HIP(ctx, hipMemcpyAsync(x->d_cnt, x->h_cnt, sizeof(unsigned long long), hipMemcpyHostToDevice, x->stream));
    PGR_NCCL(ctx, R.AllGather(x->d_cnt, x->d_cnt + 1, 1, ncclUint64, x->comm, x->stream));
    PGR_HIP(ctx, hipMemcpyAsync(x->h_cnt + 1, x->d_cnt + 1, (size_t)world * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                x->stream));
    if ((rc = exchange_sync(x, "count all-gather"))) return rc;
    uint64_t total = 0;
    for (int r = 0; r < world; ++r) total += x->h_cnt[1 + r];
    if (total >= (1ull << 32)) return ctx->fail(PGR_ERR_INVALID_ARG, "replicated index would hold 2^32 or more records");
    pgr_index *full = nullptr;
    if ((rc = pgr_index_create(ctx, &shard->spec, &full))) return rc;
    if ((rc = pgr::index_grow_raw(ctx, full, total))) {
        pgr_index_destroy(full);
        return rc;
    }
    if ((rc = order_behind_compute(x))) {
        pgr_index_destroy(full);
        return rc;
    }
    ncclResult_t nr = R.GroupStart();
    uint64_t off = 0;
    for (int r = 0; r < world && nr == ncclSuccess; ++r) {
        const uint64_t c = x->h_cnt[1 + r];
        if (c) nr = R.Broadcast(r == me ? (const void *)shard->recs : (const void *)(full->raw + off), full->raw + off, c * REC_WORDS,
                                ncclUint64, r, x->comm, x->stream);
        off += c;
    }
    const ncclResult_t ge = R.GroupEnd();
    if (nr != ncclSuccess || ge != ncclSuccess) {
        pgr_index_destroy(full);
        return ctx->fail(PGR_ERR_DEVICE, std::string("index all-gather: ") + R.GetErrorString(nr != ncclSuccess ? nr : ge));
    }
    if ((rc = exchange_sync(x, "index all-gather"))) {
        pgr_index_destroy(full);
        return rc;
    }
    full->n_raw = total;
    full->next_sid = shard->next_sid;
    if ((rc = pgr_index_finalize(ctx, full))) {
        pgr_index_destroy(full);
        return rc;
    }
    *out = full;
    return PGR_OK;
}
