// api.hip -- host side of libpgrhip.so: context, resident batches, results, the host-buffer entry points of the C ABI declared
// in include/pgr_hip.h.  The orchestration of a pass of the sequence_to_shmmrs pipeline is csrc/pipeline.hip.
#include <algorithm>
#include <chrono>
#include <atomic>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "pgr_ctx.h"
#include "pgr_index.h"
#include "pgr_small.h"

using namespace pgr;

namespace {
using Tmp_list = pgr::Tmp;  // small RAII device allocation from the context's caching allocator
}  // namespace

static std::string g_create_error;

extern "C" const char *pgr_version(void) { return "pgr-hip 0.4.0 (gfx950)"; }

extern "C" const char *pgr_last_error(const pgr_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// results of the library: plain malloc'd blocks, or (large shimmer lists) pinned blocks that go back to a process-wide pool --
// the DMA engine writes them directly, and a host that calls batch after batch pays for pinning and page faults once
extern "C" void pgr_free(void *p) { pgr::result_block_release(p); }

namespace {
struct OptionName {
    const char *name;
    int64_t pgr_ctx::Options::*field;
};
const std::vector<OptionName> &option_names() {
    using O = pgr_ctx::Options;
    static const std::vector<OptionName> v = {
        {"debug", &O::debug}, {"debug_times", &O::debug_times}, {"gpu_pack", &O::gpu_pack}, {"no_small_path", &O::no_small_path},
        {"no_pipeline", &O::no_pipeline}, {"early_sync_bp", &O::early_sync_bp}, {"index_full_sort", &O::index_full_sort},
        {"index_two_key_sort", &O::index_two_key_sort}, {"no_fused_query", &O::no_fused_query}, {"direct_query_result", &O::direct_query_result}, {"direct_query_results_delivered", &O::direct_query_results_delivered}, {"direct_query_lds_kb", &O::direct_query_lds_kb},
        {"no_query_chaining", &O::no_query_chaining}, {"no_query_level1", &O::no_query_level1}, {"no_query_keys", &O::no_query_keys}, {"lut_extra_bits", &O::lut_extra_bits}, {"query_global_sort", &O::query_global_sort},
        {"fused_query_hits", &O::fused_query_hits}, {"exchange_timeout_s", &O::exchange_timeout_s},
        {"exchange_collective_timeout_s", &O::exchange_collective_timeout_s}, {"exchange_rccl_world1", &O::exchange_rccl_world1}, {"debug_poison", &O::debug_poison}, {"debug_inject_stale_segments", &O::debug_inject_stale_segments},
        {"no_island_relay", &O::no_island_relay}, {"no_sub_tile_islands", &O::no_sub_tile_islands}, {"island_settle", &O::island_settle}, {"no_short_tiles", &O::no_short_tiles}, {"no_pre_islands", &O::no_pre_islands}, {"no_early_islands", &O::no_early_islands}, {"no_early_merge", &O::no_early_merge},  {"early_islands_in_stream", &O::early_islands_in_stream}, {"island_chunk_min", &O::island_chunk_min},
        {"back_priority", &O::back_priority}, {"no_fix_stream", &O::no_fix_stream}, {"no_stage1_only", &O::no_stage1_only}, {"pipe_staged_records", &O::pipe_staged_records},
        {"lds_match", &O::lds_match}, {"no_direct_h2d", &O::no_direct_h2d},
        {"pipe_small_list", &O::pipe_small_list}, {"pipe_persistent_list", &O::pipe_persistent_list}, {"front_priority", &O::front_priority}};
    return v;
}
}  // namespace

extern "C" int pgr_ctx_create(int device, pgr_ctx **out) {
    if (!out) return PGR_ERR_INVALID_ARG;
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e) +
                         " (libpgrhip has no CPU fallback)";
        return PGR_ERR_DEVICE;
    }
    if (device < 0 || device >= n_dev) {
        g_create_error = "device index out of range";
        return PGR_ERR_INVALID_ARG;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return PGR_ERR_DEVICE;
    }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return PGR_ERR_DEVICE;
    }
    pgr_ctx *ctx = new pgr_ctx();
    ctx->device = device;
    for (const OptionName &o : option_names()) {
        std::string env = "PGR_";
        for (const char *c = o.name; *c; ++c) env += (char)toupper((unsigned char)*c);
        if (const char *v = getenv(env.c_str())) {
            char *end = nullptr;
            const long long x = strtoll(v, &end, 10);
            const bool number = end && end != v && *end == '\0';
            // switches: PGR_X=1, PGR_X=<number>, or a bare PGR_X= / PGR_X=on.  Options that ARE a number (a time-out, a size)
            // keep their default when the value is not one: "PGR_EXCHANGE_TIMEOUT_S=5m" must not become a 1-second time-out
            static const char *const numeric[] = {"early_sync_bp", "fused_query_hits", "exchange_timeout_s", "exchange_collective_timeout_s",
                                                  "island_chunk_min", "back_priority", "direct_query_lds_kb"};
            bool is_numeric = false;
            for (const char *nm : numeric) is_numeric = is_numeric || !strcmp(nm, o.name);
            if (number) ctx->opt.*(o.field) = (int64_t)x;
            else if (!is_numeric) ctx->opt.*(o.field) = 1;
            else fprintf(stderr, "[pgr] %s=%s is not a number: the option keeps its default (%lld)\n", env.c_str(), v, (long long)(ctx->opt.*(o.field)));
        }
    }
    e = hipSetDevice(device);
    if (e == hipSuccess && ctx->opt.front_priority > 0) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, hi);
    } else if (e == hipSuccess) {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ctx->d2h_ev[i], hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ctx->pre_ev[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->pre_stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&ctx->ev[i]);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreate(&ctx->cev[i]);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_end);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_alloc);
    if (e != hipSuccess) {
        g_create_error = std::string("context setup: ") + hipGetErrorString(e);
        delete ctx;
        return PGR_ERR_DEVICE;
    }
    *out = ctx;
    return PGR_OK;
}

// A second context for work that should run BESIDE another context's: the runtime multiplexes streams onto a handful of hardware
// queues, and two contexts whose streams share one run their batches one after the other whatever the host threads do (two query
// batches in flight: 0.41 ms per batch side by side, 0.74 in order -- which of the two a plain pgr_ctx_create gives depends on how
// many streams the process has created before).  Candidates for the new context's stream are tried against `other`'s until one
// runs side by side with it (ctx.hip: streams_run_side_by_side); with none in eight the last one is kept.
extern "C" int pgr_ctx_create_beside(pgr_ctx *other, pgr_ctx **out) {
    if (!out) return PGR_ERR_INVALID_ARG;
    *out = nullptr;
    if (!other) return PGR_ERR_INVALID_ARG;
    pgr_ctx *ctx = nullptr;
    int rc = pgr_ctx_create(other->device, &ctx);
    if (rc) return rc;
    unsigned long long *d_scratch = nullptr;
    if (hipMalloc((void **)&d_scratch, 64) != hipSuccess) {
        pgr_ctx_destroy(ctx);
        g_create_error = "hipMalloc failed";
        return PGR_ERR_NOMEM;
    }
    std::vector<hipStream_t> rejected;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (int tries = 0; tries < 8; ++tries) {
        if (streams_run_side_by_side(other->stream, ctx->stream, d_scratch)) break;
        hipStream_t cand = nullptr;
        const hipError_t e = ctx->opt.front_priority > 0 ? hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, hi)
                                                         : hipStreamCreateWithFlags(&cand, hipStreamNonBlocking);
        if (e != hipSuccess) break;
        rejected.push_back(ctx->stream);  // (kept until the search is over: the next candidate lands on another queue)
        ctx->stream = cand;
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    (void)hipFree(d_scratch);
    *out = ctx;
    return PGR_OK;
}

extern "C" int pgr_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return PGR_ERR_INVALID_ARG;
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        g_create_error = std::string("hipHostRegister: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? PGR_ERR_NOMEM : PGR_ERR_DEVICE;
    }
    return PGR_OK;
}

extern "C" int pgr_host_unregister(void *p) {
    if (!p) return PGR_ERR_INVALID_ARG;
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        g_create_error = std::string("hipHostUnregister: ") + hipGetErrorString(e);
        return PGR_ERR_DEVICE;
    }
    return PGR_OK;
}

extern "C" int pgr_ctx_trim(pgr_ctx *ctx) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    PGR_ENTER(ctx);
    PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->back_stream) PGR_HIP(ctx, hipStreamSynchronize(ctx->back_stream));
    if (ctx->fix_stream) PGR_HIP(ctx, hipStreamSynchronize(ctx->fix_stream));
    for (auto &kv : ctx->free_blocks) {
        ctx->raw_free(kv.second.p);
        ctx->drop_events(kv.second);
        ctx->live_bytes -= kv.first;
    }
    ctx->free_blocks.clear();
    ctx->cached_bytes = 0;
    for (pgr::Lane *l : ctx->spare_lanes) {
        pgr::lane_release(ctx, *l);
        delete l;
    }
    ctx->spare_lanes.clear();
    return PGR_OK;
}

// One hipMalloc, touched once, that every later device allocation of the context is carved from (csrc/pgr_ctx.h: Arena).
extern "C" int pgr_ctx_reserve(pgr_ctx *ctx, uint64_t bytes) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    PGR_ENTER(ctx);
    return ctx->reserve((size_t)bytes);
}

extern "C" int pgr_ctx_arena_stats(pgr_ctx *ctx, uint64_t *reserved_bytes, uint64_t *used_bytes, uint64_t *peak_used_bytes,
                                   uint64_t *fallback_bytes, uint64_t *fallback_calls) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (reserved_bytes) *reserved_bytes = ctx->arena_bytes;
    if (used_bytes) *used_bytes = ctx->arena_used;
    if (peak_used_bytes) *peak_used_bytes = ctx->arena_peak;
    if (fallback_bytes) *fallback_bytes = ctx->fallback_bytes;
    if (fallback_calls) *fallback_calls = ctx->fallback_calls;
    return PGR_OK;
}

extern "C" int pgr_debug_take_hip_error(void) { return (int)hipGetLastError(); }

extern "C" int pgr_ctx_mem_stats(pgr_ctx *ctx, uint64_t *held_bytes, uint64_t *peak_bytes, int reset_peak) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (held_bytes) *held_bytes = ctx->live_bytes;
    if (peak_bytes) *peak_bytes = ctx->peak_bytes;
    if (reset_peak) ctx->peak_bytes = ctx->live_bytes;
    return PGR_OK;
}

extern "C" int pgr_ctx_set_option(pgr_ctx *ctx, const char *name, int64_t value) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!name) return ctx->fail(PGR_ERR_INVALID_ARG, "null option name");
    for (const OptionName &o : option_names())
        if (!strcmp(o.name, name)) {
            ctx->opt.*(o.field) = value;
            return PGR_OK;
        }
    return ctx->fail(PGR_ERR_INVALID_ARG, std::string("unknown option: ") + name);
}

extern "C" int pgr_ctx_get_option(const pgr_ctx *ctx, const char *name, int64_t *value) {
    if (!ctx || !name || !value) return PGR_ERR_INVALID_ARG;
    for (const OptionName &o : option_names())
        if (!strcmp(o.name, name)) {
            *value = ctx->opt.*(o.field);
            return PGR_OK;
        }
    return PGR_ERR_INVALID_ARG;
}

extern "C" void pgr_ctx_destroy(pgr_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->back_stream) (void)hipStreamSynchronize(ctx->back_stream);
    if (ctx->fix_stream) (void)hipStreamSynchronize(ctx->fix_stream);
    ctx->release_all();
    for (auto &ev : ctx->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->ev_alloc) (void)hipEventDestroy(ctx->ev_alloc);
    for (auto &ev : ctx->cev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : ctx->d2h_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : ctx->pre_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->back_stream) (void)hipStreamDestroy(ctx->back_stream);
    if (ctx->fix_stream) (void)hipStreamDestroy(ctx->fix_stream);
    if (ctx->pre_stream) (void)hipStreamDestroy(ctx->pre_stream);
    if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int pgr_ctx_synchronize(pgr_ctx *ctx) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PGR_OK;
}

extern "C" int pgr_ctx_last_prof(const pgr_ctx *ctx, pgr_prof *out) {
    if (!ctx || !out) return PGR_ERR_INVALID_ARG;
    *out = ctx->prof;
    return PGR_OK;
}

// ------------------------------------------------------------------------------------------------
// batches
static int batch_alloc(pgr_ctx *ctx, uint32_t n, const uint64_t *lens, pgr_batch **out) {
    pgr_batch *b = new pgr_batch();
    b->ctx = ctx;
    b->n = n;
    b->h_word_off.resize((size_t)n + 1);
    b->h_len.resize(n);
    uint64_t words = 0, bases = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (lens[i] >= (1ull << 31)) {
            delete b;
            return ctx->fail(PGR_ERR_TOO_LONG, "contig length >= 2^31 (MM128 position field is 31 bits)");
        }
        b->h_word_off[i] = words;
        b->h_len[i] = (uint32_t)lens[i];
        words += (lens[i] + 31) / 32;
        bases += lens[i];
    }
    b->h_word_off[n] = words;
    b->total_words = words;
    b->total_bases = bases;
    int rc;
    if ((rc = ctx->dmalloc((void **)&b->d.planes, std::max<uint64_t>(words, 1) * sizeof(uint2))) ||
        (rc = ctx->dmalloc((void **)&b->d.valid, std::max<uint64_t>(words, 1) * sizeof(uint32_t))) ||
        (rc = ctx->dmalloc((void **)&b->d.word_off, ((size_t)n + 1) * sizeof(uint64_t))) ||
        (rc = ctx->dmalloc((void **)&b->d.len, std::max<uint32_t>(n, 1) * sizeof(uint32_t))) ||
        (rc = ctx->dmalloc((void **)&b->d.n_invalid, std::max<uint32_t>(n, 1) * sizeof(uint32_t)))) {
        pgr_batch_destroy(b);
        return rc;
    }
    PGR_HIP(ctx, hipMemcpyAsync(b->d.word_off, b->h_word_off.data(), ((size_t)n + 1) * sizeof(uint64_t),
                                hipMemcpyHostToDevice, ctx->stream));
    if (n)
        PGR_HIP(ctx, hipMemcpyAsync(b->d.len, b->h_len.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice,
                                    ctx->stream));
    PGR_HIP(ctx, hipMemsetAsync(b->d.n_invalid, 0, std::max<uint32_t>(n, 1) * sizeof(uint32_t), ctx->stream));
    // no synchronization: the source vectors live as long as the batch and everything that reads the device arrays is
    // ordered behind these copies on the context's stream (the staging thread of the pipelined path waits for `ready`)
    // (an event of the batch's own: the staging thread of a pipelined call waits for THIS batch's copies while the calling thread
    // allocates the next sub-batches)
    if (hipEventCreateWithFlags(&b->ev_alloc, hipEventDisableTiming) != hipSuccess || hipEventRecord(b->ev_alloc, ctx->stream) != hipSuccess) {
        pgr_batch_destroy(b);
        return ctx->fail(PGR_ERR_DEVICE, "event record failed");
    }
    *out = b;
    return PGR_OK;
}

extern "C" void pgr_batch_destroy(pgr_batch *b) {
    if (!b) return;
    pgr_ctx *ctx = b->ctx;
    ctx->dfree(b->d.planes);
    ctx->dfree(b->d.valid);
    ctx->dfree(b->d.word_off);
    ctx->dfree(b->d.len);
    ctx->dfree(b->d.n_invalid);
    if (b->ev_alloc) (void)hipEventDestroy(b->ev_alloc);
    delete b;
}

extern "C" uint64_t pgr_batch_total_bases(const pgr_batch *b) { return b ? b->total_bases : 0; }

// Round-2 staging, kept behind PGR_GPU_PACK for A/B timing: the ASCII bytes themselves cross PCIe (1 B per base) and a
// kernel packs them.  Host threads fill two pinned windows, H2D + pack kernel on `st`.
// Thread-compatible with a compute call running on ctx->stream: touches only the batch, ctx->pinned, ctx->ws_ascii and
// the two events it is given.  Errors come back as a code + message (the caller owns ctx->err).
static int batch_stage_ascii_gpu(pgr_ctx *ctx, pgr_batch *b, uint32_t n, const uint8_t *const *seqs, const uint64_t *lens,
                                hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, std::string &err) {
    auto fail = [&](int code, const std::string &m) {
        err = m;
        return code;
    };
    // the pinned windows are reused from call to call: a previous staging whose copies nobody has waited for yet
    // (two batches staged back to back) must be over before the host overwrites them
    if (ctx->staged_unsynced && hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail(PGR_ERR_DEVICE, "H2D pipeline failed");
    if (st != ctx->stream && b->ev_alloc && hipStreamWaitEvent(st, b->ev_alloc, 0) != hipSuccess)  // batch_alloc's copies (main stream)
        return fail(PGR_ERR_DEVICE, "H2D pipeline failed");
    // ASCII stream: word wi of the batch <-> bytes [32*wi, 32*wi+32).  Two pinned windows of 32 MiB: while window
    // i is on its way to the GPU (H2D + pack kernel) the host threads fill window i+1.
    const uint64_t WIN_WORDS = 1ull << 20;  // 32 MiB of ASCII per window
    const uint64_t win_words = std::min<uint64_t>(std::max<uint64_t>(b->total_words, 1), WIN_WORDS);
    {   // (this may be the staging thread: the context's error string belongs to the calling thread)
        std::string why;
        if (ctx->ensure_pinned(2 * win_words * 32, &why) || ctx->ws_ascii.ensure(ctx, 2 * win_words * 32, &why))
            return fail(PGR_ERR_NOMEM, "staging buffers: " + why);
    }
    hipEvent_t done[2] = {ev0, ev1};
    bool used[2] = {false, false};
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_thr = std::max(1u, std::min(8u, hw ? hw / 2 : 1u));
    uint32_t c = 0;
    int slot = 0;
    for (uint64_t w0 = 0; w0 < b->total_words; w0 += win_words, slot ^= 1) {
        const uint64_t w1 = std::min(b->total_words, w0 + win_words);
        uint8_t *stage = (uint8_t *)ctx->pinned + (size_t)slot * win_words * 32;
        uint8_t *d_stage = (uint8_t *)ctx->ws_ascii.p + (size_t)slot * win_words * 32;
        if (used[slot] && hipEventSynchronize(done[slot]) != hipSuccess)  // this window's previous trip is over
            return fail(PGR_ERR_DEVICE, "H2D pipeline failed");
        while (c < n && b->h_word_off[c + 1] <= w0) ++c;
        // copy jobs of this window: (dst, src, len), split into <= 4 MiB pieces and spread over the threads
        struct Job {
            uint8_t *dst;
            const uint8_t *src;
            size_t len;
        };
        std::vector<Job> jobs;
        for (uint32_t cc = c; cc < n && b->h_word_off[cc] < w1; ++cc) {
            const uint64_t cw0 = b->h_word_off[cc], cw1 = b->h_word_off[cc + 1];
            const uint64_t lo = std::max(cw0, w0), hi = std::min(cw1, w1);
            if (lo >= hi) continue;
            const uint64_t b_lo = (lo - cw0) * 32, b_hi = std::min<uint64_t>((hi - cw0) * 32, lens[cc]);
            for (uint64_t o = b_lo; o < b_hi; o += (4u << 20))
                jobs.push_back(Job{stage + (lo - w0) * 32 + (o - b_lo), seqs[cc] + o,
                                   (size_t)std::min<uint64_t>(4u << 20, b_hi - o)});
        }
        if (jobs.size() <= 1 || n_thr == 1) {
            for (const Job &j : jobs) memcpy(j.dst, j.src, j.len);
        } else {
            std::atomic<size_t> next{0};
            auto work = [&]() {
                for (size_t i; (i = next.fetch_add(1)) < jobs.size();) memcpy(jobs[i].dst, jobs[i].src, jobs[i].len);
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < std::min<size_t>(n_thr, jobs.size()); ++t) th.emplace_back(work);
            work();
            for (auto &t : th) t.join();
        }
        if (hipMemcpyAsync(d_stage, stage, (w1 - w0) * 32, hipMemcpyHostToDevice, st) != hipSuccess)
            return fail(PGR_ERR_DEVICE, "H2D copy of the ASCII window failed");
        launch_pack_ascii(st, d_stage, w0, b->d, n, w1);
        if (hipEventRecord(done[slot], st) != hipSuccess) return fail(PGR_ERR_DEVICE, "H2D pipeline failed");
        used[slot] = true;
    }
    // (the per-contig counts of non-ACGT bytes stay on the device: pgr_shmmrs_compute fetches them only when a tile was
    // flagged).  On the context's own stream nothing waits here: the consumer is ordered behind the pack kernels and
    // synchronizes once at its end; the staging thread of the pipelined path (own stream) hands over finished batches.
    if (st != ctx->stream) {
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess)
            return fail(PGR_ERR_DEVICE, "pack kernel failed");
    } else {
        ctx->staged_unsynced = true;
    }
    return PGR_OK;
}

// Stage the bases of an allocated batch.  Two kinds of host input (pgr::StageSrc):
//   ASCII   host threads pack 32 bytes -> one plane word + one validity word (csrc/hostpack.cpp) straight into a pinned
//           window; 0.375 B per base cross PCIe instead of 1 B, and no pack kernel runs on the GPU;
//   packed  the caller's planes (+ validity plane) are copied into the pinned window as they are and a small kernel
//           cleans what the library relies on (bits past a contig's end, plane bits of invalid positions, the per-contig
//           counts of non-ACGT bytes).
// Two pinned windows: while window i is on its way to the GPU the host threads fill window i+1.
// Thread-compatible with a compute call running on ctx->stream: touches only the batch, ctx->pinned and the two events
// it is given.  Errors come back as a code + message (the caller owns ctx->err).
// pipe != NULL: this batch is one of a sequence staged back to back by the pipelined entry points; the two windows keep
// rolling from one batch into the next (no restart, no synchronization at the end: the caller orders its consumer behind
// an event it records on `st`).
struct StagePipe {
    bool used[2] = {false, false};
    int slot = 0;
};

// is this host address pinned (hipHostMalloc / hipHostRegister / pgr_host_register)?  The DMA engine reads such memory directly.
static bool host_ptr_is_pinned(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // (an ordinary host pointer is an "invalid value" to this call)
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// validity words [wl0, wl1) of a contig of `len` bases (v[0] = word wl0): does every base say "valid"?  (bits past the contig's end
// do not count)
static bool valid_words_all_set(const uint32_t *v, uint64_t wl0, uint64_t wl1, uint64_t len) {
    if (wl1 <= wl0) return true;
    const uint64_t nw = (len + 31) / 32;
    const uint64_t full_end = std::min(wl1, (len % 32) ? nw - 1 : nw);  // words below this one are whole
    uint32_t acc = 0xFFFFFFFFu;
    for (uint64_t w = wl0; w < full_end; ++w) acc &= v[w - wl0];
    if (acc != 0xFFFFFFFFu) return false;
    if (wl1 == nw && (len % 32) && nw - 1 >= wl0) {
        const uint32_t tail = ~(0xFFFFFFFFu >> (uint32_t)(len % 32));
        if ((v[nw - 1 - wl0] & tail) != tail) return false;
    }
    return true;
}
static int batch_stage(pgr_ctx *ctx, pgr_batch *b, uint32_t n, const StageSrc &src, hipStream_t st, hipEvent_t ev0,
                       hipEvent_t ev1, std::string &err, StagePipe *pipe = nullptr) {
    auto fail = [&](int code, const std::string &m) {
        err = m;
        return code;
    };
    // the pinned windows are reused from call to call: a previous staging whose copies nobody has waited for yet
    // (two batches staged back to back) must be over before the host overwrites them
    if (ctx->staged_unsynced && hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail(PGR_ERR_DEVICE, "H2D pipeline failed");
    if (st != ctx->stream && b->ev_alloc && hipStreamWaitEvent(st, b->ev_alloc, 0) != hipSuccess)  // batch_alloc's copies (main stream)
        return fail(PGR_ERR_DEVICE, "H2D pipeline failed");
    const bool packed = src.planes != nullptr;
    const bool gpu_pack = !packed && ctx->opt.gpu_pack;  // A/B switch: round-2 path (ASCII over PCIe + pack kernel)
    if (gpu_pack) return batch_stage_ascii_gpu(ctx, b, n, src.seqs, src.lens, st, ev0, ev1, err);
    constexpr uint64_t PIECE = 1ull << 16;        // words per host job (2 MiB of ASCII / 0.75 MiB packed)
    constexpr uint64_t WIN_WORDS = 40 * PIECE;    // 84 Mbp = 30 MiB of planes + validity per window
    // (a pipe keeps ONE window layout for all its batches: a window of the previous batch may still be on its way)
    // A batch of its own (a query batch: 100 Mbp) goes in about six windows, not in one and a bit: nothing computes before its
    // last byte is on the device, and the DMA of a window only starts when the window is full -- with 84 Mbp windows the call waited
    // for the copy of 21 MB (0.4 ms) after the fill of the first window instead of copying behind the fill
    // Packed input in PINNED host memory (hipHostMalloc / hipHostRegister / pgr_host_register): the DMA engine reads the caller's
    // planes where they lie -- no copy into the staging windows (the host copy ran at 45-49 GB/s beside a 56 GB/s link and made
    // the pre-packed route slower than the ASCII one, whose packer writes a quarter of what it reads).  All of it is queued at once
    // (windows of 8 MiB only so that a validity plane is looked at, and sent where it says something, beside the copies): nothing
    // the passes over the staged sub-batches need goes through a copy engine (pipeline.hip: the tile table and the rids are read
    // out of the pinned mailbox by a kernel), so a full queue holds nobody up.
    const bool src_pinned = packed && b->total_words && host_ptr_is_pinned(src.planes + src.word0) &&
                            (!src.valid || host_ptr_is_pinned(src.valid + src.word0)) && !ctx->opt.no_direct_h2d;
    const uint64_t win_words = src_pinned ? (src.valid ? (1ull << 20) : std::max<uint64_t>(b->total_words, 1))
                               : pipe     ? WIN_WORDS
                                          : std::min<uint64_t>(std::max<uint64_t>({(uint64_t)1, std::min<uint64_t>(b->total_words, 4 * PIECE),
                                                                                   (b->total_words + 5) / 6}),
                                                               WIN_WORDS);
    if (!src_pinned) {
        std::string why;  // (this may be the staging thread: the context's error string belongs to the calling thread)
        if (ctx->ensure_pinned(2 * win_words * 12, &why)) return fail(PGR_ERR_NOMEM, "staging buffers: " + why);
    }
    hipEvent_t done[2] = {ev0, ev1};
    StagePipe local;
    bool *used = pipe ? pipe->used : local.used;
    int &slot = pipe ? pipe->slot : local.slot;
    b->h_n_invalid.assign(std::max<uint32_t>(n, 1), 0);
    uint32_t c = 0;
    struct Job {
        uint32_t c0, c1;    // c1 == c0 + 1: words [wl0, wl1) of contig c0; otherwise the WHOLE contigs c0 .. c1 - 1, back to back
        uint64_t wl0, wl1;
        uint64_t out;       // first word inside the window
    };
    std::vector<Job> jobs;
    struct SanRange {
        uint64_t w0, w1;
        int has_valid;
    };
    std::vector<SanRange> san;
    for (uint64_t w0 = 0; w0 < b->total_words; w0 += win_words, slot ^= 1) {
        const uint64_t w1 = std::min(b->total_words, w0 + win_words);
        uint64_t *pin_planes = src_pinned ? nullptr : (uint64_t *)((uint8_t *)ctx->pinned + (size_t)slot * win_words * 12);
        uint32_t *pin_valid = src_pinned ? nullptr : (uint32_t *)(pin_planes + win_words);
        const auto tw0 = std::chrono::steady_clock::now();
        if (!src_pinned && used[slot] && hipEventSynchronize(done[slot]) != hipSuccess)  // this window's previous trip is over
            return fail(PGR_ERR_DEVICE, "H2D pipeline failed");
        if (src_pinned && hipMemcpyAsync(b->d.planes + w0, src.planes + src.word0 + w0, (w1 - w0) * sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess)
            return fail(PGR_ERR_DEVICE, "H2D copy of the caller's planes failed");
        const auto tw1 = std::chrono::steady_clock::now();
        while (c < n && b->h_word_off[c + 1] <= w0) ++c;
        jobs.clear();
        // short contigs (reads) are handed out in groups of ~PIECE / 4 words: one job per read was a million vector entries and a
        // million calls through the pool per Gbp -- the staging thread spent longer listing them than the pool packing them
        // (at least ~4 groups per thread and window: the small windows of a query batch would be two rounds and a bit otherwise)
        const uint64_t GROUP = std::max<uint64_t>(2048, std::min<uint64_t>(PIECE / 4, (w1 - w0) / (4ull * (HostPool::instance().workers() + 1))));
        // (pieces of long contigs likewise: 42 pieces of 2 MiB for 16 threads were three rounds, the last one a third full)
        const uint64_t PIECE_W = std::max<uint64_t>(8192, std::min<uint64_t>(PIECE, (((w1 - w0) / (4ull * (HostPool::instance().workers() + 1)) + 2047) / 2048) * 2048));
        for (uint32_t cc = c; cc < n && b->h_word_off[cc] < w1;) {
            const uint64_t cw0 = b->h_word_off[cc], cw1 = b->h_word_off[cc + 1];
            if (cw0 >= w0 && cw1 <= w1 && cw1 - cw0 < GROUP) {
                uint32_t ce = cc + 1;
                while (ce < n && b->h_word_off[ce + 1] <= w1 && b->h_word_off[ce + 1] - b->h_word_off[ce] < GROUP &&
                       b->h_word_off[ce] - cw0 < GROUP)
                    ++ce;
                jobs.push_back(Job{cc, ce, 0, b->h_word_off[ce] - cw0, cw0 - w0});
                cc = ce;
                continue;
            }
            const uint64_t lo = std::max(cw0, w0), hi = std::min(cw1, w1);
            for (uint64_t o = lo; o < hi; o += PIECE_W) jobs.push_back(Job{cc, cc + 1, o - cw0, std::min(hi, o + PIECE_W) - cw0, o - w0});
            ++cc;
        }
        std::atomic<uint64_t> win_bad{0};
        const auto tw2 = std::chrono::steady_clock::now();
        HostPool::instance().parallel_for(jobs.size(), [&](size_t i) {
            const Job &j = jobs[i];
            if (!packed) {
                uint64_t job_bad = 0;
                if (j.c1 == j.c0 + 1) {
                    job_bad = pack_words_stream(src.seqs[j.c0], src.lens[j.c0], j.wl0, j.wl1, pin_planes + j.out, pin_valid + j.out);
                    if (job_bad) __atomic_fetch_add(&b->h_n_invalid[j.c0], (uint32_t)job_bad, __ATOMIC_RELAXED);
                } else {
                    for (uint32_t cq = j.c0; cq < j.c1; ++cq) {
                        const uint64_t o = j.out + (b->h_word_off[cq] - b->h_word_off[j.c0]);
                        const uint64_t bad = pack_words_stream_nofence(src.seqs[cq], src.lens[cq], 0, b->h_word_off[cq + 1] - b->h_word_off[cq],
                                                                       pin_planes + o, pin_valid + o);
                        if (bad) b->h_n_invalid[cq] = (uint32_t)bad;  // (the whole contig is this job's)
                        job_bad += bad;
                    }
                    stream_fence();
                }
                if (job_bad) win_bad.fetch_add(job_bad, std::memory_order_relaxed);
            } else {
                const uint64_t g = src.word0 + b->h_word_off[j.c0] + j.wl0;  // word of the caller's arrays (contigs are back to back there too)
                if (!src_pinned) stream_copy(pin_planes + j.out, src.planes + g, (j.wl1 - j.wl0) * sizeof(uint64_t));
                // the validity plane is READ here (0.125 B per base) and travels only if it says something: a host that keeps one
                // for sequences without a single N pays nothing for it on the link
                if (src.valid && !win_bad.load(std::memory_order_relaxed)) {
                    bool all_set = true;
                    if (j.c1 == j.c0 + 1) all_set = valid_words_all_set(src.valid + g, j.wl0, j.wl1, src.lens[j.c0]);
                    else
                        for (uint32_t cq = j.c0; cq < j.c1 && all_set; ++cq)
                            all_set = valid_words_all_set(src.valid + src.word0 + b->h_word_off[cq], 0, b->h_word_off[cq + 1] - b->h_word_off[cq], src.lens[cq]);
                    if (!all_set) win_bad.store(1, std::memory_order_relaxed);
                }
            }
        });
        if (packed && src.valid && win_bad.load() && !src_pinned)  // (rare: this window's validity words do go)
            HostPool::instance().parallel_for(jobs.size(), [&](size_t i) {
                const Job &j = jobs[i];
                const uint64_t g = src.word0 + b->h_word_off[j.c0] + j.wl0;
                stream_copy(pin_valid + j.out, src.valid + g, (j.wl1 - j.wl0) * sizeof(uint32_t));
            });
        const auto tw3 = std::chrono::steady_clock::now();
        // the validity plane only travels when it says something: a window of ASCII in which the packer met no non-ACGT byte
        // (the usual case) and packed input without a validity plane put 0.25 B per base on the link, the plane is written on
        // the device from the contig lengths
        const bool has_valid = packed ? (src.valid != nullptr && win_bad.load() != 0) : win_bad.load() != 0;
        // (pinned caller memory: the planes' copy was enqueued in front of the jobs, straight from the caller's array)
        const void *valid_src = src_pinned ? (const void *)(src.valid + src.word0 + w0) : (const void *)pin_valid;
        if ((!src_pinned &&
             hipMemcpyAsync(b->d.planes + w0, pin_planes, (w1 - w0) * sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess) ||
            (has_valid && hipMemcpyAsync(b->d.valid + w0, valid_src, (w1 - w0) * sizeof(uint32_t), hipMemcpyHostToDevice, st) != hipSuccess))
            return fail(PGR_ERR_DEVICE, "H2D copy of the packed window failed");
        // The clean-up kernel of the window's words does NOT go between this window's copy and the next one's: a copy engine
        // and a compute queue hand over with a signal each way, ~10-20 us of nothing per window -- with 8 MiB windows the link
        // ran at 42 GB/s instead of 57.  The ranges are remembered and cleaned behind the batch's last copy.
        if (packed || !has_valid) {
            if (!san.empty() && san.back().w1 == w0 && san.back().has_valid == (has_valid ? 1 : 0)) san.back().w1 = w1;
            else san.push_back(SanRange{w0, w1, has_valid ? 1 : 0});
        }
        if (hipEventRecord(done[slot], st) != hipSuccess) return fail(PGR_ERR_DEVICE, "H2D pipeline failed");
        used[slot] = true;
        if (ctx->opt.debug > 1) {
            auto us = [](auto a, auto b2) { return std::chrono::duration<double, std::micro>(b2 - a).count(); };
            const auto tw4 = std::chrono::steady_clock::now();
            fprintf(stderr, "[pgr]     window of %.1f Mbp: waited %.0f us for its previous trip, %zu jobs listed in %.0f us, filled in %.0f us (%.1f GB/s of planes), enqueued in %.0f us\n",
                    (double)(w1 - w0) * 32e-6, us(tw0, tw1), jobs.size(), us(tw1, tw2), us(tw2, tw3), (double)(w1 - w0) * 8e-3 / us(tw2, tw3), us(tw3, tw4));
        }
    }
    for (const SanRange &r : san) launch_sanitize_packed(st, b->d, n, r.w0, r.w1, r.has_valid);
    // per-contig counts of non-ACGT bytes (host-packed input: counted by the packer; packed input: by the kernel above)
    if (!packed)
        for (uint32_t i = 0; i < n; ++i) b->host_saw_invalid = b->host_saw_invalid || b->h_n_invalid[i] != 0;
    if (!packed && n &&
        hipMemcpyAsync(b->d.n_invalid, b->h_n_invalid.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(PGR_ERR_DEVICE, "H2D copy of the invalid-byte counts failed");
    // On the context's own stream nothing waits here: the consumer is ordered behind the copies and synchronizes once at
    // its end; the staging thread of the pipelined path (own stream) hands over finished batches.
    if (pipe) return PGR_OK;
    if (st != ctx->stream || src_pinned) {  // (the DMA engine is still reading the CALLER's memory: not behind this call's return)
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess)
            return fail(PGR_ERR_DEVICE, "staging failed on the device");
    } else {
        ctx->staged_unsynced = true;
    }
    return PGR_OK;
}

static int batch_from_host(pgr_ctx *ctx, uint32_t n, const StageSrc &src, pgr_batch **out) {
    *out = nullptr;
    PGR_ENTER(ctx);
    pgr_batch *b = nullptr;
    const bool dbg = ctx->opt.debug != 0;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = batch_alloc(ctx, n, src.lens, &b);
    if (rc) return rc;
    std::string err;
    if ((rc = batch_stage(ctx, b, n, src, ctx->stream, ctx->ev[0], ctx->ev[1], err))) {
        pgr_batch_destroy(b);
        return ctx->fail(rc, err);
    }
    if (dbg)
        fprintf(stderr, "[pgr] batch from host (%s) %u seqs, %.1f Mbp: %.2f ms\n", src.planes ? "packed" : "ASCII", n,
                b->total_bases / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    *out = b;
    return PGR_OK;
}

extern "C" int pgr_batch_from_ascii(pgr_ctx *ctx, uint32_t n, const uint8_t *const *seqs, const uint64_t *lens,
                                    pgr_batch **out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!out || (n && (!seqs || !lens))) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    for (uint32_t i = 0; i < n; ++i)
        if (lens[i] && !seqs[i]) return ctx->fail(PGR_ERR_INVALID_ARG, "null sequence pointer");
    StageSrc src;
    src.seqs = seqs;
    src.lens = lens;
    return batch_from_host(ctx, n, src, out);
}

extern "C" int pgr_batch_from_packed(pgr_ctx *ctx, uint32_t n, const uint64_t *lens, const uint64_t *planes,
                                     const uint32_t *valid, pgr_batch **out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!out || (n && !lens)) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (pgr_packed_words(n, lens) && !planes) return ctx->fail(PGR_ERR_INVALID_ARG, "null plane array");
    static const uint64_t no_words = 0;
    StageSrc src;
    src.lens = lens;
    src.planes = planes ? planes : &no_words;
    src.valid = valid;
    return batch_from_host(ctx, n, src, out);
}

static int batch_synthetic(pgr_ctx *ctx, uint32_t n, const uint64_t *lens, uint64_t seed, uint64_t contig0,
                           const uint64_t *ids, pgr_batch **out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!out || (n && !lens)) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    PGR_ENTER(ctx);
    pgr_batch *b = nullptr;
    int rc = batch_alloc(ctx, n, lens, &b);
    if (rc) return rc;
    Tmp_list d_ids(ctx);
    if (ids && n) {
        if ((rc = d_ids.alloc((size_t)n * sizeof(uint64_t)))) {
            pgr_batch_destroy(b);
            return rc;
        }
        if (hipMemcpyAsync(d_ids.p, ids, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            pgr_batch_destroy(b);
            return ctx->fail(PGR_ERR_DEVICE, "H2D of the contig ids failed");
        }
    }
    launch_synth(ctx->stream, b->d, n, b->total_words, seed, contig0, ids && n ? (const uint64_t *)d_ids.p : nullptr);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
        pgr_batch_destroy(b);
        return ctx->fail(PGR_ERR_DEVICE, "synthetic generator kernel failed");
    }
    *out = b;
    return PGR_OK;
}

extern "C" int pgr_batch_synthetic(pgr_ctx *ctx, uint32_t n, const uint64_t *lens, uint64_t seed, uint64_t contig0,
                                   pgr_batch **out) {
    return batch_synthetic(ctx, n, lens, seed, contig0, nullptr, out);
}

extern "C" int pgr_batch_synthetic_ids(pgr_ctx *ctx, uint32_t n, const uint64_t *lens, uint64_t seed,
                                       const uint64_t *contig_ids, pgr_batch **out) {
    if (ctx && n && !contig_ids) return ctx->fail(PGR_ERR_INVALID_ARG, "null contig id list");
    return batch_synthetic(ctx, n, lens, seed, 0, contig_ids, out);
}


int pgr::check_spec(pgr_ctx *ctx, const pgr_spec *spec) {
    if (!spec) return ctx->fail(PGR_ERR_INVALID_ARG, "null spec");
    // shmmrutils.rs:443-445 / :575-576
    if (spec->k == 0 || spec->k > 56) return ctx->fail(PGR_ERR_BAD_SPEC, "spec.k must be in 1..56");
    if (spec->r == 0 || spec->r > 12) return ctx->fail(PGR_ERR_BAD_SPEC, "spec.r must be in 1..12");
    if (!spec->sketch && (spec->w == 0 || spec->w > 128)) return ctx->fail(PGR_ERR_BAD_SPEC, "spec.w must be in 1..128");
    return PGR_OK;
}


extern "C" uint64_t pgr_shmmrs_count(const pgr_shmmrs *s) { return s ? s->count : 0; }
extern "C" const pgr_mm128 *pgr_shmmrs_device_ptr(const pgr_shmmrs *s) { return s ? s->d_mm : nullptr; }
extern "C" const uint64_t *pgr_shmmrs_device_offsets(const pgr_shmmrs *s) { return s ? s->d_off : nullptr; }

extern "C" void pgr_shmmrs_destroy(pgr_shmmrs *s) {
    if (!s) return;
    s->ctx->dfree(s->d_mm);
    s->ctx->dfree(s->d_block);
    if (s->h_off.capacity() >= (1u << 17) && s->h_off.capacity() > s->ctx->spare_off.capacity()) s->h_off.swap(s->ctx->spare_off);
    delete s;
}

extern "C" int pgr_shmmrs_download(pgr_ctx *ctx, const pgr_shmmrs *s, pgr_mm128 **out_mm, uint64_t **out_off) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!s || !out_mm || !out_off) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    PGR_ENTER(ctx);
    *out_mm = nullptr;
    *out_off = nullptr;
    // a list of >= 1 MiB goes into a pinned block of the pool: ONE DMA, no staging windows, no host copy (pgr_free returns
    // the block to the pool); if the host cannot pin more memory, a pageable block filled through the staging windows
    const size_t bytes = (size_t)s->count * sizeof(pgr_mm128);
    pgr_mm128 *mm = nullptr;
    bool direct = false;
    if (bytes >= (1u << 20) && !s->host_copy) {
        size_t cap = 0;
        mm = (pgr_mm128 *)pinned_result_acquire(bytes, &cap);
        direct = mm != nullptr;
    }
    if (!mm) mm = (pgr_mm128 *)host_result_alloc(std::max<uint64_t>(s->count, 1) * sizeof(pgr_mm128));
    uint64_t *off = (uint64_t *)malloc(((size_t)s->n + 1) * sizeof(uint64_t));
    if (!mm || !off) {
        result_block_release(mm);
        free(off);
        return ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
    }
    memcpy(off, s->h_off.data(), ((size_t)s->n + 1) * sizeof(uint64_t));
    if (s->count && s->host_copy) {
        memcpy(mm, s->host_copy, s->count * sizeof(pgr_mm128));
    } else if (s->count) {
        int rc = PGR_OK;
        if (direct) {
            if (hipMemcpyAsync(mm, s->d_mm, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                rc = ctx->fail(PGR_ERR_DEVICE, "result download failed");
        } else {
            rc = ctx->d2h(mm, s->d_mm, bytes);
        }
        if (rc) {
            result_block_release(mm);
            free(off);
            return rc;
        }
    }
    *out_mm = mm;
    *out_off = off;
    return PGR_OK;
}

extern "C" int pgr_shmmrs_checksum(pgr_ctx *ctx, const pgr_shmmrs *s, uint64_t *out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!s || !out) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    PGR_ENTER(ctx);
    const uint32_t n = s->n;
    if (n == 0) return PGR_OK;
    uint64_t max_cnt = 0;
    for (uint32_t c = 0; c < n; ++c) max_cnt = std::max(max_cnt, s->h_off[c + 1] - s->h_off[c]);
    Tmp_list sums(ctx);
    int rc = sums.alloc((size_t)n * 2 * sizeof(uint64_t));
    if (rc) return rc;
    PGR_HIP(ctx, hipMemsetAsync(sums.p, 0, (size_t)n * 2 * sizeof(uint64_t), ctx->stream));
    launch_shmmr_checksum(ctx->stream, s->d_mm, s->d_off, n, max_cnt, (uint64_t *)sums.p);
    PGR_HIP(ctx, hipMemcpyAsync(out, sums.p, (size_t)n * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PGR_HIP(ctx, hipGetLastError());
    return PGR_OK;
}

extern "C" int pgr_shmmrs_copy_to_device(pgr_ctx *ctx, const pgr_shmmrs *s, pgr_mm128 *d_out, uint64_t capacity,
                                         uint32_t rid_add) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!s || (s->count && !d_out)) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (capacity < s->count) return ctx->fail(PGR_ERR_INVALID_ARG, "output buffer too small for the shimmer list");
    PGR_ENTER(ctx);
    if (s->count) {
        launch_copy_add_rid(ctx->stream, s->d_mm, s->count, rid_add, d_out);
        PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PGR_HIP(ctx, hipGetLastError());
    }
    return PGR_OK;
}

extern "C" int pgr_shmmrs_copy_to_device_rids(pgr_ctx *ctx, const pgr_shmmrs *s, pgr_mm128 *d_out, uint64_t capacity,
                                              const uint32_t *rids) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!s || (s->count && !d_out) || (s->n && !rids)) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (capacity < s->count) return ctx->fail(PGR_ERR_INVALID_ARG, "output buffer too small for the shimmer list");
    if (!s->rid_is_index) return ctx->fail(PGR_ERR_STATE, "the result already carries caller rids (computed with rids / padding)");
    PGR_ENTER(ctx);
    if (s->count) {
        int rc;
        if ((rc = ctx->ws_rids.ensure(ctx, (size_t)s->n * sizeof(uint32_t)))) return rc;
        PGR_HIP(ctx, hipMemcpyAsync(ctx->ws_rids.p, rids, (size_t)s->n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        launch_copy_map_rid(ctx->stream, s->d_mm, s->count, (const uint32_t *)ctx->ws_rids.p, d_out);
        PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PGR_HIP(ctx, hipGetLastError());
    }
    return PGR_OK;
}

extern "C" int pgr_shmmrs_offsets(const pgr_shmmrs *s, uint64_t *out) {
    if (!s || !out) return PGR_ERR_INVALID_ARG;
    memcpy(out, s->h_off.data(), ((size_t)s->n + 1) * sizeof(uint64_t));
    return PGR_OK;
}

extern "C" uint64_t pgr_shmmrs_n_pairs(const pgr_shmmrs *s) {
    if (!s) return 0;
    uint64_t np = 0;
    for (uint32_t c = 0; c < s->n; ++c) {
        const uint64_t cnt = s->h_off[c + 1] - s->h_off[c];
        if (cnt > 1) np += cnt - 1;
    }
    return np;
}

int pgr::shmmrs_to_frag_recs_enqueue(pgr_ctx *ctx, const pgr_shmmrs *s, const uint32_t *sids, int query_side,
                                     pgr_frag_rec *d_out, uint64_t capacity) {
    const uint32_t n = s->n;
    // the source of an asynchronous H2D copy lives in the context, not on this stack frame (large pageable sources are
    // pinned and read by the DMA engine after hipMemcpyAsync has returned).  It is rewritten by the next call only: `s` is
    // the product of a pgr_shmmrs_compute, which synchronized the stream behind any earlier copy from this vector, and the
    // callers that enqueue twice on one result synchronize in between
    std::vector<uint64_t> &rec_off = ctx->keep_rec_off;
    rec_off.resize((size_t)n + 1);
    uint64_t np = 0;
    for (uint32_t c = 0; c < n; ++c) {
        rec_off[c] = np;
        const uint64_t cnt = s->h_off[c + 1] - s->h_off[c];
        if (cnt > 1) np += cnt - 1;
    }
    rec_off[n] = np;
    if (np == 0) return PGR_OK;
    if (!d_out || capacity < np) return ctx->fail(PGR_ERR_INVALID_ARG, "output buffer too small for the pair records");
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = ctx->ws_rec_off.ensure(ctx, ((size_t)n + 1) * sizeof(uint64_t)))) return rc;
    PGR_HIP(ctx, hipMemcpyAsync(ctx->ws_rec_off.p, rec_off.data(), ((size_t)n + 1) * sizeof(uint64_t),
                                hipMemcpyHostToDevice, st));
    uint32_t *d_sids = nullptr;
    if (sids) {
        if ((rc = ctx->ws_rids.ensure(ctx, (size_t)n * sizeof(uint32_t)))) return rc;
        PGR_HIP(ctx, hipMemcpyAsync(ctx->ws_rids.p, sids, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        d_sids = (uint32_t *)ctx->ws_rids.p;
    }
    launch_frag_recs(st, s->d_mm, s->d_off, (const uint64_t *)ctx->ws_rec_off.p, n, s->count, d_sids, query_side,
                     s->rid_is_index ? 1 : 0, d_out);
    return PGR_OK;
}

extern "C" int pgr_shmmrs_to_frag_recs_device(pgr_ctx *ctx, const pgr_shmmrs *s, const uint32_t *sids, int query_side,
                                              pgr_frag_rec *d_out, uint64_t capacity, uint64_t *n_out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!s || !n_out) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    PGR_ENTER(ctx);
    *n_out = pgr_shmmrs_n_pairs(s);
    if (*n_out == 0) return PGR_OK;
    const int rc = shmmrs_to_frag_recs_enqueue(ctx, s, sids, query_side, d_out, capacity);
    if (rc) return rc;
    PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PGR_HIP(ctx, hipGetLastError());
    return PGR_OK;
}

// ------------------------------------------------------------------------------------------------
// host-buffer conveniences (the B1 drop-in)
// Large host inputs are cut into sub-batches of 256 Mbp - 1 Gbp: a staging thread pushes sub-batch i+1 through the
// pinned windows and the pack kernel (copy stream) while sub-batch i is consumed (shimmers, records, download) on the
// context's stream.  The PCIe transfer of the ASCII input is the longest stage (48 GB/s); the pipeline hides the rest
// behind it.  consume(batch, c0, c1) gets contigs [c0, c1) of the call, resident on the GPU.
bool pgr::worth_pipelining(const pgr_ctx *ctx, uint32_t n, const uint64_t *lens) {
    if (n < 2 || ctx->opt.no_pipeline) return false;
    uint64_t total_bp = 0;
    for (uint32_t i = 0; i < n; ++i) total_bp += lens[i];
    return total_bp >= (512ull << 20);
}

int pgr::for_each_staged(pgr_ctx *ctx, uint32_t n, const StageSrc &src,
                         const std::function<int(pgr_batch *, uint32_t, uint32_t)> &consume) {
    const auto t_start = std::chrono::steady_clock::now();
    PGR_ENTER(ctx);
    const uint64_t *lens = src.lens;
    for (uint32_t i = 0; i < n && !src.planes; ++i)
        if (lens[i] && !src.seqs[i]) return ctx->fail(PGR_ERR_INVALID_ARG, "null sequence pointer");
    // Sub-batch sizes.  Nothing computes before the first sub-batch is staged and nothing overlaps the processing of the last
    // one, so the call ramps up and down: 64 Mbp first, doubling to the steady size (>= 256 Mbp keeps the per-sub-batch costs of
    // the consumer -- launches, one synchronization, the result's way back -- behind the staging), and halving towards the end
    // (never more than half of what is left, down to 48 Mbp).  Round 3 cut equal pieces of 256 Mbp: 2.3 ms went by before the
    // first kernel ran and 1.7 ms after the last byte was staged (rocprofv3 timeline of a 1.04 Gbp call).
    uint64_t total_bp = 0;
    for (uint32_t i = 0; i < n; ++i) total_bp += lens[i];
    const uint64_t SUB_BP = std::min<uint64_t>(std::max<uint64_t>(total_bp / 8, 256ull << 20), 1ull << 30);
    struct Sub {
        uint32_t c0, c1;
        pgr_batch *b = nullptr;
        uint64_t word0 = 0;  // first word of contig c0 in the caller's packed arrays
    };
    std::vector<Sub> subs;
    {
        uint64_t target = 64ull << 20, left = total_bp;
        for (uint32_t c = 0; c < n;) {
            // (measured in round 5: a coarser ramp-down -- 128 Mbp floor, the remainder joined to its predecessor -- changes nothing:
            // one long last pass instead of four short ones, 5.9-6.1 ms either way)
            const uint64_t want = std::max<uint64_t>(48ull << 20, std::min<uint64_t>(std::min(target, SUB_BP), left / 2));
            uint32_t e = c;
            uint64_t tot = 0;
            while (e < n && (e == c || tot + lens[e] <= want)) tot += lens[e++];
            Sub sb;
            sb.c0 = c;
            sb.c1 = e;
            subs.push_back(sb);
            c = e;
            left -= tot;
            target *= 2;
        }
    }
    {
        uint64_t w = 0;
        uint32_t c = 0;
        for (Sub &sb : subs) {
            for (; c < sb.c0; ++c) w += (lens[c] + 31) / 32;
            sb.word0 = w;
        }
    }
    auto destroy_all = [&]() {
        for (Sub &sb : subs) {
            pgr_batch_destroy(sb.b);
            sb.b = nullptr;
        }
    };
    if (subs.empty()) return PGR_OK;
    int rc = PGR_OK;
    // Device allocations stay on the calling thread (the caching allocator is not thread safe) -- but only the first sub-batch
    // is allocated before the staging thread starts: the tables of a million reads (lengths, word offsets: 12 MB through
    // pageable copies) took 2 ms during which nothing was staged.  The staging thread waits for `n_alloc`.
    if ((rc = batch_alloc(ctx, subs[0].c1 - subs[0].c0, lens + subs[0].c0, &subs[0].b))) return rc;
    const bool dbg = ctx->opt.debug != 0;
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    if (dbg) fprintf(stderr, "[pgr] pipelined call: %zu sub-batches, the first allocated at %.2f ms\n", subs.size(), since());
    std::mutex mu;
    std::condition_variable cv;
    size_t n_alloc = 1;  // sub-batches with their device arrays (guarded by mu)
    size_t n_ready = 0;
    int stage_rc = PGR_OK;
    std::string stage_err;
    std::atomic<bool> cancel{false};
    // one event per sub-batch, recorded behind its last copy on the copy stream: the consumer's stream waits for it on the
    // device, the staging thread never blocks on a finished sub-batch and keeps its two windows rolling into the next one
    const bool legacy = ctx->opt.gpu_pack && !src.planes;
    std::vector<hipEvent_t> ready(subs.size(), nullptr);
    for (auto &e : ready)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            for (auto &q : ready)
                if (q) (void)hipEventDestroy(q);
            destroy_all();
            return ctx->fail(PGR_ERR_DEVICE, "event creation failed");
        }
    StagePipe pipe;
    // Pinned packed input goes to the copy engine sub-batch by sub-batch, nothing holding it back (batch_stage) -- but the engine
    // takes its work in the order it was queued, downloads of results included: sub-batch j is queued only when sub-batch j - 2
    // has been consumed (its result's download is in the queue by then), so that a download waits for ONE staging copy, not
    // for all that are left (measured: the second sub-batch's consumer sat 4.5 ms behind the whole call's copies).
    const bool throttle = src.planes && host_ptr_is_pinned(src.planes + src.word0) && !ctx->opt.no_direct_h2d;
    size_t n_consumed = 0;  // (guarded by mu)
    std::thread stager([&]() {
        (void)hipSetDevice(ctx->device);
        for (size_t i = 0; i < subs.size() && !cancel.load(); ++i) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return (n_alloc > i && (!throttle || n_consumed + 2 > i)) || cancel.load(); });
                if (n_alloc <= i || cancel.load()) return;
            }
            std::string err;
            StageSrc ss = src;
            if (ss.seqs) ss.seqs += subs[i].c0;
            ss.lens += subs[i].c0;
            ss.word0 = src.word0 + subs[i].word0;
            int r = batch_stage(ctx, subs[i].b, subs[i].c1 - subs[i].c0, ss, ctx->copy_stream, ctx->cev[0], ctx->cev[1], err,
                                legacy ? nullptr : &pipe);
            if (!r && hipEventRecord(ready[i], ctx->copy_stream) != hipSuccess) {
                r = PGR_ERR_DEVICE;
                err = "event record failed";
            }
            std::lock_guard<std::mutex> lk(mu);
            if (r) {
                stage_rc = r;
                stage_err = err;
                cv.notify_all();
                return;
            }
            n_ready = i + 1;
            cv.notify_all();
            if (dbg) fprintf(stderr, "[pgr]   sub-batch %zu staged at %.2f ms\n", i, since());
        }
    });
    for (size_t i = 1; i < subs.size() && !rc; ++i) {  // the other sub-batches, while the first is being staged
        rc = batch_alloc(ctx, subs[i].c1 - subs[i].c0, lens + subs[i].c0, &subs[i].b);
        std::lock_guard<std::mutex> lk(mu);
        if (rc) cancel.store(true);
        else n_alloc = i + 1;
        cv.notify_all();
    }
    if (dbg) fprintf(stderr, "[pgr]   all sub-batches allocated at %.2f ms\n", since());
    for (size_t i = 0; i < subs.size() && !rc; ++i) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return n_ready > i || stage_rc != PGR_OK; });
            if (n_ready <= i) {
                rc = ctx->fail(stage_rc, stage_err);
                break;
            }
        }
        const double tc0 = since();
        if (hipStreamWaitEvent(ctx->stream, ready[i], 0) != hipSuccess) {
            rc = ctx->fail(PGR_ERR_DEVICE, "stream wait failed");
            break;
        }
        rc = consume(subs[i].b, subs[i].c0, subs[i].c1);
        {
            std::lock_guard<std::mutex> lk(mu);
            n_consumed = i + 1;
            cv.notify_all();
        }
        if (dbg) fprintf(stderr, "[pgr]   sub-batch %zu consumed %.2f -> %.2f ms\n", i, tc0, since());
        if (rc) {
            // a consumer that failed before its own synchronization: this sub-batch's copies (the batch owns the pageable source
            // of one of them) and kernels may still be in flight
            (void)hipEventSynchronize(ready[i]);
            (void)hipStreamSynchronize(ctx->stream);
        }
        pgr_batch_destroy(subs[i].b);
        subs[i].b = nullptr;
    }
    if (rc) cancel.store(true);
    stager.join();
    if (rc) (void)hipStreamSynchronize(ctx->copy_stream);  // nothing of a cancelled call may still read the pinned windows
    for (auto &e : ready) (void)hipEventDestroy(e);
    destroy_all();
    return rc;
}

static int shmmr_batch_pipelined(pgr_ctx *ctx, const pgr_spec *spec, uint32_t n, const StageSrc &src,
                                 const uint32_t *rids, int padding, pgr_mm128 **out_mm, uint64_t **out_off) {
    const uint64_t *lens = src.lens;
    std::vector<uint32_t> rr;  // rid of contig i of the CALL (sub-batches must not restart at 0)
    if (!rids) {
        rr.resize(n);
        for (uint32_t i = 0; i < n; ++i) rr[i] = i;
        rids = rr.data();
    }
    pgr_mm128 *mm = nullptr;
    uint64_t cap = 0, total = 0;
    uint64_t *off = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    if (!off) return ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
    bool first = true;
    uint64_t total_bp = 0;
    for (uint32_t i = 0; i < n; ++i) total_bp += lens[i];
    const bool dbg = ctx->opt.debug != 0;
    // The download of sub-batch i runs on its own stream while sub-batch i + 1 computes.  The result block is PINNED memory
    // of the process-wide pool (pgr_free puts it back): the DMA engine writes every sub-batch's list where it belongs, no
    // staging block and no host copy -- round 3 went through two pinned blocks and copied 50 MB per Gbp with the pool's
    // threads (which the ASCII packer needs), into a fresh malloc whose page faults and munmap cost another ~2 ms per call.
    // `direct` == false (the host cannot pin more memory): the round-3 route.
    bool direct = true;
    struct Pending {
        pgr_shmmrs *s = nullptr;
        uint64_t dst = 0;  // first element in mm
        int slot = 0;
    } pend;
    int slot = 0;
    auto finish_pending = [&]() -> int {
        if (!pend.s) return PGR_OK;
        pgr_shmmrs *ps = pend.s;
        pend.s = nullptr;
        int r = PGR_OK;
        if (hipEventSynchronize(ctx->d2h_ev[pend.slot]) != hipSuccess) r = ctx->fail(PGR_ERR_DEVICE, "result download failed");
        if (!r && ps->count && !direct) {
            const uint8_t *srcp = (const uint8_t *)ctx->pinned_out + (size_t)pend.slot * ctx->d2h_slot_bytes;
            uint8_t *dstp = (uint8_t *)(mm + pend.dst);
            const size_t len = ps->count * sizeof(pgr_mm128);
            constexpr size_t PIECE = 1u << 20;
            HostPool::instance().parallel_for((len + PIECE - 1) / PIECE, [&](size_t i) {
                const size_t o = i * PIECE;
                memcpy(dstp + o, srcp + o, std::min(PIECE, len - o));
            });
        }
        pgr_shmmrs_destroy(ps);
        return r;
    };
    const auto t_call = std::chrono::steady_clock::now();
    int rc = for_each_staged(ctx, n, src, [&](pgr_batch *b, uint32_t c0, uint32_t c1) -> int {
        pgr_shmmrs *s = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        int r = pgr_shmmrs_compute(ctx, b, spec, rids + c0, padding, &s);
        if (r) return r;
        if (dbg)
            fprintf(stderr, "[pgr]     compute %.2f ms (%.1f Mbp, %llu shimmers)\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), b->total_bases / 1e6,
                    (unsigned long long)s->count);
        if (total + s->count > cap || !direct) {
            if ((r = finish_pending())) {  // (before mm may move / the staging block is reused)
                pgr_shmmrs_destroy(s);
                return r;
            }
        }
        if (total + s->count > cap) {
            // the first sub-batch predicts the rest (shimmer density is a property of the spec)
            uint64_t guess = cap + cap / 2;
            if (first && b->total_bases)
                guess = (uint64_t)((double)s->count * ((double)total_bp / (double)b->total_bases) * 1.1) + 4096;
            cap = std::max<uint64_t>(total + s->count, guess);
            pgr_mm128 *nm = nullptr;
            if (direct) {
                size_t got = 0;
                nm = (pgr_mm128 *)pinned_result_acquire(cap * sizeof(pgr_mm128), &got);
                if (nm) cap = got / sizeof(pgr_mm128);
                else if (!mm) direct = false;  // nothing pinned yet: the pageable route for the whole call
                else {
                    pgr_shmmrs_destroy(s);
                    return ctx->fail(PGR_ERR_NOMEM, "cannot pin a larger result block");
                }
            }
            // (pageable: grown by allocate + copy: the block is huge-page advised, realloc would hand back plain pages)
            if (!nm) nm = (pgr_mm128 *)host_result_alloc(cap * sizeof(pgr_mm128));
            if (!nm) {
                pgr_shmmrs_destroy(s);
                return ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
            }
            if (total) memcpy(nm, mm, total * sizeof(pgr_mm128));
            result_block_release(mm);
            mm = nm;
        }
        first = false;
        for (uint32_t c = c0; c < c1; ++c) off[c] = total + s->h_off[c - c0];
        const size_t bytes = s->count * sizeof(pgr_mm128);
        uint8_t *dst = (uint8_t *)(mm + total);
        if (!direct) {
            // two pinned blocks of the largest sub-batch result seen so far (growing them waits for nothing: none is pending here)
            if (bytes > ctx->d2h_slot_bytes || !ctx->pinned_out) {
                const size_t want = std::max<size_t>(bytes + bytes / 4, 1u << 20);
                if ((r = ctx->ensure_pinned_out(2 * want))) {
                    pgr_shmmrs_destroy(s);
                    return r;
                }
                ctx->d2h_slot_bytes = ctx->pinned_out_cap / 2;
            }
            dst = (uint8_t *)ctx->pinned_out + (size_t)slot * ctx->d2h_slot_bytes;
        } else if ((r = finish_pending())) {  // the download before this one: its device list can go now (it is over long ago)
            pgr_shmmrs_destroy(s);
            return r;
        }
        if (bytes && hipMemcpyAsync(dst, s->d_mm, bytes, hipMemcpyDeviceToHost, ctx->d2h_stream) != hipSuccess) {
            pgr_shmmrs_destroy(s);
            return ctx->fail(PGR_ERR_DEVICE, "result download failed");
        }
        if (hipEventRecord(ctx->d2h_ev[slot], ctx->d2h_stream) != hipSuccess) {
            (void)hipStreamSynchronize(ctx->d2h_stream);
            pgr_shmmrs_destroy(s);
            return ctx->fail(PGR_ERR_DEVICE, "event record failed");
        }
        pend.s = s;
        pend.dst = total;
        pend.slot = slot;
        slot ^= 1;
        total += s->count;
        return PGR_OK;
    });
    if (dbg) fprintf(stderr, "[pgr] pipelined call: sub-batches allocated, staged and consumed in %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
    if (!rc) rc = finish_pending();
    if (pend.s) {  // an error left a download in flight
        (void)hipStreamSynchronize(ctx->d2h_stream);
        pgr_shmmrs_destroy(pend.s);
        pend.s = nullptr;
    }
    if (rc) {
        result_block_release(mm);
        free(off);
        return rc;
    }
    off[n] = total;
    if (!mm) mm = (pgr_mm128 *)malloc(sizeof(pgr_mm128));
    *out_mm = mm;
    *out_off = off;
    return mm ? PGR_OK : ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
}

// Small calls (the reference's real callers: <= 129 contigs per batch, seq_db.rs:549-564; one query, ext.rs:252-282): ONE kernel
// launch and ONE synchronization.  The host packs the bases into a pinned block, small_shmmr_kernel (csrc/small.hip, one
// workgroup per contig) reads it over PCIe and writes the final MM128 lists into a second pinned block; the host copies them
// out.  handled == false: not a small call, or the kernel handed a contig back (palindromic k-mer, low-complexity list) --
// the caller takes the general path.
static int shmmr_batch_small(pgr_ctx *ctx, const pgr_spec *spec, uint32_t n, const StageSrc &src, const uint32_t *rids,
                             pgr_mm128 **out_mm, uint64_t **out_off, bool &handled) {
    handled = false;
    if (spec->sketch || spec->w < (uint32_t)L1_MIN_W || n == 0 || n > SMALL_MAX_CONTIGS || ctx->opt.no_small_path) return PGR_OK;
    uint64_t total_bp = 0, total_words = 0, total_slots = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (src.lens[i] > SMALL_MAX_LEN) return PGR_OK;
        total_bp += src.lens[i];
        total_words += (src.lens[i] + 31) / 32;
        total_slots += src.lens[i] / 32 + 64;
    }
    if (total_bp > SMALL_MAX_BASES) return PGR_OK;
    PGR_ENTER(ctx);
    if (ctx->staged_unsynced) {  // the pinned windows may still be the source of an earlier staging
        PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->staged_unsynced = false;
    }
    const size_t desc_bytes = ((size_t)n * sizeof(SmallContig) + 15) & ~(size_t)15;
    const size_t in_bytes = desc_bytes + (size_t)std::max<uint64_t>(total_words, 1) * 12;  // planes, then the packer's validity words
    const size_t cnt_bytes = (((size_t)n + 1) * sizeof(uint32_t) + 15) & ~(size_t)15;
    const size_t out_bytes = cnt_bytes + (size_t)total_slots * sizeof(pgr_mm128);
    int rc;
    if ((rc = ctx->ensure_pinned(in_bytes)) || (rc = ctx->ensure_pinned_out(out_bytes))) return rc;
    SmallContig *desc = (SmallContig *)ctx->pinned;
    uint64_t *planes = (uint64_t *)((uint8_t *)ctx->pinned + desc_bytes);
    uint32_t *valid = (uint32_t *)(planes + std::max<uint64_t>(total_words, 1));
    uint32_t *counts = (uint32_t *)ctx->pinned_out;
    pgr_mm128 *slots = (pgr_mm128 *)((uint8_t *)ctx->pinned_out + cnt_bytes);
    uint64_t w_off = 0, s_off = 0;
    for (uint32_t i = 0; i < n; ++i) {
        desc[i].word_off = w_off;
        desc[i].len = (uint32_t)src.lens[i];
        desc[i].rid = rids ? rids[i] : i;
        desc[i].out_off = (uint32_t)s_off;
        desc[i].out_cap = (uint32_t)(src.lens[i] / 32 + 64);
        w_off += (src.lens[i] + 31) / 32;
        s_off += desc[i].out_cap;
    }
    // ---- the bases: packed on the host (every base must be valid: anything else belongs to the exact-island path)
    uint64_t bad = 0;
    if (!src.planes) {
        if (total_bp < (1u << 20)) {
            for (uint32_t i = 0; i < n && !bad; ++i)
                bad += pack_words(src.seqs[i], src.lens[i], 0, (src.lens[i] + 31) / 32, planes + desc[i].word_off, valid + desc[i].word_off);
        } else {
            std::atomic<uint64_t> nb{0};
            HostPool::instance().parallel_for(n, [&](size_t i) {
                const uint64_t b2 = pack_words(src.seqs[i], src.lens[i], 0, (src.lens[i] + 31) / 32, planes + desc[i].word_off,
                                               valid + desc[i].word_off);
                if (b2) nb.fetch_add(b2);
            });
            bad = nb.load();
        }
    } else {
        memcpy(planes, src.planes + src.word0, (size_t)total_words * sizeof(uint64_t));
        if (src.valid) {  // all bases valid?  (bits past a contig's end do not count)
            for (uint32_t i = 0; i < n && !bad; ++i) {
                const uint64_t nw = (src.lens[i] + 31) / 32;
                const uint32_t *v = src.valid + src.word0 + desc[i].word_off;
                for (uint64_t j = 0; j + 1 < nw && !bad; ++j) bad += v[j] != 0xFFFFFFFFu;
                if (nw) {
                    const uint32_t nbits = (uint32_t)(src.lens[i] - (nw - 1) * 32);
                    const uint32_t tail = nbits == 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> nbits);
                    bad += (v[nw - 1] & tail) != tail;
                }
            }
        }
    }
    if (bad) return PGR_OK;
    counts[n] = 0;  // the fallback flag word
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < n; ++i) max_len = std::max<uint32_t>(max_len, (uint32_t)src.lens[i]);
    SmallArgs a;
    a.planes = (const uint2 *)planes;
    a.valid = nullptr;  // checked above
    a.l1_cap = small_l1_cap(max_len, spec->w);
    a.desc = desc;
    a.n = n;
    a.w = spec->w;
    a.k = spec->k;
    a.r = spec->r;
    a.min_span = spec->min_span;
    a.tc = ((L1_EXT - 2 * (spec->w - 1)) / 64) * 64;
    a.out = slots;
    a.counts = counts;
    a.flags = counts + n;
    launch_small_shmmr(ctx->stream, a);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
        return ctx->fail(PGR_ERR_DEVICE, "small-batch kernel failed on the device");
    if (counts[n]) {  // a contig needs the exact state machine / a bigger list: general path, and not this kernel again
        ctx->skip_small_once = true;
        return PGR_OK;
    }
    uint64_t *off = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    if (!off) return ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n; ++i) {
        off[i] = tot;
        tot += counts[i];
    }
    off[n] = tot;
    pgr_mm128 *mm = (pgr_mm128 *)host_result_alloc(std::max<uint64_t>(tot, 1) * sizeof(pgr_mm128));
    if (!mm) {
        free(off);
        return ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
    }
    for (uint32_t i = 0; i < n; ++i)
        if (counts[i]) memcpy(mm + off[i], slots + desc[i].out_off, (size_t)counts[i] * sizeof(pgr_mm128));
    *out_mm = mm;
    *out_off = off;
    handled = true;
    return PGR_OK;
}

static int shmmr_batch_host(pgr_ctx *ctx, const pgr_spec *spec, uint32_t n, const StageSrc &src, const uint32_t *rids,
                            int padding, pgr_mm128 **out_mm, uint64_t **out_off) {
    int rc = check_spec(ctx, spec);
    if (rc) return rc;
    *out_mm = nullptr;
    *out_off = nullptr;
    if (!padding || spec->r <= 1) {  // (the padding artefact of reduce_shmmr only exists for r > 1)
        bool handled = false;
        if ((rc = shmmr_batch_small(ctx, spec, n, src, rids, out_mm, out_off, handled)) || handled) return rc;
    }
    // (skip_small_once, set by shmmr_batch_small when its kernel handed the batch back, belongs to THIS call: whatever way the
    // general path leaves, the next call on the context starts without it)
    struct ClearSkip {
        pgr_ctx *c;
        ~ClearSkip() { c->skip_small_once = false; }
    } clear_skip{ctx};
    if (worth_pipelining(ctx, n, src.lens)) return shmmr_batch_pipelined(ctx, spec, n, src, rids, padding, out_mm, out_off);
    pgr_batch *b = nullptr;
    if ((rc = batch_from_host(ctx, n, src, &b))) return rc;
    pgr_shmmrs *s = nullptr;
    ctx->want_host_copy = true;
    rc = pgr_shmmrs_compute(ctx, b, spec, rids, padding, &s);
    ctx->want_host_copy = false;
    pgr_batch_destroy(b);
    if (rc) return rc;
    rc = pgr_shmmrs_download(ctx, s, out_mm, out_off);
    pgr_shmmrs_destroy(s);
    return rc;
}

extern "C" int pgr_shmmr_batch(pgr_ctx *ctx, const pgr_spec *spec, uint32_t n, const uint8_t *const *seqs,
                               const uint64_t *lens, const uint32_t *rids, int padding, pgr_mm128 **out_mm,
                               uint64_t **out_off) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!out_mm || !out_off) return ctx->fail(PGR_ERR_INVALID_ARG, "null output pointer");
    if (n && (!seqs || !lens)) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    for (uint32_t i = 0; i < n; ++i)
        if (lens[i] && !seqs[i]) return ctx->fail(PGR_ERR_INVALID_ARG, "null sequence pointer");
    StageSrc src;
    src.seqs = seqs;
    src.lens = lens;
    return shmmr_batch_host(ctx, spec, n, src, rids, padding, out_mm, out_off);
}

extern "C" int pgr_shmmr_batch_packed(pgr_ctx *ctx, const pgr_spec *spec, uint32_t n, const uint64_t *lens,
                                      const uint64_t *planes, const uint32_t *valid, const uint32_t *rids, int padding,
                                      pgr_mm128 **out_mm, uint64_t **out_off) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!out_mm || !out_off) return ctx->fail(PGR_ERR_INVALID_ARG, "null output pointer");
    if (n && !lens) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (pgr_packed_words(n, lens) && !planes) return ctx->fail(PGR_ERR_INVALID_ARG, "null plane array");
    static const uint64_t no_words = 0;
    StageSrc src;
    src.lens = lens;
    src.planes = planes ? planes : &no_words;
    src.valid = valid;
    return shmmr_batch_host(ctx, spec, n, src, rids, padding, out_mm, out_off);
}

extern "C" int pgr_frag_recs_batch(pgr_ctx *ctx, const pgr_spec *spec, uint32_t n, const uint8_t *const *seqs,
                                   const uint64_t *lens, const uint32_t *sids, int query_side, pgr_frag_rec **out_recs,
                                   uint64_t **out_off) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!out_recs || !out_off) return ctx->fail(PGR_ERR_INVALID_ARG, "null output pointer");
    *out_recs = nullptr;
    *out_off = nullptr;
    int rc = check_spec(ctx, spec);
    if (rc) return rc;
    pgr_batch *b = nullptr;
    if ((rc = pgr_batch_from_ascii(ctx, n, seqs, lens, &b))) return rc;
    pgr_shmmrs *s = nullptr;
    rc = pgr_shmmrs_compute(ctx, b, spec, nullptr, 0, &s);
    pgr_batch_destroy(b);
    if (rc) return rc;
    const uint64_t np = pgr_shmmrs_n_pairs(s);
    DevBuf tmp;
    if ((rc = tmp.ensure(ctx, std::max<uint64_t>(np, 1) * sizeof(pgr_frag_rec)))) {
        pgr_shmmrs_destroy(s);
        return rc;
    }
    uint64_t n_out = 0;
    rc = pgr_shmmrs_to_frag_recs_device(ctx, s, sids, query_side, (pgr_frag_rec *)tmp.p, np, &n_out);
    uint64_t *off = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    pgr_frag_rec *recs = (pgr_frag_rec *)malloc(std::max<uint64_t>(np, 1) * sizeof(pgr_frag_rec));
    if (!rc && (!off || !recs)) rc = ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
    if (!rc) {
        uint64_t acc = 0;
        for (uint32_t c = 0; c < n; ++c) {
            off[c] = acc;
            const uint64_t cnt = s->h_off[c + 1] - s->h_off[c];
            if (cnt > 1) acc += cnt - 1;
        }
        off[n] = acc;
        if (np) rc = ctx->d2h(recs, tmp.p, np * sizeof(pgr_frag_rec));
    }
    tmp.release(ctx);
    pgr_shmmrs_destroy(s);
    if (rc) {
        free(off);
        free(recs);
        return rc;
    }
    *out_recs = recs;
    *out_off = off;
    return PGR_OK;
}
