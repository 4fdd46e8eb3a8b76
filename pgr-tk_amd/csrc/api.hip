// api.hip -- host side of libpgrhip.so: context, resident batches, orchestration of the
// sequence_to_shmmrs pipeline, and the C ABI declared in include/pgr_hip.h.
//
// Pipeline of one pgr_shmmrs_compute:
//   level1_tile_kernel  (dominant)  -> unordered per-tile segments of level-1 minimizers
//   level1_tail_kernel              -> per-contig tail segment (rescan-only positions)
//   level1_chunk_kernel             -> exact state machine (chunked, verified seams) for contigs the
//                                      closed form cannot do
//   scan(seg counts) + gather       -> ordered per-contig level-1 lists
//   select(reduce) x2, select(min_span) -> final MM128 lists (+ rid patch)
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
#include "pgr_small.h"

using namespace pgr;

namespace {
struct Tmp_list {  // small RAII device allocation from the context's caching allocator
    pgr_ctx *ctx;
    void *p = nullptr;
    explicit Tmp_list(pgr_ctx *c) : ctx(c) {}
    ~Tmp_list() { ctx->dfree(p); }
    int alloc(size_t bytes) { return ctx->dmalloc(&p, bytes < 16 ? 16 : bytes); }
};
}  // namespace

static std::string g_create_error;

extern "C" const char *pgr_version(void) { return "pgr-hip 0.3.0 (gfx950)"; }

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
        {"index_two_key_sort", &O::index_two_key_sort}, {"no_fused_query", &O::no_fused_query},
        {"no_query_chaining", &O::no_query_chaining}, {"query_global_sort", &O::query_global_sort},
        {"fused_query_hits", &O::fused_query_hits}, {"exchange_timeout_s", &O::exchange_timeout_s},
        {"no_island_relay", &O::no_island_relay}, {"no_short_tiles", &O::no_short_tiles}, {"no_pre_islands", &O::no_pre_islands}, {"island_chunk_min", &O::island_chunk_min}};
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
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
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
    for (const OptionName &o : option_names()) {
        std::string env = "PGR_";
        for (const char *c = o.name; *c; ++c) env += (char)toupper((unsigned char)*c);
        if (const char *v = getenv(env.c_str())) {
            char *end = nullptr;
            const long long x = strtoll(v, &end, 10);
            ctx->opt.*(o.field) = (end && end != v) ? (int64_t)x : 1;  // PGR_X=1, PGR_X=<number>, or a bare PGR_X= / PGR_X=on
        }
    }
    *out = ctx;
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
    if (hipEventRecord(ctx->ev_alloc, ctx->stream) != hipSuccess) {
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
    if (st != ctx->stream && hipStreamWaitEvent(st, ctx->ev_alloc, 0) != hipSuccess)  // batch_alloc's copies (main stream)
        return fail(PGR_ERR_DEVICE, "H2D pipeline failed");
    // ASCII stream: word wi of the batch <-> bytes [32*wi, 32*wi+32).  Two pinned windows of 32 MiB: while window
    // i is on its way to the GPU (H2D + pack kernel) the host threads fill window i+1.
    const uint64_t WIN_WORDS = 1ull << 20;  // 32 MiB of ASCII per window
    const uint64_t win_words = std::min<uint64_t>(std::max<uint64_t>(b->total_words, 1), WIN_WORDS);
    if (ctx->ensure_pinned(2 * win_words * 32) || ctx->ws_ascii.ensure(ctx, 2 * win_words * 32))
        return fail(PGR_ERR_NOMEM, "staging buffers: " + ctx->err);
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
    if (st != ctx->stream && hipStreamWaitEvent(st, ctx->ev_alloc, 0) != hipSuccess)  // batch_alloc's copies (main stream)
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
    const uint64_t win_words = pipe ? WIN_WORDS
                                    : std::min<uint64_t>(std::max<uint64_t>({(uint64_t)1, std::min<uint64_t>(b->total_words, 4 * PIECE),
                                                                             (b->total_words + 5) / 6}),
                                                         WIN_WORDS);
    if (ctx->ensure_pinned(2 * win_words * 12)) return fail(PGR_ERR_NOMEM, "staging buffers: " + ctx->err);
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
    for (uint64_t w0 = 0; w0 < b->total_words; w0 += win_words, slot ^= 1) {
        const uint64_t w1 = std::min(b->total_words, w0 + win_words);
        uint64_t *pin_planes = (uint64_t *)((uint8_t *)ctx->pinned + (size_t)slot * win_words * 12);
        uint32_t *pin_valid = (uint32_t *)(pin_planes + win_words);
        const auto tw0 = std::chrono::steady_clock::now();
        if (used[slot] && hipEventSynchronize(done[slot]) != hipSuccess)  // this window's previous trip is over
            return fail(PGR_ERR_DEVICE, "H2D pipeline failed");
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
                stream_copy(pin_planes + j.out, src.planes + g, (j.wl1 - j.wl0) * sizeof(uint64_t));
                if (src.valid) stream_copy(pin_valid + j.out, src.valid + g, (j.wl1 - j.wl0) * sizeof(uint32_t));
            }
        });
        const auto tw3 = std::chrono::steady_clock::now();
        // the validity plane only travels when it says something: a window of ASCII in which the packer met no non-ACGT byte
        // (the usual case) and packed input without a validity plane put 0.25 B per base on the link, the plane is written on
        // the device from the contig lengths
        const bool has_valid = packed ? src.valid != nullptr : win_bad.load() != 0;
        if (hipMemcpyAsync(b->d.planes + w0, pin_planes, (w1 - w0) * sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess ||
            (has_valid &&
             hipMemcpyAsync(b->d.valid + w0, pin_valid, (w1 - w0) * sizeof(uint32_t), hipMemcpyHostToDevice, st) != hipSuccess))
            return fail(PGR_ERR_DEVICE, "H2D copy of the packed window failed");
        if (packed || !has_valid) launch_sanitize_packed(st, b->d, n, w0, w1, has_valid ? 1 : 0);
        if (hipEventRecord(done[slot], st) != hipSuccess) return fail(PGR_ERR_DEVICE, "H2D pipeline failed");
        used[slot] = true;
        if (ctx->opt.debug > 1) {
            auto us = [](auto a, auto b2) { return std::chrono::duration<double, std::micro>(b2 - a).count(); };
            const auto tw4 = std::chrono::steady_clock::now();
            fprintf(stderr, "[pgr]     window of %.1f Mbp: waited %.0f us for its previous trip, %zu jobs listed in %.0f us, filled in %.0f us (%.1f GB/s of planes), enqueued in %.0f us\n",
                    (double)(w1 - w0) * 32e-6, us(tw0, tw1), jobs.size(), us(tw1, tw2), us(tw2, tw3), (double)(w1 - w0) * 8e-3 / us(tw2, tw3), us(tw3, tw4));
        }
    }
    // per-contig counts of non-ACGT bytes (host-packed input: counted by the packer; packed input: by the kernel above)
    if (!packed)
        for (uint32_t i = 0; i < n; ++i) b->host_saw_invalid = b->host_saw_invalid || b->h_n_invalid[i] != 0;
    if (!packed && n &&
        hipMemcpyAsync(b->d.n_invalid, b->h_n_invalid.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(PGR_ERR_DEVICE, "H2D copy of the invalid-byte counts failed");
    // On the context's own stream nothing waits here: the consumer is ordered behind the copies and synchronizes once at
    // its end; the staging thread of the pipelined path (own stream) hands over finished batches.
    if (pipe) return PGR_OK;
    if (st != ctx->stream) {
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess)
            return fail(PGR_ERR_DEVICE, "staging failed on the device");
    } else {
        ctx->staged_unsynced = true;
    }
    return PGR_OK;
}

static int batch_from_host(pgr_ctx *ctx, uint32_t n, const StageSrc &src, pgr_batch **out) {
    *out = nullptr;
    PGR_HIP(ctx, hipSetDevice(ctx->device));
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
    PGR_HIP(ctx, hipSetDevice(ctx->device));
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

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Exact state machine (level1_chunk_kernel) over ISLANDS: position ranges [B, E) of a contig (tile aligned, or
// the whole contig) that replace the closed-form tiles near an irregularity (non-ACGT byte, palindromic
// k-mer) or everything when the spec has no tile path (w < 17).  Inside an island the chunks of 32 kbp are
// seamed by emission step and every seam is verified against the previous chunk's end state; the island's
// left edge trusts the warm-up (everything before it is regular by construction: islands keep a clean tile
// on both sides), its right edge is verified with a probe (warm-up only) at E and the island grows when the
// machine has not yet returned to its regular regime there.
struct Island {
    uint32_t contig;
    uint64_t B, E;
    bool whole;  // one chunk for the whole contig (last resort)
    // a tile of the island saw a palindromic k-mer: inside a stretch of skipped pushes a chunk needs the true state of the
    // chunk in front (one seam per round), so such islands keep long chunks; islands around non-ACGT bytes verify at the
    // first try and are cut short for parallelism
    bool pal = true;
};

static int run_exact_islands(pgr_ctx *ctx, const pgr_batch *b, L1Args &a, std::vector<Island> &islands,
                             const std::vector<uint32_t> &tile_first, uint32_t tc, uint64_t region_base,
                             const std::vector<uint32_t> &empty_seg_ranges) {
    hipStream_t st = ctx->stream;
    // chunk length: 32 kbp for big jobs, shorter when the islands are few so that there are still thousands of wavefronts
    // (one per chunk), down to 1024 positions: a round costs what its slowest chunk costs -- ~3 us per step of 64 positions,
    // 225 us for the 4096-position chunks that were the minimum while a chunk had to own the segment-table entry of the tile it
    // starts in.  The lists of the chunks that start in one tile are put together behind the last round (below).
    uint64_t island_bases = 0;
    for (const Island &is : islands) island_bases += is.E - is.B;
    const uint64_t CS_MIN = ctx->opt.island_chunk_min > 0 ? (uint64_t)((ctx->opt.island_chunk_min + 63) / 64 * 64) : 1024;
    // (~2.5 wavefronts per SIMD: below that a round waits for dependent instructions, above it the SIMDs are busy -- a step is
    // ~1.5 us of issue -- and shorter chunks only add warm-up steps)
    const uint64_t CS_SHORT = std::min<uint64_t>(32768, std::max<uint64_t>(CS_MIN, ((island_bases / 2560 + 1023) / 1024) * 1024));
    // segment ranges of (re)built islands -- and of the tiles the caller leaves out --, cleared by ONE kernel before the next chunk launch
    std::vector<uint32_t> zero_ranges(empty_seg_ranges);
    struct HChunk {
        ChunkDesc d;
        size_t island;
        bool full_cap = false, probe = false, retired = false;
        bool final = false;     // the state this chunk started from is known to be the true one
        ChunkState t_out;       // final: the true state at ce
        uint32_t ring_src = 0;  // final: ring slot that holds the true ring at ce (this chunk's, or the one it passed through)
        uint64_t n_push = 0, bmin = 0;  // of the last run: pushes at the steps [cs, ce), smallest x of those with branch 2 enabled
        uint64_t n_out = 0;             // of the last run: elements in the chunk's region
        bool dropped = false;           // its output is not part of the list (a stuck machine passed through it)
    };
    std::vector<HChunk> ch;
    std::vector<ChunkState> s_in, s_out;
    std::vector<uint32_t> status;
    std::vector<size_t> todo;
    auto cap_of = [&](uint64_t len, bool full) -> uint64_t {
        return full ? len + a.w + 320 : std::min<uint64_t>(len + a.w + 320, len / 4 + 1024);
    };
    int rc;
    // (re)build the chunks of one island; its tile (and tail) segments become empty first
    auto build = [&](size_t ii) -> int {
        const Island &is = islands[ii];
        const uint32_t c = is.contig;
        const uint64_t L = b->h_len[c];
        const uint32_t nt = tile_first[c + 1] - tile_first[c];
        const uint32_t seg0 = tile_first[c] + c;
        // (a contig that is one tile may be longer than a tile core: clamped)
        uint32_t rng[2] = {seg0 + (uint32_t)std::min<uint64_t>(nt ? nt - 1 : 0, is.B / tc), seg0 + (uint32_t)std::min<uint64_t>(nt, (is.E + tc - 1) / tc)};
        if (is.E >= L) rng[1] = seg0 + nt + 1;  // including the tail segment
        zero_ranges.push_back(rng[0]);
        zero_ranges.push_back(rng[1]);
        // (round 3 kept 32 kbp chunks for islands around palindromic k-mers: their seams were corrected one per host round.  A
        // state now passes through chunks without pushes and through chunks a stuck machine cannot emit in, on the host)
        const uint64_t CS = (is.pal && ctx->opt.no_island_relay) ? 32768 : CS_SHORT;
        const uint64_t nch = is.whole ? 1 : (is.E - is.B + CS - 1) / CS;
        for (uint64_t j = 0; j < nch; ++j) {
            HChunk h;
            memset(&h.d, 0, sizeof(h.d));
            memset(&h.t_out, 0, sizeof(h.t_out));
            h.island = ii;
            h.d.contig = c;
            h.d.cs = is.whole ? 0 : is.B + j * CS;
            h.d.ce = is.whole ? L : std::min<uint64_t>(is.E, is.B + (j + 1) * CS);
            h.d.emit_lo_pos = (j == 0) ? is.B : 0;
            h.d.drain_end = h.d.ce;
            if (j + 1 == nch && is.E < L) h.d.drain_end = std::min<uint64_t>(L, is.E + 320);
            h.d.seg = seg0 + (uint32_t)std::min<uint64_t>(nt ? nt - 1 : 0, h.d.cs / tc);
            h.d.warm = 256;
            // a long island is a long irregular stretch (a run of N, low-complexity sequence): every position emits there
            // (ties, shmmrutils.rs:516-527), the sparse region estimate would overflow and the chunk run twice
            // (so is an island around non-ACGT bytes, however short: it may be one of the two ends of a long gap)
            h.full_cap = nch >= 8 || !is.pal;
            todo.push_back(ch.size());
            ch.push_back(h);
        }
        if (is.E < L) {  // probe: what a warmed-up (regular) machine looks like at E
            HChunk h;
            memset(&h.d, 0, sizeof(h.d));
            memset(&h.t_out, 0, sizeof(h.t_out));
            h.island = ii;
            h.probe = true;
            h.d.contig = c;
            h.d.cs = h.d.ce = h.d.drain_end = is.E;
            h.d.seg = 0xFFFFFFFFu;
            h.d.warm = 256;
            todo.push_back(ch.size());
            ch.push_back(h);
        }
        return PGR_OK;
    };
    for (size_t ii = 0; ii < islands.size(); ++ii)
        if ((rc = build(ii))) return rc;

    uint64_t next_region = region_base;
    const auto t_isl0 = std::chrono::steady_clock::now();
    auto isl_lap = [&](const char *what, int round) {
        if (ctx->opt.debug_times)
            fprintf(stderr, "[pgr]     islands round %d %-34s at %7.1f us\n", round, what,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_isl0).count());
    };
    for (int round = 0; !todo.empty(); ++round) {
        // Every round settles at least one seam or grows / merges an island, so the number of rounds is bounded by the number
        // of chunks plus the growth steps; in practice it is 1-3: a state is handed through chunks that cannot change it on the
        // host (see the verification below), and only a chunk whose predecessor's state is final runs again.
        if (round > 1024 + 4 * (int)ch.size()) return ctx->fail(PGR_ERR_INTERNAL, "exact-machine islands did not converge");
        const size_t nq = todo.size();
        std::vector<ChunkDesc> descs(nq);
        for (size_t q = 0; q < nq; ++q) {
            HChunk &h = ch[todo[q]];
            h.d.ring_out = h.probe ? 0xFFFFFFFFu : (uint32_t)todo[q];
            if (!h.d.override_state) h.d.ring_in = 0xFFFFFFFFu;
            h.d.region_off = next_region;
            h.d.region_cap = h.probe ? 1 : cap_of(h.d.drain_end - h.d.cs, h.full_cap);
            next_region += h.d.region_cap;
            descs[q] = h.d;
        }
        if ((rc = ctx->ws_l1.ensure_keep(ctx, (next_region + 1) * sizeof(L1Rec), st))) return rc;
        a.out = (L1Rec *)ctx->ws_l1.p;
        // one block on the device and its pinned image on the host: [descriptors | states at cs | states at ce | push info | status]
        const size_t desc_bytes = nq * sizeof(ChunkDesc);
        const size_t down_bytes = nq * (2 * sizeof(ChunkState) + 4 * sizeof(uint64_t) + sizeof(uint32_t));
        if ((rc = ctx->ws_serial.ensure(ctx, desc_bytes + down_bytes)) || (rc = ctx->ensure_imail(desc_bytes + down_bytes))) return rc;
        ChunkDesc *d_desc = (ChunkDesc *)ctx->ws_serial.p;
        ChunkState *d_in = (ChunkState *)(d_desc + nq);
        ChunkState *d_out = d_in + nq;
        uint64_t *d_info = (uint64_t *)(d_out + nq);
        uint32_t *d_stat = (uint32_t *)(d_info + 4 * nq);
        uint8_t *h_img = (uint8_t *)ctx->imail;
        memcpy(h_img, descs.data(), desc_bytes);
        Tmp_list d_zr(ctx);  // (the source vector and this block live until the synchronization at the end of the round)
        if (!zero_ranges.empty()) {
            if ((rc = d_zr.alloc(zero_ranges.size() * sizeof(uint32_t)))) return rc;
            PGR_HIP(ctx, hipMemcpyAsync(d_zr.p, zero_ranges.data(), zero_ranges.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            launch_zero_seg_ranges(st, a, (const uint32_t *)d_zr.p, (uint32_t)(zero_ranges.size() / 2));
        }
        PGR_HIP(ctx, hipMemcpyAsync(d_desc, h_img, desc_bytes, hipMemcpyHostToDevice, st));
        PGR_HIP(ctx, hipMemsetAsync(d_in, 0, nq * sizeof(ChunkState), st));
        // one ring slot per chunk ever built (ids = indices into `ch`), kept across the rounds
        if ((rc = ctx->ws_flags.ensure_keep(ctx, ch.size() * CHUNK_RING_WORDS * sizeof(uint64_t), st))) return rc;
        isl_lap("chunks listed, buffers ready", round);
        launch_level1_chunks(st, a, d_desc, (uint32_t)nq, d_in, d_out, d_stat, (uint64_t *)ctx->ws_flags.p, d_info);
        PGR_HIP(ctx, hipMemcpyAsync(h_img + desc_bytes, d_in, down_bytes, hipMemcpyDeviceToHost, st));
        isl_lap("chunk kernel enqueued", round);
        PGR_HIP(ctx, hipStreamSynchronize(st));
        isl_lap("states back on the host", round);
        const ChunkState *r_in = (const ChunkState *)(h_img + desc_bytes), *r_out = r_in + nq;
        const uint64_t *r_info = (const uint64_t *)(r_out + nq);
        const uint32_t *r_stat = (const uint32_t *)(r_info + 4 * nq);
        PGR_HIP(ctx, hipGetLastError());
        zero_ranges.clear();
        s_in.resize(ch.size());
        s_out.resize(ch.size());
        status.resize(ch.size());
        for (size_t q = 0; q < nq; ++q) {
            s_in[todo[q]] = r_in[q];
            s_out[todo[q]] = r_out[q];
            status[todo[q]] = r_stat[q];
            ch[todo[q]].n_push = r_info[4 * q];
            ch[todo[q]].bmin = r_info[4 * q + 1];
            ch[todo[q]].n_out = r_info[4 * q + 3];
            ch[todo[q]].dropped = false;
        }
        if (ctx->opt.debug) {
            uint32_t worst = 0;
            size_t wq = 0;
            uint64_t steps = 0;
            uint64_t worst_t = 0;
            for (size_t q = 0; q < nq; ++q) {
                steps += r_stat[q] >> 8;
                if ((r_info[4 * q + 2] & 0xFFFFFFFFull) > (worst_t & 0xFFFFFFFFull)) {
                    worst_t = r_info[4 * q + 2];
                    worst = r_stat[q] >> 8;
                    wq = q;
                }
            }
            fprintf(stderr, "[pgr]   slowest chunk: %.1f us (%.1f us before its first step), warm %u override %u drain_end %llu\n",
                    (worst_t & 0xFFFFFFFFull) / 100.0, (worst_t >> 32) / 100.0, descs[wq].warm,
                    descs[wq].override_state, (unsigned long long)descs[wq].drain_end);
            fprintf(stderr, "[pgr] exact islands round %d: %zu chunks run, %zu islands, region end %llu; %llu steps of 64 positions, "
                    "the longest chunk %u (chunk [%llu, %llu) of contig %u%s)\n", round, nq, islands.size(),
                    (unsigned long long)next_region, (unsigned long long)steps, worst, (unsigned long long)descs[wq].cs,
                    (unsigned long long)descs[wq].ce, descs[wq].contig, descs[wq].seg == 0xFFFFFFFFu ? ", a probe" : "");
        }
        // ---- verify seams (chunks of an island are contiguous in `ch`, the probe comes last).  A chunk is FINAL once the state
        // it started from is known to be the true one: the island's first chunk (regular by construction), a chunk whose
        // recorded state at cs equals the true state its final predecessor left at ce (the warm-up was right, or the state was
        // installed), and a chunk without a push in [cs, ce) behind a final predecessor -- the machine does not move there
        // (shmmrutils.rs:477-480: a skipped position touches neither ring nor mdist), so its end state is its predecessor's
        // with the k-mer rolled on, and its (empty) output is right whatever state it ran with.  Only a chunk with a final
        // predecessor is corrected, with that predecessor's true state and ring: a correction never builds on a stale state.
        std::vector<size_t> next;
        std::vector<size_t> rebuild;  // islands to rebuild (grown or turned into one whole-contig chunk)
        const bool relay = !ctx->opt.no_island_relay;
        for (size_t i = 0; i < ch.size(); ++i) {
            HChunk &h = ch[i];
            if (h.retired) continue;
            Island &is = islands[h.island];
            if (status[i] & 2u) {  // the true state could not be installed: the contig as one chunk
                if (!is.whole) {
                    is.whole = true;
                    is.B = 0;
                    is.E = b->h_len[is.contig];
                    rebuild.push_back(h.island);
                }
                continue;
            }
            bool again = false;
            if (status[i] & 1u) {  // region overflow
                h.full_cap = true;
                again = true;
            }
            const bool has_prev = i > 0 && !ch[i - 1].retired && ch[i - 1].island == h.island;
            auto grow = [&]() {
                // the machine is not back in its regular regime at E: grow the island
                const uint64_t L = b->h_len[is.contig];
                is.E = std::min<uint64_t>(L, is.E + 4ull * tc);
                if (L - is.E < 2ull * tc) is.E = L;
                rebuild.push_back(h.island);
            };
            if (!relay) {  // the round-3 scheme (A/B): every seam against whatever the chunk in front produced last
                if (h.probe) {
                    if (has_prev && memcmp(&s_in[i], &s_out[i - 1], sizeof(ChunkState)) != 0) grow();
                } else if (!is.whole && h.d.cs > is.B && has_prev && memcmp(&s_in[i], &s_out[i - 1], sizeof(ChunkState)) != 0) {
                    h.d.override_state = 1;
                    h.d.in_state = s_out[i - 1];
                    h.d.ring_in = (uint32_t)(i - 1);
                    h.d.warm = 1024;
                    again = true;
                }
                if (again) next.push_back(i);
                continue;
            }
            if (h.probe) {
                if (!has_prev) h.final = true;
                else if (ch[i - 1].final && !h.final) {
                    if (memcmp(&s_in[i], &ch[i - 1].t_out, sizeof(ChunkState)) != 0) {
                        if (ctx->opt.debug) {
                            const ChunkState &p = s_in[i], &q = ch[i - 1].t_out;
                            fprintf(stderr, "[pgr] probe mismatch contig %u E=%llu: min_x %llx/%llx min_y %llx/%llx mdist %llu/%llu "
                                    "F0 %llx/%llx R0 %llx/%llx sig %llx/%llx\n", is.contig, (unsigned long long)is.E,
                                    (unsigned long long)p.min_x, (unsigned long long)q.min_x, (unsigned long long)p.min_y,
                                    (unsigned long long)q.min_y, (unsigned long long)p.mdist, (unsigned long long)q.mdist,
                                    (unsigned long long)p.F0, (unsigned long long)q.F0, (unsigned long long)p.R0,
                                    (unsigned long long)q.R0, (unsigned long long)p.ring_sig, (unsigned long long)q.ring_sig);
                        }
                        grow();
                    } else {
                        h.final = true;
                    }
                }
            } else if (is.whole || !has_prev || h.d.cs <= is.B) {  // the island's first chunk
                h.final = true;
                h.t_out = s_out[i];
                h.ring_src = (uint32_t)i;
            } else if (ch[i - 1].final) {
                const HChunk &pv = ch[i - 1];
                const ChunkState &t = pv.t_out;
                const bool kmer_ok = s_in[i].F0 == t.F0 && s_in[i].F1 == t.F1 && s_in[i].R0 == t.R0 && s_in[i].R1 == t.R1;
                if ((status[i] & 4u) && kmer_ok) {  // no push in [cs, ce): the state passes through
                    h.final = true;
                    h.t_out = t;
                    h.t_out.F0 = s_out[i].F0;
                    h.t_out.F1 = s_out[i].F1;
                    h.t_out.R0 = s_out[i].R0;
                    h.t_out.R1 = s_out[i].R1;
                    h.ring_src = pv.ring_src;
                } else if (memcmp(&s_in[i], &t, sizeof(ChunkState)) == 0) {
                    h.final = true;
                    h.t_out = s_out[i];
                    h.ring_src = (uint32_t)i;
                } else if (kmer_ok && !a.sketch && t.mdist > (uint64_t)(a.w - 1) && h.n_push >= a.w && h.bmin > t.min_x &&
                           h.d.drain_end <= h.d.ce && !(status[i] & 1u)) {
                    // The machine arrives STUCK: mdist is beyond w - 1 (a rescan measured the distance to a minimum from in
                    // front of a stretch of skipped pushes, shmmrutils.rs:505-514), so no rescan can fire, and no push of this
                    // chunk reaches down to min_mer (branch 2, :516-520) -- nothing is emitted, min_mer stays, mdist counts the
                    // pushes, and with >= w pushes the ring at ce holds this chunk's own last w pushes: exactly what its run
                    // from a warmed-up state left there.  Its output of that run is dropped.
                    h.final = true;
                    h.t_out = s_out[i];
                    h.t_out.min_x = t.min_x;
                    h.t_out.min_y = t.min_y;
                    h.t_out.mdist = t.mdist + h.n_push;
                    h.ring_src = (uint32_t)i;
                    h.dropped = true;
                } else {
                    if (ctx->opt.debug)
                        fprintf(stderr, "[pgr]   chunk %zu [%llu, %llu) of contig %u runs again from the true state: mdist %llu (warm-up %llu), "
                                "min_x %llx (%llx), %llu pushes, smallest branch-2 x %llx\n", i, (unsigned long long)h.d.cs,
                                (unsigned long long)h.d.ce, is.contig, (unsigned long long)t.mdist, (unsigned long long)s_in[i].mdist,
                                (unsigned long long)t.min_x, (unsigned long long)s_in[i].min_x, (unsigned long long)h.n_push,
                                (unsigned long long)h.bmin);
                    h.final = false;
                    h.d.override_state = 1;
                    h.d.in_state = t;
                    h.d.ring_in = pv.ring_src;  // the ring the last chunk with a push left at its end
                    h.d.warm = 1024;
                    again = true;
                }
            }  // else: the chunk in front is not settled yet
            if (again) next.push_back(i);
        }
        if (!rebuild.empty()) {
            std::sort(rebuild.begin(), rebuild.end());
            rebuild.erase(std::unique(rebuild.begin(), rebuild.end()), rebuild.end());
            for (auto &h : ch)
                if (std::binary_search(rebuild.begin(), rebuild.end(), h.island)) h.retired = true;
            next.erase(std::remove_if(next.begin(), next.end(), [&](size_t i) { return ch[i].retired; }), next.end());
            todo.swap(next);
            for (size_t ii : rebuild) {
                // merge with later islands of the same contig that the grown island now touches
                for (size_t jj = 0; jj < islands.size(); ++jj)
                    if (jj != ii && islands[jj].contig == islands[ii].contig && islands[jj].B < islands[ii].E + tc &&
                        islands[jj].B >= islands[ii].B && islands[jj].E > islands[ii].B && !islands[jj].whole &&
                        islands[jj].E != 0) {
                        islands[ii].E = std::max(islands[ii].E, islands[jj].E);
                        islands[ii].pal = islands[ii].pal || islands[jj].pal;
                        for (auto &h : ch)
                            if (h.island == jj) h.retired = true;
                        islands[jj].E = islands[jj].B = 0;  // absorbed
                    }
                todo.erase(std::remove_if(todo.begin(), todo.end(), [&](size_t i) { return ch[i].retired; }), todo.end());
                if ((rc = build(ii))) return rc;
            }
        } else {
            todo.swap(next);
        }
    }
    if (!zero_ranges.empty()) {  // (segment ranges of islands built in the last round: none in practice)
        Tmp_list d_zr(ctx);
        if ((rc = d_zr.alloc(zero_ranges.size() * sizeof(uint32_t)))) return rc;
        PGR_HIP(ctx, hipMemcpyAsync(d_zr.p, zero_ranges.data(), zero_ranges.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        launch_zero_seg_ranges(st, a, (const uint32_t *)d_zr.p, (uint32_t)(zero_ranges.size() / 2));
        PGR_HIP(ctx, hipStreamSynchronize(st));
    }
    isl_lap("seams verified", -1);
    // ---- the lists of the chunks that start in one tile become that tile's segment: copied back to back into a fresh region
    // (the chunks of an island are contiguous in `ch`, in position order; their counts came back with the states)
    {
        std::vector<uint64_t> img;  // copies (3 words each), then segment entries (3 words each)
        std::vector<uint64_t> segs;
        uint32_t cur_seg = 0xFFFFFFFFu;
        for (const HChunk &h : ch) {
            if (h.retired || h.probe || h.d.seg == 0xFFFFFFFFu) continue;
            if (h.d.seg != cur_seg) {
                cur_seg = h.d.seg;
                segs.push_back((uint64_t)cur_seg | ((uint64_t)h.d.contig << 32));
                segs.push_back(next_region);
                segs.push_back(0);
            }
            if (h.dropped || h.n_out == 0) continue;
            img.push_back(h.d.region_off);
            img.push_back(next_region);
            img.push_back(h.n_out);
            segs[segs.size() - 1] += h.n_out;
            next_region += h.n_out;
        }
        const size_t n_copies = img.size() / 3, n_set = segs.size() / 3;
        if (n_set) {
            for (size_t i = 0; i < n_set; ++i)
                if (segs[3 * i + 2] > 0xFFFFFFFFull) return ctx->fail(PGR_ERR_INTERNAL, "a tile's exact list exceeds 2^32 elements");
            img.insert(img.end(), segs.begin(), segs.end());
            const size_t bytes = img.size() * sizeof(uint64_t);
            if ((rc = ctx->ws_l1.ensure_keep(ctx, (next_region + 1) * sizeof(L1Rec), st)) || (rc = ctx->ws_serial.ensure(ctx, bytes)) ||
                (rc = ctx->ensure_imail(bytes)))
                return rc;
            a.out = (L1Rec *)ctx->ws_l1.p;
            memcpy(ctx->imail, img.data(), bytes);  // (pinned, and untouched until this context's next island call: no wait here)
            PGR_HIP(ctx, hipMemcpyAsync(ctx->ws_serial.p, ctx->imail, bytes, hipMemcpyHostToDevice, st));
            launch_assemble_chunks(st, a, (const uint64_t *)ctx->ws_serial.p, (uint32_t)n_copies,
                                   (const uint64_t *)ctx->ws_serial.p + 3 * n_copies, (uint32_t)n_set);
        }
    }
    isl_lap("tile lists assembled (enqueued)", -1);
    return PGR_OK;
}

static int check_spec(pgr_ctx *ctx, const pgr_spec *spec) {
    if (!spec) return ctx->fail(PGR_ERR_INVALID_ARG, "null spec");
    // shmmrutils.rs:443-445 / :575-576
    if (spec->k == 0 || spec->k > 56) return ctx->fail(PGR_ERR_BAD_SPEC, "spec.k must be in 1..56");
    if (spec->r == 0 || spec->r > 12) return ctx->fail(PGR_ERR_BAD_SPEC, "spec.r must be in 1..12");
    if (!spec->sketch && (spec->w == 0 || spec->w > 128)) return ctx->fail(PGR_ERR_BAD_SPEC, "spec.w must be in 1..128");
    return PGR_OK;
}

// One pass of the hot path over a resident batch of SHORT contigs (query batches, reads, fragmented assemblies): the
// one-workgroup-per-contig kernel of csrc/small.hip replaces tiles + tails + segment scans + the fused list kernel -- 4 launches
// and one synchronization instead of ~15 dependent operations.  handled == false: not eligible, or a contig was handed back
// (non-ACGT byte, palindromic k-mer, low-complexity list): the caller runs the general pipeline.
static int shmmrs_compute_small(pgr_ctx *ctx, const pgr_batch *b, const pgr_spec *spec, const uint32_t *rids, pgr_shmmrs **out,
                                bool &handled) {
    handled = false;
    const uint32_t n = b->n;
    // Small batches only: the kernel trades throughput for latency (a workgroup walks its contig serially: tiles, then the tail
    // on one wavefront, then the list stage between barriers).  Measured on 10 kbp contigs: 0.045 ms + 58 ns per contig against
    // 0.13 ms + 34 ns per contig for the general pipeline -- the lines cross near 3500 contigs (35 Mbp).
    if (n == 0 || n > SMALL_MAX_CONTIGS || b->total_bases > SMALL_MAX_BASES || spec->sketch || spec->w < (uint32_t)L1_MIN_W ||
        b->host_saw_invalid || ctx->opt.no_small_path)
        return PGR_OK;
    if (ctx->skip_small_once) {  // shmmr_batch_small has just run this kernel on these contigs and was handed them back
        ctx->skip_small_once = false;
        return PGR_OK;
    }
    uint32_t max_len = 0;
    uint64_t total_slots = 0;
    for (uint32_t c = 0; c < n; ++c) {
        if (b->h_len[c] > SMALL_MAX_LEN) return PGR_OK;
        max_len = std::max(max_len, b->h_len[c]);
        total_slots += b->h_len[c] / 32 + 64;
    }
    if (total_slots >= (1ull << 32)) return PGR_OK;
    hipStream_t st = ctx->stream;
    std::vector<SmallContig> &desc = ctx->keep_small_desc;  // source of an async H2D copy: lives in the context
    desc.resize(n);
    uint64_t s_off = 0;
    for (uint32_t c = 0; c < n; ++c) {
        desc[c].word_off = b->h_word_off[c];
        desc[c].len = b->h_len[c];
        desc[c].rid = rids ? rids[c] : c;
        desc[c].out_off = (uint32_t)s_off;
        desc[c].out_cap = b->h_len[c] / 32 + 64;
        s_off += desc[c].out_cap;
    }
    constexpr size_t N_STATUS = 10;  // (same result layout as the general path: status words in front of the offsets)
    int rc;
    if ((rc = ctx->ws_small_desc.ensure(ctx, (size_t)n * sizeof(SmallContig))) ||
        (rc = ctx->ws_small_cnt.ensure(ctx, 2 * ((size_t)n + 1) * sizeof(uint32_t))) ||
        (rc = ctx->ws_list_a.ensure(ctx, (size_t)total_slots * sizeof(pgr_mm128))) ||
        (rc = ctx->ws_scan_tmp.ensure(ctx, scan_counts_temp_bytes(n + 1))) ||
        (rc = ctx->ensure_mailbox(((size_t)n + 2) * sizeof(uint64_t))))
        return rc;
    uint32_t *d_counts = (uint32_t *)ctx->ws_small_cnt.p, *d_clean = d_counts + (n + 1);
    pgr_shmmrs *res = new pgr_shmmrs();
    res->ctx = ctx;
    res->n = n;
    auto bail = [&](int code) {
        pgr_shmmrs_destroy(res);
        return code;
    };
    if ((rc = ctx->dmalloc((void **)&res->d_block, (N_STATUS + (size_t)n + 1) * sizeof(uint64_t)))) return bail(rc);
    res->d_off = res->d_block + N_STATUS;
    const double dens = 2.0 / (double)(spec->w + 1);
    const double spec_key = (double)spec->w * 1e9 + spec->k * 1e6 + spec->r * 1e4 + spec->min_span;
    const double ratio = (ctx->est_spec_key == spec_key && ctx->est_final_ratio > 0) ? ctx->est_final_ratio * 1.15 : dens / 3.0 + 1e-4;
    uint64_t cap_res = std::max<uint64_t>((uint64_t)((double)b->total_bases * ratio) + 64ull * n + 1024, 16);
    uint64_t *mbox = (uint64_t *)ctx->mailbox;
    hipError_t e = hipEventRecord(ctx->ev[0], st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->ws_small_desc.p, desc.data(), (size_t)n * sizeof(SmallContig), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_counts + n, 0, sizeof(uint32_t), st);  // the fallback flag word
    if (e != hipSuccess) return bail(ctx->fail(PGR_ERR_DEVICE, std::string("small path set-up: ") + hipGetErrorString(e)));
    SmallArgs a;
    a.planes = b->d.planes;
    a.valid = b->d.valid;
    a.l1_cap = small_l1_cap(max_len, spec->w);
    a.desc = (const SmallContig *)ctx->ws_small_desc.p;
    a.n = n;
    a.w = spec->w;
    a.k = spec->k;
    a.r = spec->r;
    a.min_span = spec->min_span;
    a.tc = ((L1_EXT - 2 * (spec->w - 1)) / 64) * 64;
    a.out = (pgr_mm128 *)ctx->ws_list_a.p;
    a.counts = d_counts;
    a.flags = d_counts + n;
    launch_small_shmmr(st, a);
    launch_small_counts(st, d_counts, n, d_clean);
    if (scan_counts(st, ctx->ws_scan_tmp.p, scan_counts_temp_bytes(n + 1), d_clean, res->d_off, n + 1) != hipSuccess)
        return bail(ctx->fail(PGR_ERR_DEVICE, "scan failed"));
    for (int attempt = 0;; ++attempt) {
        ctx->dfree(res->d_mm);
        res->d_mm = nullptr;
        if ((rc = ctx->dmalloc((void **)&res->d_mm, cap_res * sizeof(pgr_mm128)))) return bail(rc);
        launch_small_gather(st, (const pgr_mm128 *)ctx->ws_list_a.p, a.desc, d_counts, res->d_off, n, res->d_mm, cap_res);
        e = hipEventRecord(ctx->ev_end, st);
        // (a consumer that does not wait for the host -- the query path -- enqueues its kernels here, see pgr_shmmrs_compute)
        if (e == hipSuccess && ctx->post_enqueue && (rc = ctx->post_enqueue(res->d_mm, res->d_off, cap_res, res->d_off + n)))
            return bail(rc);
        if (e == hipSuccess) e = hipMemcpyAsync(mbox, res->d_off, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(mbox + n + 1, d_counts + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) return bail(ctx->fail(PGR_ERR_DEVICE, std::string("small path: ") + hipGetErrorString(e)));
        if ((uint32_t)mbox[n + 1]) {  // a contig needs the general pipeline
            pgr_shmmrs_destroy(res);
            return PGR_OK;
        }
        if (mbox[n] <= cap_res || attempt > 0) break;
        cap_res = mbox[n] + 16;  // more survivors than estimated: gather again into a bigger buffer
    }
    res->h_off.assign(mbox, mbox + n + 1);
    res->count = mbox[n];
    res->rid_is_index = rids == nullptr;
    ctx->staged_unsynced = false;
    ctx->want_host_copy = false;
    if (b->total_bases) {
        ctx->est_spec_key = spec_key;
        ctx->est_final_ratio = (double)res->count / (double)b->total_bases;
    }
    pgr_prof prof;
    memset(&prof, 0, sizeof(prof));
    (void)hipEventElapsedTime(&prof.total_ms, ctx->ev[0], ctx->ev_end);
    prof.level1_ms = prof.total_ms;  // (one kernel does levels 1 and 2)
    prof.bases_tiled = b->total_bases;
    prof.n_tiles = n;
    ctx->prof = prof;
    *out = res;
    handled = true;
    return PGR_OK;
}

// One pass of the hot path over a resident batch.  Everything is enqueued on the context's stream with sizes that are
// upper bounds or estimates; the host reads the true counts ONCE at the end (one hipStreamSynchronize per call in the
// common case) and repeats a stage only when an estimate turned out too small:
//   stage 1  level-1 tiles + tails (+ exact islands when a tile flagged a palindromic k-mer / non-ACGT byte)
//   stage 2  scan of the segment counts (the level-1 total stays on the device)
//   stage 3  fused reduce x2 + min_span (grid = upper bound, surplus workgroups exit on the device-side total)
//   stage 4  scan of the block counts, ordered gather into the result, per-contig offsets, rid patch
// Batches of >= 64 Mbp synchronize once more after stage 1 (a 30 us round trip is nothing there and a flagged batch
// does not run stages 2-4 twice); smaller ones (the reference's real callers: <= 129 contigs per call, seq_db.rs:561, and
// single queries, ext.rs:252) run optimistically and redo stages 2-4 after the islands in the rare flagged case.
extern "C" int pgr_shmmrs_compute(pgr_ctx *ctx, const pgr_batch *b, const pgr_spec *spec, const uint32_t *rids,
                                  int padding, pgr_shmmrs **out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!b || !out) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    const bool dbg_t = ctx->opt.debug_times != 0;  // host-side timeline of the call on stderr
    const auto dbg_t0 = std::chrono::steady_clock::now();
    auto dbg_lap = [&](const char *what) {
        if (dbg_t)
            fprintf(stderr, "[pgr] shmmrs_compute %-28s at %7.1f us\n", what,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t0).count());
    };
    int rc = check_spec(ctx, spec);
    if (rc) return rc;
    if (b->ctx != ctx) return ctx->fail(PGR_ERR_STATE, "batch belongs to another context");
    PGR_HIP(ctx, hipSetDevice(ctx->device));
    if (!(padding && !spec->sketch && spec->r > 1)) {  // batches of short contigs: one workgroup per contig, 4 launches
        bool handled = false;
        if ((rc = shmmrs_compute_small(ctx, b, spec, rids, out, handled)) || handled) return rc;
    }
    hipStream_t st = ctx->stream;
    const uint32_t n = b->n;
    const bool sketch = spec->sketch != 0;
    const bool tiled = sketch || spec->w >= (uint32_t)L1_MIN_W;
    const uint32_t w_eff = sketch ? 1u : spec->w;
    // tile core: extended tile minus both halos, rounded down to the 64-position step of the exact kernel so
    // that islands of exact tiles start and end on step boundaries
    // (batches of short contigs -- reads --: tiles of one wavefront's 1024 positions, pgr_internal.h)
    const bool short_tiles = tiled && n && b->total_bases / n <= (uint64_t)L1_SHORT_MEAN_LEN && w_eff <= (uint32_t)L1_SHORT_MAX_W &&
                             !ctx->opt.no_short_tiles;
    const uint32_t ext = short_tiles ? (uint32_t)L1_EXT_SHORT : (uint32_t)L1_EXT;
    const uint32_t tc = ((ext - 2 * (w_eff - 1)) / 64) * 64;

    // ---- host plan: tiles for the closed form, list for the serial kernel.  (The table lives in the context: a million reads
    // are 4 MB that a fresh vector would fault in page by page on every call; big batches fill it on the pool's threads.)
    std::vector<uint32_t> &tile_first = ctx->h_tile_first;
    tile_first.resize((size_t)n + 1);
    std::vector<uint32_t> serial;
    uint64_t n_tiles64 = 0, bases_tiled = 0;
    if (tiled && n >= (1u << 17)) {
        constexpr uint32_t PIECE = 1u << 14;
        const uint32_t n_pieces = (n + PIECE - 1) / PIECE;
        std::vector<uint64_t> piece_tiles((size_t)n_pieces + 1, 0);
        HostPool::instance().parallel_for(n_pieces, [&](size_t p) {
            const uint32_t c0 = (uint32_t)p * PIECE, c1 = std::min<uint32_t>(n, c0 + PIECE);
            uint64_t t = 0;
            for (uint32_t c = c0; c < c1; ++c) t += l1_tiles_of(b->h_len[c], tc, ext);
            piece_tiles[p + 1] = t;
        });
        for (uint32_t p = 0; p < n_pieces; ++p) piece_tiles[p + 1] += piece_tiles[p];
        n_tiles64 = piece_tiles[n_pieces];
        bases_tiled = b->total_bases;
        if (n_tiles64 + n + 1 < (1ull << 31))
            HostPool::instance().parallel_for(n_pieces, [&](size_t p) {
                const uint32_t c0 = (uint32_t)p * PIECE, c1 = std::min<uint32_t>(n, c0 + PIECE);
                uint64_t t = piece_tiles[p];
                for (uint32_t c = c0; c < c1; ++c) {
                    tile_first[c] = (uint32_t)t;
                    t += l1_tiles_of(b->h_len[c], tc, ext);
                }
            });
    } else {
        for (uint32_t c = 0; c < n; ++c) {
            tile_first[c] = (uint32_t)n_tiles64;
            const uint64_t L = b->h_len[c];
            if (L == 0) continue;
            n_tiles64 += l1_tiles_of(L, tc, ext);  // every contig owns tile segments (the chunk kernel reuses them)
            if (tiled) bases_tiled += L;  // contigs with non-ACGT bytes too: only islands around them are replaced
            else serial.push_back(c);     // w < 17: the whole contig goes through the exact kernel
        }
    }
    if (n_tiles64 + n + 1 >= (1ull << 31)) return ctx->fail(PGR_ERR_INVALID_ARG, "batch too large (tile count)");
    tile_first[n] = (uint32_t)n_tiles64;
    const uint32_t n_tiles = (uint32_t)n_tiles64;
    const uint32_t n_segs = n_tiles + n;

    // level-1 buffer: one fixed slot per tile (2x the expected count: density 2/(w+1), sketch 2^-(4+r)) and
    // a cursor-allocated overflow region for dense tiles and the per-contig tails
    const double dens = sketch ? 1.0 / (double)(1ull << (4 + spec->r)) : 2.0 / (double)(spec->w + 1);
    const uint32_t slot = std::min<uint32_t>(tc, (((uint32_t)((double)tc * dens * 2.0) + 64 + 63) / 64) * 64);
    const uint64_t slots_total = (uint64_t)n_tiles * slot;
    uint64_t cap_par = (uint64_t)((double)bases_tiled * dens * 0.02) + 65536 + 32ull * n;  // (a contig's tail has its own slot)
    // low-complexity / N-rich input overflows the fixed tile slots by far more than that: remember what the last call with
    // this spec needed per base (a genome comes as many similar batches) instead of running stage 1 twice every time
    const double l1_key = (double)spec->w * 1e3 + spec->k + (sketch ? 0.5 : 0.0);
    if (ctx->est_l1_key == l1_key && ctx->est_ovf_ratio > 0)
        cap_par = std::max<uint64_t>(cap_par, (uint64_t)((double)bases_tiled * ctx->est_ovf_ratio * 1.1) + 65536 + 32ull * n);

    constexpr size_t N_CURSOR = 8;  // [0..2] level 1 (L1Args::cursor), [4..5] fused list stage
    constexpr size_t N_STATUS = 10;
    if ((rc = ctx->ws_tile_first.ensure(ctx, ((size_t)n + 1) * sizeof(uint32_t))) ||
        (rc = ctx->ws_seg_off.ensure(ctx, ((size_t)n_segs + 1) * sizeof(uint64_t))) ||
        (rc = ctx->ws_tile_desc.ensure(ctx, ((size_t)n_tiles + 1) * sizeof(TileDesc))) ||
        (rc = ctx->ws_seg_cnt.ensure(ctx, ((size_t)n_segs + 1) * sizeof(uint32_t))) ||
        (rc = ctx->ws_seg_cid.ensure(ctx, ((size_t)n_segs + 1) * sizeof(uint32_t))) ||
        (rc = ctx->ws_seg_dst.ensure(ctx, ((size_t)n_segs + 1) * sizeof(uint64_t))) ||
        (rc = ctx->ws_tile_lv.ensure(ctx, ((size_t)n_tiles + 1) * sizeof(uint64_t))) ||
        // one block that a single memset clears per call: cursors | contig flags | tile flags  (+ the status words)
        (rc = ctx->ws_cursor.ensure(ctx, (N_CURSOR + N_STATUS) * sizeof(unsigned long long) +
                                             std::max<size_t>(n, 1) * sizeof(uint32_t) + (size_t)n_tiles + 64)) ||
        (rc = ctx->ws_off_a.ensure(ctx, ((size_t)n + 1) * sizeof(uint64_t))) ||
        (rc = ctx->ws_off_b.ensure(ctx, (N_STATUS + (size_t)n + 1) * sizeof(uint64_t))) ||
        (rc = ctx->ensure_mailbox((N_STATUS + (size_t)n + 1) * sizeof(uint64_t))))
        return rc;
    unsigned long long *d_cursor = (unsigned long long *)ctx->ws_cursor.p;
    uint32_t *d_cflags = (uint32_t *)(d_cursor + N_CURSOR);
    uint8_t *d_tflags = (uint8_t *)(d_cflags + std::max<size_t>(n, 1));
    const size_t zero_bytes = N_CURSOR * sizeof(unsigned long long) + std::max<size_t>(n, 1) * sizeof(uint32_t) + (size_t)n_tiles + 16;
    uint64_t *mbox = (uint64_t *)ctx->mailbox;  // pinned: [0, N_STATUS) status, then the n+1 result offsets
    // (through the pinned mailbox -- free until this call's results come back into it, and the stream orders the two: a copy
    // from pageable memory is staged by the runtime, ~15 us during which nothing else is enqueued)
    memcpy(mbox, tile_first.data(), ((size_t)n + 1) * sizeof(uint32_t));
    PGR_HIP(ctx, hipMemcpyAsync(ctx->ws_tile_first.p, mbox, ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    dbg_lap("plan + tile table uploaded");
    uint32_t *d_rids = nullptr;
    if (rids && n) {
        if ((rc = ctx->ws_rids.ensure(ctx, (size_t)n * sizeof(uint32_t)))) return rc;
        PGR_HIP(ctx, hipMemcpyAsync(ctx->ws_rids.p, rids, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        d_rids = (uint32_t *)ctx->ws_rids.p;
    }

    L1Args a;
    a.b = b->d;
    a.n_contigs = n;
    a.n_tiles = n_tiles;
    a.tile_first = (const uint32_t *)ctx->ws_tile_first.p;
    a.desc = (TileDesc *)ctx->ws_tile_desc.p;
    a.tile_flags = d_tflags;
    a.tile_lv = nullptr;  // set once mark_invalid_tiles has filled it and run_islands has made it cumulative
    a.w = w_eff;
    a.k = spec->k;
    a.r = spec->r;
    a.tc = tc;
    a.ext = ext;
    a.sketch = sketch ? 1u : 0u;
    a.cursor = d_cursor;
    a.seg_off = (uint64_t *)ctx->ws_seg_off.p;
    a.seg_cnt = (uint32_t *)ctx->ws_seg_cnt.p;
    a.seg_cid = (uint32_t *)ctx->ws_seg_cid.p;
    a.contig_flags = d_cflags;

    pgr_prof prof;
    memset(&prof, 0, sizeof(prof));
    prof.n_tiles = n_tiles;
    prof.bases_tiled = bases_tiled;
    // big batches look at the level-1 status words once before the list stage is enqueued (one more round trip, ~45 us):
    // if a tile asked for the exact path the islands are fixed first and the list stage runs once.  Smaller batches run
    // optimistically and repeat stages 2-4 in the (rare) flagged case: cheaper than the round trip below ~1 Gbp.
    const uint64_t early_bp = (uint64_t)std::max<int64_t>(0, ctx->opt.early_sync_bp);
    // (a batch the host packer has counted non-ACGT bytes in is known to need islands: look at the flags before the list stage)
    const bool early_sync = b->total_bases >= early_bp || !serial.empty() || b->host_saw_invalid;
    const bool pad_fix = padding && !sketch && spec->r > 1;
    const bool do_reduce = !sketch && spec->r > 1;
    const uint32_t halo = do_reduce ? 2 * spec->r * spec->r : 1;
    const uint32_t slot2 = do_reduce ? 256u : FUSED_BLOCK_ELEMS;
    uint64_t serial_base = 0;  // first element of the serial regions inside the level-1 buffer
    bool islands_done = false;
    bool l2_cursor_clean = false;  // the list stage's cursor words were cleared by stage 1's memset

    // islands of exact tiles from the flags: flags[c] bit 0 = a tile of contig c saw a palindromic k-mer, n_invalid[c] = its non-ACGT
    // bytes, tf[tile] = tile flags (bit 0 palindromic k-mer, bit 1 non-ACGT byte in reach, bit 2 nothing but such bytes; bit 3 is set here)
    auto list_islands = [&](const uint32_t *flags, const uint32_t *n_invalid, uint8_t *tf, std::vector<Island> &islands,
                            std::vector<uint32_t> &gap_segs) {
        for (uint32_t c = 0; c < n; ++c) {
            if (n_invalid[c] == 0 && (sketch || !(flags[c] & 1u))) continue;
            const uint32_t t0 = tile_first[c], nt = tile_first[c + 1] - t0;
            const uint64_t L = b->h_len[c];
            // The inside of a long run of non-ACGT bytes (the gaps of a reference chromosome: up to 30 Mbp) needs no
            // machine at all.  Every position there pushes the same stale k-mer (shmmrutils.rs:461-476), so the level-1
            // list holds one element per position, all with one x -- ties keep them through both reductions
            // (:359-415) and the min_span stencil drops every one of them for having a neighbour with its x (:545-550).
            // What an element further than 2 r^2 list places from both ends of such a run contributes to the rest of
            // the list is nothing: tiles whose whole extended range is invalid AND whose two neighbours on either side
            // are too (>= 7 kbp of the run kept at each end) are left out -- their segments stay empty, the islands on
            // both sides end inside the run, where a warmed-up machine is exact.  (Round 3 pushed 40.9 Mbp of such
            // positions of a chromosome-like contig through the chunk kernel and the list stage: half of its 2.4 ms.)
            for (uint32_t t = 0, run = 0; t < nt; ++t) {  // tile t - 2 is deep when tiles t - 4 .. t are all inside a gap
                run = (tf[t0 + t] & 4) ? run + 1 : 0;
                if (run >= 5) tf[t0 + t - 2] |= 8;  // (bit 3: host only)
            }
            for (uint32_t t = 0; t < nt; ++t)
                if (tf[t0 + t] & 8) {
                    uint32_t e = t;
                    while (e + 1 < nt && (tf[t0 + e + 1] & 8)) ++e;
                    gap_segs.push_back(t0 + c + t);      // segment index of tile t of contig c
                    gap_segs.push_back(t0 + c + e + 1);
                    for (uint32_t q = t; q <= e; ++q) tf[t0 + q] = 0;  // not flagged: no island over them
                    t = e;
                }
            uint32_t n_flag = 0;
            for (uint32_t t = 0; t < nt; ++t) n_flag += tf[t0 + t] != 0;
            if (n_flag == 0) continue;
            if (sketch && n_invalid[c] == 0) continue;  // sketch has no state machine: palindromes are exact
            if (3ull * n_flag > nt) {  // mostly irregular: one island
                bool pal = false;
                for (uint32_t t = 0; t < nt; ++t) pal = pal || (tf[t0 + t] & 1);
                islands.push_back(Island{c, 0, L, false, pal});
                continue;
            }
            for (uint32_t t = 0; t < nt;) {
                if (!tf[t0 + t]) {
                    ++t;
                    continue;
                }
                // (no tile in front of the first flagged one: tile t - 1 is clean, so nothing irregular lies within its reach -- which
                // ends w - 1 + 64 positions INTO tile t --, and the machine that starts 256 positions in front of tile t is regular
                // at its first step by construction; behind tiles deep inside a gap it starts inside the run, where a warmed-up
                // machine is exact)
                uint32_t ta = t, tb = t;
                while (tb + 1 < nt && (tf[t0 + tb + 1] || (tb + 2 < nt && tf[t0 + tb + 2]))) ++tb;  // bridge 1-tile gaps
                bool any_pal = false;
                for (uint32_t q = ta; q <= tb; ++q) any_pal = any_pal || (tf[t0 + q] & 1);
                // a clean neighbour on the right for the machine to find back into its regular regime -- behind skipped pushes
                // (palindromic k-mers) it may arrive stuck; behind a non-ACGT byte it cannot: the byte lies >= w + k + 64
                // positions in front of the first clean tile (or that tile would be flagged), every position pushes, and the
                // ring holds only pushes from behind the byte when the island ends.  The probe at the island's end checks it.
                if (tb + 1 < nt && any_pal) ++tb;
                Island is{c, (uint64_t)ta * tc, tb + 1 == nt ? L : std::min<uint64_t>(L, (uint64_t)(tb + 1) * tc), false, false};
                is.pal = any_pal;
                if (L - is.E < 2ull * tc) is.E = L;  // the contig's tail region joins the island
                if (!islands.empty() && islands.back().contig == c && islands.back().E + tc >= is.B) {
                    islands.back().E = std::max(islands.back().E, is.E);
                    islands.back().pal = islands.back().pal || is.pal;
                } else {
                    islands.push_back(is);
                }
                t = tb + 1;
            }
        }
    };
    // Islands around non-ACGT bytes, listed while the tile kernel runs: a batch the host packer has counted such bytes in gets the
    // tile flags (all but the palindrome bit, which the tile kernel sets) on the host as soon as mark_invalid_tiles has run -- the
    // copy, the listing (and the cumulative last-valid table) used to sit between the tile kernel and the chunk kernel, 0.15 ms
    // of a chromosome-like contig's 1.2.  Used when the tile kernel reports no palindromic k-mer; otherwise listed again.
    std::vector<Island> pre_islands;
    std::vector<uint32_t> pre_gap_segs;
    bool pre_listed = false;
    // ---- stage 1
    auto stage1 = [&]() -> int {
        uint64_t serial_total = 0;
        for (uint32_t c : serial) serial_total += (uint64_t)b->h_len[c] / 4 + 4096;
        int r;
        if ((r = ctx->ws_l1.ensure(ctx, (slots_total + cap_par + serial_total + (uint64_t)n * L1_TAIL_SLOT + 1) * sizeof(L1Rec)))) return r;
        a.out = (L1Rec *)ctx->ws_l1.p;
        a.slot = slot;
        a.ovf_base = slots_total;
        a.cap = cap_par;
        a.tail_base = slots_total + cap_par;  // the contigs' tail slots sit between the overflow region and the exact regions
        serial_base = a.tail_base + (uint64_t)n * L1_TAIL_SLOT;  // (run_exact_islands grows the buffer behind this point)
        // cursors (both stages), contig flags, tile flags: cleared by the tile descriptor kernel when there are tiles
        if (!(tiled && bases_tiled)) PGR_HIP(ctx, hipMemsetAsync(d_cursor, 0, zero_bytes, st));
        if (n == 0) PGR_HIP(ctx, hipMemsetAsync((uint32_t *)ctx->ws_seg_cnt.p + n_segs, 0, sizeof(uint32_t), st));
        l2_cursor_clean = true;  // (otherwise the tail kernel writes the scan sentinel)
        if (!(tiled && bases_tiled)) PGR_HIP(ctx, hipEventRecord(ctx->ev[1], st));
        pre_listed = false;
        if (tiled && bases_tiled) {
            launch_level1_pre(st, a, (uint64_t *)ctx->ws_tile_lv.p);
            const bool pre = b->host_saw_invalid && !b->h_n_invalid.empty() && n_tiles && !ctx->opt.no_pre_islands;
            if (pre) {
                const size_t tb = scan_max_temp_bytes(n_tiles);
                if ((r = ctx->ws_scan_tmp.ensure(ctx, tb)) || (r = ctx->ensure_imail(n_tiles))) return r;
                PGR_HIP(ctx, scan_max_inplace(st, ctx->ws_scan_tmp.p, tb, (uint64_t *)ctx->ws_tile_lv.p, n_tiles));
                PGR_HIP(ctx, hipEventRecord(ctx->pre_ev[0], st));
                PGR_HIP(ctx, hipStreamWaitEvent(ctx->pre_stream, ctx->pre_ev[0], 0));
                PGR_HIP(ctx, hipMemcpyAsync(ctx->imail, d_tflags, n_tiles, hipMemcpyDeviceToHost, ctx->pre_stream));
                PGR_HIP(ctx, hipEventRecord(ctx->pre_ev[1], ctx->pre_stream));
            }
            PGR_HIP(ctx, hipEventRecord(ctx->ev[1], st));  // (prof.level1_ms is the tile kernel alone: descriptors and flags are in front of it)
            launch_level1_tiles(st, a);
            if (pre) {
                PGR_HIP(ctx, hipEventSynchronize(ctx->pre_ev[1]));
                std::vector<uint32_t> no_flags(n, 0u);
                pre_islands.clear();
                pre_gap_segs.clear();
                list_islands(no_flags.data(), b->h_n_invalid.data(), (uint8_t *)ctx->imail, pre_islands, pre_gap_segs);
                pre_listed = true;
                dbg_lap("islands around non-ACGT bytes listed");
            }
        }
        PGR_HIP(ctx, hipEventRecord(ctx->ev[2], st));
        if (!(tiled && bases_tiled)) launch_level1_tails(st, a);  // (otherwise every contig's last tile has run its tail)
        islands_done = false;
        return PGR_OK;
    };
    // ---- islands of exact tiles: around palindromic k-mers (skipped pushes, flagged by the tile kernel) and
    // non-ACGT bytes (flagged by mark_invalid_tiles); whole contigs when the spec has no tile path.  Synchronizes.
    auto run_islands = [&](uint64_t need_word) -> int {
        dbg_lap("islands: level-1 flags seen");
        std::vector<Island> islands;
        std::vector<uint32_t> gap_segs;  // [first, last + 1) segment ranges of tiles deep inside runs of non-ACGT bytes: emptied
        for (uint32_t c : serial) islands.push_back(Island{c, 0, b->h_len[c], false, true});
        const bool use_pre = pre_listed && !(need_word & 1ull) && serial.empty();  // (no tile saw a palindromic k-mer)
        if (use_pre) {
            islands = pre_islands;
            gap_segs = pre_gap_segs;
        } else if (tiled && bases_tiled && need_word) {
            std::vector<uint32_t> flags(n), n_invalid(n);
            std::vector<uint8_t> tf(n_tiles);
            if (n) {
                PGR_HIP(ctx, hipMemcpyAsync(flags.data(), d_cflags, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                PGR_HIP(ctx, hipMemcpyAsync(n_invalid.data(), b->d.n_invalid, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            }
            PGR_HIP(ctx, hipMemcpyAsync(tf.data(), d_tflags, n_tiles, hipMemcpyDeviceToHost, st));
            PGR_HIP(ctx, hipStreamSynchronize(st));
            dbg_lap("islands: tile flags on the host");
            list_islands(flags.data(), n_invalid.data(), tf.data(), islands, gap_segs);
        }
        {
            std::vector<uint32_t> cs;
            for (const auto &is : islands) cs.push_back(is.contig);
            std::sort(cs.begin(), cs.end());
            prof.n_serial_contigs = std::unique(cs.begin(), cs.end()) - cs.begin();
        }
        if (!islands.empty()) {
            L1Args as = a;
            as.w = sketch ? 1u : spec->w;  // the exact machine follows the spec literally (sketch ignores w)
            if (tiled && bases_tiled && n_tiles && !use_pre) {  // (use_pre: done in front of the tile kernel)
                // per-tile "last valid position" (written by mark_invalid_tiles) -> cumulative: the chunks' k-mer look-back
                // and forward roll cross a run of N of any length in one step
                const size_t tb = scan_max_temp_bytes(n_tiles);
                int r2;
                if ((r2 = ctx->ws_scan_tmp.ensure(ctx, tb))) return r2;
                PGR_HIP(ctx, scan_max_inplace(st, ctx->ws_scan_tmp.p, tb, (uint64_t *)ctx->ws_tile_lv.p, n_tiles));
            }
            if (tiled && bases_tiled && n_tiles) as.tile_lv = (uint64_t *)ctx->ws_tile_lv.p;
            dbg_lap("islands: listed");
            int r = run_exact_islands(ctx, b, as, islands, tile_first, tc, serial_base, gap_segs);
            if (r) return r;
            dbg_lap("islands: exact");
            a.out = as.out;
            prof.exact_bases = 0;
            for (const auto &is : islands) prof.exact_bases += is.E - is.B;  // final extents (islands may have grown)
        }
        islands_done = true;
        return PGR_OK;
    };

    // ---- stages 2-4.  n_blocks: grid of the fused kernel; cap2: its overflow region; cap_res: result capacity
    pgr_shmmrs *res = new pgr_shmmrs();
    res->ctx = ctx;
    res->n = n;
    // (the offsets of a million reads are 8 MB: the block of a destroyed result of this context is used again)
    if ((size_t)n + 1 >= (1u << 17) && ctx->spare_off.capacity() >= (size_t)n + 1) res->h_off.swap(ctx->spare_off);
    res->h_off.resize((size_t)n + 1);
    auto bail = [&](int code) {
        pgr_shmmrs_destroy(res);
        return code;
    };
// a failing HIP call behind this point must release `res` (and its device blocks) on its way out
#define PGR_HIP_BAIL(expr)                                                                                        \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess) return bail(ctx->fail(PGR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e))); \
    } while (0)
    // the pipeline's status words sit right in front of the result offsets: ONE copy brings both to the mailbox
    if ((rc = ctx->dmalloc((void **)&res->d_block, (N_STATUS + (size_t)n + 1) * sizeof(uint64_t)))) return bail(rc);
    res->d_off = res->d_block + N_STATUS;
    // estimates: level-1 count from the density (low-complexity sequence exceeds it: retried with the true count),
    // final count from this context's last result with the same spec (first call: a third of the level-1 estimate)
    const uint64_t l1_bound = slots_total + cap_par + b->total_bases / 4 + 4096ull * n + 4096;  // what stage 1 can emit at all
    // (per contig: the first window and the tail emit a few elements on top of the density -- NOT thousands: 2048 per contig made
    // the list stage of 10^6 reads a grid of 2 x 10^6 workgroups for 25 x 10^3 of work, 0.6 of its 0.8 ms, and 8 GB of slots)
    const uint64_t l1_est = std::min<uint64_t>(l1_bound, (uint64_t)((double)b->total_bases * dens * 1.06) + 16ull * n + 8192);
    uint32_t n_blocks = (uint32_t)((l1_est + FUSED_BLOCK_ELEMS - 1) / FUSED_BLOCK_ELEMS);
    uint64_t cap2 = (uint64_t)((double)l1_est * 0.01) + 65536;
    const double spec_key = (double)spec->w * 1e9 + spec->k * 1e6 + spec->r * 1e4 + spec->min_span + (sketch ? 0.5 : 0.0) + (padding ? 0.25 : 0.0);
    const double ratio = (ctx->est_spec_key == spec_key && ctx->est_final_ratio > 0) ? ctx->est_final_ratio * 1.15 : dens / 3.0 + 1e-4;
    uint64_t cap_res = std::max<uint64_t>((uint64_t)((double)b->total_bases * ratio) + 64ull * n + 1024, 16);
    pgr_mm128 *d_list = nullptr;  // ordered final list (before the padding artefact)
    uint64_t *d_loff = nullptr;
    uint64_t *d_total1 = (uint64_t *)ctx->ws_seg_dst.p + n_segs;
    std::vector<uint64_t> l1_off;  // only needed for the padding artefact
    auto stage2 = [&]() -> int {
        const size_t tb = scan_counts_temp_bytes(n_segs + 1);
        int r;
        if ((r = ctx->ws_scan_tmp.ensure(ctx, tb))) return r;
        PGR_HIP(ctx, scan_counts(st, ctx->ws_scan_tmp.p, tb, (const uint32_t *)ctx->ws_seg_cnt.p,
                                 (uint64_t *)ctx->ws_seg_dst.p, n_segs + 1));
        if (pad_fix)
            launch_contig_offsets(st, (const uint64_t *)ctx->ws_seg_dst.p, (const uint32_t *)ctx->ws_tile_first.p, n, n_segs,
                                  (uint64_t *)ctx->ws_off_a.p);
        return PGR_OK;
    };
    auto stage3 = [&]() -> int {
        int r;
        if ((r = ctx->ws_blk_cnt.ensure(ctx, ((size_t)n_blocks + 1) * sizeof(uint32_t))) ||
            (r = ctx->ws_blk_base.ensure(ctx, ((size_t)n_blocks + 1) * sizeof(uint64_t))) ||
            (r = ctx->ws_blk_off.ensure(ctx, ((size_t)n_blocks + 1) * sizeof(uint64_t))) ||
            (r = ctx->ws_start_rank.ensure(ctx, ((size_t)n_blocks + 1) * sizeof(uint32_t))))
            return r;
        // fixed output slot per fused workgroup (expected survivors: ~12 % after two reductions + min_span), plus
        // a cursor-allocated overflow region
        const uint64_t slots2 = (uint64_t)n_blocks * slot2;
        if ((r = ctx->ws_list_a.ensure(ctx, (slots2 + cap2 + 1) * sizeof(pgr_mm128)))) return r;
        if (!l2_cursor_clean) PGR_HIP(ctx, hipMemsetAsync(d_cursor + 4, 0, 2 * sizeof(unsigned long long), st));
        l2_cursor_clean = false;  // (a repeat of this stage alone clears it again)
        if (n_blocks == 0) PGR_HIP(ctx, hipMemsetAsync((uint32_t *)ctx->ws_blk_cnt.p, 0, sizeof(uint32_t), st));
        FusedArgsPub fa;
        fa.l1 = (const L1Rec *)ctx->ws_l1.p;
        fa.seg_cid = (const uint32_t *)ctx->ws_seg_cid.p;
        fa.k = spec->k;
        fa.seg_off = (const uint64_t *)ctx->ws_seg_off.p;
        fa.seg_cnt = (const uint32_t *)ctx->ws_seg_cnt.p;
        fa.seg_dst = (const uint64_t *)ctx->ws_seg_dst.p;
        fa.n_segs = n_segs;
        fa.total = d_total1;
        fa.r = spec->r;
        fa.padding = padding ? 1u : 0u;
        fa.min_span = spec->min_span;
        fa.do_reduce = do_reduce ? 1u : 0u;
        fa.halo = halo;
        fa.out = (pgr_mm128 *)ctx->ws_list_a.p;
        fa.slot = slot2;
        fa.ovf_base = slots2;
        fa.cap = cap2;
        fa.cursor = d_cursor + 4;
        fa.blk_off = (uint64_t *)ctx->ws_blk_off.p;
        fa.blk_cnt = (uint32_t *)ctx->ws_blk_cnt.p;
        fa.blk_first_seg = (uint32_t *)ctx->ws_start_rank.p;
        launch_fused_select_pub(st, fa, n_blocks);
        return PGR_OK;
    };
    bool scanned4 = false;
    uint64_t host_copy_elems = 0;
    auto stage4 = [&]() -> int {
        int r;
        if (!scanned4) {  // (a repeat of stage 4 alone only re-gathers into a bigger result buffer)
            const size_t tb2 = scan_counts_temp_bytes(n_blocks + 1);
            if ((r = ctx->ws_scan_tmp.ensure(ctx, tb2))) return r;
            PGR_HIP(ctx, scan_counts(st, ctx->ws_scan_tmp.p, tb2, (const uint32_t *)ctx->ws_blk_cnt.p,
                                     (uint64_t *)ctx->ws_blk_base.p, n_blocks + 1));
            scanned4 = true;
        }
        const uint64_t *d_nfinal = (const uint64_t *)ctx->ws_blk_base.p + n_blocks;
        if (!pad_fix) {  // common case: gather straight into the result buffer
            ctx->dfree(res->d_mm);
            res->d_mm = nullptr;
            if ((r = ctx->dmalloc((void **)&res->d_mm, cap_res * sizeof(pgr_mm128)))) return r;
            d_list = res->d_mm;
            d_loff = res->d_off;
        } else {
            if ((r = ctx->ws_list_b.ensure(ctx, cap_res * sizeof(pgr_mm128)))) return r;
            d_list = (pgr_mm128 *)ctx->ws_list_b.p;
            d_loff = (uint64_t *)ctx->ws_off_b.p + N_STATUS;
        }
        launch_gather_segments(st, (const pgr_mm128 *)ctx->ws_list_a.p, (const uint64_t *)ctx->ws_blk_off.p,
                               (const uint32_t *)ctx->ws_blk_cnt.p, (const uint64_t *)ctx->ws_blk_base.p, n_blocks, d_list,
                               cap_res);
        launch_offsets_by_rid(st, d_list, d_nfinal, cap_res, n, d_loff, d_cursor, d_total1, d_loff - N_STATUS);  // + status words
        if (d_rids) launch_patch_rid(st, d_list, d_nfinal, cap_res, d_rids, n);
        PGR_HIP(ctx, hipEventRecord(ctx->ev_end, st));
        // a consumer of the result that does not want to wait for the host (the query path: pair records, lookup, chaining)
        // enqueues its kernels here, behind stage 4 and in front of the one synchronization; a repeated pass calls it again.
        // (In front of the copies to the host as well: a DMA between two kernels costs ~20 us of bubbles.)
        if (ctx->post_enqueue && !pad_fix && (r = ctx->post_enqueue(d_list, d_loff, cap_res, d_nfinal))) return r;
        PGR_HIP(ctx, hipMemcpyAsync(mbox, d_loff - N_STATUS, (N_STATUS + (size_t)n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        // a small result that the caller wants on the host anyway (pgr_shmmr_batch) rides along with this round trip
        host_copy_elems = 0;
        if (ctx->want_host_copy && !pad_fix && cap_res * sizeof(pgr_mm128) <= (256u << 10) &&
            ctx->ensure_pinned_out(256u << 10) == PGR_OK) {
            PGR_HIP(ctx, hipMemcpyAsync(ctx->pinned_out, d_list, cap_res * sizeof(pgr_mm128), hipMemcpyDeviceToHost, st));
            host_copy_elems = cap_res;
        }
        return PGR_OK;
    };

    PGR_HIP_BAIL(hipEventRecord(ctx->ev[0], st));
    int from = 1;  // first stage to (re)run
    uint64_t n_final = 0;
    uint64_t l1_alloc_seen = 0;  // elements the level-1 kernels took from the overflow region
    for (int attempt = 0;; ++attempt) {
        if (attempt > 8) return bail(ctx->fail(PGR_ERR_INTERNAL, "shimmer pipeline: buffers kept overflowing"));
        if (from <= 1) {
            if ((rc = stage1())) return bail(rc);
            if (early_sync) {
                PGR_HIP_BAIL(hipMemcpyAsync(mbox, d_cursor, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
                if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess)
                    return bail(ctx->fail(PGR_ERR_DEVICE, "level-1 kernels failed on the device"));
                l1_alloc_seen = mbox[0];
                if (mbox[1] || mbox[0] > cap_par) {  // cursor region too small: grow and redo
                    cap_par = (uint64_t)((double)mbox[0] * 1.1) + 65536;
                    continue;
                }
                if ((mbox[2] || !serial.empty()) && (rc = run_islands(mbox[2]))) return bail(rc);
                islands_done = true;
            }
            PGR_HIP_BAIL(hipEventRecord(ctx->ev[3], st));
        }
        if (from <= 2 && (rc = stage2())) return bail(rc);
        if (from <= 3) {
            scanned4 = false;
            if ((rc = stage3())) return bail(rc);
        }
        if ((rc = stage4())) return bail(rc);
        if (pad_fix) {
            l1_off.resize((size_t)n + 1);
            PGR_HIP_BAIL(hipMemcpyAsync(l1_off.data(), ctx->ws_off_a.p, ((size_t)n + 1) * sizeof(uint64_t),
                                        hipMemcpyDeviceToHost, st));
        }
        dbg_lap("all stages enqueued");
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess)
            return bail(ctx->fail(PGR_ERR_DEVICE, "pipeline failed on the device"));
        dbg_lap("synchronized");
        // ---- the one look at the device-side counts
        const uint64_t l1_alloc = mbox[0], l1_ovf = mbox[1], need_islands = mbox[2], l2_alloc = mbox[4], l2_ovf = mbox[5];
        const uint64_t total1 = mbox[8];
        n_final = mbox[9];
        l1_alloc_seen = l1_alloc;
        if (l1_ovf || l1_alloc > cap_par) {  // (only possible without the early synchronization)
            cap_par = (uint64_t)((double)l1_alloc * 1.1) + 65536;
            from = 1;
            continue;
        }
        if (!islands_done && (need_islands || !serial.empty())) {
            if ((rc = run_islands(need_islands))) return bail(rc);
            PGR_HIP_BAIL(hipEventRecord(ctx->ev[3], st));
            from = 2;
            continue;
        }
        prof.n_level1 = total1;
        if (total1 > (uint64_t)n_blocks * FUSED_BLOCK_ELEMS) {  // denser than the estimate: the grid missed the tail
            n_blocks = (uint32_t)((total1 + FUSED_BLOCK_ELEMS - 1) / FUSED_BLOCK_ELEMS);
            cap2 = std::max<uint64_t>(cap2, (uint64_t)((double)total1 * 0.01) + 65536);
            from = 3;
            continue;
        }
        if (l2_ovf || l2_alloc > cap2) {
            cap2 = (uint64_t)((double)l2_alloc * 1.05) + 65536;
            from = 3;
            continue;
        }
        if (n_final > cap_res) {  // more survivors than estimated: bigger result buffer, gather again
            cap_res = n_final + 16;
            from = 4;
            continue;
        }
        break;
    }
    memcpy(res->h_off.data(), mbox + N_STATUS, ((size_t)n + 1) * sizeof(uint64_t));
    res->count = n_final;
    if (host_copy_elems >= n_final && host_copy_elems) res->host_copy = (const pgr_mm128 *)ctx->pinned_out;
    ctx->want_host_copy = false;
    res->rid_is_index = (d_rids == nullptr) && !pad_fix;
    if (b->total_bases) {
        ctx->est_spec_key = spec_key;
        ctx->est_final_ratio = (double)n_final / (double)b->total_bases;
    }
    if (bases_tiled) {
        ctx->est_l1_key = l1_key;
        ctx->est_ovf_ratio = (double)l1_alloc_seen / (double)bases_tiled;
    }
    if (pad_fix) {
        // reference artefact: reduce_shmmr on an EMPTY list with padding emits its sentinels
        // (shmmrutils.rs:367-380), which survive as exactly two {MAX,MAX} after the second pass + filter
        std::vector<uint64_t> final_off((size_t)n + 1);
        uint64_t add = 0;
        for (uint32_t c = 0; c < n; ++c) {
            final_off[c] = res->h_off[c] + add;
            if (l1_off[c + 1] == l1_off[c]) add += 2;
        }
        final_off[n] = res->h_off[n] + add;
        res->count = final_off[n];
        if ((rc = ctx->dmalloc((void **)&res->d_mm, std::max<uint64_t>(res->count, 1) * sizeof(pgr_mm128)))) return bail(rc);
        if (hipMemcpyAsync(res->d_off, final_off.data(), ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice,
                           st) != hipSuccess)
            return bail(ctx->fail(PGR_ERR_DEVICE, "H2D of the result offsets failed"));
        launch_copy_or_sentinel(st, d_list, d_loff, res->d_off, n, res->d_mm);
        res->h_off = final_off;
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess)
            return bail(ctx->fail(PGR_ERR_DEVICE, "pipeline failed on the device"));
    }
#undef PGR_HIP_BAIL
    hipEvent_t ev_end = ctx->ev_end;  // recorded at the end of stage 4, complete since the synchronization above
    ctx->staged_unsynced = false;
    (void)hipEventElapsedTime(&prof.level1_ms, ctx->ev[1], ctx->ev[2]);
    (void)hipEventElapsedTime(&prof.level1_aux_ms, ctx->ev[2], ctx->ev[3]);
    (void)hipEventElapsedTime(&prof.level2_ms, ctx->ev[3], ev_end);
    (void)hipEventElapsedTime(&prof.total_ms, ctx->ev[0], ev_end);
    ctx->prof = prof;
    *out = res;
    dbg_lap("done");
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
    PGR_HIP(ctx, hipSetDevice(ctx->device));
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
    PGR_HIP(ctx, hipSetDevice(ctx->device));
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
    PGR_HIP(ctx, hipSetDevice(ctx->device));
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
    PGR_HIP(ctx, hipSetDevice(ctx->device));
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
    PGR_HIP(ctx, hipSetDevice(ctx->device));
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
    PGR_HIP(ctx, hipSetDevice(ctx->device));
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
    std::thread stager([&]() {
        (void)hipSetDevice(ctx->device);
        for (size_t i = 0; i < subs.size() && !cancel.load(); ++i) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return n_alloc > i || cancel.load(); });
                if (n_alloc <= i) return;
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
    PGR_HIP(ctx, hipSetDevice(ctx->device));
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
