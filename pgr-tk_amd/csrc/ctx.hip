// ctx.hip -- context memory management (workspaces, pinned staging, caching device allocator).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <thread>
#include <vector>

#include "pgr_ctx.h"

namespace pgr {

int DevBuf::ensure(pgr_ctx *ctx, size_t bytes, std::string *err) {
    if (bytes <= cap && p) return PGR_OK;
    if (p) {
        ctx->raw_free(p);
        p = nullptr;
        cap = 0;
    }
    // grow with head-room so that repeated calls with slightly different sizes do not reallocate
    size_t want = std::max<size_t>(bytes + bytes / 8, 256);
    hipError_t e = ctx->raw_alloc(&p, want);
    if (e != hipSuccess) {
        want = std::max<size_t>(bytes, 256);
        e = ctx->raw_alloc(&p, want);
    }
    if (e != hipSuccess) {
        p = nullptr;
        const std::string msg = std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e);
        if (err) {
            *err = msg;
            return PGR_ERR_NOMEM;
        }
        return ctx->fail(PGR_ERR_NOMEM, msg);
    }
    cap = want;
    if (ctx->opt.debug_poison) {  // a fresh block is nobody's yet: fill it and wait (debugging aid, not a fast path)
        (void)hipMemsetAsync(p, 0xFF, cap, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
    }
    return PGR_OK;
}

int DevBuf::ensure_keep(pgr_ctx *ctx, size_t bytes, hipStream_t st) {
    if (bytes <= cap && p) return PGR_OK;
    void *np = nullptr;
    const size_t want = std::max<size_t>(bytes + bytes / 8, 256);
    hipError_t e = ctx->raw_alloc(&np, want);
    if (e != hipSuccess)
        return ctx->fail(PGR_ERR_NOMEM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    if (p && cap) {
        e = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            ctx->raw_free(np);
            return ctx->fail(PGR_ERR_DEVICE, std::string("workspace grow copy: ") + hipGetErrorString(e));
        }
        ctx->raw_free(p);
    }
    p = np;
    cap = want;
    return PGR_OK;
}

void DevBuf::release(pgr_ctx *ctx) {
    if (p) ctx->raw_free(p);
    p = nullptr;
    cap = 0;
}

}  // namespace pgr

hipEvent_t pgr_ctx::take_event() {
    if (!ev_pool.empty()) {
        hipEvent_t e = ev_pool.back();
        ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}

// the block goes to work on `user`: that stream waits for whatever the OTHER stream had queued when the block was freed
void pgr_ctx::wait_and_recycle(FreeBlock &fb, hipStream_t user) {
    if (fb.ev_front) {
        if (user != stream) (void)hipStreamWaitEvent(user, fb.ev_front, 0);
        ev_pool.push_back(fb.ev_front);  // (a wait that is already queued keeps the record it saw: the event can be recorded again)
        fb.ev_front = nullptr;
    }
    if (fb.ev_back) {
        if (user != back_stream) (void)hipStreamWaitEvent(user, fb.ev_back, 0);
        ev_pool.push_back(fb.ev_back);
        fb.ev_back = nullptr;
    }
    if (fb.ev_fix) {
        if (user != fix_stream) (void)hipStreamWaitEvent(user, fb.ev_fix, 0);
        ev_pool.push_back(fb.ev_fix);
        fb.ev_fix = nullptr;
    }
}

void pgr_ctx::drop_events(FreeBlock &fb) {
    if (fb.ev_front) ev_pool.push_back(fb.ev_front);
    if (fb.ev_back) ev_pool.push_back(fb.ev_back);
    if (fb.ev_fix) ev_pool.push_back(fb.ev_fix);
    fb.ev_front = fb.ev_back = fb.ev_fix = nullptr;
}

namespace {
// one workgroup that stays on the device for ~`ticks` of the shader clock's real-time counter (100 MHz): long enough for the host
// to put a second kernel on another stream beside it
__global__ void pgr_linger_kernel(unsigned long long ticks, unsigned long long *out) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t = t0;
    while (t - t0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
        t = __builtin_readcyclecounter();
    }
    if (out) *out = t - t0;
}
__global__ void pgr_touch_kernel(unsigned long long *out) {
    if (out) *out = 1;
}
}  // namespace

// The runtime multiplexes its streams onto a handful of hardware queues (round robin, per priority class); two streams that
// share one are in order with each other, whatever the API says -- the back stream of a pipe would then run its list stage
// BEHIND the next batch's tiles instead of beside them (measured: 21.5 instead of 20.6 ms per batch).  Which queue a new stream
// gets depends on how many streams the process has created before.  So: try it.  A kernel that lingers ~0.4 ms on `a`, a trivial
// one on `b` right after it: if `b`'s is done while `a`'s is still there, the two streams do not share a queue.
bool pgr::streams_run_side_by_side(hipStream_t a, hipStream_t b, unsigned long long *d_scratch) {
    hipEvent_t ea = nullptr, eb = nullptr;
    if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) {
        (void)hipEventDestroy(ea);
        return false;
    }
    bool side_by_side = false;
    for (int attempt = 0; attempt < 2 && !side_by_side; ++attempt) {
        hipLaunchKernelGGL(pgr_linger_kernel, dim3(1), dim3(64), 0, a, 40000ull << attempt, d_scratch);  // 0.4 ms, then 0.8 ms
        (void)hipEventRecord(ea, a);
        hipLaunchKernelGGL(pgr_touch_kernel, dim3(1), dim3(64), 0, b, d_scratch + 1);
        (void)hipEventRecord(eb, b);
        (void)hipEventSynchronize(eb);
        side_by_side = hipEventQuery(ea) == hipErrorNotReady;
        (void)hipEventSynchronize(ea);
    }
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    return side_by_side;
}

int pgr_ctx::enable_multi_stream() {
    if (multi_stream) return PGR_OK;
    if (!back_stream) {
        int lo = 0, hi = 0;  // (numerically lower = higher priority)
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const int prio = opt.back_priority > 0 ? hi : opt.back_priority < 0 ? lo : 0;
        unsigned long long *d_scratch = nullptr;
        if (hipMalloc((void **)&d_scratch, 64) != hipSuccess) return fail(PGR_ERR_NOMEM, "hipMalloc failed");
        std::vector<hipStream_t> rejected;
        hipError_t e = hipSuccess;
        for (int tries = 0; tries < 8; ++tries) {
            hipStream_t cand = nullptr;
            e = hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, prio);
            if (e != hipSuccess) break;
            if (pgr::streams_run_side_by_side(stream, cand, d_scratch)) {
                back_stream = cand;
                break;
            }
            rejected.push_back(cand);
        }
        back_shares_queue = back_stream == nullptr;
        if (!back_stream && !rejected.empty()) {  // no luck: the pipe still works, its list stages run in order with the tiles
            back_stream = rejected.back();
            rejected.pop_back();
        }
        for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
        rejected.clear();
        // the stream a job's second pass runs on: beside the context's stream AND beside the back stream
        if (!back_shares_queue && !fix_stream) {
            for (int tries = 0; tries < 8; ++tries) {
                hipStream_t cand = nullptr;
                if (hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, prio) != hipSuccess) break;
                if (pgr::streams_run_side_by_side(stream, cand, d_scratch) && pgr::streams_run_side_by_side(back_stream, cand, d_scratch)) {
                    fix_stream = cand;
                    break;
                }
                rejected.push_back(cand);
            }
            for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
            if (opt.debug) fprintf(stderr, "[pgr] fix stream: %s (%zu candidates shared a hardware queue with one of the two others)\n",
                                   fix_stream ? "found" : "none", rejected.size());
        }
        (void)hipFree(d_scratch);
        if (opt.debug) fprintf(stderr, "[pgr] back stream: priority %d, %zu candidates shared a hardware queue with the context's stream%s\n", prio,
                               rejected.size() + (back_shares_queue ? 1 : 0), back_shares_queue ? " -- none did not" : "");
        if (!back_stream)
            return fail(PGR_ERR_DEVICE, std::string("hipStreamCreateWithPriority: ") + hipGetErrorString(e));
    }
    // every block freed so far was freed under the one-stream rule: make that true for two streams by waiting once
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(PGR_ERR_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    multi_stream = true;
    return PGR_OK;
}

// ---- the arena (pgr_ctx.h)
int pgr_ctx::reserve(size_t bytes) {
    if (bytes == 0) return PGR_OK;
    bytes = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    Arena a;
    hipError_t e = hipMalloc((void **)&a.base, bytes);
    if (e != hipSuccess) return fail(PGR_ERR_NOMEM, std::string("pgr_ctx_reserve: hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    // touched once, here: the first use of untouched device memory is the other half of a cold start
    e = hipMemsetAsync(a.base, opt.debug_poison ? 0xFF : 0, bytes, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        (void)hipFree(a.base);
        return fail(PGR_ERR_DEVICE, std::string("pgr_ctx_reserve: first touch: ") + hipGetErrorString(e));
    }
    a.list.reset(bytes);
    arenas.push_back(std::move(a));
    arena_bytes += bytes;
    return PGR_OK;
}

hipError_t pgr_ctx::raw_alloc(void **out, size_t bytes) {
    *out = nullptr;
    const size_t want = (std::max<size_t>(bytes, 1) + 4095) & ~(size_t)4095;
    for (size_t ai = 0; ai < arenas.size(); ++ai) {
        const size_t off = arenas[ai].list.take(want);
        if (off == pgr::ArenaList::NONE) continue;
        *out = arenas[ai].base + off;
        arena_live[*out] = {(int)ai, want};
        arena_used += want;
        arena_peak = std::max(arena_peak, arena_used);
        return hipSuccess;
    }
    const hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess && !arenas.empty()) {
        fallback_bytes += bytes;
        ++fallback_calls;
        if (opt.debug) fprintf(stderr, "[pgr] arena exhausted: %zu bytes from hipMalloc (%zu of %zu arena bytes in use)\n", bytes, arena_used, arena_bytes);
    }
    return e;
}

void pgr_ctx::raw_free(void *p) {
    if (!p) return;
    auto it = arena_live.find(p);
    if (it == arena_live.end()) {
        (void)hipFree(p);
        return;
    }
    // what hipFree guarantees, callers rely on (a grown workspace, a dropped cache): nothing of the block's past is still running
    (void)hipDeviceSynchronize();
    Arena &a = arenas[(size_t)it->second.first];
    a.list.give_back((size_t)((char *)p - a.base), it->second.second);
    arena_used -= it->second.second;
    arena_live.erase(it);
}

int pgr_ctx::dmalloc(void **out, size_t bytes) {
    *out = nullptr;
    bytes = std::max<size_t>((bytes + 255) & ~(size_t)255, 256);
    hipStream_t user = alloc_stream ? alloc_stream : stream;
    // best fit within 1.5x from the cache; with two streams in play, a block that was last used on the requesting stream is
    // taken first (no wait), any other only when no such block fits
    {
        const bool want_back = multi_stream && user == back_stream;
        auto hit = free_blocks.end();
        for (auto it = free_blocks.lower_bound(bytes); it != free_blocks.end() && it->first <= bytes + bytes / 2 + 4096; ++it) {
            if (!multi_stream || it->second.on_back == want_back) {
                hit = it;
                break;
            }
            // (the back stream may wait for the context's stream -- it is behind it anyway --, the context's stream does not wait
            // for a list stage: a fresh block instead)
            if (hit == free_blocks.end() && want_back) hit = it;
        }
        if (hit != free_blocks.end()) {
            wait_and_recycle(hit->second, user);
            // (on the stream the block goes to work on, behind everything that stream was made to wait for: whoever still reads the
            // block on a stream the allocator was not told about reads 0xFF from here on)
            if (opt.debug_poison) (void)hipMemsetAsync(hit->second.p, 0xFF, hit->first, user);
            *out = hit->second.p;
            live_blocks[hit->second.p] = LiveBlock{hit->first, want_back, multi_stream && fix_stream && user == fix_stream};
            cached_bytes -= hit->first;
            free_blocks.erase(hit);
            return PGR_OK;
        }
    }
    void *p = nullptr;
    hipError_t e = raw_alloc(&p, bytes);
    if (e != hipSuccess && !free_blocks.empty()) {  // drop the cache and retry
        for (auto &kv : free_blocks) {
            raw_free(kv.second.p);  // (synchronizes the device: nothing of a freed block's past is still running behind it)
            drop_events(kv.second);
            live_bytes -= kv.first;
        }
        free_blocks.clear();
        cached_bytes = 0;
        e = raw_alloc(&p, bytes);
    }
    if (e != hipSuccess)
        return fail(PGR_ERR_NOMEM, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    live_blocks[p] = LiveBlock{bytes, multi_stream && user == back_stream, multi_stream && fix_stream && user == fix_stream};
    live_bytes += bytes;
    peak_bytes = std::max(peak_bytes, live_bytes);
    if (opt.debug_poison) (void)hipMemsetAsync(p, 0xFF, bytes, user);
    *out = p;
    return PGR_OK;
}

void pgr_ctx::dfree(void *p) {
    if (!p) return;
    auto it = live_blocks.find(p);
    if (it == live_blocks.end()) {
        raw_free(p);
        return;
    }
    const size_t bytes = it->second.bytes;
    const bool on_back = it->second.on_back, on_fix = it->second.on_fix;
    live_blocks.erase(it);
    // keep at most 160 GiB cached (288 GB of HBM3E per GPU; a failing hipMalloc drops the cache and retries).  Beyond that
    // the smallest cached blocks make room: hipFree synchronizes the device, so a full cache that frees every incoming
    // block (round 2) turned a streaming index build into a chain of device-wide stalls
    const size_t CAP = 160ull << 30;
    if (bytes > CAP) {
        raw_free(p);
        live_bytes -= bytes;
        return;
    }
    while (cached_bytes + bytes > CAP && !free_blocks.empty()) {
        auto it2 = free_blocks.begin();
        raw_free(it2->second.p);
        drop_events(it2->second);
        cached_bytes -= it2->first;
        live_bytes -= it2->first;
        free_blocks.erase(it2);
    }
    FreeBlock fb;
    fb.p = p;
    fb.on_back = on_back;
    if (multi_stream) {
        // Where the streams that worked on the block stand now: whoever takes it for ANOTHER stream waits for that.
        //   the context's stream: always -- except for a block freed inside a job's second pass on the fix stream (pgr_pipe_collect):
        //     what is pending on it is pending on the fix stream (the first pass has been waited for by the host), and the context's
        //     stream is busy with the NEXT job's tiles: an event recorded there would make this pass's next allocation, which is likely
        //     to get this very block, wait for them;
        //   the back stream: blocks handed out for it, or marked (an index's records);
        //   the fix stream: blocks handed out for it or marked, and every block freed inside a second pass.
        const bool in_fix_pass = fix_stream && alloc_stream == fix_stream;
        bool ok = true;
        auto mark = [&](hipEvent_t &ev, hipStream_t st) {
            if (!ok) return;
            ev = take_event();
            ok = ev && hipEventRecord(ev, st) == hipSuccess;
        };
        if (!in_fix_pass) mark(fb.ev_front, stream);
        if (on_back && back_stream) mark(fb.ev_back, back_stream);
        if ((on_fix || in_fix_pass) && fix_stream) mark(fb.ev_fix, fix_stream);
        if (!ok) {  // no event to be had: the slow, safe way
            (void)hipStreamSynchronize(stream);
            if (back_stream) (void)hipStreamSynchronize(back_stream);
            if (fix_stream) (void)hipStreamSynchronize(fix_stream);
            drop_events(fb);
        }
    }
    free_blocks.emplace(bytes, fb);
    cached_bytes += bytes;
}

void pgr_ctx::block_on_back(void *p) {
    auto it = live_blocks.find(p);
    if (it != live_blocks.end()) it->second.on_back = true;
}

void pgr_ctx::block_on_fix(void *p) {
    auto it = live_blocks.find(p);
    if (it != live_blocks.end()) it->second.on_fix = true;
}

int pgr_ctx::poison_workspaces(hipStream_t st) {
    pgr::DevBuf *bufs[] = {&ws_tile_first, &ws_seg_off, &ws_seg_cnt, &ws_seg_dst, &ws_cursor, &ws_flags, &ws_l1, &ws_serial, &ws_scan_tmp,
                           &ws_list_a, &ws_list_b, &ws_off_a, &ws_off_b, &ws_blk_cnt, &ws_blk_base, &ws_start_rank, &ws_rids, &ws_rec_off,
                           &ws_blk_off, &ws_tile_desc, &ws_tile_flags, &ws_seg_cid, &ws_tile_lv, &ws_recs};
    for (pgr::DevBuf *b : bufs)
        if (b->p && b->cap && hipMemsetAsync(b->p, 0xFF, b->cap, st) != hipSuccess) return fail(PGR_ERR_DEVICE, "debug_poison: memset failed");
    return PGR_OK;
}

void pgr_ctx::swap_lane(pgr::Lane &l) {
    std::swap(ws_tile_first, l.ws_tile_first);
    std::swap(ws_seg_off, l.ws_seg_off);
    std::swap(ws_seg_cnt, l.ws_seg_cnt);
    std::swap(ws_seg_dst, l.ws_seg_dst);
    std::swap(ws_cursor, l.ws_cursor);
    std::swap(ws_flags, l.ws_flags);
    std::swap(ws_l1, l.ws_l1);
    std::swap(ws_serial, l.ws_serial);
    std::swap(ws_scan_tmp, l.ws_scan_tmp);
    std::swap(ws_list_a, l.ws_list_a);
    std::swap(ws_list_b, l.ws_list_b);
    std::swap(ws_off_a, l.ws_off_a);
    std::swap(ws_off_b, l.ws_off_b);
    std::swap(ws_blk_cnt, l.ws_blk_cnt);
    std::swap(ws_blk_base, l.ws_blk_base);
    std::swap(ws_start_rank, l.ws_start_rank);
    std::swap(ws_rids, l.ws_rids);
    std::swap(ws_rec_off, l.ws_rec_off);
    std::swap(ws_blk_off, l.ws_blk_off);
    std::swap(ws_tile_desc, l.ws_tile_desc);
    std::swap(ws_tile_flags, l.ws_tile_flags);
    std::swap(ws_seg_cid, l.ws_seg_cid);
    std::swap(ws_tile_lv, l.ws_tile_lv);
    std::swap(ws_recs, l.ws_recs);
    std::swap(mailbox, l.mailbox);
    std::swap(mailbox_cap, l.mailbox_cap);
    for (int i = 0; i < 4; ++i) std::swap(ev[i], l.ev[i]);
    std::swap(ev_end, l.ev_end);
    h_tile_first.swap(l.h_tile_first);
    keep_rec_off.swap(l.keep_rec_off);
}

int pgr_ctx::ensure_pinned(size_t bytes, std::string *err_out) {
    if (bytes <= pinned_cap && pinned) return PGR_OK;
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr;
    pinned_cap = 0;
    hipError_t e = hipHostMalloc(&pinned, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        pinned = nullptr;
        const std::string msg = std::string("hipHostMalloc: ") + hipGetErrorString(e);
        if (err_out) {
            *err_out = msg;
            return PGR_ERR_NOMEM;
        }
        return fail(PGR_ERR_NOMEM, msg);
    }
    pinned_cap = bytes;
    return PGR_OK;
}

int pgr_ctx::ensure_pinned_out(size_t bytes) {
    if (bytes <= pinned_out_cap && pinned_out) return PGR_OK;
    if (pinned_out) (void)hipHostFree(pinned_out);
    pinned_out = nullptr;
    pinned_out_cap = 0;
    hipError_t e = hipHostMalloc(&pinned_out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        pinned_out = nullptr;
        return fail(PGR_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    pinned_out_cap = bytes;
    return PGR_OK;
}

int pgr_ctx::ensure_qmail() {
    if (qmail) return PGR_OK;
    if (hipHostMalloc(&qmail, 256, hipHostMallocDefault) != hipSuccess) {
        qmail = nullptr;
        return fail(PGR_ERR_NOMEM, "hipHostMalloc of the query mailbox failed");
    }
    memset(qmail, 0, 256);
    return PGR_OK;
}

int pgr_ctx::ensure_mailbox(size_t bytes) {
    if (bytes <= mailbox_cap && mailbox) return PGR_OK;
    if (mailbox) (void)hipHostFree(mailbox);
    mailbox = nullptr;
    mailbox_cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    hipError_t e = hipHostMalloc(&mailbox, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        mailbox = nullptr;
        return fail(PGR_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    mailbox_cap = want;
    return PGR_OK;
}

int pgr_ctx::ensure_imail(size_t bytes) {
    if (bytes <= imail_cap && imail) return PGR_OK;
    if (imail) (void)hipHostFree(imail);
    imail = nullptr;
    imail_cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 1u << 16);
    hipError_t e = hipHostMalloc(&imail, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        imail = nullptr;
        return fail(PGR_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    imail_cap = want;
    return PGR_OK;
}

int pgr_ctx::d2h(void *dst, const void *src_dev, size_t bytes) {
    if (bytes == 0) return PGR_OK;
    if (bytes < (4u << 20)) {  // stream ordered like the pipelined path (the context's stream is non-blocking)
        hipError_t e = hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        return e == hipSuccess ? PGR_OK : fail(PGR_ERR_DEVICE, std::string("D2H copy: ") + hipGetErrorString(e));
    }
    const size_t WIN = 16u << 20;
    int rc = ensure_pinned_out(2 * WIN);
    if (rc) return rc;
    // the destination is usually fresh malloc'd memory: its first touch (page faults) is spread over the pool's threads
    auto host_copy = [&](uint8_t *d, const uint8_t *s, size_t len) {
        constexpr size_t PIECE = 1u << 20;
        if (len < 2 * PIECE) {
            memcpy(d, s, len);
            return;
        }
        pgr::HostPool::instance().parallel_for((len + PIECE - 1) / PIECE, [&](size_t i) {
            const size_t o = i * PIECE;
            memcpy(d + o, s + o, std::min(PIECE, len - o));
        });
    };
    const uint8_t *sd = (const uint8_t *)src_dev;
    uint8_t *dh = (uint8_t *)dst, *pin = (uint8_t *)pinned_out;
    size_t issued = 0, copied = 0;
    int slot = 0;
    hipError_t e = hipSuccess;
    // prime window 0, then: issue window i+1, wait for window i, copy it out
    auto issue = [&](int sl) {
        const size_t len = std::min(WIN, bytes - issued);
        e = hipMemcpyAsync(pin + (size_t)sl * WIN, sd + issued, len, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipEventRecord(ev[2 + sl], stream);
        issued += len;
    };
    issue(0);
    while (e == hipSuccess && copied < bytes) {
        const int cur = slot;
        slot ^= 1;
        if (issued < bytes) issue(slot);
        if (e != hipSuccess) break;
        e = hipEventSynchronize(ev[2 + cur]);
        if (e != hipSuccess) break;
        const size_t len = std::min(WIN, bytes - copied);
        host_copy(dh + copied, pin + (size_t)cur * WIN, len);
        copied += len;
    }
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(stream);
        return fail(PGR_ERR_DEVICE, std::string("D2H copy: ") + hipGetErrorString(e));
    }
    return PGR_OK;
}

void pgr_ctx::release_all() {
    pgr::DevBuf *bufs[] = {&ws_ascii,    &ws_tile_first, &ws_seg_off,    &ws_seg_cnt, &ws_seg_dst,
                           &ws_cursor,   &ws_flags,      &ws_l1,         &ws_serial,  &ws_scan_tmp,
                           &ws_list_a,   &ws_list_b,     &ws_off_a,      &ws_off_b,   &ws_blk_cnt,
                           &ws_blk_base, &ws_start_rank, &ws_rids,       &ws_rec_off,    &ws_blk_off,    &ws_tile_desc,  &ws_tile_flags, &ws_seg_cid, &ws_tile_lv, &ws_small_desc, &ws_small_cnt, &ws_recs};
    for (auto *b : bufs) b->release(this);
    for (pgr::Lane *l : spare_lanes) {
        pgr::lane_release(this, *l);
        delete l;
    }
    spare_lanes.clear();
    for (auto &kv : free_blocks) {
        raw_free(kv.second.p);
        drop_events(kv.second);
    }
    free_blocks.clear();
    for (auto &kv : live_blocks) raw_free(kv.first);
    live_blocks.clear();
    cached_bytes = 0;
    live_bytes = 0;
    (void)hipDeviceSynchronize();
    for (Arena &a : arenas)
        if (a.base) (void)hipFree(a.base);
    arenas.clear();
    arena_live.clear();
    arena_bytes = arena_used = 0;
    for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    ev_pool.clear();
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr;
    pinned_cap = 0;
    if (pinned_out) (void)hipHostFree(pinned_out);
    pinned_out = nullptr;
    pinned_out_cap = 0;
    if (mailbox) (void)hipHostFree(mailbox);
    mailbox = nullptr;
    mailbox_cap = 0;
    if (qmail) (void)hipHostFree(qmail);
    qmail = nullptr;
    if (imail) (void)hipHostFree(imail);
    imail = nullptr;
    imail_cap = 0;
}

// ------------------------------------------------------------------------------------------------
// Pinned host blocks for results a kernel writes DIRECTLY into host memory (query_fused.hip): the block the library hands to
// the caller IS the memory the GPU wrote over PCIe -- no staging copy, no second synchronization.  hipHostMalloc costs about a
// millisecond, so released blocks are kept (process wide: a result may outlive its context) up to a cap.
namespace {
struct PinnedPool {
    std::mutex mu;
    std::multimap<size_t, void *> free_by_cap;
    std::unordered_map<void *, size_t> live;  // blocks handed out
    size_t cached = 0;
    static constexpr size_t CACHE_CAP = 1ull << 30;
    static PinnedPool &instance() {
        static PinnedPool *p = new PinnedPool();  // never destroyed: results may be released during process teardown
        return *p;
    }
};
}  // namespace

void *pgr::pinned_result_acquire(size_t min_bytes, size_t *cap) {
    PinnedPool &P = PinnedPool::instance();
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.free_by_cap.lower_bound(min_bytes);
        if (it != P.free_by_cap.end() && it->first <= 4 * std::max<size_t>(min_bytes, 1u << 20)) {
            void *p = it->second;
            *cap = it->first;
            P.cached -= it->first;
            P.free_by_cap.erase(it);
            P.live[p] = *cap;
            return p;
        }
    }
    size_t want = 1u << 20;
    while (want < min_bytes) want <<= 1;
    void *p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> g(P.mu);
    P.live[p] = want;
    *cap = want;
    return p;
}

void pgr::result_block_release(void *p) {
    if (!p) return;
    PinnedPool &P = PinnedPool::instance();
    size_t cap = 0;
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.live.find(p);
        if (it == P.live.end()) {
            cap = 0;
        } else {
            cap = it->second;
            P.live.erase(it);
            if (P.cached + cap <= PinnedPool::CACHE_CAP) {
                P.free_by_cap.emplace(cap, p);
                P.cached += cap;
                return;
            }
        }
    }
    if (cap) (void)hipHostFree(p);
    else free(p);  // an ordinary host block (host_result_alloc)
}
