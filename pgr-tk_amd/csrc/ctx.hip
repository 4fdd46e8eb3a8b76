// ctx.hip -- context memory management (workspaces, pinned staging, caching device allocator).
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <thread>
#include <vector>

#include "pgr_ctx.h"

namespace pgr {

int DevBuf::ensure(pgr_ctx *ctx, size_t bytes) {
    if (bytes <= cap && p) return PGR_OK;
    if (p) {
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    // grow with head-room so that repeated calls with slightly different sizes do not reallocate
    size_t want = std::max<size_t>(bytes + bytes / 8, 256);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        want = std::max<size_t>(bytes, 256);
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        p = nullptr;
        return ctx->fail(PGR_ERR_NOMEM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    }
    cap = want;
    return PGR_OK;
}

int DevBuf::ensure_keep(pgr_ctx *ctx, size_t bytes, hipStream_t st) {
    if (bytes <= cap && p) return PGR_OK;
    void *np = nullptr;
    const size_t want = std::max<size_t>(bytes + bytes / 8, 256);
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess)
        return ctx->fail(PGR_ERR_NOMEM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    if (p && cap) {
        e = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            (void)hipFree(np);
            return ctx->fail(PGR_ERR_DEVICE, std::string("workspace grow copy: ") + hipGetErrorString(e));
        }
        (void)hipFree(p);
    }
    p = np;
    cap = want;
    return PGR_OK;
}

void DevBuf::release(pgr_ctx *) {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

}  // namespace pgr

int pgr_ctx::dmalloc(void **out, size_t bytes) {
    *out = nullptr;
    bytes = std::max<size_t>((bytes + 255) & ~(size_t)255, 256);
    // best fit within 1.5x from the cache
    auto it = free_blocks.lower_bound(bytes);
    if (it != free_blocks.end() && it->first <= bytes + bytes / 2 + 4096) {
        *out = it->second;
        live_blocks[it->second] = it->first;
        cached_bytes -= it->first;
        free_blocks.erase(it);
        return PGR_OK;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess && !free_blocks.empty()) {  // drop the cache and retry
        for (auto &kv : free_blocks) (void)hipFree(kv.second);
        free_blocks.clear();
        cached_bytes = 0;
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess)
        return fail(PGR_ERR_NOMEM, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    live_blocks[p] = bytes;
    *out = p;
    return PGR_OK;
}

void pgr_ctx::dfree(void *p) {
    if (!p) return;
    auto it = live_blocks.find(p);
    if (it == live_blocks.end()) {
        (void)hipFree(p);
        return;
    }
    const size_t bytes = it->second;
    live_blocks.erase(it);
    // keep at most 160 GiB cached (288 GB of HBM3E per GPU; a failing hipMalloc drops the cache and retries).  Beyond that
    // the smallest cached blocks make room: hipFree synchronizes the device, so a full cache that frees every incoming
    // block (round 2) turned a streaming index build into a chain of device-wide stalls
    const size_t CAP = 160ull << 30;
    if (bytes > CAP) {
        (void)hipFree(p);
        return;
    }
    while (cached_bytes + bytes > CAP && !free_blocks.empty()) {
        auto it2 = free_blocks.begin();
        (void)hipFree(it2->second);
        cached_bytes -= it2->first;
        free_blocks.erase(it2);
    }
    free_blocks.emplace(bytes, p);
    cached_bytes += bytes;
}

int pgr_ctx::ensure_pinned(size_t bytes) {
    if (bytes <= pinned_cap && pinned) return PGR_OK;
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr;
    pinned_cap = 0;
    hipError_t e = hipHostMalloc(&pinned, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        pinned = nullptr;
        return fail(PGR_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
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
                           &ws_blk_base, &ws_start_rank, &ws_rids,       &ws_rec_off,    &ws_blk_off,    &ws_tile_desc,  &ws_tile_flags, &ws_seg_cid, &ws_tile_lv, &ws_small_desc, &ws_small_cnt};
    for (auto *b : bufs) b->release(this);
    for (auto &kv : free_blocks) (void)hipFree(kv.second);
    free_blocks.clear();
    for (auto &kv : live_blocks) (void)hipFree(kv.first);
    live_blocks.clear();
    cached_bytes = 0;
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
