// pgr_ctx.h -- the context object behind the C ABI: one GPU, one stream, grow-only workspaces and a
// small caching device allocator (so the steady state of repeated calls does no hipMalloc/hipFree).
#pragma once
#include "arena_list.h"
#include <algorithm>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "pgr_internal.h"
#include "pgr_host.h"
#include "pgr_small.h"

namespace pgr {
// true when a kernel on b runs while a kernel on a is still there: the two streams do not share a hardware queue (ctx.hip)
bool streams_run_side_by_side(hipStream_t a, hipStream_t b, unsigned long long *d_scratch);
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(pgr_ctx *ctx, size_t bytes, std::string *err = nullptr);  // contents NOT preserved; err != NULL: the message goes there, not into the context
    int ensure_keep(pgr_ctx *ctx, size_t bytes, hipStream_t st);   // contents preserved
    void release(pgr_ctx *ctx);
};
// Everything one pass of the shimmer pipeline keeps between its stages: the workspaces, the pinned mailbox its counts come
// back into, the events that time it, the host-side tables that are sources of asynchronous copies.  The context holds ONE
// such set in its own members (every synchronous call uses it); a pgr_pipe (csrc/pipeline.hip) owns two more and swaps the
// one of the job it is working on into the context's members for the duration of that work (pgr_ctx::swap_lane), so that
// two passes can be in flight -- the list stage of batch i on the back stream beside the tiles of batch i + 1.
struct Lane {
    DevBuf ws_tile_first, ws_seg_off, ws_seg_cnt, ws_seg_dst, ws_cursor, ws_flags, ws_l1, ws_serial, ws_scan_tmp, ws_list_a,
        ws_list_b, ws_off_a, ws_off_b, ws_blk_cnt, ws_blk_base, ws_start_rank, ws_rids, ws_rec_off, ws_blk_off, ws_tile_desc,
        ws_tile_flags, ws_seg_cid, ws_tile_lv, ws_recs;
    void *mailbox = nullptr;
    size_t mailbox_cap = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_end = nullptr;
    std::vector<uint32_t> h_tile_first;
    std::vector<uint64_t> keep_rec_off;
};
}  // namespace pgr

namespace pgr {
void drop_stale_hip_error(const char *who);  // (scan.hip) takes the thread's stale HIP error away; says so under PGR_DEBUG_STALE=1
}

struct pgr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_end = nullptr;
    hipEvent_t ev_alloc = nullptr;  // (unused since every batch has an event of its own: pgr_batch::ev_alloc)
    std::string err;
    pgr_prof prof = {};
    pgr_query_prof qprof = {};

    // pinned staging for H2D (owned by whoever stages a batch: the calling thread, or the staging thread of the
    // pipelined pgr_shmmr_batch) and, separately, for result downloads (d2h, calling thread)
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    void *pinned_out = nullptr;
    size_t pinned_out_cap = 0;
    // pinned mailbox: the handful of device-side counts (+ result offsets) a call reads after its one synchronization
    void *mailbox = nullptr;
    size_t mailbox_cap = 0;
    // Tuning and A/B switches.  Read from the environment ONCE, when the context is created (PGR_<NAME IN UPPER CASE>), and
    // changed afterwards only through pgr_ctx_set_option: no call looks at the environment.
    struct Options {
        int64_t debug = 0;               // progress lines of the staging / island / query paths on stderr
        int64_t debug_times = 0;         // host-side timeline of a call on stderr
        int64_t gpu_pack = 0;            // A/B: the round-2 host path (ASCII over PCIe, packed by a kernel)
        int64_t no_small_path = 0;       // never take the one-workgroup-per-contig kernel (csrc/small.hip)
        int64_t no_pipeline = 0;         // never cut a large host batch into staged sub-batches
        int64_t early_sync_bp = 1ll << 30;  // batches of at least this many bases look at the level-1 flags before the list stage
        int64_t index_full_sort = 0;     // pgr_index_finalize: always the four-field sort
        int64_t index_two_key_sort = 0;  // pgr_index_finalize: never the one-key sort + run fix-ups
        int64_t no_fused_query = 0;      // never the one-wavefront-per-query kernel (csrc/query_fused.hip)
        int64_t direct_query_lds_kb = 0;     // experiment: LDS per workgroup of the single-pass query kernel (fewer queries resident at once); -1: its block in device memory (timing only)
        int64_t direct_query_results_delivered = 0;  // (a counter, read with pgr_ctx_get_option: batches whose result the single-pass kernel wrote)
        int64_t direct_query_result = 0;     // experiment: that kernel writes the host's result block itself (single pass; measured slower)
        int64_t no_query_chaining = 0;   // do not enqueue the query stage behind the shimmer pipeline
        int64_t lut_extra_bits = 0;      // pgr_index_finalize: the bucket table over the keys gets 2^this times as many buckets (0 .. 4)
        int64_t no_query_keys = 0;       // pgr_index_finalize builds no per-query-kernel key table (pgr_index.h: qkeys), for A/B
        int64_t no_query_level1 = 0;     // query batches never take the level-1 form of the per-query kernel (tile kernel + per-query kernel, no list stage), for A/B
        int64_t query_global_sort = 0;   // group the hits of a batch with the global radix sort
        int64_t fused_query_hits = 0;    // > 0: fixed slot size H of the per-query kernel
        int64_t exchange_timeout_s = 300;  // bound on ncclCommInitRank (all ranks must arrive); 0 = wait for ever
        int64_t exchange_collective_timeout_s = 1800;  // bound on every wait for a collective -- which includes waiting for a SLOWER peer to
                                                       // get there (an imbalanced rank is not a dead one): generous; 0 = wait for ever
        int64_t debug_poison = 0;        // every device block that is about to be used again is filled with 0xFF first (blocks leaving the caching
                                         // allocator, grown workspaces, a job's workspaces when the job is planned), and a list stage checks the
                                         // segment table it is about to read: reading what an earlier call left behind fails the call
        int64_t debug_inject_stale_segments = 0;  // FAULT INJECTION for the tests of debug_poison: a pass without a tile kernel does not clear the
                                                  // segment counts (the round-5 defect, profiles/r05_fuzz/cursor_block_size_fault.txt)
        int64_t exchange_rccl_world1 = 0;  // an exchange of ONE rank goes through a real RCCL communicator all the same (default: plain copies, RCCL not loaded)
        int64_t no_island_relay = 0;     // exact islands: correct seams one per host round (the round-3 scheme), for A/B
        int64_t island_settle = 0;       // > 0: positions behind the last palindromic block at which a sub-tile island ends (default 2 w + k + 64, rounded up to 64)
        int64_t no_sub_tile_islands = 0; // exact islands of whole tiles only (rounds 3-5), for A/B: round 6 begins / ends them inside tiles
        int64_t island_chunk_min = 0;    // > 0: shortest chunk of the exact machine (positions; default 1024), for A/B
        int64_t early_islands_in_stream = 0;  // ... behind the tile kernel on its stream, not beside it on a stream of their own (for A/B)
        int64_t no_early_islands = 0;    // the islands around non-ACGT bytes never start their first round behind the tile kernel, before its flags are seen (for A/B)
        int64_t no_early_merge = 0;      // a tile that reports a palindromic k-mer drops the early round of the islands around non-ACGT bytes (the first form), instead of keeping it and adding the new islands (for A/B)
        int64_t no_pre_islands = 0;      // never list the islands around non-ACGT bytes while the tile kernel runs (for A/B)
        int64_t no_short_tiles = 0;      // batches of short contigs: the 4096-position tiles all the same, for A/B
        int64_t pipe_persistent_list = 0;  // pgr_pipe: > 0 = the list kernel of a pipelined job is this many persistent 512-element workgroups (about one per CU: beside the tiles, never in a tile's slot), for A/B
        int64_t pipe_small_list = 0;     // pgr_pipe: 1 = the list kernel runs 512-element workgroups (14 KB of LDS: they fit beside a CU's four tile workgroups), for A/B
        int64_t front_priority = 0;      // 1: the context's stream is created with the device's highest priority (read at pgr_ctx_create only), for A/B
        int64_t no_direct_h2d = 0;       // packed input in pinned host memory goes through the staging windows all the same, for A/B
        int64_t lds_match = 0;           // pgr_pipe: 1 = the back stream's kernels occupy the tile kernel's LDS size or none (padded list kernel, LDS-free scans), for A/B
        int64_t pipe_staged_records = 0; // pgr_pipe: never place an index job's records through the device cursor (always stage + copy), for A/B
        int64_t no_stage1_only = 0;      // pgr_pipe: the first pass of a job always includes its list stage, even when the last job needed islands, for A/B
        int64_t no_fix_stream = 0;       // pgr_pipe: a job that needs a second pass is finished on the context's stream (behind the next job's tiles), for A/B
        int64_t back_priority = 0;       // pgr_pipe: priority of the back stream (list stages): 1 = the device's highest, 0 = the default, -1 = the lowest
    } opt;
    std::vector<uint32_t> h_tile_first;  // pgr_shmmrs_compute: first tile of every contig (host copy, kept between calls)
    std::vector<uint64_t> spare_off;     // offsets block of the last destroyed big result (used again by the next one)
    bool skip_small_once = false;  // the host entry point's one-workgroup kernel handed the batch back: do not try it again
    bool want_host_copy = false;   // set by the host-buffer entry points: a small result rides along with the final round trip
    bool staged_unsynced = false;  // a batch was staged on `stream` and nobody has synchronized since
    // result-size estimate: final shimmers per base of the last pgr_shmmrs_compute with the spec `est_spec_key`
    double est_spec_key = -1.0, est_final_ratio = 0.0;
    // overflow-region need of the level-1 kernels per tiled base, last call with the spec `est_l1_key`
    double est_l1_key = -1.0, est_ovf_ratio = 0.0;
    // ... level-1 minimizers per base and the list kernel's overflow need per base of that call, and whether it needed islands
    double est_l1_dens = 0.0, est_l2_ovf = 0.0;
    bool est_flagged = false;
    std::vector<pgr::SmallContig> keep_small_desc;  // source of the async H2D copy of shmmrs_compute_small
    std::vector<uint64_t> keep_rec_off;  // source of the async H2D copy of shmmrs_to_frag_recs_enqueue
    // second stream + events: staging of sub-batch i+1 (H2D + pack) while sub-batch i computes on `stream`
    hipStream_t copy_stream = nullptr;
    hipEvent_t cev[2] = {nullptr, nullptr};
    // third stream + events: the result of sub-batch i goes to the host while sub-batch i + 1 computes (pipelined host calls)
    hipStream_t d2h_stream = nullptr;
    hipEvent_t d2h_ev[2] = {nullptr, nullptr};
    hipEvent_t pre_ev[2] = {nullptr, nullptr};  // pgr_shmmrs_compute: tile flags known / on the host, while the tiles still run
    hipStream_t pre_stream = nullptr;           // (their copy: a stream of its own, the download stream may be busy with a result)
    size_t d2h_slot_bytes = 0;  // half of pinned_out when the pipelined download uses it as two blocks

    // workspaces
    pgr::DevBuf ws_ascii, ws_tile_first, ws_seg_off, ws_seg_cnt, ws_seg_dst, ws_cursor, ws_flags, ws_l1, ws_serial,
        ws_scan_tmp, ws_list_a, ws_list_b, ws_off_a, ws_off_b, ws_blk_cnt, ws_blk_base, ws_start_rank, ws_rids,
        ws_rec_off, ws_blk_off, ws_tile_desc, ws_tile_flags, ws_seg_cid, ws_tile_lv, ws_small_desc, ws_small_cnt,
        ws_recs;  // (ws_recs: pair records of a pipelined job on their way into an index)

    // caching allocator for result buffers: size -> free blocks.
    // One stream: a freed block may still be in use by work that is queued on the context's stream, and whoever takes it next
    // queues behind that work on the same stream -- no waiting, no events.  With a pgr_pipe the back stream runs beside the
    // context's stream (`multi_stream`): a block that is freed then remembers where both streams stood at that moment (an
    // event each; the back stream's only for blocks that were handed out for it, so that the context's stream never waits for
    // a list stage it has nothing to do with) and whoever takes it for the OTHER stream makes that stream wait for the event --
    // event-ordered frees.
    struct FreeBlock {
        void *p = nullptr;
        hipEvent_t ev_front = nullptr, ev_back = nullptr;  // recorded on stream / back_stream when the block was freed
        hipEvent_t ev_fix = nullptr;  // a block freed while a job's second pass runs on the fix stream (alloc_stream == fix_stream): its
                                      // pending work is there and nowhere else -- the pass that used it on the back stream has been waited for
        bool on_back = false;  // last used for work on the back stream: handed to the back stream's requests first, and to the
                               // context's stream only when nothing else fits (it would wait for a list stage: a bubble)
    };
    struct LiveBlock {
        size_t bytes = 0;
        bool on_back = false;  // handed out for work on the back stream (or marked: block_on_back): a free records that stream too
        bool on_fix = false;   // ... the same for the fix stream (handed out inside a second pass, or marked: block_on_fix)
    };
    // ---- the arena under everything above (pgr_ctx_reserve, include/pgr_hip.h).  hipMalloc / hipFree of GiB blocks are not cheap
    // calls on this platform: a multi-GB hipMalloc in a process that has freed as much before blocks for seconds
    // (profiles/r05_target/big_malloc_probe.txt), and memory that has never been touched costs ~26 ms per GiB at first use.  A host
    // that knows roughly what a build needs reserves it ONCE: one hipMalloc, touched once; from then on every workspace, result
    // block and batch of this context is carved from it (address-ordered free list, best fit, neighbours coalesced) and goes back
    // to it; the runtime's allocator is asked only when the arena cannot serve a request (counted: mem_fallback_bytes).
    struct Arena {
        char *base = nullptr;
        pgr::ArenaList list;  // (csrc/arena_list.h: the free ranges)
    };
    std::vector<Arena> arenas;                 // grow-only: a later reserve adds one
    std::map<void *, std::pair<int, size_t>> arena_live;  // block -> (arena, length)
    size_t arena_bytes = 0, arena_used = 0, arena_peak = 0;
    size_t fallback_bytes = 0, fallback_calls = 0;        // device memory taken from the runtime although an arena exists
    int reserve(size_t bytes);
    hipError_t raw_alloc(void **out, size_t bytes);  // arena first, then hipMalloc
    void raw_free(void *p);                          // (like hipFree: nothing of the block's past is running when it returns)
    std::multimap<size_t, FreeBlock> free_blocks;
    std::map<void *, LiveBlock> live_blocks;
    size_t cached_bytes = 0;
    size_t live_bytes = 0, peak_bytes = 0;  // bytes handed out + cached (what the allocator holds of the device), its high-water mark
    bool multi_stream = false;
    int n_pipes = 0;
    bool back_shares_queue = false;  // no candidate for the back stream ran beside the context's stream (ctx.hip: enable_multi_stream)
    std::vector<pgr::Lane *> spare_lanes;  // lanes of destroyed pipes (their workspaces stay allocated for the next pipe)
    hipStream_t back_stream = nullptr;   // list stages of a pgr_pipe (created with the first pipe; higher priority than `stream`)
    // A job of a pipe that cannot keep its optimistic pass (flagged tiles: every batch of a real assembly; undersized estimates) is
    // finished on a stream of its own: the back stream holds the NEXT job's list stage, which waits for that job's tiles -- behind it
    // the islands and the second list stage of this job would wait for them too, and the context's stream would idle meanwhile.
    // nullptr when no stream was found that runs beside both others (then such a job is finished on the context's stream).
    hipStream_t fix_stream = nullptr;
    hipStream_t alloc_stream = nullptr;  // the stream the NEXT dmalloc's block will be used on (nullptr: `stream`)
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t take_event();
    void wait_and_recycle(FreeBlock &fb, hipStream_t user);
    void drop_events(FreeBlock &fb);
    void block_on_back(void *p);  // a block of the context's stream that the back stream works on as well (an index's records)
    void block_on_fix(void *p);   // ... that a second pass on the fix stream works on as well
    int enable_multi_stream();  // creates back_stream; synchronizes once so that earlier frees need no events
    void swap_lane(pgr::Lane &l);
    int poison_workspaces(hipStream_t st);  // debug_poison: 0xFF over the workspaces a job works in (whatever they hold is an earlier call's)

    int fail(int code, const std::string &msg) {
        err = msg;
        return code;
    }
    int dmalloc(void **out, size_t bytes);
    void dfree(void *p);
    int ensure_pinned(size_t bytes, std::string *err = nullptr);  // err != NULL: the message goes there (a thread other than the caller's)
    int ensure_pinned_out(size_t bytes);
    int ensure_mailbox(size_t bytes);
    // a second, small pinned mailbox for a consumer whose kernels run behind the shimmer pipeline's (which owns `mailbox`)
    void *qmail = nullptr;
    int ensure_qmail();
    // pinned block of the exact-island rounds (api.hip: run_exact_islands): descriptors up, states / status words down, one
    // DMA each way per round (four pageable copies cost ~25 us each)
    void *imail = nullptr;
    size_t imail_cap = 0;
    int ensure_imail(size_t bytes);
    // set by a caller of pgr_shmmrs_compute around the call: invoked (general pipeline only) with the device list, its offsets
    // [n+1], the list's capacity and the device address of the true count, right before the pipeline's one synchronization
    std::function<int(const pgr_mm128 *, const uint64_t *, uint64_t, const uint64_t *)> post_enqueue;
    // device -> pageable host memory through the pinned buffer (two windows, D2H of window i+1 overlaps the host
    // copy of window i, which is spread over a few threads); small transfers go straight through hipMemcpy
    int d2h(void *dst, const void *src_dev, size_t bytes);
    void release_all();
};

namespace pgr {
// host input of a staged batch: ASCII contigs (seqs), or the packed planes of pgr_batch_from_packed (planes != NULL; the
// arrays cover the contigs of the WHOLE call, word0 = first word of the sub-batch's first contig)
struct StageSrc {
    const uint8_t *const *seqs = nullptr;
    const uint64_t *lens = nullptr;
    const uint64_t *planes = nullptr;
    const uint32_t *valid = nullptr;
    uint64_t word0 = 0;
};
// shmmrutils.rs:443-445 / :575-576: the reference's assert!s as an error code
int check_spec(pgr_ctx *ctx, const pgr_spec *spec);
// pair records of a resident result into d_out (capacity >= pgr_shmmrs_n_pairs), stream ordered: returns without waiting
int shmmrs_to_frag_recs_enqueue(pgr_ctx *ctx, const pgr_shmmrs *s, const uint32_t *sids, int query_side,
                                pgr_frag_rec *d_out, uint64_t capacity);
// host inputs of >= 512 Mbp: contigs [c0, c1) are staged on the copy stream while the previous range is consumed
bool worth_pipelining(const pgr_ctx *ctx, uint32_t n, const uint64_t *lens);
int for_each_staged(pgr_ctx *ctx, uint32_t n, const StageSrc &src,
                    const std::function<int(pgr_batch *, uint32_t, uint32_t)> &consume);
void lane_release(pgr_ctx *ctx, Lane &l);  // frees a lane's workspaces, mailbox and events (pipeline.hip)
// pinned host blocks a kernel writes a result into (ctx.hip): acquire returns nullptr when the host cannot pin more memory;
// result_block_release takes any result block of the library -- pinned ones go back to the pool, the others to free()
void *pinned_result_acquire(size_t min_bytes, size_t *cap);
void result_block_release(void *p);
}  // namespace pgr

// Entry of an API call: the context's device, and a clean slate in the HIP runtime's per-thread "last error".  Calls below look at
// that state after their launches (hipGetLastError: did one of MINE fail to launch?), and so does rocPRIM -- an error some earlier
// call of this thread left there (another library's, PyTorch's, an ignored return value of this one) would be taken for theirs:
// round 6 saw "scan_counts(...): invalid argument" once in a 149-test run from exactly that.  The stale value is dropped, not lost:
// PGR_DEBUG_STALE=1 prints it.
#define PGR_ENTER(ctx)                                                                                       \
    do {                                                                                                     \
        hipError_t _e = hipSetDevice((ctx)->device);                                                         \
        if (_e != hipSuccess)                                                                                \
            return (ctx)->fail(PGR_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(_e));      \
        pgr::drop_stale_hip_error(__func__);                                                                 \
    } while (0)
#define PGR_HIP(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            return (ctx)->fail(PGR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));          \
    } while (0)
