// pgr_aln.h -- device helpers shared by the query kernels of index.hip (one kernel per stage, any batch) and query_fused.hip
// (one wavefront per short query, every stage behind the query's pair records in one kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>

#include "pgr_index.h"

namespace pgr {

// count filters of aln::query_fragment_to_hps (aln.rs:197-228)
struct QParams {
    uint32_t max_count, max_count_query, max_count_target;
};

// a pair whose key holds more records than this is walked by a whole wavefront in hits_kernel
constexpr uint64_t HITS_HEAVY = 64;

// aln::sparse_aln (aln.rs:12-142)
struct AlnParams {
    uint32_t max_span;
    float penalty;
    int has_max_gap;
    uint32_t max_gap;
    int oriented;
};

constexpr uint32_t MAX_SPAN_CAP = 64;

__device__ __forceinline__ bool same_q(const pgr_hitpair &a, const pgr_hitpair &b) {
    return a.qb == b.qb && a.qe == b.qe && a.qo == b.qo;
}
__device__ __forceinline__ bool same_hp(const pgr_hitpair &a, const pgr_hitpair &b) {
    return same_q(a, b) && a.tb == b.tb && a.te == b.te && a.to == b.to;
}
__device__ __forceinline__ float absf(float v) { return v < 0.0f ? -v : v; }

// maximum over the wavefront (DPP row shifts + row broadcasts; lanes without a source see -inf)
__device__ __forceinline__ float wave_max_f32(float v) {
    const int ninf = __float_as_int(-INFINITY);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), 0x111, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), 0x112, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), 0x114, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), 0x118, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), 0x142, 0xa, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), 0x143, 0xc, 0xf, false)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// The largest POSITIVE value over the wavefront, as its bit pattern (0: no lane holds a positive value).  Positive f32 values
// order like their bit patterns, and an unsigned maximum has the identity a DPP row shift supplies for lanes without a source:
// the compiler folds the shifts into six v_max_u32_dpp, where the f32 form above costs five instructions per step (a fill with
// -inf, the shift, two canonicalizing maxima, a hazard nop).  What the chaining DP asks of its maximum -- is it above zero, and
// which lanes hold it -- is answered by the bits (aln.rs:86-89, :105-131: scores are compared with > 0 and with each other).
__device__ __forceinline__ uint32_t wave_max_pos_bits(float x) {
    const int b = __float_as_int(x);
    uint32_t v = (uint32_t)(b > 0 ? b : 0);
    auto mx = [](uint32_t p, uint32_t q) { return p > q ? p : q; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ void wave_sync() {  // single-wave workgroup: orders LDS and global accesses of the wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// raw_query_fragment lookup (seq_db.rs:1200-1228): [a, b) = records of the key (h0, h1), empty when absent.  Keys are window
// minima of a hash: nearly all of them lie in the lowest few percent of the 56-bit range, where the bucket table
// (pgr_index.h) cuts the binary search over all keys (25 dependent steps of two loads for 3x10^7 keys) down to the few keys of
// one bucket.
__device__ __forceinline__ void lookup_range(uint64_t h0, uint64_t h1, const pgr_frag_rec *__restrict__ recs,
                                             const uint64_t *__restrict__ key_off, uint64_t n_keys,
                                             const uint32_t *__restrict__ lut, uint32_t lut_bits, uint32_t lut_shift,
                                             const ulonglong2 *__restrict__ keys, uint64_t &a, uint64_t &b) {
    uint64_t lo = 0, hi = n_keys;  // first key >= (h0,h1)
    if (lut) {
        const uint64_t top = (1ull << lut_bits) - 1;
        const uint64_t bk = (h0 >> lut_shift) < top ? (h0 >> lut_shift) : top;
        lo = lut[bk];
        hi = lut[bk + 1];
    }
    a = 0;
    b = 0;
    if (keys) {  // the bucket's keys by themselves: one or two cache lines for the whole search
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            const ulonglong2 kk = keys[mid];
            if (kk.x < h0 || (kk.x == h0 && kk.y < h1)) lo = mid + 1;
            else hi = mid;
        }
        if (lo < n_keys) {
            const ulonglong2 kk = keys[lo];
            if (kk.x == h0 && kk.y == h1) {
                a = key_off[lo];
                b = key_off[lo + 1];
            }
        }
    } else {
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            const pgr_frag_rec &r = recs[key_off[mid]];
            if (r.h0 < h0 || (r.h0 == h0 && r.h1 < h1)) lo = mid + 1;
            else hi = mid;
        }
        if (lo < n_keys) {
            const pgr_frag_rec &r = recs[key_off[lo]];
            if (r.h0 == h0 && r.h1 == h1) {
                a = key_off[lo];
                b = key_off[lo + 1];
            }
        }
    }
}

// ---- query_fused.hip: the stages behind the queries' pair records for batches of SHORT queries, one wavefront per query.
// declined = the batch does not fit the path (a query with more pairs / hits than the kernel's LDS image, a key with more
// than HITS_HEAVY records, a (query, target) group too long for the register DP): the caller takes the general path.
struct QueryFusedCounts {
    uint64_t n_signatures = 0, n_hits = 0;
    uint64_t n_pairs = 0;  // (level-1 form: the queries' shimmer pairs, which only the device has counted)
};
constexpr uint32_t QF_MAX_PAIRS = 128;  // shimmer pairs of one query
bool query_fused_eligible(const pgr_ctx *ctx, uint32_t n_queries, uint64_t max_pairs, uint32_t max_aln_span);

// The level-1 form of the per-query kernel: no list stage of the batch at all.  The shimmer pipeline's list stage (two scans,
// block_first_seg, fused_select, gather, offsets_by_rid, pair_offsets, frag_recs_dev: 11 dependent launches) exists to put the
// level-1 minimizers of ALL contigs of a batch into one global order; a query's wavefront needs its own ~250 only.  It reads them
// from the tile kernel's segments (contig q owns the segments tile_first[q] + q .. tile_first[q + 1] + q), reduces twice, applies
// min_span and forms the pairs in LDS (shmmrutils.rs:359-415, 533-555; seq_db.rs:1205-1217), then goes on as before.  What the
// level-1 kernels flag (islands needed: a palindromic k-mer, a non-ACGT byte; an overflow) declines the batch on the device: the
// shimmer pipeline takes it as it always did.
struct L1Rec;
struct QfLevel1View {
    const L1Rec *l1 = nullptr;            // the level-1 buffer (tile slots, overflow region, tail slots)
    const uint64_t *seg_off = nullptr;    // [n_tiles + n]
    const uint32_t *seg_cnt = nullptr;
    const uint32_t *tile_first = nullptr;  // [n + 1] (device)
    const unsigned long long *status = nullptr;  // the level-1 cursor words: [0] overflow taken [1] overflow too small [2] islands needed
    uint64_t ovf_cap = 0;
    uint32_t *flags = nullptr;  // six 32-bit words cleared in front of the tile kernel (the cursor words of the list stage that does not run)
    uint32_t r = 0, min_span = 0;
};
constexpr uint32_t QF_C1_MAX = 2048;  // level-1 minimizers of one query (16 B of LDS each)
// level-1 minimizers to make room for, for queries of up to max_len bases (density 2 / (w + 1) with room); 0: too long for the form
uint32_t query_fused_level1_cap(uint32_t max_len, uint32_t w);
// pipeline.hip: stage 1 of the shimmer pipeline alone (tile descriptors, flags, tiles + tails) and `consumer` right behind it
int shmmr_level1_then(pgr_ctx *ctx, const pgr_batch *b, const pgr_spec *spec, const std::function<int(const QfLevel1View &)> &consumer,
                      bool *taken);

// One batch through the per-query kernel.  enqueue*() puts the kernels and the first download on the context's stream (nothing
// waits); finish() runs behind a synchronization of that stream and hands out the result -- or says `declined`.
//   enqueue(qrec, pair_off)          the queries' pair records are already on the device
//   enqueue_from_shimmers(...)       builds them from the shimmer pipeline's device result first: this is what
//                                    pgr_ctx::post_enqueue calls so that the whole query needs ONE host wait
struct QueryFusedRun {
    pgr_ctx *ctx;
    const pgr_index *ix;
    uint32_t n_queries;
    QParams qp;
    AlnParams ap;
    uint32_t P = 0, H = 0;
    bool enqueued = false, no_pinned = false, finish_called = false;
    // a run of a pipelined query job (pgr_pipe_submit_query): its kernels and its download go to the pipe's back stream, its
    // totals to the slot's own pinned words (NULL: the context's stream / the context's mailbox)
    hipStream_t stream = nullptr;
    uint64_t *mail = nullptr;
    // ... and the download of its chains to a third stream (behind an event), so that the back stream is free for the next job's
    // list stage while 7.5 MB cross PCIe; ev_copied is what the collector waits for
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_copied = nullptr;
    QueryFusedRun(pgr_ctx *ctx, const pgr_index *ix, uint32_t n_queries, uint64_t max_pairs, const QParams &qp, const AlnParams &ap);
    ~QueryFusedRun();
    QueryFusedRun(const QueryFusedRun &) = delete;
    QueryFusedRun &operator=(const QueryFusedRun &) = delete;
    int enqueue(const pgr_frag_rec *d_qrec, const uint64_t *d_pair_off, bool flags_cleared = false);
    int enqueue_from_shimmers(const pgr_mm128 *d_mm, const uint64_t *d_off, uint64_t cap, const uint64_t *d_count);
    // the level-1 form: behind the tile kernel of the queries (C1 from query_fused_level1_cap); finish() as for the others --
    // `declined` with l1_flagged set means the level-1 kernels asked for the shimmer pipeline (islands, overflow)
    int enqueue_from_level1(const QfLevel1View &v, uint32_t c1);
    bool from_l1 = false, l1_flagged = false;
    uint32_t C1 = 0;
    int finish(pgr_hps_result *out, QueryFusedCounts *counts, bool *declined);

private:
    void *d_cnt = nullptr, *d_offs = nullptr, *d_shp = nullptr, *d_sf = nullptr, *d_img = nullptr, *d_qrec = nullptr, *d_rec_off = nullptr;
    size_t cnt_bytes = 0, offs_bytes = 0, shp_bytes = 0, sf_bytes = 0, img_bytes = 0, qrec_bytes = 0, rec_off_bytes = 0;
    void *d_desc = nullptr;  // the look-back's descriptors (single-pass form)
    size_t desc_bytes = 0;
    bool direct = false, direct_failed = false;  // single pass: the kernel writes the host's block itself
    uint64_t cap_t = 0, cap_c = 0, cap_h = 0;    // ... whose sections hold this many targets / chains / hit pairs
    uint8_t *block = nullptr;  // pinned host block of the result
    size_t cap = 0, first = 0;
    const pgr_frag_rec *qrec_used = nullptr;
    const uint64_t *pair_off_used = nullptr;
    QfLevel1View l1v;
};

}  // namespace pgr
