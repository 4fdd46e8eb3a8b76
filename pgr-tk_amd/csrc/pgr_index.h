// pgr_index.h -- the GPU-resident ShmmrToFrags index object shared by index.hip (build / query) and
// mapgraph.hip (MAP-graph adjacency list, principal bundles).
#pragma once
#include <algorithm>
#include <atomic>

#include "pgr_ctx.h"

struct pgr_index {
    pgr_ctx *ctx = nullptr;
    pgr_spec spec = {};
    pgr_frag_rec *raw = nullptr;  // appended records (device)
    uint64_t n_raw = 0, cap_raw = 0;
    pgr_frag_rec *recs = nullptr;  // sorted records (device), valid when finalized
    uint64_t n = 0;
    uint64_t *key_off = nullptr;  // [n_keys + 1]
    uint64_t n_keys = 0;
    // bucket table over the keys: bucket(h0) = min(h0 >> lut_shift, 2^lut_bits - 1); lut[b] = first key whose bucket is >= b,
    // lut[2^lut_bits] = n_keys.  A lookup searches one bucket (a few keys) instead of all n_keys.  nullptr: no table.
    uint32_t *lut = nullptr;
    uint32_t lut_bits = 0, lut_shift = 0;
    // the keys by themselves ((h0, h1) per key, 16 B each, same order as key_off): the few keys of a bucket share one or two
    // cache lines, a search step does not have to go through key_off into the 40-byte records.  nullptr: no table.
    ulonglong2 *keys = nullptr;
    // the per-query kernel's table (query_fused.hip), 32 B per key, same order: the key AND, for a key with exactly one record (nearly
    // all keys of a genome index), that record's sid / bgn / end / orient -- a query pair's lookup (seq_db.rs:1200-1228) is then two
    // dependent trips to memory (bucket table, the bucket's entries) instead of six (bucket table, keys, key_off, the record's sid
    // for the count filters, the sid again and the record for the hit).  x = h0 | single << 63, y = h1 | orient << 63,
    // z = sid | bgn << 32, w = end.  nullptr: no table (fewer than 4096 keys, or external records with hashes of more than 56 bits).
    ulonglong4 *qkeys = nullptr;
    bool finalized = false;
    uint32_t next_sid = 0;
    int pipe_jobs = 0;  // jobs of a pgr_pipe in flight that place records in `raw` (the block must not move under them: pgr_index_reserve)
    uint64_t sid_bound = 0;  // max(sid) + 1 over the finalized records (0: unknown)
    // query path: calls left that go straight to the stage-by-stage kernels after the one-wavefront-per-query kernel
    // (query_fused.hip) declined a batch on this index
    // (hints only, relaxed atomics: contexts on several threads may query one index)
    mutable std::atomic<uint32_t> fused_skip{0};
    mutable std::atomic<uint32_t> fused_l1_skip{0};  // ... that skip the level-1 form (its last batch was flagged by the tile kernel)
    mutable std::atomic<uint32_t> fused_pairs{0};  // most shimmer pairs of one query in the last batch (0: no batch yet)
    mutable std::atomic<uint32_t> fused_hits{0};  // slot size (hits per query) the last batch of short queries needed, 0: the minimum
    mutable std::atomic<float> fused_per_q[3] = {{-1.0f}, {-1.0f}, {-1.0f}};  // targets / chains / hit pairs per query of that batch (-1: none yet)
    mutable std::atomic<float> fused_bytes_per_q{0.0f};  // result bytes per query of that batch (sizes the next first download)
};

namespace pgr {

// small RAII device temp (from the context's caching allocator)
struct Tmp {
    pgr_ctx *ctx;
    void *p = nullptr;
    explicit Tmp(pgr_ctx *c) : ctx(c) {}
    ~Tmp() { ctx->dfree(p); }
    int alloc(size_t bytes) { return ctx->dmalloc(&p, std::max<size_t>(bytes, 16)); }
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p); }
    Tmp(const Tmp &) = delete;
    Tmp &operator=(const Tmp &) = delete;
};

inline unsigned bits_for(uint64_t n_values) {  // bits needed for values 0 .. n_values - 1 (at least 1)
    unsigned b = 1;
    while (b < 64 && (1ull << b) < n_values) ++b;
    return b;
}

inline dim3 grid_for(uint64_t n, uint32_t block = 256) { return dim3((uint32_t)((n + block - 1) / block)); }

// record fields usable as sort keys
enum RecField { F_FRG_ID = 0, F_SID = 1, F_H1 = 2, F_H0 = 3, F_BGN = 4, F_END = 5, F_ORIENT = 6 };

// multi-pass stable LSD radix sort of a permutation of recs; fields are given least significant first.
// idx_a: in = initial permutation, out = sorted permutation
int sort_perm(pgr_ctx *ctx, const pgr_frag_rec *recs, uint64_t n, const int *fields, const unsigned *bits, int n_fields,
              uint32_t *idx_a, uint32_t *idx_b, uint64_t *keys_a, uint64_t *keys_b);
// room for `need` appended records (contents kept)
int index_grow_raw(pgr_ctx *ctx, pgr_index *ix, uint64_t need);
void launch_iota(hipStream_t st, uint32_t *idx, uint64_t n);
void launch_gather_recs(hipStream_t st, const pgr_frag_rec *in, const uint32_t *idx, pgr_frag_rec *out, uint64_t n);

}  // namespace pgr
