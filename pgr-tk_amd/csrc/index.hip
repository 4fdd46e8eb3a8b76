// index.hip -- ShmmrToFrags index (frag_map) and the query path on the GPU.
//
// Index  : FxHashMap<(u64,u64), Vec<FragmentSignature>> of pgr-db/src/seq_db.rs:75-76, built by
//          load_index_from_seq_vec (seq_db.rs:573-615), as a CSR: records sorted by (h0, h1, sid, frg_id)
//          (= the reference's per-key Vec order, which is insertion order = ascending (sid, frg_id)),
//          key_off[] = first record of every distinct key.
// Query  : raw_query_fragment (seq_db.rs:1200-1228) + aln::query_fragment_to_hps (aln.rs:147-242) +
//          aln::sparse_aln (aln.rs:12-142) for a whole batch of queries.
//
// Sorting / scans are rocPRIM library calls; lookups, count filters, hit expansion and the chaining DP
// are hand-written kernels.  f32 arithmetic in the DP is IEEE without contraction (-ffp-contract=off),
// in the reference's operation order.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <unistd.h>

#include "pgr_index.h"
#include "pgr_device.h"
#include "pgr_aln.h"

using namespace pgr;

namespace {

// the few device-side counts the host reads after a stage, gathered into one block: ONE small copy instead of several
__global__ void gather_words_kernel(const uint64_t *a, const uint64_t *b, const uint64_t *c, const uint32_t *d,
                                    uint64_t *__restrict__ out) {
    if (threadIdx.x == 0) out[0] = a ? *a : 0;
    if (threadIdx.x == 1) out[1] = b ? *b : 0;
    if (threadIdx.x == 2) out[2] = c ? *c : 0;
    if (threadIdx.x == 3) out[3] = d ? (uint64_t)*d : 0;
}

__global__ void iota_kernel(uint32_t *idx, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)i;
}

// field: pgr::RecField
__global__ void rec_key_kernel(const pgr_frag_rec *__restrict__ recs, const uint32_t *__restrict__ idx, int field,
                               uint64_t *__restrict__ keys, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pgr_frag_rec &r = recs[idx[i]];
    uint64_t k;
    switch (field) {
    case 0: k = r.frg_id; break;
    case 1: k = r.sid; break;
    case 2: k = r.h1; break;
    case 3: k = r.h0; break;
    case 4: k = r.bgn; break;
    case 5: k = r.end; break;
    default: k = r.orient; break;
    }
    keys[i] = k;
}

// one pass over the appended records (grid-stride, a few thousand workgroups: one atomic per workgroup and statistic):
// stats[0] |= 1 not in (sid, frg_id) append order, stats[1] |= 1 not already in (h0, h1, sid, frg_id) order,
// stats[2] = max(sid) + 1, stats[3] = max(h0, h1) over all records (bounds the radix passes of both key fields; records that come
// in through pgr_index_add_records need not be canonical, so h0 <= h1 is NOT assumed)
__global__ __launch_bounds__(256) void raw_stats_kernel(const pgr_frag_rec *__restrict__ recs, uint64_t n,
                                                        unsigned long long *__restrict__ stats, uint64_t *__restrict__ h0_out,
                                                        uint64_t *__restrict__ h1_out) {
    bool bad = false, bad_key = false;
    unsigned long long msid = 0, mh = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const pgr_frag_rec a = recs[i];
        h0_out[i] = a.h0;  // the two hashes by themselves, in append order: sort keys without another pass over the 40-byte records
        h1_out[i] = a.h1;
        msid = umax64(msid, (unsigned long long)a.sid + 1ull);
        mh = umax64(mh, umax64(a.h0, a.h1));
        if (i + 1 < n) {
            const pgr_frag_rec &b = recs[i + 1];
            const bool id_gt = a.sid > b.sid || (a.sid == b.sid && a.frg_id > b.frg_id);
            bad = bad || id_gt;
            bad_key = bad_key || a.h0 > b.h0 || (a.h0 == b.h0 && (a.h1 > b.h1 || (a.h1 == b.h1 && id_gt)));
        }
    }
    __shared__ unsigned long long s_sid[4], s_h[4];
    __shared__ uint32_t s_bad[2];
    if (threadIdx.x < 2) s_bad[threadIdx.x] = 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        msid = umax64(msid, shfl_xor64(msid, d));
        mh = umax64(mh, shfl_xor64(mh, d));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        s_sid[threadIdx.x >> 6] = msid;
        s_h[threadIdx.x >> 6] = mh;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) s_bad[0] = 1;      // (benign race: all writers store 1)
    if (__ballot(bad_key) && (threadIdx.x & 63) == 0) s_bad[1] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(stats + 2, umax64(umax64(s_sid[0], s_sid[1]), umax64(s_sid[2], s_sid[3])));
        atomicMax(stats + 3, umax64(umax64(s_h[0], s_h[1]), umax64(s_h[2], s_h[3])));
        if (s_bad[0]) atomicOr(stats, 1ull);
        if (s_bad[1]) atomicOr(stats + 1, 1ull);
    }
}

__global__ void gather_recs_kernel(const pgr_frag_rec *__restrict__ in, const uint32_t *__restrict__ idx,
                                   pgr_frag_rec *__restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}

// ---- sorting by ONE key.  Records appended in (sid, frg_id) order need the order (h0, h1, append position).  A stable sort by
// h0 alone (half the radix passes) leaves every run of equal h0 in append order; what is missing is a stable sort by h1 INSIDE
// the runs -- and nearly all runs are short: a shimmer that is smaller than both its neighbours is the h0 of two pairs (run of
// 2), a key of a G-haplotype pangenome makes runs of ~2G.
//   h1s[i] = h1 of the i-th record in h0 order
__global__ void gather_h1_kernel(const uint32_t *__restrict__ idx, const uint64_t *__restrict__ h1_app, uint64_t n,
                                 uint64_t *__restrict__ h1s) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) h1s[i] = h1_app[idx[i]];
}
//   runs of up to RUN_SMALL records: every record finds its run and its rank in it (h1, then position: stable) by itself.
//   Longer runs: their first record finds the end (galloping + binary search) and puts (start, length) on a list for
//   run_long_kernel; a run of more than RUN_LONG_MAX records raises counters[1] (the caller sorts by both keys instead).
constexpr int RUN_SMALL = 16;
constexpr uint32_t RUN_LONG_MAX = 4096;
__global__ __launch_bounds__(256) void run_rank_kernel(const uint64_t *__restrict__ k, const uint64_t *__restrict__ h1s,
                                                       const uint32_t *__restrict__ idx_in, uint64_t n, uint32_t *__restrict__ idx_out,
                                                       uint2 *__restrict__ long_list, uint32_t *__restrict__ counters) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const uint64_t key = live ? k[i] : 0;
    bool leader = false;  // first record of a long run
    uint32_t len = 0;
    if (live) {
        uint64_t s = i;
        int steps = 0;
        while (s > 0 && steps < RUN_SMALL && k[s - 1] == key) {
            --s;
            ++steps;
        }
        const bool start_found = s == 0 || k[s - 1] != key;
        bool small = false;
        uint64_t e = s + 1;
        if (start_found) {
            while (e < n && e - s <= (uint64_t)RUN_SMALL && k[e] == key) ++e;
            small = e - s <= (uint64_t)RUN_SMALL;
        }
        if (small) {
            const uint64_t mine = h1s[i];
            uint32_t rank = 0;
            for (uint64_t j = s; j < e; ++j) {
                const uint64_t o = h1s[j];
                rank += (o < mine || (o == mine && j < i)) ? 1u : 0u;
            }
            idx_out[s + rank] = idx_in[i];
        } else if (start_found && s == i) {
            uint64_t lo = e, step = 32;  // k[lo - 1] == key; first index behind the run lies in (lo - 1, n]
            uint64_t hi = lo;
            while (hi < n && k[hi] == key) {
                lo = hi + 1;
                hi = hi + step < n ? hi + step : n;
                step <<= 1;
            }
            while (lo < hi) {  // k[lo - 1] == key, (hi == n or k[hi] != key)
                const uint64_t mid = (lo + hi) >> 1;
                if (k[mid] == key) lo = mid + 1;
                else hi = mid;
            }
            const uint64_t L = lo - s;
            if (L > (uint64_t)RUN_LONG_MAX) counters[1] = 1u;  // (benign race: every writer stores 1)
            else {
                leader = true;
                len = (uint32_t)L;
            }
        }
    }
    const uint64_t m = __ballot(leader);
    if (m) {  // one atomic per wavefront for its leaders
        const uint32_t lane = threadIdx.x & 63;
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(counters, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(m), 64);
        if (leader) long_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = make_uint2((uint32_t)i, len);
    }
}
//   one workgroup per long run: the run's h1 in LDS, every record's rank by counting (O(L^2 / 256) steps: L is a few hundred for
//   pangenome keys).
__global__ __launch_bounds__(256) void run_long_kernel(const uint2 *__restrict__ long_list, const uint64_t *__restrict__ h1s,
                                                       const uint32_t *__restrict__ idx_in, uint32_t *__restrict__ idx_out) {
    __shared__ uint64_t h[RUN_LONG_MAX];
    const uint2 r = long_list[blockIdx.x];
    const uint64_t s = r.x;
    const uint32_t L = r.y;
    for (uint32_t j = threadIdx.x; j < L; j += 256) h[j] = h1s[s + j];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < L; i += 256) {
        const uint64_t mine = h[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < L; ++j) {
            const uint64_t o = h[j];
            rank += (o < mine || (o == mine && j < i)) ? 1u : 0u;
        }
        idx_out[s + rank] = idx_in[s + i];
    }
}

__global__ void key_flags_kernel(const pgr_frag_rec *__restrict__ recs, uint64_t n, uint32_t *__restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) {
        flags[i] = 0;
        return;
    }
    flags[i] = (i == 0 || recs[i].h0 != recs[i - 1].h0 || recs[i].h1 != recs[i - 1].h1) ? 1u : 0u;
}

// keys[i] = (h0, h1) of key i
__global__ void gather_keys_kernel(const pgr_frag_rec *__restrict__ recs, const uint64_t *__restrict__ key_off, uint64_t n_keys,
                                   ulonglong2 *__restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_keys) return;
    const pgr_frag_rec &r = recs[key_off[i]];
    keys[i] = make_ulonglong2(r.h0, r.h1);
}

// qkeys[i] = key i with its record when it has exactly one (pgr_index.h)
__global__ void gather_qkeys_kernel(const pgr_frag_rec *__restrict__ recs, const uint64_t *__restrict__ key_off, uint64_t n_keys,
                                    ulonglong4 *__restrict__ qkeys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_keys) return;
    const uint64_t s = key_off[i], e = key_off[i + 1];
    const pgr_frag_rec r = recs[s];
    ulonglong4 q;
    q.x = r.h0 | (e - s == 1 ? 1ull << 63 : 0ull);
    q.y = r.h1 | ((uint64_t)(r.orient & 1u) << 63);
    q.z = (uint64_t)r.sid | ((uint64_t)r.bgn << 32);
    q.w = (uint64_t)r.end;
    qkeys[i] = q;
}

// lut[b] = first key whose bucket is >= b (b = 2^bits: n_keys); one thread per bucket
__global__ void build_lut_kernel(const pgr_frag_rec *__restrict__ recs, const uint64_t *__restrict__ key_off, uint64_t n_keys,
                                 uint32_t bits, uint32_t shift, uint32_t *__restrict__ lut) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nb = 1ull << bits;
    if (b > nb) return;
    uint64_t lo = 0, hi = n_keys;
    if (b == nb) lo = n_keys;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const uint64_t h0 = recs[key_off[mid]].h0;
        const uint64_t bk = (h0 >> shift) < nb - 1 ? (h0 >> shift) : nb - 1;
        if (bk < b) lo = mid + 1;
        else hi = mid;
    }
    lut[b] = (uint32_t)lo;
}

__global__ void scatter_starts_kernel(const uint32_t *__restrict__ flags, const uint64_t *__restrict__ rank, uint64_t n,
                                      uint64_t *__restrict__ starts, uint32_t *__restrict__ zero4 = nullptr) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) {
        starts[rank[n]] = n;
        if (zero4) zero4[0] = zero4[1] = zero4[2] = zero4[3] = 0;  // counters of the kernels behind this one (saves a memset)
        return;
    }
    if (flags[i]) starts[rank[i]] = i;
}

// multi-pass stable LSD sort of a permutation; fields are sorted in the given order (least significant first)
}  // namespace

int pgr::sort_perm(pgr_ctx *ctx, const pgr_frag_rec *recs, uint64_t n, const int *fields, const unsigned *bits, int n_fields,
                   uint32_t *idx_a /*in: perm, out: sorted perm*/, uint32_t *idx_b, uint64_t *keys_a, uint64_t *keys_b) {
    hipStream_t st = ctx->stream;
    const size_t tb = sort_pairs_temp_bytes(n);
    int rc;
    if ((rc = ctx->ws_scan_tmp.ensure(ctx, tb))) return rc;
    uint32_t *cur = idx_a, *nxt = idx_b;
    for (int f = 0; f < n_fields; ++f) {
        hipLaunchKernelGGL(rec_key_kernel, grid_for(n), dim3(256), 0, st, recs, cur, fields[f], keys_a, n);
        PGR_HIP(ctx, sort_pairs(st, ctx->ws_scan_tmp.p, tb, keys_a, keys_b, cur, nxt, n, bits[f]));
        std::swap(cur, nxt);
    }
    if (cur != idx_a) PGR_HIP(ctx, hipMemcpyAsync(idx_a, cur, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    return PGR_OK;
}

void pgr::launch_iota(hipStream_t st, uint32_t *idx, uint64_t n) {
    if (n) hipLaunchKernelGGL(iota_kernel, grid_for(n), dim3(256), 0, st, idx, n);
}
void pgr::launch_gather_recs(hipStream_t st, const pgr_frag_rec *in, const uint32_t *idx, pgr_frag_rec *out, uint64_t n) {
    if (n) hipLaunchKernelGGL(gather_recs_kernel, grid_for(n), dim3(256), 0, st, in, idx, out, n);
}

int pgr::index_grow_raw(pgr_ctx *ctx, pgr_index *ix, uint64_t need) {
    if (need <= ix->cap_raw) return PGR_OK;
    const uint64_t cap = std::max<uint64_t>(need + need / 2, 1024);
    pgr_frag_rec *np = nullptr;
    int rc = ctx->dmalloc((void **)&np, cap * sizeof(pgr_frag_rec));
    if (rc) return rc;
    {
        // The append block is written by whatever stream the caller's jobs run on (the context's stream, a pipe's back stream for
        // commits and direct placement, its fix stream for second passes) -- the allocator has ordered it, and under debug_poison
        // filled it, for ONE stream: the one it was asked for.  The old records are copied on that stream and the host waits for
        // it, ALWAYS: when this returns the block's previous life (other streams' work the allocator made this stream wait for)
        // and the fill are over, and any stream may write it.  (Round 6, fuzz_pipe under debug_poison: the first commit into a
        // fresh block has no old records to copy, so nothing was waited for, and the fill -- queued on the context's stream behind
        // a tile kernel -- landed on top of the back stream's commit copy: a job's records replaced by 0xFF, 4 runs in 12.
        // Without the fill the same gap is a block's previous life racing its first writer on another stream.)
        hipStream_t st = ctx->alloc_stream ? ctx->alloc_stream : ctx->stream;
        hipError_t e = hipSuccess;
        if (ix->n_raw) e = hipMemcpyAsync(np, ix->raw, ix->n_raw * sizeof(pgr_frag_rec), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            ctx->dfree(np);
            return ctx->fail(PGR_ERR_DEVICE, hipGetErrorString(e));
        }
    }
    ctx->dfree(ix->raw);
    ix->raw = np;
    ix->cap_raw = cap;
    return PGR_OK;
}
static int grow_raw(pgr_ctx *ctx, pgr_index *ix, uint64_t need) { return pgr::index_grow_raw(ctx, ix, need); }

// ------------------------------------------------------------------------------------------------
extern "C" int pgr_index_create(pgr_ctx *ctx, const pgr_spec *spec, pgr_index **out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!spec || !out) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (spec->k == 0 || spec->k > 56) return ctx->fail(PGR_ERR_BAD_SPEC, "spec.k must be in 1..56");
    if (spec->r == 0 || spec->r > 12) return ctx->fail(PGR_ERR_BAD_SPEC, "spec.r must be in 1..12");
    if (!spec->sketch && (spec->w == 0 || spec->w > 128)) return ctx->fail(PGR_ERR_BAD_SPEC, "spec.w must be in 1..128");
    pgr_index *ix = new pgr_index();
    ix->ctx = ctx;
    ix->spec = *spec;
    *out = ix;
    return PGR_OK;
}

extern "C" void pgr_index_destroy(pgr_index *ix) {
    if (!ix) return;
    ix->ctx->dfree(ix->raw);
    ix->ctx->dfree(ix->recs);
    ix->ctx->dfree(ix->key_off);
    ix->ctx->dfree(ix->lut);
    ix->ctx->dfree(ix->keys);
    ix->ctx->dfree(ix->qkeys);
    delete ix;
}

extern "C" uint64_t pgr_index_n_keys(const pgr_index *ix) { return ix ? ix->n_keys : 0; }
extern "C" uint64_t pgr_index_n_records(const pgr_index *ix) { return ix ? (ix->finalized ? ix->n : ix->n_raw) : 0; }

extern "C" int pgr_index_reserve(pgr_ctx *ctx, pgr_index *ix, uint64_t n_records) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix) return ctx->fail(PGR_ERR_INVALID_ARG, "null index");
    if (ix->ctx != ctx) return ctx->fail(PGR_ERR_STATE, "index belongs to another context");
    if (n_records <= ix->cap_raw) return PGR_OK;
    // (jobs of a pipe in flight write their records through the device cursor into the block that is about to move)
    if (ix->pipe_jobs > 0) return ctx->fail(PGR_ERR_STATE, "pgr_index_reserve while jobs of a pgr_pipe on this index are in flight: collect them first");
    PGR_ENTER(ctx);
    pgr_frag_rec *np = nullptr;
    int rc = ctx->dmalloc((void **)&np, n_records * sizeof(pgr_frag_rec));  // (exactly what was asked for: no head-room on top)
    if (rc) return rc;
    if (ix->n_raw) {
        hipError_t e = hipMemcpyAsync(np, ix->raw, ix->n_raw * sizeof(pgr_frag_rec), hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            ctx->dfree(np);
            return ctx->fail(PGR_ERR_DEVICE, hipGetErrorString(e));
        }
    }
    ctx->dfree(ix->raw);
    ix->raw = np;
    ix->cap_raw = n_records;
    return PGR_OK;
}

extern "C" int pgr_index_add_records(pgr_ctx *ctx, pgr_index *ix, const pgr_frag_rec *recs, uint64_t n, int on_device) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || (n && !recs)) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (n == 0) return PGR_OK;
    PGR_ENTER(ctx);
    int rc = grow_raw(ctx, ix, ix->n_raw + n);
    if (rc) return rc;
    PGR_HIP(ctx, hipMemcpyAsync(ix->raw + ix->n_raw, recs, n * sizeof(pgr_frag_rec),
                                on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ix->n_raw += n;
    ix->finalized = false;
    return PGR_OK;
}

namespace {
// pair records from a concatenation of per-sequence shimmer lists (y >> 32 = sequence id):
// flags[i] = 1 iff (i, i+1) is a pair; rank = exclusive scan; run_start = first index of i's sequence
__global__ void mm_pair_flags_kernel(const pgr_mm128 *__restrict__ mm, uint64_t n, uint32_t *__restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    flags[i] = (i + 1 < n && (mm[i].y >> 32) == (mm[i + 1].y >> 32)) ? 1u : 0u;
}
__global__ void mm_to_recs_kernel(const pgr_mm128 *__restrict__ mm, uint64_t n, const uint32_t *__restrict__ flags,
                                  const uint64_t *__restrict__ rank, pgr_frag_rec *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const pgr_mm128 s0 = mm[i], s1 = mm[i + 1];
    // ordinal of the pair in its sequence = number of pairs of the same sequence before it: walk back over
    // the run start with the ranks (pairs of one sequence are consecutive ranks)
    uint64_t lo = 0, hi = i;  // first index of the run of this sid: binary search on the sid field
    const uint64_t sid = s0.y >> 32;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        // runs are contiguous: every element in [run_start, i] has this sid; elements before have another
        if ((mm[mid].y >> 32) == sid && rank[i] - rank[mid] == i - mid) hi = mid;
        else lo = mid + 1;
    }
    const uint64_t h0 = s0.x >> 8, h1 = s1.x >> 8;
    const bool keep = h0 <= h1;  // index side (seq_db.rs:391)
    pgr_frag_rec r;
    r.h0 = keep ? h0 : h1;
    r.h1 = keep ? h1 : h0;
    r.frg_id = (uint32_t)(i - lo);
    r.sid = (uint32_t)sid;
    r.bgn = (uint32_t)((s0.y & 0xFFFFFFFFull) >> 1) + 1;
    r.end = (uint32_t)((s1.y & 0xFFFFFFFFull) >> 1) + 1;
    r.orient = keep ? 0u : 1u;
    r._pad = 0;
    out[rank[i]] = r;
}
}  // namespace

extern "C" int pgr_index_add_shmmrs(pgr_ctx *ctx, pgr_index *ix, const pgr_mm128 *mm, uint64_t n, int on_device) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || (n && !mm)) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (n < 2) return PGR_OK;
    if (n >= (1ull << 32)) return ctx->fail(PGR_ERR_INVALID_ARG, "more than 2^32-1 shimmers in one call");
    PGR_ENTER(ctx);
    hipStream_t st = ctx->stream;
    int rc;
    Tmp d_mm(ctx), flags(ctx), rank(ctx);
    const pgr_mm128 *dm = mm;
    if (!on_device) {
        if ((rc = d_mm.alloc(n * sizeof(pgr_mm128)))) return rc;
        PGR_HIP(ctx, hipMemcpyAsync(d_mm.p, mm, n * sizeof(pgr_mm128), hipMemcpyHostToDevice, st));
        dm = d_mm.as<pgr_mm128>();
    }
    if ((rc = flags.alloc((n + 1) * 4)) || (rc = rank.alloc((n + 1) * 8))) return rc;
    hipLaunchKernelGGL(mm_pair_flags_kernel, grid_for(n + 1), dim3(256), 0, st, dm, n, flags.as<uint32_t>());
    const size_t tb = scan_counts_temp_bytes((uint32_t)(n + 1));
    if ((rc = ctx->ws_scan_tmp.ensure(ctx, tb))) return rc;
    PGR_HIP(ctx, scan_counts(st, ctx->ws_scan_tmp.p, tb, flags.as<uint32_t>(), rank.as<uint64_t>(), (uint32_t)(n + 1)));
    uint64_t np = 0;
    PGR_HIP(ctx, hipMemcpyAsync(&np, rank.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
    PGR_HIP(ctx, hipStreamSynchronize(st));
    if ((rc = grow_raw(ctx, ix, ix->n_raw + np))) return rc;
    hipLaunchKernelGGL(mm_to_recs_kernel, grid_for(n), dim3(256), 0, st, dm, n, flags.as<uint32_t>(), rank.as<uint64_t>(),
                       ix->raw + ix->n_raw);
    PGR_HIP(ctx, hipStreamSynchronize(st));
    PGR_HIP(ctx, hipGetLastError());
    ix->n_raw += np;
    ix->finalized = false;
    return PGR_OK;
}

extern "C" int pgr_index_add_resident(pgr_ctx *ctx, pgr_index *ix, const pgr_batch *b, const uint32_t *sids) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || !b) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    pgr_shmmrs *s = nullptr;
    int rc = pgr_shmmrs_compute(ctx, b, &ix->spec, nullptr, 0, &s);  // index path: padding = false (seq_db.rs:462)
    if (rc) return rc;
    const uint64_t np = pgr_shmmrs_n_pairs(s);
    std::vector<uint32_t> auto_sids;
    if (!sids) {  // load_index_from_reader: running sid (seq_db.rs:543-553)
        auto_sids.resize(b->n);
        for (uint32_t i = 0; i < b->n; ++i) auto_sids[i] = ix->next_sid + i;
        sids = auto_sids.data();
    }
    rc = grow_raw(ctx, ix, ix->n_raw + np);
    uint64_t n_out = 0;
    if (!rc) rc = pgr_shmmrs_to_frag_recs_device(ctx, s, sids, 0, ix->raw + ix->n_raw, ix->cap_raw - ix->n_raw, &n_out);
    pgr_shmmrs_destroy(s);
    if (rc) return rc;
    ix->n_raw += n_out;
    uint32_t mx = ix->next_sid;
    for (uint32_t i = 0; i < b->n; ++i) mx = std::max(mx, sids[i] + 1);
    ix->next_sid = mx;
    ix->finalized = false;
    return PGR_OK;
}

static int index_add_host(pgr_ctx *ctx, pgr_index *ix, uint32_t n, const StageSrc &src, const uint32_t *sids) {
    if (worth_pipelining(ctx, n, src.lens))  // stage the next ~Gbp while this one is turned into records (running sids keep order)
        return for_each_staged(ctx, n, src, [&](pgr_batch *sb, uint32_t c0, uint32_t) {
            return pgr_index_add_resident(ctx, ix, sb, sids ? sids + c0 : nullptr);
        });
    pgr_batch *b = nullptr;
    int rc = src.planes ? pgr_batch_from_packed(ctx, n, src.lens, src.planes, src.valid, &b)
                        : pgr_batch_from_ascii(ctx, n, src.seqs, src.lens, &b);
    if (rc) return rc;
    rc = pgr_index_add_resident(ctx, ix, b, sids);
    pgr_batch_destroy(b);
    return rc;
}

extern "C" int pgr_index_add_batch(pgr_ctx *ctx, pgr_index *ix, uint32_t n, const uint8_t *const *seqs, const uint64_t *lens,
                                   const uint32_t *sids) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix) return ctx->fail(PGR_ERR_INVALID_ARG, "null index");
    if (n && (!seqs || !lens)) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    StageSrc src;
    src.seqs = seqs;
    src.lens = lens;
    return index_add_host(ctx, ix, n, src, sids);
}

extern "C" int pgr_index_add_packed(pgr_ctx *ctx, pgr_index *ix, uint32_t n, const uint64_t *lens, const uint64_t *planes,
                                    const uint32_t *valid, const uint32_t *sids) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix) return ctx->fail(PGR_ERR_INVALID_ARG, "null index");
    if (n && !lens) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (pgr_packed_words(n, lens) && !planes) return ctx->fail(PGR_ERR_INVALID_ARG, "null plane array");
    static const uint64_t no_words = 0;
    StageSrc src;
    src.lens = lens;
    src.planes = planes ? planes : &no_words;
    src.valid = valid;
    return index_add_host(ctx, ix, n, src, sids);
}

extern "C" int pgr_index_finalize(pgr_ctx *ctx, pgr_index *ix) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix) return ctx->fail(PGR_ERR_INVALID_ARG, "null index");
    if (ix->finalized) return PGR_OK;
    PGR_ENTER(ctx);
    hipStream_t st = ctx->stream;
    const uint64_t n = ix->n_raw;
    if (n >= (1ull << 32)) return ctx->fail(PGR_ERR_INVALID_ARG, "index holds more than 2^32-1 records per GPU");
    ctx->dfree(ix->recs);
    ctx->dfree(ix->key_off);
    ctx->dfree(ix->lut);
    ctx->dfree(ix->keys);
    ctx->dfree(ix->qkeys);
    ix->recs = nullptr;
    ix->key_off = nullptr;
    ix->lut = nullptr;
    ix->keys = nullptr;
    ix->qkeys = nullptr;
    ix->n = n;
    ix->n_keys = 0;
    int rc;
    if ((rc = ctx->dmalloc((void **)&ix->recs, std::max<uint64_t>(n, 1) * sizeof(pgr_frag_rec)))) return rc;
    if (n == 0) {
        if ((rc = ctx->dmalloc((void **)&ix->key_off, sizeof(uint64_t)))) return rc;
        PGR_HIP(ctx, hipMemsetAsync(ix->key_off, 0, sizeof(uint64_t), st));
        PGR_HIP(ctx, hipStreamSynchronize(st));
        ix->finalized = true;
        return PGR_OK;
    }
    Tmp idx_a(ctx), idx_b(ctx), keys_a(ctx), keys_b(ctx), h1_app(ctx), flags(ctx), rank(ctx);
    if ((rc = idx_a.alloc(n * 4)) || (rc = idx_b.alloc(n * 4)) || (rc = keys_a.alloc(n * 8)) || (rc = keys_b.alloc(n * 8)) ||
        (rc = h1_app.alloc(n * 8)))
        return rc;
    hipLaunchKernelGGL(iota_kernel, grid_for(n), dim3(256), 0, st, idx_a.as<uint32_t>(), n);
    // one look at the appended records: append order, key order, largest sid, largest hash (and h0 / h1 by themselves)
    //   records appended in (sid, frg_id) order need no passes over the ids (64 of 176 key bits less);
    //   records already in full key order (concatenated key-range shards, csrc/exchange.hip) need no sort at all;
    //   the largest hash bounds the radix passes (shimmer hashes are minima of minima: a few bits below 2^56)
    Tmp d_stats(ctx);
    if ((rc = d_stats.alloc(32))) return rc;
    uint64_t stats[4] = {1, 1, 0, 0};
    PGR_HIP(ctx, hipMemsetAsync(d_stats.p, 0, 32, st));
    hipLaunchKernelGGL(raw_stats_kernel, dim3((uint32_t)std::min<uint64_t>(2048, (n + 255) / 256)), dim3(256), 0, st, ix->raw, n,
                       d_stats.as<unsigned long long>(), keys_a.as<uint64_t>(), h1_app.as<uint64_t>());
    PGR_HIP(ctx, hipMemcpyAsync(stats, d_stats.p, 32, hipMemcpyDeviceToHost, st));
    PGR_HIP(ctx, hipStreamSynchronize(st));
    ix->sid_bound = stats[2];
    const bool full_sort = ctx->opt.index_full_sort != 0;
    if (!stats[1] && !full_sort) {
        // already in (h0, h1, sid, frg_id) order
        PGR_HIP(ctx, hipMemcpyAsync(ix->recs, ix->raw, n * sizeof(pgr_frag_rec), hipMemcpyDeviceToDevice, st));
    } else {
        // the library's own hashes have 56 bits (MM128.x >> 8); external records may carry anything
        const unsigned need = stats[3] >> 56 ? 64u : std::max(1u, bits_for(stats[3] + 1));
        const unsigned hb = full_sort ? std::max(56u, need) : need;
        bool sorted = false;
        if (!stats[0] && !full_sort && !ctx->opt.index_two_key_sort) {
            // append-ordered records: ONE stable radix sort by h0, then the runs of equal h0 are put in h1 order (see above)
            const size_t tb = sort_pairs_temp_bytes(n);
            Tmp h1s(ctx), longs(ctx), cnt(ctx);
            if ((rc = ctx->ws_scan_tmp.ensure(ctx, tb)) || (rc = h1s.alloc(n * 8)) ||
                (rc = longs.alloc((n / (RUN_SMALL + 1) + 64) * sizeof(uint2))) || (rc = cnt.alloc(16)))
                return rc;
            PGR_HIP(ctx, sort_pairs(st, ctx->ws_scan_tmp.p, tb, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), idx_a.as<uint32_t>(),
                                    idx_b.as<uint32_t>(), n, hb));
            hipLaunchKernelGGL(gather_h1_kernel, grid_for(n), dim3(256), 0, st, idx_b.as<uint32_t>(), h1_app.as<uint64_t>(), n,
                               h1s.as<uint64_t>());
            PGR_HIP(ctx, hipMemsetAsync(cnt.p, 0, 16, st));
            hipLaunchKernelGGL(run_rank_kernel, grid_for(n), dim3(256), 0, st, keys_b.as<uint64_t>(), h1s.as<uint64_t>(),
                               idx_b.as<uint32_t>(), n, idx_a.as<uint32_t>(), longs.as<uint2>(), cnt.as<uint32_t>());
            uint32_t c2[2] = {0, 0};
            PGR_HIP(ctx, hipMemcpyAsync(c2, cnt.p, 8, hipMemcpyDeviceToHost, st));
            PGR_HIP(ctx, hipStreamSynchronize(st));
            if (!c2[1]) {
                if (c2[0])
                    hipLaunchKernelGGL(run_long_kernel, dim3(c2[0]), dim3(256), 0, st, longs.as<uint2>(), h1s.as<uint64_t>(),
                                       idx_b.as<uint32_t>(), idx_a.as<uint32_t>());
                sorted = true;
            } else {  // a run of thousands of records with one h0 (a repeat family): the two-key sort below
                hipLaunchKernelGGL(iota_kernel, grid_for(n), dim3(256), 0, st, idx_a.as<uint32_t>(), n);
            }
        }
        if (!sorted) {
            const int fields[4] = {0, 1, 2, 3};  // frg_id, sid, h1, h0 (least significant first)
            const unsigned bits[4] = {32, 32, hb, hb};
            const int skip = (stats[0] || full_sort) ? 0 : 2;
            if ((rc = sort_perm(ctx, ix->raw, n, fields + skip, bits + skip, 4 - skip, idx_a.as<uint32_t>(), idx_b.as<uint32_t>(),
                                keys_a.as<uint64_t>(), keys_b.as<uint64_t>())))
                return rc;
        }
        hipLaunchKernelGGL(gather_recs_kernel, grid_for(n), dim3(256), 0, st, ix->raw, idx_a.as<uint32_t>(), ix->recs, n);
    }
    // distinct keys -> key_off
    if ((rc = flags.alloc((n + 1) * 4)) || (rc = rank.alloc((n + 1) * 8))) return rc;
    hipLaunchKernelGGL(key_flags_kernel, grid_for(n + 1), dim3(256), 0, st, ix->recs, n, flags.as<uint32_t>());
    const size_t tb = scan_counts_temp_bytes((uint32_t)(n + 1));
    if ((rc = ctx->ws_scan_tmp.ensure(ctx, tb))) return rc;
    PGR_HIP(ctx, scan_counts(st, ctx->ws_scan_tmp.p, tb, flags.as<uint32_t>(), rank.as<uint64_t>(), (uint32_t)(n + 1)));
    uint64_t n_keys = 0;
    PGR_HIP(ctx, hipMemcpyAsync(&n_keys, rank.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
    PGR_HIP(ctx, hipStreamSynchronize(st));
    if ((rc = ctx->dmalloc((void **)&ix->key_off, (n_keys + 1) * sizeof(uint64_t)))) return rc;
    hipLaunchKernelGGL(scatter_starts_kernel, grid_for(n + 1), dim3(256), 0, st, flags.as<uint32_t>(),
                       rank.as<uint64_t>(), n, ix->key_off);
    ix->n_keys = n_keys;
    // bucket table for the lookups (indexes of >= 4096 keys): ~4 keys per bucket on average, scaled to the key at the
    // 99th percentile (the rest shares the last bucket)
    if (n_keys >= 4096) {
        uint64_t k99 = 0, h99 = 0;
        PGR_HIP(ctx, hipMemcpyAsync(&k99, ix->key_off + (n_keys - 1 - n_keys / 100), 8, hipMemcpyDeviceToHost, st));
        PGR_HIP(ctx, hipStreamSynchronize(st));
        PGR_HIP(ctx, hipMemcpyAsync(&h99, &ix->recs[k99].h0, 8, hipMemcpyDeviceToHost, st));
        PGR_HIP(ctx, hipStreamSynchronize(st));
        // (lut_extra_bits: round 6 -- window minima crowd towards small hashes, so the equal-width buckets near zero hold several
        // times the average; profiles/r06_query/lut_bits.txt)
        const unsigned xb = (unsigned)std::max<int64_t>(0, std::min<int64_t>(4, ctx->opt.lut_extra_bits));
        const unsigned bits = std::min(22u + xb, std::max(4u, bits_for(n_keys) - 2 + xb));
        const unsigned hb = bits_for(h99 + 1);  // bits of the largest bucketed h0
        ix->lut_bits = bits;
        ix->lut_shift = hb > bits ? hb - bits : 0;
        if ((rc = ctx->dmalloc((void **)&ix->lut, ((1ull << bits) + 1) * sizeof(uint32_t))) ||
            (rc = ctx->dmalloc((void **)&ix->keys, n_keys * sizeof(ulonglong2))))
            return rc;
        hipLaunchKernelGGL(gather_keys_kernel, grid_for(n_keys), dim3(256), 0, st, ix->recs, ix->key_off, n_keys, ix->keys);
        if (!(stats[3] >> 56) && !ctx->opt.no_query_keys) {  // (the table's flag bits sit above the library's 56-bit hashes)
            if ((rc = ctx->dmalloc((void **)&ix->qkeys, n_keys * sizeof(ulonglong4)))) return rc;
            hipLaunchKernelGGL(gather_qkeys_kernel, grid_for(n_keys), dim3(256), 0, st, ix->recs, ix->key_off, n_keys, ix->qkeys);
        }
        hipLaunchKernelGGL(build_lut_kernel, grid_for((1ull << bits) + 1), dim3(256), 0, st, ix->recs, ix->key_off, n_keys, bits,
                           ix->lut_shift, ix->lut);
    }
    PGR_HIP(ctx, hipStreamSynchronize(st));
    PGR_HIP(ctx, hipGetLastError());
    ix->finalized = true;
    return PGR_OK;
}

extern "C" int pgr_index_download(pgr_ctx *ctx, const pgr_index *ix, pgr_frag_rec **out, uint64_t *n) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || !out || !n) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (!ix->finalized) return ctx->fail(PGR_ERR_STATE, "index not finalized");
    *out = (pgr_frag_rec *)host_result_alloc(std::max<uint64_t>(ix->n, 1) * sizeof(pgr_frag_rec));
    if (!*out) return ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
    *n = ix->n;
    if (ix->n) {
        const int rc = ctx->d2h(*out, ix->recs, ix->n * sizeof(pgr_frag_rec));
        if (rc) {
            free(*out);
            *out = nullptr;
            return rc;
        }
    }
    return PGR_OK;
}

// ================================================================================================
// query path
namespace {

// raw_query_fragment lookup: [lo, hi) = records of the query pair's key (empty when absent).  Keys are window minima of a
// hash: nearly all of them lie in the lowest few percent of the 56-bit range, where the bucket table (pgr_index.h) cuts
// the binary search over all keys (25 dependent steps of two loads for 3x10^7 keys) down to the few keys of one bucket.
__global__ void lookup_kernel(const pgr_frag_rec *__restrict__ q, uint64_t nq, const pgr_frag_rec *__restrict__ recs,
                              const uint64_t *__restrict__ key_off, uint64_t n_keys, const uint32_t *__restrict__ lut,
                              uint32_t lut_bits, uint32_t lut_shift, const ulonglong2 *__restrict__ keys,
                              uint64_t *__restrict__ lo_out, uint64_t *__restrict__ hi_out) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nq) return;
    uint64_t a, b;
    lookup_range(q[p].h0, q[p].h1, recs, key_off, n_keys, lut, lut_bits, lut_shift, keys, a, b);
    lo_out[p] = a;
    hi_out[p] = b;
}

// shmmr_pair_hash_count (aln.rs:180-181): number of pairs of the same query with the same key.
// sorted: permutation of the query pairs ordered by (query, h0, h1); one thread per run start.
__global__ void run_count_kernel(const pgr_frag_rec *__restrict__ q, const uint32_t *__restrict__ sorted, uint64_t nq,
                                 uint32_t *__restrict__ count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const pgr_frag_rec &a = q[sorted[i]];
    if (i > 0) {
        const pgr_frag_rec &p = q[sorted[i - 1]];
        if (p.sid == a.sid && p.h0 == a.h0 && p.h1 == a.h1) return;  // not a run start
    }
    uint64_t j = i + 1;
    while (j < nq) {
        const pgr_frag_rec &b = q[sorted[j]];
        if (b.sid != a.sid || b.h0 != a.h0 || b.h1 != a.h1) break;
        ++j;
    }
    const uint32_t len = (uint32_t)(j - i);
    for (uint64_t t = i; t < j; ++t) count[sorted[t]] = len;
}

// the same for queries with few pairs, without sorting: the pairs of one query are contiguous (q_off = record offsets
// per query), every thread compares its pair with the others of its query
__global__ void pair_count_kernel(const pgr_frag_rec *__restrict__ q, const uint64_t *__restrict__ q_off, uint64_t nq,
                                  uint32_t *__restrict__ count) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nq) return;
    const pgr_frag_rec a = q[p];
    const uint64_t lo = q_off[a.sid], hi = q_off[a.sid + 1];
    uint32_t c = 0;
    for (uint64_t i = lo; i < hi; ++i) c += (q[i].h0 == a.h0 && q[i].h1 == a.h1) ? 1u : 0u;
    count[p] = c;
}

// aln.rs:197-228: number of hits a query pair contributes (mode 0) or the hits themselves (mode 1).
// Records of one key are sorted by sid, so target_shmer_pair_count[(key,sid)] = count * run length.
// A pair whose key holds up to HITS_HEAVY records is walked by its own thread.  A key of a repeat can hold 10^5 records in a
// pangenome index (a serial thread pays ~0.4 us per record): those pairs are taken by the whole wavefront afterwards, 64
// records per step -- a lane at the start of a sid run finds the run's end by binary search (the run either passes the
// target filter as a whole, and is then at most max_count_target long, or is skipped as a whole).
__device__ __forceinline__ void emit_hit(const pgr_frag_rec &qp, const pgr_frag_rec &r, uint64_t o, uint64_t *__restrict__ hit_key,
                                         pgr_hitpair *__restrict__ hit_hp) {
    pgr_hitpair h;
    h.qb = qp.bgn;
    h.qe = qp.end;
    h.qo = qp.orient;
    h.tb = r.bgn;
    h.te = r.end;
    h.to = r.orient;
    hit_hp[o] = h;
    hit_key[o] = ((uint64_t)qp.sid << 32) | r.sid;  // (query, target)
}

__global__ __launch_bounds__(256) void hits_kernel(const pgr_frag_rec *__restrict__ q, uint64_t nq,
                                                   const uint32_t *__restrict__ count, const uint64_t *__restrict__ lo,
                                                   const uint64_t *__restrict__ hi, const pgr_frag_rec *__restrict__ recs,
                                                   QParams prm, int mode, uint32_t *__restrict__ n_hits,
                                                   const uint64_t *__restrict__ hit_off, uint64_t *__restrict__ hit_key,
                                                   pgr_hitpair *__restrict__ hit_hp) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    // (no early return: the lanes of a wavefront work together on its heavy pairs below)
    if (p == nq && mode == 0) n_hits[nq] = 0;
    const bool live = p < nq;
    const uint32_t c = live ? count[p] : 0;
    const bool pass = live && c <= prm.max_count && c <= prm.max_count_query;
    const uint64_t s0 = pass ? lo[p] : 0, e0 = pass ? hi[p] : 0;
    const bool heavy = e0 - s0 > HITS_HEAVY;
    uint32_t n = 0;
    if (pass && !heavy) {
        const pgr_frag_rec qp = q[p];
        uint64_t o = mode ? hit_off[p] : 0;
        uint64_t s = s0;
        while (s < e0) {
            const uint32_t sid = recs[s].sid;
            uint64_t t = s + 1;
            while (t < e0 && recs[t].sid == sid) ++t;
            if ((uint64_t)(t - s) * c <= prm.max_count_target) {
                if (mode)
                    for (uint64_t u = s; u < t; ++u) emit_hit(qp, recs[u], o++, hit_key, hit_hp);
                n += (uint32_t)(t - s);
            }
            s = t;
        }
    }
    for (uint64_t hm = __ballot(heavy); hm; hm &= hm - 1) {  // wave-uniform loop over the heavy pairs of these 64
        const int src = __builtin_ctzll(hm);
        const uint64_t ps = shfl64(s0, src), pe = shfl64(e0, src);
        const uint32_t pc = (uint32_t)__shfl((int)c, src, 64);
        const uint64_t pp = p - lane + (uint64_t)src;  // index of that pair
        const pgr_frag_rec qp = q[pp];
        uint64_t kept = 0;  // kept records of the pair so far (all lanes hold the same value)
        const uint64_t obase = mode ? hit_off[pp] : 0;
        for (uint64_t b0 = ps; b0 < pe; b0 += 64) {
            const uint64_t u = b0 + lane;
            uint32_t run = 0;  // > 0: a sid run that passes the filter starts at u and is `run` records long
            if (u < pe) {
                const uint32_t sid = recs[u].sid;
                if (u == ps || recs[u - 1].sid != sid) {
                    uint64_t a = u + 1, bnd = pe;  // first index in (u, pe] whose sid differs (sids ascend inside a key)
                    while (a < bnd) {
                        const uint64_t mid = (a + bnd) >> 1;
                        if (recs[mid].sid == sid) a = mid + 1;
                        else bnd = mid;
                    }
                    const uint64_t len = a - u;
                    if (len * pc <= prm.max_count_target) run = (uint32_t)len;
                }
            }
            const uint32_t incl = wave_incl_sum(run);
            if (mode && run) {
                uint64_t o = obase + kept + (incl - run);
                for (uint32_t k = 0; k < run; ++k) emit_hit(qp, recs[u + k], o++, hit_key, hit_hp);
            }
            kept += (uint32_t)__shfl((int)incl, 63, 64);
        }
        if ((int)lane == src) n = (uint32_t)kept;
    }
    if (mode == 0 && live) n_hits[p] = n;
}

// field 0: qb, 1: the group key (query << 32 | sid) squeezed to (query << sid_bits | sid): fewer radix passes
__global__ void hit_key_kernel(const uint64_t *__restrict__ hit_key, const pgr_hitpair *__restrict__ hp,
                               const uint32_t *__restrict__ idx, int field, unsigned sid_bits, uint64_t *__restrict__ keys,
                               uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = idx[i];
    const uint64_t k = hit_key[j];
    keys[i] = field == 0 ? (uint64_t)hp[j].qb : (((k >> 32) << sid_bits) | (k & 0xFFFFFFFFull));
}

__global__ void gather_hits_kernel(const uint64_t *__restrict__ hit_key, const pgr_hitpair *__restrict__ hp,
                                   const uint32_t *__restrict__ idx, uint64_t n, uint64_t *__restrict__ key_out,
                                   pgr_hitpair *__restrict__ hp_out, uint32_t *__restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) {
        flags[n] = 0;
        return;
    }
    const uint32_t j = idx[i];
    const uint64_t k = hit_key[j];
    key_out[i] = k;
    hp_out[i] = hp[j];
    flags[i] = (i == 0 || hit_key[idx[i - 1]] != k) ? 1u : 0u;
}

// ---- grouping of the hits of the query path without a global sort.  hits_kernel emits the hits pair by pair, so the hits
// of one query are contiguous ([q_hit_lo(q), q_hit_lo(q+1)) = hit_off[pair_off[q] ..]) and in query-position order; the
// group key is (query, target sid): a STABLE sort by sid inside every query's segment is the whole job (aln.rs:21 wants the
// hits of a group in ascending query bgn = the order they already have).  One wavefront per query, ranks by counting in
// LDS (O(m^2 / 64) per query of m hits: m is a few dozen to a few thousand).  Queries with more hits than SEG_SORT_MAX
// send the batch to the global radix sort instead (the host knows the maximum from query_hit_max_kernel).
constexpr uint32_t SEG_SORT_MAX = 4096;

__global__ void query_hit_max_kernel(const uint64_t *__restrict__ pair_off, const uint64_t *__restrict__ hit_off,
                                     uint32_t n_queries, unsigned long long *__restrict__ out_max) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t m = 0;
    if (q < n_queries) m = hit_off[pair_off[q + 1]] - hit_off[pair_off[q]];
    for (int off = 32; off; off >>= 1) {
        const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out_max, (unsigned long long)m);
}

// blockDim.x = 64 (queries of up to a few hundred hits: one wavefront each) or 256 (longer ones)
__global__ __launch_bounds__(256) void group_sort_kernel(const uint64_t *__restrict__ hit_key, const pgr_hitpair *__restrict__ hp,
                                                         const uint64_t *__restrict__ pair_off,
                                                         const uint64_t *__restrict__ hit_off, uint64_t n,
                                                         uint64_t *__restrict__ key_out, pgr_hitpair *__restrict__ hp_out,
                                                         uint32_t *__restrict__ flags) {
    __shared__ __attribute__((aligned(16))) uint32_t sid[SEG_SORT_MAX];
    __shared__ uint32_t srt[SEG_SORT_MAX];
    const uint32_t q = blockIdx.x, t = threadIdx.x, T = blockDim.x;
    if (q == 0 && t == 0) flags[n] = 0;  // sentinel of the scan over the group starts
    const uint64_t s = hit_off[pair_off[q]];
    const uint32_t m = (uint32_t)(hit_off[pair_off[q + 1]] - s);
    if (m == 0 || m > SEG_SORT_MAX) return;  // (m > SEG_SORT_MAX: the host does not take this path)
    for (uint32_t i = t; i < m; i += T) sid[i] = (uint32_t)hit_key[s + i];
    __syncthreads();
    for (uint32_t i0 = 0; i0 < m; i0 += T) {
        const uint32_t i = i0 + t;
        const uint32_t mine = i < m ? sid[i] : 0xFFFFFFFFu;
        uint32_t rank = 0;
        // elements before the block of i count when <=, elements behind it when <, the block itself with the index
        // compare (all lanes read the same LDS words: broadcasts, four words per read where the range is aligned)
        const uint32_t split = i0 + T < m ? i0 + T : m;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(sid);
        for (uint32_t j = 0; j < i0 / 4; ++j) {  // i0 is a multiple of 64
            const uint4 v = s4[j];
            rank += (v.x <= mine ? 1u : 0u) + (v.y <= mine ? 1u : 0u) + (v.z <= mine ? 1u : 0u) + (v.w <= mine ? 1u : 0u);
        }
        for (uint32_t j = i0; j < split; ++j) rank += (sid[j] < mine || (sid[j] == mine && j < i)) ? 1u : 0u;
        uint32_t j = split;
        if (split == i0 + T)  // aligned: full groups of four first
            for (; j + 4 <= m; j += 4) {
                const uint4 v = s4[j / 4];
                rank += (v.x < mine ? 1u : 0u) + (v.y < mine ? 1u : 0u) + (v.z < mine ? 1u : 0u) + (v.w < mine ? 1u : 0u);
            }
        for (; j < m; ++j) rank += sid[j] < mine ? 1u : 0u;
        if (i < m) {
            srt[rank] = mine;
            key_out[s + rank] = hit_key[s + i];
            hp_out[s + rank] = hp[s + i];
        }
    }
    __syncthreads();
    for (uint32_t r = t; r < m; r += T) flags[s + r] = (r == 0 || srt[r - 1] != srt[r]) ? 1u : 0u;
}

constexpr int ALN_WAVE_MIN = 64;    // groups with at least this many hits are chained by a whole wavefront ...
constexpr int ALN_WAVE_MIN_FEW = 16;  // ... from this size already when the call has few groups (latency, not throughput)
constexpr uint64_t ALN_FEW_GROUPS = 4096;
constexpr int ALN_LDS_SMALL = 256;  // ... in a 9 KB LDS image (many workgroups per CU) up to this many hits,
constexpr int ALN_LDS_MAX = 3584;   // in a 129 KB image up to this many (36 B per hit), in global memory above
constexpr int ALN_ROW_LDS = 2048;   // hits of the (short) groups of one wavefront of sparse_aln_kernel staged in LDS

// The dynamic programme of aln::sparse_aln (aln.rs:25-103) for one group, run by ONE thread.  h / vs / pv / sl are the
// group's hits and work arrays -- in LDS for the short groups of the query path, in global memory otherwise (generic
// pointers).  span[] is the span set of the current look-back (aln.rs:70, :91) as a list of candidate INDICES: the
// candidates come in descending index = non-increasing query bgn order, so an interval that is already in the set can only
// sit in the trailing entries with the same bgn.
template <class IdxT, class H, class F, class I, class S>
__device__ __forceinline__ void aln_dp_thread(H h, int n, F vs, I pv, I sl, S span, const AlnParams &prm) {
    for (int i = 0; i < n; ++i) {
        int s = i;
        for (int j = i - 1; j >= 0 && h[j].qb == h[i].qb; --j)
            if (same_hp(h[j], h[i])) {
                s = sl[j];
                break;
            }
        sl[i] = s;
    }
    vs[sl[0]] = (float)h[0].qe - (float)h[0].qb;  // aln.rs:25-27
    pv[sl[0]] = -1;
    for (int i = 1; i < n; ++i) {  // aln.rs:29-103
        const pgr_hitpair cur = h[i];
        const float cur_len = (float)cur.qe - (float)cur.qb;
        int best_v = -1;
        float best_s = 0.0f;
        uint32_t span_n = 0, t_qb = 0, t_qe = 0, t_qo = 0;
        for (int j = i - 1; j >= 0; --j) {
            const pgr_hitpair p = h[j];
            if (prm.oriented && ((p.qo ^ p.to) != (cur.qo ^ cur.to))) continue;  // :43-50
            float a = (float)cur.qb - (float)p.qe;
            float b = (cur.qo == cur.to) ? ((float)cur.tb - (float)p.te) : ((float)cur.te - (float)p.tb);
            a = absf(a);
            b = absf(b);
            if (prm.has_max_gap) {  // :52-65
                const float mg = (float)prm.max_gap;
                if (a > mg || b > mg) continue;
            }
            if (same_q(p, cur)) continue;  // :67
            bool found = false;            // :70 (the newest entry is kept in registers: most checks end there)
            if (span_n && p.qb == t_qb) {
                if (p.qe == t_qe && p.qo == t_qo) {
                    found = true;
                } else {
                    for (int t = (int)span_n - 2; t >= 0; --t) {
                        const pgr_hitpair e = h[span[t]];
                        if (e.qb != p.qb) break;
                        if (e.qe == p.qe && e.qo == p.qo) {
                            found = true;
                            break;
                        }
                    }
                }
            }
            if (!found) {
                span[span_n++] = (IdxT)j;
                t_qb = p.qb;
                t_qe = p.qe;
                t_qo = p.qo;
            }
            const int slj = sl[j];
            const float p_s = vs[slj];      // :71
            float s = p_s + cur_len;        // :72
            const float sum = a + b;        // :74-84
            const float pen = prm.penalty * sum;
            s = s - pen;
            if (s > best_s) {  // :86-89
                best_s = s;
                best_v = slj;
            }
            if (span_n >= prm.max_span) break;  // :91
        }
        const int si = sl[i];
        if (best_s > 0.0f) {  // :96-102
            vs[si] = best_s;
            pv[si] = best_v;
        } else {
            vs[si] = cur_len;
            pv[si] = -1;
        }
    }
}

// chain extraction (aln.rs:105-140) by one thread.  A visited value-slot is marked by sl[v] = -1 - v (the DP is done, so sl
// is free to carry the flag); unvisited representatives have sl[i] == i.  Returns true when only non-positive scores are
// left (aln.rs:129-131 would spin forever: the group ends there).
template <class H, class F, class I>
__device__ __forceinline__ bool aln_extract_thread(H h, int n, F vs, I pv, I sl, pgr_hitpair *__restrict__ o_hp,
                                                   uint32_t *__restrict__ o_len, float *__restrict__ o_score,
                                                   uint32_t &n_ch, uint32_t &n_out) {
    int n_unvisited = 0;
    for (int i = 0; i < n; ++i) n_unvisited += (sl[i] == i);
    while (n_unvisited > 0) {
        float best_s = 0.0f;
        int best_v = -1;
        for (int i = 0; i < n; ++i)
            if (sl[i] == i && vs[i] > best_s) {  // strict >, first (lowest sorted index) wins
                best_s = vs[i];
                best_v = i;
            }
        if (best_v < 0) return true;
        uint32_t len = 0;
        int v = best_v, first_v = best_v;
        while (v >= 0 && sl[v] == v) {  // :121-128 (counted back to front, written in chain order below)
            ++len;
            first_v = v;
            const int nv = pv[v];
            sl[v] = -1 - v;  // :133-137
            --n_unvisited;
            v = nv;
        }
        v = best_v;  // :132 reverse: the pv links are intact, walk again and store from the back
        for (uint32_t a = 0; a < len; ++a) {
            o_hp[n_out + (len - 1 - a)] = h[v];
            v = pv[v];
        }
        o_len[n_ch] = len;
        o_score[n_ch] = best_s - vs[first_v];  // :138-139
        ++n_ch;
        n_out += len;
    }
    return false;
}

// aln::sparse_aln (aln.rs:12-142), one thread per (query, target) group.  The hits of a group are
// already stably sorted by query bgn (aln.rs:21).  v_s / best_pre_v are FxHashMaps keyed by the
// HitPair VALUE in the reference, so identical hit pairs share one slot: slot[i] = first index with
// the same value.  Tie-break of the chain extraction (FxHashSet order in the reference, unspecified):
// lowest sorted index.  Outputs are written into the group's own range [gs, gs+n).
// Groups of fewer than ALN_WAVE_MIN hits (nearly all groups of a query batch) are chained here, their hits and work arrays
// staged in LDS: 64 independent serial programmes per wavefront cost ~60x fewer instructions than a wavefront per group,
// and LDS keeps the dependent accesses at ~100 cycles.  Longer groups go to sparse_aln_wave_kernel.
struct AlnRowLds {
    pgr_hitpair h[ALN_ROW_LDS];
    float vs[ALN_ROW_LDS];
    int sl[ALN_ROW_LDS];
    int pv[ALN_ROW_LDS];
    uint8_t span[64][ALN_WAVE_MIN];
};

__global__ __launch_bounds__(64) void sparse_aln_kernel(
    const pgr_hitpair *__restrict__ hp, const uint64_t *__restrict__ g_start, const uint64_t *__restrict__ n_groups_ptr,
    AlnParams prm, float *v_s, int *pre, int *slot, pgr_hitpair *__restrict__ out_hp, uint32_t *__restrict__ chain_len,
    float *__restrict__ chain_score, uint32_t *__restrict__ g_nchains, uint32_t *__restrict__ g_nhp,
    uint32_t *__restrict__ err, uint32_t *__restrict__ big_list, uint32_t *__restrict__ n_big, uint32_t cls_stride,
    uint32_t *__restrict__ span_buf /* max_span > MAX_SPAN_CAP: one word per hit */) {
    __shared__ AlnRowLds L;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (no early return before the ballots below: every lane of the wave takes part in them)
    const bool live = g < *n_groups_ptr;  // the grid is sized by the number of hits (an upper bound known without a round trip)
    const uint64_t gs = live ? g_start[g] : 0;
    const int n = live ? (int)(g_start[g + 1] - gs) : 0;
    if (live) {
        g_nchains[g] = 0;
        g_nhp[g] = 0;
    }
    // max_span above the LDS span set of the wave kernel (aln.rs:91 accepts any value): every group stays on this
    // one-thread path with its span set in global memory -- a span set never holds more entries than the group has hits
    // groups of >= ALN_WAVE_MIN hits: one wavefront each (sparse_aln_wave_kernel), two size classes.  The list slots are
    // claimed with ONE atomic per wavefront and class (same-address atomics run at ~88 per us on gfx950)
    // a thread chains a 30-hit group in ~80 us, a wavefront in ~25: with few groups in the call (a single query) the machine
    // is empty either way and the wavefront is quicker; with thousands of groups the threads win by ~60x fewer instructions
    const int wave_min = *n_groups_ptr < ALN_FEW_GROUPS ? ALN_WAVE_MIN_FEW : ALN_WAVE_MIN;
    const bool to_wave = n >= wave_min && span_buf == nullptr;
    const int cls = n > ALN_LDS_SMALL ? 1 : 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const uint64_t m = __ballot(to_wave && cls == c);
        if (m == 0) continue;
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(n_big + c, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(m), 64);
        if (to_wave && cls == c) big_list[(size_t)c * cls_stride + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)g;
    }
    const bool mine = !to_wave && n >= 2;  // n < 2: aln.rs:234, targets with a single hit are dropped (and lanes beyond the last group)
    if (__ballot(mine) == 0) return;
    // LDS placement of the short groups of this wavefront.  The groups of a wavefront are consecutive, so their hits are one
    // contiguous range of the sorted hit array: when that range fits it is copied as it lies (coalesced, independent loads);
    // otherwise (a long group in between, or many groups near the limit) the short groups are packed by an exclusive prefix
    // sum of their sizes and copied one group per step.
    const uint32_t want = (mine && n < ALN_WAVE_MIN) ? (uint32_t)n : 0u;
    const uint64_t live_m = __ballot(live);
    const int first_l = __builtin_ctzll(live_m), last_l = 63 - __builtin_clzll(live_m);  // (live_m != 0: some lane is `mine`)
    const uint64_t ge = gs + (uint64_t)n;
    const uint64_t r_lo = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(gs >> 32), first_l) << 32) |
                          (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)gs, first_l);
    const uint64_t r_hi = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ge >> 32), last_l) << 32) |
                          (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ge, last_l);
    uint32_t lds_base;
    bool in_lds;
    if (r_hi - r_lo <= (uint64_t)ALN_ROW_LDS) {  // wave-uniform
        const uint32_t R = (uint32_t)(r_hi - r_lo);
        for (uint32_t e = lane; e < R; e += 64) L.h[e] = hp[r_lo + e];
        lds_base = (uint32_t)(gs - r_lo);
        in_lds = want != 0;
    } else {
        lds_base = wave_incl_sum(want) - want;
        in_lds = want != 0 && lds_base + want <= (uint32_t)ALN_ROW_LDS;
        for (uint64_t m = __ballot(in_lds); m; m &= m - 1) {
            const int src = __builtin_ctzll(m);
            const uint64_t gs_l = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(gs >> 32), src) << 32) |
                                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)gs, src);
            const uint32_t n_l = (uint32_t)__builtin_amdgcn_readlane(n, src);
            const uint32_t b_l = (uint32_t)__builtin_amdgcn_readlane((int)lds_base, src);
            if (lane < n_l) L.h[b_l + lane] = hp[gs_l + lane];
        }
    }
    __syncthreads();
    if (!mine) return;
    // two call sites so that each one sees pointers of ONE address space (LDS reads and writes become ds_ instructions; a
    // pointer that may be either costs a flat access with several times the latency on every dependent step)
    uint32_t n_ch = 0, n_out = 0;
    bool stuck = false;
    if (in_lds) {
        aln_dp_thread<uint8_t>(&L.h[lds_base], n, &L.vs[lds_base], &L.pv[lds_base], &L.sl[lds_base], L.span[lane], prm);
        stuck = aln_extract_thread(&L.h[lds_base], n, &L.vs[lds_base], &L.pv[lds_base], &L.sl[lds_base], out_hp + gs,
                                   chain_len + gs, chain_score + gs, n_ch, n_out);
    } else {
        if (span_buf) {
            aln_dp_thread<uint32_t>(hp + gs, n, v_s + gs, pre + gs, slot + gs, span_buf + gs, prm);
        } else {
            uint32_t span_loc[MAX_SPAN_CAP];
            aln_dp_thread<uint32_t>(hp + gs, n, v_s + gs, pre + gs, slot + gs, span_loc, prm);
        }
        stuck = aln_extract_thread(hp + gs, n, v_s + gs, pre + gs, slot + gs, out_hp + gs, chain_len + gs, chain_score + gs,
                                   n_ch, n_out);
    }
    if (stuck) atomicAdd(err, 1u);
    g_nchains[g] = n_ch;
    g_nhp[g] = n_out;
}

// ------------------------------------------------------------------------------------------------
// sparse_aln for ONE long group per wavefront.  Same recurrence, same f32 operation order and the same tie
// rules as sparse_aln_kernel; what changes is where the latency goes: the look-back of a hit evaluates 64
// candidates per step in parallel (the "stop after max_span distinct query intervals" rule becomes ballots in
// candidate order), the arg-max scans of the extraction are lane-strided, and groups of up to ALN_LDS_MAX hits
// live in LDS for the whole computation (a serial thread pays ~1 us of HBM/L2 latency per dependent access).
template <int NMAX>
struct AlnWaveLds {
    pgr_hitpair h[NMAX];
    float vs[NMAX];
    int sl[NMAX];
    int pv[NMAX];
    uint32_t span_q[MAX_SPAN_CAP][3];
    uint32_t cand[4][64];  // qb, qe, qo, considered-flag of the 64 candidates of the current look-back batch
};

template <int NMAX>
__global__ __launch_bounds__(64) void sparse_aln_wave_kernel(
    const pgr_hitpair *__restrict__ hp, const uint64_t *__restrict__ g_start, const uint32_t *__restrict__ big_list,
    const uint32_t *__restrict__ n_big, AlnParams prm, float *v_s, int *pre, int *slot, int *track, pgr_hitpair *out_hp,
    uint32_t *__restrict__ chain_len, float *__restrict__ chain_score, uint32_t *__restrict__ g_nchains,
    uint32_t *__restrict__ g_nhp, uint32_t *__restrict__ err) {
    static_assert(NMAX % 64 == 0, "the LDS image is also used as a ring of 64-hit blocks");
    __shared__ AlnWaveLds<NMAX> L;
    // the grid is small and fixed (the number of long groups is only known on the device): every workgroup takes the
    // groups blockIdx.x, blockIdx.x + gridDim.x, ...
    for (uint32_t gi = blockIdx.x; gi < *n_big; gi += gridDim.x) {
    const uint32_t g = big_list[gi];
    const uint64_t gs = g_start[g];
    const int n = (int)(g_start[g + 1] - gs);
    const int lane = (int)threadIdx.x;
    // in_lds: the whole group lives in LDS.  Otherwise the arrays stay in global memory and LDS is a RING over the
    // last NMAX hits (blocks of 64 are loaded as the DP reaches them): the look-back of a hit almost always stays
    // inside it; index k is in the ring iff k >= ring_lo.
    const bool in_lds = n <= NMAX;
    const pgr_hitpair *gh = hp + gs;
    float *gvs = v_s + gs;
    int *gsl = slot + gs, *gpv = pre + gs, *gtrack = track + gs;
    int ring_lo = 0;
    auto H = [&](int k) -> pgr_hitpair { return (in_lds || k >= ring_lo) ? L.h[in_lds ? k : k % NMAX] : gh[k]; };
    auto SL = [&](int k) -> int { return (in_lds || k >= ring_lo) ? L.sl[in_lds ? k : k % NMAX] : gsl[k]; };
    auto VS = [&](int k) -> float { return (in_lds || k >= ring_lo) ? L.vs[in_lds ? k : k % NMAX] : gvs[k]; };
    auto load_block = [&](int blk) {  // hits [blk, blk + 64) -> ring; everything below blk + 64 - NMAX drops out
        const int k = blk + lane;
        if (k < n) {
            L.h[k % NMAX] = gh[k];
            L.sl[k % NMAX] = gsl[k];
        }
        ring_lo = blk + 64 > NMAX ? blk + 64 - NMAX : 0;
    };

    if (in_lds)
        for (int i = lane; i < n; i += 64) L.h[i] = gh[i];
    wave_sync();
    // value slots: the earliest identical hit pair of the (equal qb) run
    for (int i = lane; i < n; i += 64) {
        int s = i;
        const pgr_hitpair hi = in_lds ? L.h[i] : gh[i];
        for (int j = i - 1; j >= 0; --j) {
            const pgr_hitpair hj = in_lds ? L.h[j] : gh[j];
            if (hj.qb != hi.qb) break;
            if (same_hp(hj, hi)) s = j;
        }
        if (in_lds) L.sl[i] = s;
        else gsl[i] = s;
    }
    wave_sync();
    if (!in_lds) {
        load_block(0);
        wave_sync();
    }
    if (lane == 0) {  // aln.rs:25-27
        const pgr_hitpair h0 = H(0);
        const int s0 = SL(0);
        const float v0 = (float)h0.qe - (float)h0.qb;
        if (in_lds) {
            L.vs[s0] = v0;
            L.pv[s0] = -1;
        } else {
            L.vs[s0 % NMAX] = v0;
            gvs[s0] = v0;
            gpv[s0] = -1;
        }
    }
    wave_sync();
    for (int i = 1; i < n; ++i) {  // aln.rs:29-103
        if (!in_lds && (i & 63) == 0) {
            load_block(i);
            wave_sync();
        }
        const pgr_hitpair cur = H(i);
        const float cur_len = (float)cur.qe - (float)cur.qb;
        float best_s = 0.0f;
        int best_v = -1;
        uint32_t span_n = 0;
        bool stop = false;
        for (int jb = i - 1; jb >= 0 && !stop; jb -= 64) {
            const int j = jb - lane;
            bool cons = j >= 0;
            const pgr_hitpair p = cons ? H(j) : cur;
            float a = 0.0f, b = 0.0f;
            if (cons) {
                if (prm.oriented && ((p.qo ^ p.to) != (cur.qo ^ cur.to))) cons = false;  // :43-50
                a = (float)cur.qb - (float)p.qe;
                b = (cur.qo == cur.to) ? ((float)cur.tb - (float)p.te) : ((float)cur.te - (float)p.tb);
                a = absf(a);
                b = absf(b);
                if (prm.has_max_gap) {  // :52-65
                    const float mg = (float)prm.max_gap;
                    if (a > mg || b > mg) cons = false;
                }
                if (same_q(p, cur)) cons = false;  // :67
            }
            const uint64_t cm = __ballot(cons);
            if (!cm) continue;
            // span_set (:70, :91) in candidate order, lane parallel.  A candidate opens a new distinct query interval
            // unless an earlier batch (span_q) or an earlier considered lane of this batch has the same interval;
            // equal intervals share their qb, and the candidates are sorted by qb, so the lanes to look at are the
            // run of equal qb right before this lane.
            L.cand[0][lane] = p.qb;
            L.cand[1][lane] = p.qe;
            L.cand[2][lane] = p.qo;
            L.cand[3][lane] = cons ? 1u : 0u;
            wave_sync();
            bool fresh = cons;
            for (uint32_t t = 0; t < span_n; ++t)  // only in the second and later batches of a look-back
                if (p.qb == L.span_q[t][0] && p.qe == L.span_q[t][1] && p.qo == L.span_q[t][2]) fresh = false;
            if (fresh)
                for (int l = lane - 1; l >= 0 && L.cand[0][l] == p.qb; --l)
                    if (L.cand[3][l] && L.cand[1][l] == p.qe && L.cand[2][l] == p.qo) {
                        fresh = false;
                        break;
                    }
            const uint64_t lt = lane ? (U64MAX >> (64 - lane)) : 0ull;
            const uint64_t nm = __ballot(fresh);
            const uint32_t before = (uint32_t)__popcll(nm & lt);  // distinct intervals opened by earlier lanes
            const uint32_t need = prm.max_span - span_n;           // >= 1: the look-back has not stopped yet
            const uint64_t sm = __ballot(fresh && before + 1 == need);
            const int stop_lane = sm ? __builtin_ctzll(sm) : 64;  // the candidate completing the span set is scored
            if (fresh && lane <= stop_lane) {
                L.span_q[span_n + before][0] = p.qb;
                L.span_q[span_n + before][1] = p.qe;
                L.span_q[span_n + before][2] = p.qo;
            }
            span_n += (uint32_t)__popcll(stop_lane < 63 ? nm & (U64MAX >> (63 - stop_lane)) : nm);
            wave_sync();  // span_q is read back in the next batch
            const bool proc = cons && lane <= stop_lane;
            float s = -INFINITY;
            int slj = 0;
            if (proc) {
                slj = SL(j);
                const float p_s = VS(slj);      // :71
                s = p_s + cur_len;              // :72
                const float sum = a + b;        // :74-84
                const float pen = prm.penalty * sum;
                s = s - pen;
            }
            // strict >, candidates in look-back order (:86-89): the batch maximum replaces the running best only
            // when strictly larger; among equal maxima the first lane (= nearest candidate) wins
            const float m = wave_max_f32(s);
            if (m > best_s) {
                const int l = __builtin_ctzll(__ballot(proc && s == m));
                best_s = m;
                best_v = __builtin_amdgcn_readlane(slj, l);
            }
            if (stop_lane < 64) stop = true;
        }
        if (lane == 0) {  // :96-102
            const int si = SL(i);
            const float v = best_s > 0.0f ? best_s : cur_len;
            const int pvv = best_s > 0.0f ? best_v : -1;
            if (in_lds) {
                L.vs[si] = v;
                L.pv[si] = pvv;
            } else {
                if (si >= ring_lo) L.vs[si % NMAX] = v;
                gvs[si] = v;
                gpv[si] = pvv;
            }
        }
        wave_sync();
    }
    // extraction (aln.rs:105-140); a visited value slot is marked by sl[v] = -1 - v.  From here on the global-mode
    // arrays are read from global memory (the ring is no longer meaningful).
    const float *xvs = in_lds ? L.vs : gvs;
    int *xsl = in_lds ? L.sl : gsl;
    int n_unvisited = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        n_unvisited += (int)__popcll(__ballot(i < n && xsl[i] == i));
    }
    uint32_t n_ch = 0, n_out = 0;
    while (n_unvisited > 0) {
        float bs = 0.0f;
        int bv = -1;
        for (int i = lane; i < n; i += 64)
            if (xsl[i] == i && xvs[i] > bs) {  // strict >: the lowest index of this lane's maxima
                bs = xvs[i];
                bv = i;
            }
        for (int off = 32; off; off >>= 1) {  // wave arg-max: highest score, ties to the lowest sorted index
            const float os = __shfl_xor(bs, off, 64);
            const int ov = __shfl_xor(bv, off, 64);
            if (ov >= 0 && (bv < 0 || os > bs || (os == bs && ov < bv))) {
                bs = os;
                bv = ov;
            }
        }
        if (bv < 0) {  // aln.rs:129-131 would spin forever (only non-positive scores left): this group ends here
            if (lane == 0) atomicAdd(err, 1u);
            break;
        }
        int len = 0, first_v = bv;
        if (in_lds) {
            if (lane == 0) {
                int v = bv;
                while (v >= 0 && L.sl[v] == v) {  // :121-128
                    out_hp[gs + n_out + len] = L.h[v];
                    ++len;
                    first_v = v;
                    const int nv = L.pv[v];
                    L.sl[v] = -1 - v;  // :133-137
                    v = nv;
                }
            }
            len = __builtin_amdgcn_readfirstlane(len);
            first_v = __builtin_amdgcn_readfirstlane(first_v);
            wave_sync();
            for (int a = lane; a < len / 2; a += 64) {  // :132 reverse
                const pgr_hitpair t = out_hp[gs + n_out + a];
                out_hp[gs + n_out + a] = out_hp[gs + n_out + len - 1 - a];
                out_hp[gs + n_out + len - 1 - a] = t;
            }
        } else {
            // the walk is a pointer chase (a serial thread pays a memory round trip per step): predecessors are
            // almost always a few positions back, so the wave keeps 64 consecutive (pv, sl) entries in registers
            // and steps through them with readlane; visited indices go to `track`, the hit pairs are gathered in
            // parallel afterwards (already in reversed = chain order)
            int v = bv, base = -1, pvb = -1, slb = -2;
            while (v >= 0) {
                if (base < 0 || v < base || v >= base + 64) {
                    base = v >= 63 ? v - 63 : 0;
                    const int k = base + lane;
                    pvb = k < n ? gpv[k] : -1;
                    slb = k < n ? gsl[k] : -2;
                }
                const int o = v - base;
                if (__builtin_amdgcn_readlane(slb, o) != v) break;  // :121-128: stop at a visited node
                if (lane == 0) {
                    gtrack[len] = v;
                    gsl[v] = -1 - v;  // :133-137
                }
                ++len;
                first_v = v;
                v = __builtin_amdgcn_readlane(pvb, o);
            }
            wave_sync();
            for (int a = lane; a < len; a += 64) out_hp[gs + n_out + (len - 1 - a)] = gh[gtrack[a]];  // :132
        }
        if (lane == 0) {
            chain_len[gs + n_ch] = (uint32_t)len;
            chain_score[gs + n_ch] = bs - xvs[first_v];  // :138-139
        }
        ++n_ch;
        n_out += (uint32_t)len;
        n_unvisited -= len;
        wave_sync();
    }
    if (lane == 0) {
        g_nchains[g] = n_ch;
        g_nhp[g] = n_out;
    }
    wave_sync();  // the LDS image is reused by the next group
    }
}


// ------------------------------------------------------------------------------------------------
// dense outputs: the chaining kernels leave chains / hit pairs at their group's offset in hit-sized arrays;
// pack them by group (chain_off / hp_off = exclusive scans of the per-group counts) so that the host receives
// exactly what it returns
__global__ void group_counts_kernel(const uint64_t *__restrict__ g_start, const uint64_t *__restrict__ n_groups_ptr,
                                    uint64_t n, const uint32_t *__restrict__ g_nch, const uint32_t *__restrict__ g_nhp,
                                    const uint64_t *__restrict__ skey, uint32_t *__restrict__ nch_out,
                                    uint32_t *__restrict__ nhp_out, uint64_t *__restrict__ gkey_out) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n) return;
    if (g >= *n_groups_ptr) {  // beyond the last group (arrays are sized by the number of hits) + the scan sentinel
        nch_out[g] = 0;
        nhp_out[g] = 0;
        return;
    }
    const uint64_t gs = g_start[g];
    const bool real = g_start[g + 1] - gs >= 2;  // aln.rs:234: targets with a single hit are dropped
    nch_out[g] = real ? g_nch[g] : 0u;
    nhp_out[g] = real ? g_nhp[g] : 0u;
    gkey_out[g] = skey[gs];
}

__global__ void pack_chains_kernel(const uint64_t *__restrict__ g_start, uint64_t n,
                                   const uint32_t *__restrict__ flags, const uint64_t *__restrict__ rank,
                                   const uint32_t *__restrict__ nch, const uint32_t *__restrict__ nhp,
                                   const uint64_t *__restrict__ chain_off, const uint64_t *__restrict__ hp_off,
                                   const uint32_t *__restrict__ c_len, const float *__restrict__ c_score,
                                   const pgr_hitpair *__restrict__ o_hp, uint32_t *__restrict__ d_clen,
                                   float *__restrict__ d_cscore, pgr_hitpair *__restrict__ d_hp,
                                   const uint64_t *__restrict__ gkey, uint64_t n_groups, uint64_t *__restrict__ d_gstart,
                                   uint64_t *__restrict__ d_gkey, uint32_t *__restrict__ d_nch) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n_groups) {  // the per-group arrays of the image (n_groups <= n: every group has a hit)
        d_gstart[i] = g_start[i];
        d_nch[i] = nch[i];
        if (i < n_groups) d_gkey[i] = gkey[i];
    }
    if (i >= n) return;
    const uint64_t g = rank[i] - (flags[i] ? 0u : 1u);  // rank = exclusive scan of the group-start flags
    const uint64_t k = i - g_start[g];
    if (k < nhp[g]) d_hp[hp_off[g] + k] = o_hp[i];
    if (k < nch[g]) {
        d_clen[chain_off[g] + k] = c_len[i];
        d_cscore[chain_off[g] + k] = c_score[i];
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host orchestration of the chaining stage: hits (key = group id, hp) -> per-group chains.
// Two host round trips: one for the totals (chains, hit pairs, groups) that size the packed output, one for the
// download.  Every per-group array is sized by the number of hits (an upper bound on the number of groups) and the
// kernels read the true group count from device memory, so nothing waits for it.
namespace {

inline size_t up8(size_t v) { return (v + 7) & ~(size_t)7; }

// one host block = the packed device image (downloaded in one piece) + the small offset arrays built on the host;
// pgr_hps_result's pointers point into it and pgr_hps_result_free releases it as a whole (`_owner`)
struct ChainOut {
    uint8_t *block = nullptr;
    uint64_t n_groups = 0, n_chains = 0, n_hps = 0, n_nonterm = 0;
    // views into block (device image)
    const uint64_t *g_start = nullptr, *g_key = nullptr;
    pgr_hitpair *hps = nullptr;
    float *c_score = nullptr;
    const uint32_t *c_len = nullptr, *g_nch = nullptr;
    size_t image_bytes = 0;
    ChainOut() = default;
    ChainOut(const ChainOut &) = delete;
    ChainOut &operator=(const ChainOut &) = delete;
    ~ChainOut() { free(block); }
};

// extra = bytes the caller wants behind the image in the same host block (the offset arrays of fill_result)

// sorted_by_qb: the hits of every group already come in ascending qb (the query path emits them pair by pair in query
// position order), so only the grouping sort is needed.  sid_bound / n_queries bound the two halves of the group key.
// seg (query path): per-query hit segments {pair_off, hit_off} on the device when no query has more than SEG_SORT_MAX hits:
// the grouping runs per query in LDS (group_sort_kernel) instead of through the global sort.
struct HitSegments {
    const uint64_t *pair_off = nullptr, *hit_off = nullptr;
    uint64_t max_hits = 0;  // most hits of one query
};
int chain_hits(pgr_ctx *ctx, const uint64_t *d_key, const pgr_hitpair *d_hp, uint64_t n, const AlnParams &prm,
               uint32_t n_queries, uint64_t sid_bound, bool sorted_by_qb, ChainOut &out, HitSegments seg = HitSegments()) {
    if (n == 0) return PGR_OK;
    if (n >= (1ull << 32)) return ctx->fail(PGR_ERR_INVALID_ARG, "more than 2^32-1 hits in one batch");
    hipStream_t st = ctx->stream;
    int rc;
    const bool dbg = ctx->opt.debug != 0;
    const auto c0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(st);
        fprintf(stderr, "[pgr]   chain_hits %-18s at %.2f ms\n", what,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count());
    };
    Tmp idx_a(ctx), idx_b(ctx), keys_a(ctx), keys_b(ctx), skey(ctx), shp(ctx), flags(ctx), rank(ctx), gstart(ctx);
    if ((rc = idx_a.alloc(n * 4)) || (rc = idx_b.alloc(n * 4)) || (rc = keys_a.alloc(n * 8)) || (rc = keys_b.alloc(n * 8)) ||
        (rc = skey.alloc(n * 8)) || (rc = shp.alloc(n * sizeof(pgr_hitpair))) || (rc = flags.alloc((n + 1) * 4)) ||
        (rc = rank.alloc((n + 1) * 8)) || (rc = gstart.alloc((n + 2) * 8)))
        return rc;
    const size_t tb = sort_pairs_temp_bytes(n);
    if ((rc = ctx->ws_scan_tmp.ensure(ctx, std::max(tb, scan_counts_temp_bytes((uint32_t)(n + 1)))))) return rc;
    if (sorted_by_qb && seg.pair_off && n_queries) {
        hipLaunchKernelGGL(group_sort_kernel, dim3(n_queries), dim3(seg.max_hits <= 256 ? 64 : 256), 0, st, d_key, d_hp, seg.pair_off, seg.hit_off, n,
                           skey.as<uint64_t>(), shp.as<pgr_hitpair>(), flags.as<uint32_t>());
    } else {
        hipLaunchKernelGGL(iota_kernel, grid_for(n), dim3(256), 0, st, idx_a.as<uint32_t>(), n);
        uint32_t *cur = idx_a.as<uint32_t>(), *nxt = idx_b.as<uint32_t>();
        const unsigned sid_bits = std::min(32u, bits_for(sid_bound ? sid_bound : (1ull << 32)));
        const unsigned key_bits = std::min(64u, sid_bits + bits_for(n_queries));
        for (int f = sorted_by_qb ? 1 : 0; f < 2; ++f) {  // stable: by qb (aln.rs:21), then by group
            hipLaunchKernelGGL(hit_key_kernel, grid_for(n), dim3(256), 0, st, d_key, d_hp, cur, f, sid_bits, keys_a.as<uint64_t>(), n);
            PGR_HIP(ctx, sort_pairs(st, ctx->ws_scan_tmp.p, tb, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), cur, nxt, n,
                                    f == 0 ? 32u : key_bits));
            std::swap(cur, nxt);
        }
        hipLaunchKernelGGL(gather_hits_kernel, grid_for(n + 1), dim3(256), 0, st, d_key, d_hp, cur, n, skey.as<uint64_t>(),
                           shp.as<pgr_hitpair>(), flags.as<uint32_t>());
    }
    PGR_HIP(ctx, scan_counts(st, ctx->ws_scan_tmp.p, scan_counts_temp_bytes((uint32_t)(n + 1)), flags.as<uint32_t>(),
                             rank.as<uint64_t>(), (uint32_t)(n + 1)));
    lap("sorted + grouped");
    const uint64_t *d_ngroups = rank.as<uint64_t>() + n;  // number of groups, on the device
    Tmp err(ctx);  // [0] groups the reference never finishes, [1], [2] groups per wave class: cleared by scatter_starts_kernel
    if ((rc = err.alloc(16))) return rc;
    hipLaunchKernelGGL(scatter_starts_kernel, grid_for(n + 1), dim3(256), 0, st, flags.as<uint32_t>(), rank.as<uint64_t>(),
                       n, gstart.as<uint64_t>(), err.as<uint32_t>());
    Tmp v_s(ctx), pre(ctx), slot(ctx), o_hp(ctx), c_len(ctx), c_score(ctx), g_nch(ctx), g_nhp(ctx), big(ctx), trk(ctx);
    const uint64_t max_big = n / ALN_WAVE_MIN_FEW + 1;  // a wave-chained group has at least ALN_WAVE_MIN_FEW hits
    if ((rc = v_s.alloc(n * 4)) || (rc = pre.alloc(n * 4)) || (rc = slot.alloc(n * 4)) ||
        (rc = o_hp.alloc(n * sizeof(pgr_hitpair))) || (rc = c_len.alloc(n * 4)) || (rc = c_score.alloc(n * 4)) ||
        (rc = g_nch.alloc(n * 4)) || (rc = g_nhp.alloc(n * 4)) ||
        (rc = big.alloc(2 * max_big * 4)) || (rc = trk.alloc(n * 4)))
        return rc;
    Tmp span_g(ctx);  // only for max_span > MAX_SPAN_CAP
    if (prm.max_span > MAX_SPAN_CAP && (rc = span_g.alloc(n * 4))) return rc;
    hipLaunchKernelGGL(sparse_aln_kernel, grid_for(n, 64), dim3(64), 0, st, shp.as<pgr_hitpair>(),
                       gstart.as<uint64_t>(), d_ngroups, prm, v_s.as<float>(), pre.as<int>(), slot.as<int>(),
                       o_hp.as<pgr_hitpair>(), c_len.as<uint32_t>(), c_score.as<float>(), g_nch.as<uint32_t>(),
                       g_nhp.as<uint32_t>(), err.as<uint32_t>(), big.as<uint32_t>(), err.as<uint32_t>() + 1,
                       (uint32_t)max_big, span_g.as<uint32_t>());
    // one wavefront per longer group; the grids are upper bounds, surplus workgroups exit on the device-side counts
    hipLaunchKernelGGL(sparse_aln_wave_kernel<ALN_LDS_SMALL>, dim3((uint32_t)std::min<uint64_t>(max_big, 4096)), dim3(64), 0, st,
                       shp.as<pgr_hitpair>(), gstart.as<uint64_t>(), big.as<uint32_t>(), err.as<uint32_t>() + 1, prm,
                       v_s.as<float>(), pre.as<int>(), slot.as<int>(), trk.as<int>(), o_hp.as<pgr_hitpair>(), c_len.as<uint32_t>(),
                       c_score.as<float>(), g_nch.as<uint32_t>(), g_nhp.as<uint32_t>(), err.as<uint32_t>());
    const uint64_t max_long = n / (ALN_LDS_SMALL + 1) + 1;
    hipLaunchKernelGGL(sparse_aln_wave_kernel<ALN_LDS_MAX>, dim3((uint32_t)std::min<uint64_t>(max_long, 512)), dim3(64), 0, st,
                       shp.as<pgr_hitpair>(), gstart.as<uint64_t>(), big.as<uint32_t>() + max_big, err.as<uint32_t>() + 2,
                       prm, v_s.as<float>(), pre.as<int>(), slot.as<int>(), trk.as<int>(), o_hp.as<pgr_hitpair>(), c_len.as<uint32_t>(),
                       c_score.as<float>(), g_nch.as<uint32_t>(), g_nhp.as<uint32_t>(), err.as<uint32_t>());
    lap("chained");
    // per-group counts -> offsets of the packed output
    Tmp d_nch(ctx), d_nhp(ctx), d_gkey(ctx), d_choff(ctx), d_hpoff(ctx);
    if ((rc = d_nch.alloc((n + 1) * 4)) || (rc = d_nhp.alloc((n + 1) * 4)) || (rc = d_gkey.alloc(n * 8)) ||
        (rc = d_choff.alloc((n + 1) * 8)) || (rc = d_hpoff.alloc((n + 1) * 8)))
        return rc;
    if ((rc = ctx->ensure_mailbox(64))) return rc;
    uint64_t *mb = (uint64_t *)ctx->mailbox;
    Tmp words(ctx);
    if ((rc = words.alloc(32))) return rc;
    hipLaunchKernelGGL(group_counts_kernel, grid_for(n + 1), dim3(256), 0, st, gstart.as<uint64_t>(), d_ngroups, n,
                       g_nch.as<uint32_t>(), g_nhp.as<uint32_t>(), skey.as<uint64_t>(), d_nch.as<uint32_t>(),
                       d_nhp.as<uint32_t>(), d_gkey.as<uint64_t>());
    const size_t tbg = scan_counts_temp_bytes((uint32_t)(n + 1));
    PGR_HIP(ctx, scan_counts(st, ctx->ws_scan_tmp.p, tbg, d_nch.as<uint32_t>(), d_choff.as<uint64_t>(), (uint32_t)(n + 1)));
    PGR_HIP(ctx, scan_counts(st, ctx->ws_scan_tmp.p, tbg, d_nhp.as<uint32_t>(), d_hpoff.as<uint64_t>(), (uint32_t)(n + 1)));
    hipLaunchKernelGGL(gather_words_kernel, dim3(1), dim3(64), 0, st, d_choff.as<uint64_t>() + n, d_hpoff.as<uint64_t>() + n,
                       d_ngroups, err.as<uint32_t>(), words.as<uint64_t>());
    // ---- round trip 1: the totals
    PGR_HIP(ctx, hipMemcpyAsync(mb, words.p, 32, hipMemcpyDeviceToHost, st));
    PGR_HIP(ctx, hipStreamSynchronize(st));
    PGR_HIP(ctx, hipGetLastError());
    const uint64_t n_chains = mb[0], n_hps = mb[1], n_groups = mb[2];
    out.n_nonterm = (uint32_t)mb[3];
    // ---- packed device image: g_start | g_key | hps | c_score | c_len | g_nch, one download
    const size_t o_gstart = 0, o_gkey = o_gstart + (n_groups + 1) * 8, o_hps = o_gkey + n_groups * 8,
                 o_cscore = o_hps + n_hps * sizeof(pgr_hitpair), o_clen = up8(o_cscore + n_chains * 4),
                 o_nch = up8(o_clen + n_chains * 4), image = up8(o_nch + (n_groups + 1) * 4);
    Tmp img(ctx);
    if ((rc = img.alloc(image))) return rc;
    uint8_t *dI = img.as<uint8_t>();
    hipLaunchKernelGGL(pack_chains_kernel, grid_for(n + 1), dim3(256), 0, st, gstart.as<uint64_t>(), n,
                       flags.as<uint32_t>(), rank.as<uint64_t>(), d_nch.as<uint32_t>(), d_nhp.as<uint32_t>(),
                       d_choff.as<uint64_t>(), d_hpoff.as<uint64_t>(), c_len.as<uint32_t>(), c_score.as<float>(),
                       o_hp.as<pgr_hitpair>(), (uint32_t *)(dI + o_clen), (float *)(dI + o_cscore), (pgr_hitpair *)(dI + o_hps),
                       d_gkey.as<uint64_t>(), n_groups, (uint64_t *)(dI + o_gstart), (uint64_t *)(dI + o_gkey),
                       (uint32_t *)(dI + o_nch));
    // host block: image + room for q_off, t_off, c_off (u64) and t_sid (u32) that fill_result builds
    const size_t extra = ((size_t)n_queries + 1) * 8 + (n_groups + 1) * 8 + (n_chains + 1) * 8 + up8(n_groups * 4);
    out.block = (uint8_t *)host_result_alloc(image + extra + 8);
    if (!out.block) return ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
    if ((rc = ctx->d2h(out.block, dI, image))) return rc;  // ---- round trip 2
    PGR_HIP(ctx, hipGetLastError());
    out.image_bytes = image;
    out.n_groups = n_groups;
    out.n_chains = n_chains;
    out.n_hps = n_hps;
    out.g_start = (const uint64_t *)(out.block + o_gstart);
    out.g_key = (const uint64_t *)(out.block + o_gkey);
    out.hps = (pgr_hitpair *)(out.block + o_hps);
    out.c_score = (float *)(out.block + o_cscore);
    out.c_len = (const uint32_t *)(out.block + o_clen);
    out.g_nch = (const uint32_t *)(out.block + o_nch);
    lap("downloaded");
    return PGR_OK;
}

// groups are sorted by key = (query << 32 | sid): flat result.  The offset arrays go behind the image in co.block.
int fill_result(pgr_ctx *ctx, uint32_t n_queries, ChainOut &co, pgr_hps_result *out) {
    if (!co.block) {  // no hits at all
        const size_t bytes = ((size_t)n_queries + 1) * 8 + 3 * 8 + 64;
        co.block = (uint8_t *)calloc(1, bytes);
        if (!co.block) return ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
        co.image_bytes = 0;
    }
    uint8_t *p = co.block + co.image_bytes;
    uint64_t *q_off = (uint64_t *)p;
    p += ((size_t)n_queries + 1) * 8;
    uint64_t *t_off = (uint64_t *)p;
    p += (co.n_groups + 1) * 8;
    uint64_t *c_off = (uint64_t *)p;
    p += (co.n_chains + 1) * 8;
    uint32_t *t_sid = (uint32_t *)p;
    uint64_t n_targets = 0, n_chains = 0;
    uint64_t g = 0;
    for (uint32_t q = 0; q < n_queries; ++q) {
        q_off[q] = n_targets;
        for (; g < co.n_groups && (uint32_t)(co.g_key[g] >> 32) <= q; ++g) {
            if ((uint32_t)(co.g_key[g] >> 32) != q) continue;
            if (co.g_start[g + 1] - co.g_start[g] < 2) continue;  // aln.rs:234: targets with a single hit are dropped
            t_sid[n_targets] = (uint32_t)(co.g_key[g] & 0xFFFFFFFFull);
            t_off[n_targets] = n_chains;
            n_chains += co.g_nch[g];
            ++n_targets;
        }
    }
    q_off[n_queries] = n_targets;
    t_off[n_targets] = n_chains;
    if (n_chains != co.n_chains) return ctx->fail(PGR_ERR_INTERNAL, "chain bookkeeping mismatch");
    uint64_t n_hp = 0;
    for (uint64_t c = 0; c < n_chains; ++c) {
        c_off[c] = n_hp;
        n_hp += co.c_len[c];
    }
    c_off[n_chains] = n_hp;
    out->n_queries = n_queries;
    out->q_off = q_off;
    out->n_targets = n_targets;
    out->t_sid = t_sid;
    out->t_off = t_off;
    out->n_chains = n_chains;
    out->c_score = co.c_score ? co.c_score : (float *)t_off;  // (never dereferenced when n_chains == 0)
    out->c_off = c_off;
    out->n_hps = co.n_hps;
    out->hps = co.hps ? co.hps : (pgr_hitpair *)t_off;
    out->n_nonterminating = co.n_nonterm;
    out->_owner = co.block;
    co.block = nullptr;
    return PGR_OK;
}

}  // namespace

extern "C" void pgr_hps_result_free(pgr_hps_result *r) {
    if (!r) return;
    result_block_release(r->_owner);  // every array of the result lives in this one block (malloc'd, or pinned: query_fused.hip)
    memset(r, 0, sizeof(*r));
}

extern "C" int pgr_ctx_last_query_prof(const pgr_ctx *ctx, pgr_query_prof *out) {
    if (!ctx || !out) return PGR_ERR_INVALID_ARG;
    *out = ctx->qprof;
    return PGR_OK;
}

namespace {
// sum over the query pairs of the number of fragment signatures their key holds in the index (before the count
// filters): what the lookup stage reads, 17 B each in the reference's layout
__global__ void sum_ranges_kernel(const uint64_t *__restrict__ lo, const uint64_t *__restrict__ hi, uint64_t nq,
                                  unsigned long long *__restrict__ total) {
    // 64-bit sums: ONE key holds < 2^32 signatures, the pairs of a repeat-heavy query batch together can hold more
    uint64_t v = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nq; p += (uint64_t)gridDim.x * blockDim.x)
        v += hi[p] - lo[p];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += shfl_xor64(v, d);
    // few workgroups: same-address atomics run at ~88 per us on gfx950
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(total, (unsigned long long)v);
}
}  // namespace

// B2 on a resident batch of queries (the H2D of the ASCII queries already happened: pgr_batch_from_ascii).
// (Measured and rejected in round 3: cutting a 10 000-query batch into 2-4 parts that run concurrently on contexts of their own
// -- own host threads, streams, workspaces, one shared index -- so that one part's host round trips hide behind the other's
// kernels: 1.34 / 1.62 / 1.34 ms against 1.12 ms in one piece.  A part does not get faster by being smaller: its time is the
// chain of ~45 dependent operations, and two such chains submitted from two threads slow each other down.)
extern "C" int pgr_query_hps_resident(pgr_ctx *ctx, const pgr_index *ix, const pgr_batch *b, float penalty,
                                      uint32_t max_count, uint32_t max_count_query, uint32_t max_count_target,
                                      uint32_t max_aln_span, int has_max_gap, uint32_t max_gap, int oriented,
                                      pgr_hps_result *out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || !out || !b) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    memset(out, 0, sizeof(*out));
    if (!ix->finalized) return ctx->fail(PGR_ERR_STATE, "index not finalized (call pgr_index_finalize)");
    if (b->ctx != ctx) return ctx->fail(PGR_ERR_STATE, "batch belongs to another context");
    if (max_aln_span == 0) return ctx->fail(PGR_ERR_INVALID_ARG, "max_aln_span must be at least 1");
    PGR_ENTER(ctx);
    hipStream_t st = ctx->stream;
    const bool dbg = ctx->opt.debug != 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) {
        return (float)std::chrono::duration<double, std::milli>(b2 - a).count();
    };
    const uint32_t n_queries = b->n;
    pgr_query_prof qp = {};
    qp.n_queries = n_queries;
    qp.query_bases = b->total_bases;
    const auto t1 = now();
    const QParams fqp{max_count, max_count_query, max_count_target};
    const AlnParams fap{max_aln_span, penalty, has_max_gap, max_gap, oriented};
    auto fused_done = [&](const QueryFusedCounts &fc, std::chrono::steady_clock::time_point t2, int path) {
        const auto t5 = now();
        qp.n_signatures = fc.n_signatures;
        qp.n_hits = fc.n_hits;
        qp.n_groups = out->n_targets;
        qp.n_chains = out->n_chains;
        qp.n_hps = out->n_hps;
        qp.shmmr_ms = ms(t1, t2);
        qp.chain_ms = ms(t2, t5);  // (lookup, hits, chaining, packing, download: one stage here)
        qp.total_ms = ms(t1, t5);
        qp.path = (uint32_t)path;
        ctx->qprof = qp;
        if (ctx->opt.debug_times) fprintf(stderr, "[pgr] query: result ready at %.1f us\n", qp.total_ms * 1e3);
        if (dbg)
            fprintf(stderr, "[pgr] query batch %u (one wavefront per query%s): shimmers %.2f ms, the rest %.2f\n", n_queries,
                    path == 2 ? ", enqueued behind the shimmer pipeline" : "", qp.shmmr_ms, qp.chain_ms);
    };
    // Batches of short queries (query_fused.hip).  When an earlier batch on this index has shown what its queries look like
    // (pairs per query), the per-query kernel is enqueued BEHIND the shimmer pipeline, in front of that pipeline's one
    // synchronization: pair records and offsets are derived on the device, the whole query is one host wait.  A batch that does
    // not fit the guess declines on the device and is done again below with what the host knows by then.
    const bool fused_on = ix->n && ix->fused_skip.load(std::memory_order_relaxed) == 0;
    uint32_t pairs_hint = ix->fused_pairs.load(std::memory_order_relaxed);
    if (!pairs_hint && fused_on && n_queries) {
        // first batch on this index: guess the pairs of the longest query from the spec's shimmer density (level-1 minimizers
        // 2 / (w + 1) per base, each of the two reductions keeps ~2 / (r + 1) of them) with room; a wrong guess costs a repeat
        uint32_t max_len = 0;
        for (uint32_t c = 0; c < b->n; ++c) max_len = std::max(max_len, b->h_len[c]);
        const pgr_spec &sp = ix->spec;
        const double keep = sp.r > 1 ? 2.0 / (double)(sp.r + 1) : 1.0;
        const double dens = sp.sketch ? 1.0 / (double)(1ull << (4 + sp.r)) : 2.0 / (double)(sp.w + 1) * keep * keep;
        pairs_hint = (uint32_t)std::min<double>((double)max_len * dens * 1.6 + 4.0, 1e9);
    }
    // The level-1 form first (pgr_aln.h: QfLevel1View): the tile kernel of the queries, then the per-query kernel straight on its
    // segments -- no list stage of the batch, one host wait.  What it declines (islands needed, a query that outgrows the LDS
    // image) goes on below as before; an index whose batches are flagged skips the attempt for its next calls.
    if (fused_on && pairs_hint && n_queries && !ctx->opt.no_query_level1 && ix->fused_l1_skip.load(std::memory_order_relaxed) == 0 &&
        query_fused_eligible(ctx, n_queries, pairs_hint, max_aln_span)) {
        uint32_t max_len = 0;
        for (uint32_t c = 0; c < b->n; ++c) max_len = std::max(max_len, b->h_len[c]);
        const uint32_t c1 = query_fused_level1_cap(max_len, ix->spec.w);
        if (c1) {
            QueryFusedRun run(ctx, ix, n_queries, pairs_hint, fqp, fap);
            bool taken = false;
            int rc1 = shmmr_level1_then(ctx, b, &ix->spec, [&](const QfLevel1View &v) { return run.enqueue_from_level1(v, c1); }, &taken);
            if (rc1) return rc1;
            if (taken) {
                PGR_HIP(ctx, hipStreamSynchronize(st));
                ctx->staged_unsynced = false;
                ctx->want_host_copy = false;
                const auto t2 = now();
                if (run.enqueued) {
                    QueryFusedCounts fc;
                    bool declined = false;
                    if ((rc1 = run.finish(out, &fc, &declined))) return rc1;
                    if (!declined) {
                        qp.n_query_pairs = fc.n_pairs;
                        fused_done(fc, t2, 3);
                        return PGR_OK;
                    }
                    if (run.l1_flagged) ix->fused_l1_skip.store(4, std::memory_order_relaxed);
                    if (ctx->opt.debug)
                        fprintf(stderr, "[pgr] query batch %u: the level-1 form declined (%s): the shimmer pipeline takes it\n", n_queries,
                                run.l1_flagged ? "the tile kernel asked for islands" : "a query outgrows its LDS image");
                }
            }
        }
    } else if (const uint32_t left = ix->fused_l1_skip.load(std::memory_order_relaxed)) {
        ix->fused_l1_skip.store(left - 1, std::memory_order_relaxed);
    }
    std::unique_ptr<QueryFusedRun> chained;
    if (fused_on && pairs_hint && query_fused_eligible(ctx, n_queries, pairs_hint, max_aln_span) && !ctx->opt.no_query_chaining) {
        chained.reset(new QueryFusedRun(ctx, ix, n_queries, pairs_hint, fqp, fap));
        QueryFusedRun *run = chained.get();
        ctx->post_enqueue = [run](const pgr_mm128 *d_mm, const uint64_t *d_off, uint64_t cap, const uint64_t *d_count) {
            return run->enqueue_from_shimmers(d_mm, d_off, cap, d_count);
        };
    }
    // queries -> shimmer-pair records, query side (strict <, seq_db.rs:1213); sid field = query index
    pgr_shmmrs *s = nullptr;
    int rc = pgr_shmmrs_compute(ctx, b, &ix->spec, nullptr, 0, &s);
    ctx->post_enqueue = nullptr;
    if (rc) return rc;
    const auto t2 = now();
    if (ctx->opt.debug_times) fprintf(stderr, "[pgr] query: shimmers returned at %.1f us\n", ms(t1, t2) * 1e3);
    auto t3 = t2, t4 = t2;
    const uint64_t nq = pgr_shmmrs_n_pairs(s);
    qp.n_query_pairs = nq;
    uint64_t max_pairs = 0;  // most shimmer pairs of one query
    for (uint32_t c = 0; c < s->n; ++c) max_pairs = std::max(max_pairs, s->h_off[c + 1] - s->h_off[c]);
    ix->fused_pairs.store((uint32_t)std::min<uint64_t>(std::max<uint64_t>(max_pairs, 1), 1u << 30), std::memory_order_relaxed);
    bool structural_decline = false;  // the per-query kernel cannot hold this batch whatever its sizes
    if (chained && chained->enqueued) {
        QueryFusedCounts fc;
        bool declined = false;
        if ((rc = chained->finish(out, &fc, &declined))) {
            pgr_shmmrs_destroy(s);
            return rc;
        }
        if (!declined) {
            pgr_shmmrs_destroy(s);
            fused_done(fc, t2, 2);
            return PGR_OK;
        }
        structural_decline = max_pairs <= chained->P;  // (otherwise the guess was too small: once more below)
    }
    chained.reset();
    ChainOut co;
    if (nq && ix->n) {
        Tmp qrec(ctx), lo(ctx), hi(ctx), cnt(ctx), idx_a(ctx), idx_b(ctx), keys_a(ctx), keys_b(ctx), nh(ctx), hoff(ctx), nsig(ctx);
        if ((rc = qrec.alloc(nq * sizeof(pgr_frag_rec)))) {
            pgr_shmmrs_destroy(s);
            return rc;
        }
        rc = shmmrs_to_frag_recs_enqueue(ctx, s, nullptr, 1, qrec.as<pgr_frag_rec>(), nq);  // stream ordered, no wait
        pgr_shmmrs_destroy(s);
        s = nullptr;
        if (rc) return rc;
        if (fused_on && !structural_decline && query_fused_eligible(ctx, n_queries, max_pairs, max_aln_span)) {
            QueryFusedRun run(ctx, ix, n_queries, max_pairs, fqp, fap);
            QueryFusedCounts fc;
            bool declined = false;
            if ((rc = run.enqueue(qrec.as<pgr_frag_rec>(), (const uint64_t *)ctx->ws_rec_off.p))) return rc;
            PGR_HIP(ctx, hipStreamSynchronize(st));
            if ((rc = run.finish(out, &fc, &declined))) return rc;
            if (!declined) {
                fused_done(fc, t2, 1);
                return PGR_OK;
            }
            structural_decline = true;
        }
        if (structural_decline) {
            // a long query, a repeat key, a long group: the stage-by-stage kernels below take the batch, and the next calls
            // on this index do not try again for a while
            ix->fused_skip.store(16, std::memory_order_relaxed);
        } else {
            const uint32_t left = ix->fused_skip.load(std::memory_order_relaxed);
            if (left) ix->fused_skip.store(left - 1, std::memory_order_relaxed);
        }
        if ((rc = lo.alloc(nq * 8)) || (rc = hi.alloc(nq * 8)) || (rc = cnt.alloc(nq * 4)) || (rc = idx_a.alloc(nq * 4)) ||
            (rc = idx_b.alloc(nq * 4)) || (rc = keys_a.alloc(nq * 8)) || (rc = keys_b.alloc(nq * 8)) ||
            (rc = nh.alloc((nq + 1) * 4)) || (rc = hoff.alloc((nq + 1) * 8)) || (rc = nsig.alloc(16)))
            return rc;
        hipLaunchKernelGGL(lookup_kernel, grid_for(nq), dim3(256), 0, st, qrec.as<pgr_frag_rec>(), nq, ix->recs,
                           ix->key_off, ix->n_keys, ix->lut, ix->lut_bits, ix->lut_shift, ix->keys, lo.as<uint64_t>(), hi.as<uint64_t>());
        PGR_HIP(ctx, hipMemsetAsync(nsig.p, 0, 16, st));
        hipLaunchKernelGGL(sum_ranges_kernel, dim3((uint32_t)std::min<uint64_t>(64, (nq + 255) / 256)), dim3(256), 0, st,
                           lo.as<uint64_t>(), hi.as<uint64_t>(), nq, nsig.as<unsigned long long>());
        // per-query key multiplicities (aln.rs:180-181).  Queries of up to a few thousand pairs: every pair is compared
        // with the other pairs of its query (they are contiguous); longer ones: sort by (query, h0, h1), count runs
        if (max_pairs <= 4096) {
            hipLaunchKernelGGL(pair_count_kernel, grid_for(nq), dim3(256), 0, st, qrec.as<pgr_frag_rec>(),
                               (const uint64_t *)ctx->ws_rec_off.p, nq, cnt.as<uint32_t>());
        } else {
            hipLaunchKernelGGL(iota_kernel, grid_for(nq), dim3(256), 0, st, idx_a.as<uint32_t>(), nq);
            const int fields[3] = {2, 3, 1};  // h1, h0, sid(=query)
            const unsigned bits[3] = {56, 56, 32};
            if ((rc = sort_perm(ctx, qrec.as<pgr_frag_rec>(), nq, fields, bits, 3, idx_a.as<uint32_t>(), idx_b.as<uint32_t>(),
                                keys_a.as<uint64_t>(), keys_b.as<uint64_t>())))
                return rc;
            hipLaunchKernelGGL(run_count_kernel, grid_for(nq), dim3(256), 0, st, qrec.as<pgr_frag_rec>(),
                               idx_a.as<uint32_t>(), nq, cnt.as<uint32_t>());
        }
        QParams qprm{max_count, max_count_query, max_count_target};
        hipLaunchKernelGGL(hits_kernel, grid_for(nq + 1), dim3(256), 0, st, qrec.as<pgr_frag_rec>(), nq,
                           cnt.as<uint32_t>(), lo.as<uint64_t>(), hi.as<uint64_t>(), ix->recs, qprm, 0, nh.as<uint32_t>(),
                           (const uint64_t *)nullptr, (uint64_t *)nullptr, (pgr_hitpair *)nullptr);
        const size_t tb = scan_counts_temp_bytes((uint32_t)(nq + 1));
        if ((rc = ctx->ws_scan_tmp.ensure(ctx, tb))) return rc;
        PGR_HIP(ctx, scan_counts(st, ctx->ws_scan_tmp.p, tb, nh.as<uint32_t>(), hoff.as<uint64_t>(), (uint32_t)(nq + 1)));
        if ((rc = ctx->ensure_mailbox(64))) return rc;
        uint64_t *mb = (uint64_t *)ctx->mailbox;
        Tmp words(ctx);
        if ((rc = words.alloc(32))) return rc;
        // nsig: [0] looked-up signatures, [1] most hits of one query (decides how the hits are grouped, chain_hits)
        hipLaunchKernelGGL(query_hit_max_kernel, grid_for(n_queries), dim3(256), 0, st, (const uint64_t *)ctx->ws_rec_off.p,
                           hoff.as<uint64_t>(), n_queries, nsig.as<unsigned long long>() + 1);
        hipLaunchKernelGGL(gather_words_kernel, dim3(1), dim3(64), 0, st, hoff.as<uint64_t>() + nq, nsig.as<uint64_t>(),
                           nsig.as<uint64_t>() + 1, (const uint32_t *)nullptr, words.as<uint64_t>());
        PGR_HIP(ctx, hipMemcpyAsync(mb, words.p, 32, hipMemcpyDeviceToHost, st));
        PGR_HIP(ctx, hipStreamSynchronize(st));  // the number of hits sizes everything behind this point
        const uint64_t n_hits = mb[0];
        qp.n_signatures = mb[1];
        qp.n_hits = n_hits;
        t3 = now();
        if (n_hits) {
            Tmp hkey(ctx), hhp(ctx);
            if ((rc = hkey.alloc(n_hits * 8)) || (rc = hhp.alloc(n_hits * sizeof(pgr_hitpair)))) return rc;
            hipLaunchKernelGGL(hits_kernel, grid_for(nq + 1), dim3(256), 0, st, qrec.as<pgr_frag_rec>(), nq,
                               cnt.as<uint32_t>(), lo.as<uint64_t>(), hi.as<uint64_t>(), ix->recs, qprm, 1,
                               (uint32_t *)nullptr, hoff.as<uint64_t>(), hkey.as<uint64_t>(), hhp.as<pgr_hitpair>());
            AlnParams ap{max_aln_span, penalty, has_max_gap, max_gap, oriented};
            HitSegments seg;
            if (mb[2] <= SEG_SORT_MAX && !ctx->opt.query_global_sort) {
                seg.pair_off = (const uint64_t *)ctx->ws_rec_off.p;
                seg.hit_off = hoff.as<uint64_t>();
                seg.max_hits = mb[2];
            }
            if ((rc = chain_hits(ctx, hkey.as<uint64_t>(), hhp.as<pgr_hitpair>(), n_hits, ap, n_queries, ix->sid_bound, true, co, seg)))
                return rc;
        }
        t4 = now();
    } else {
        pgr_shmmrs_destroy(s);
    }
    rc = fill_result(ctx, n_queries, co, out);
    const auto t5 = now();
    qp.n_groups = out->n_targets;
    qp.n_chains = out->n_chains;
    qp.n_hps = out->n_hps;
    qp.shmmr_ms = ms(t1, t2);
    qp.lookup_ms = ms(t2, t3);
    qp.chain_ms = ms(t3, t4);
    qp.result_ms = ms(t4, t5);
    qp.total_ms = ms(t1, t5);
    ctx->qprof = qp;
    if (dbg)
        fprintf(stderr, "[pgr] query batch %u: shimmers %.2f ms, lookup+counts %.2f, hits+chain %.2f, result %.2f\n", n_queries,
                qp.shmmr_ms, qp.lookup_ms, qp.chain_ms, qp.result_ms);
    return rc;
}

extern "C" int pgr_query_hps_batch(pgr_ctx *ctx, const pgr_index *ix, uint32_t n_queries, const uint8_t *const *seqs,
                                   const uint64_t *lens, float penalty, uint32_t max_count, uint32_t max_count_query,
                                   uint32_t max_count_target, uint32_t max_aln_span, int has_max_gap, uint32_t max_gap,
                                   int oriented, pgr_hps_result *out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || !out) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    memset(out, 0, sizeof(*out));
    if (!ix->finalized) return ctx->fail(PGR_ERR_STATE, "index not finalized (call pgr_index_finalize)");
    if (max_aln_span == 0) return ctx->fail(PGR_ERR_INVALID_ARG, "max_aln_span must be at least 1");
    const auto t0 = std::chrono::steady_clock::now();
    pgr_batch *b = nullptr;
    int rc = pgr_batch_from_ascii(ctx, n_queries, seqs, lens, &b);  // enqueues H2D + pack; nothing waits here
    if (rc) return rc;
    const float stage = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    rc = pgr_query_hps_resident(ctx, ix, b, penalty, max_count, max_count_query, max_count_target, max_aln_span, has_max_gap,
                                max_gap, oriented, out);
    pgr_batch_destroy(b);
    ctx->qprof.stage_ms = stage;  // host time of the staging (pinned memcpy + enqueue); the copy itself overlaps the rest
    ctx->qprof.total_ms += stage;
    return rc;
}

extern "C" int pgr_sparse_aln_batch(pgr_ctx *ctx, uint32_t n_groups, const pgr_hitpair *hits, const uint64_t *g_off,
                                    uint32_t max_span, float penalty, int has_max_gap, uint32_t max_gap, int oriented,
                                    pgr_hps_result *out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!out || (n_groups && (!hits || !g_off))) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    memset(out, 0, sizeof(*out));
    if (max_span == 0) return ctx->fail(PGR_ERR_INVALID_ARG, "max_span must be at least 1");
    PGR_ENTER(ctx);
    const uint64_t n = n_groups ? g_off[n_groups] : 0;
    for (uint32_t g = 0; g < n_groups; ++g)
        if (g_off[g + 1] - g_off[g] < 2)  // aln.rs:24 assert!(sp_hits.len() > 1)
            return ctx->fail(PGR_ERR_INVALID_ARG, "sparse_aln needs at least 2 hit pairs per group");
    ChainOut co;
    if (n) {
        std::vector<uint64_t> keys(n);
        for (uint32_t g = 0; g < n_groups; ++g)
            for (uint64_t i = g_off[g]; i < g_off[g + 1]; ++i) keys[i] = (uint64_t)g;  // query 0, "sid" = group
        Tmp dk(ctx), dh(ctx);
        int rc;
        if ((rc = dk.alloc(n * 8)) || (rc = dh.alloc(n * sizeof(pgr_hitpair)))) return rc;
        PGR_HIP(ctx, hipMemcpyAsync(dk.p, keys.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
        PGR_HIP(ctx, hipMemcpyAsync(dh.p, hits, n * sizeof(pgr_hitpair), hipMemcpyHostToDevice, ctx->stream));
        PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        AlnParams ap{max_span, penalty, has_max_gap, max_gap, oriented};
        if ((rc = chain_hits(ctx, dk.as<uint64_t>(), dh.as<pgr_hitpair>(), n, ap, 1, n_groups, false, co))) return rc;
    }
    return fill_result(ctx, 1, co, out);
}

// ------------------------------------------------------------------------------------------------
// .mdb on-disk format (pgr-db/src/seq_db.rs:1291-1326 writer, :1328-1471 readers):
//   "mdb" | w,k,r,min_span,flags : 5 x u32 LE | n_keys : u64 | n_keys x { h0,h1,n : u64 ; n x {frg_id,sid,bgn,end : u32 ; orient : u8} }
// The reference writes keys in FxHashMap order (unspecified); here keys are written sorted.  Readers are
// order agnostic, and so is pgr_index_load_mdb.
extern "C" int pgr_index_write_mdb(pgr_ctx *ctx, const pgr_index *ix, const char *path) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || !path) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    pgr_frag_rec *recs = nullptr;
    uint64_t n = 0;
    int rc = pgr_index_download(ctx, ix, &recs, &n);
    if (rc) return rc;
    // written to "<path>.tmp.<pid>" and renamed on success: a failed or short write (disk full, quota) never leaves
    // a truncated .mdb under the final name
    const std::string tmp_path = std::string(path) + ".tmp." + std::to_string((long)getpid());
    FILE *f = fopen(tmp_path.c_str(), "wb");
    if (!f) {
        free(recs);
        return ctx->fail(PGR_ERR_INVALID_ARG, std::string("cannot open for writing: ") + tmp_path);
    }
    std::vector<uint8_t> buf;
    buf.reserve(1 << 20);
    bool ok = true;
    auto flush = [&]() {
        if (!buf.empty() && ok) ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
        buf.clear();
    };
    auto put = [&](const void *p, size_t len) {
        const uint8_t *b = (const uint8_t *)p;
        buf.insert(buf.end(), b, b + len);
        if (buf.size() >= (1 << 20)) flush();
    };
    put("mdb", 3);
    const uint32_t hdr[5] = {ix->spec.w, ix->spec.k, ix->spec.r, ix->spec.min_span, ix->spec.sketch ? 1u : 0u};
    put(hdr, sizeof(hdr));
    const uint64_t nk = ix->n_keys;
    put(&nk, 8);
    for (uint64_t i = 0; i < n && ok;) {
        uint64_t j = i;
        while (j < n && recs[j].h0 == recs[i].h0 && recs[j].h1 == recs[i].h1) ++j;
        const uint64_t head[3] = {recs[i].h0, recs[i].h1, j - i};
        put(head, sizeof(head));
        for (uint64_t t = i; t < j; ++t) {
            const uint32_t q[4] = {recs[t].frg_id, recs[t].sid, recs[t].bgn, recs[t].end};
            put(q, sizeof(q));
            const uint8_t o = (uint8_t)recs[t].orient;
            put(&o, 1);
        }
        i = j;
    }
    flush();
    ok = ok && !ferror(f);
    ok = (fclose(f) == 0) && ok;
    free(recs);
    if (ok && rename(tmp_path.c_str(), path) != 0) ok = false;
    if (!ok) {
        (void)remove(tmp_path.c_str());
        return ctx->fail(PGR_ERR_INVALID_ARG, std::string("write failed: ") + path);
    }
    return PGR_OK;
}

extern "C" int pgr_index_load_mdb(pgr_ctx *ctx, const char *path, pgr_index **out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!path || !out) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return ctx->fail(PGR_ERR_INVALID_ARG, std::string("cannot open: ") + path);
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> d((size_t)std::max<long>(sz, 0));
    const size_t got = d.empty() ? 0 : fread(d.data(), 1, d.size(), f);
    fclose(f);
    if (got != d.size() || d.size() < 31 || memcmp(d.data(), "mdb", 3) != 0)
        return ctx->fail(PGR_ERR_INVALID_ARG, std::string("not an .mdb file: ") + path);
    uint32_t hdr[5];
    memcpy(hdr, d.data() + 3, 20);
    uint64_t nk;
    memcpy(&nk, d.data() + 23, 8);
    pgr_spec spec = {hdr[0], hdr[1], hdr[2], hdr[3], hdr[4] & 1u};
    pgr_index *ix = nullptr;
    int rc = pgr_index_create(ctx, &spec, &ix);
    if (rc) return rc;
    // the lengths stored in the file are not trusted: every count is checked against the bytes that are really
    // there, without multiplying (a huge count must not wrap the bound)
    auto bad = [&](const char *what) {
        pgr_index_destroy(ix);
        return ctx->fail(PGR_ERR_INVALID_ARG, std::string(what) + ": " + path);
    };
    if (nk > (d.size() - 31) / 24) return bad("truncated .mdb (key count exceeds the file size)");
    std::vector<pgr_frag_rec> recs;
    recs.reserve((d.size() - 31) / 17);
    size_t off = 31;
    uint32_t max_sid = 0;
    for (uint64_t kx = 0; kx < nk; ++kx) {
        if (d.size() - off < 24) return bad("truncated .mdb");
        uint64_t head[3];
        memcpy(head, d.data() + off, 24);
        off += 24;
        if (head[2] > (d.size() - off) / 17) return bad("truncated .mdb");
        if (recs.size() + head[2] >= (1ull << 32)) return bad(".mdb holds 2^32 or more fragment signatures");
        for (uint64_t t = 0; t < head[2]; ++t) {
            uint32_t q[4];
            memcpy(q, d.data() + off, 16);
            pgr_frag_rec r;
            r.h0 = head[0];
            r.h1 = head[1];
            r.frg_id = q[0];
            r.sid = q[1];
            r.bgn = q[2];
            r.end = q[3];
            r.orient = d[off + 16];
            r._pad = 0;
            recs.push_back(r);
            max_sid = std::max(max_sid, r.sid);
            off += 17;
        }
    }
    rc = pgr_index_add_records(ctx, ix, recs.data(), recs.size(), 0);
    if (!rc) rc = pgr_index_finalize(ctx, ix);
    if (rc) {
        pgr_index_destroy(ix);
        return rc;
    }
    ix->next_sid = recs.empty() ? 0u : (max_sid == 0xFFFFFFFFu ? max_sid : max_sid + 1);  // no wrap at u32::MAX
    *out = ix;
    return PGR_OK;
}

extern "C" int pgr_index_spec(const pgr_index *ix, pgr_spec *out) {
    if (!ix || !out) return PGR_ERR_INVALID_ARG;
    *out = ix->spec;
    return PGR_OK;
}
