// level2.hip -- list-level stages of sequence_to_shmmrs: ordered gather of the level-1 segments,
// the two hierarchical reductions (reduce_shmmr, pgr-db/src/shmmrutils.rs:359-415), the min_span
// stencil (:536-555) and the shimmer-pair records (pgr-db/src/seq_db.rs:381-400, 1205-1217).
//
// The level-1 list is ~2.5 % of the positions (12 B each in HBM: L1Rec).  It is read ONCE, straight from the unordered
// per-tile segments, by fused_select_kernel (reduce x2 + min_span in LDS); only the survivors (~12 %) are
// written, ordered by a scan + gather_segments_kernel.  Plain HBM streaming code, no MFMA.
#include <algorithm>

#include "pgr_device.h"
#include "pgr_internal.h"

#ifndef PGR_ABLATE_L2
#define PGR_ABLATE_L2 0  // timing experiments only
#endif
#ifndef PGR_L2_PACKED
#define PGR_L2_PACKED 1  // 1: keys in LDS carry a block-local contig ordinal in their top byte (no contig-id reads in the reduce)
#endif

namespace pgr {

// ------------------------------------------------------------------ ordered gather
// one wavefront per segment
__global__ __launch_bounds__(256) void gather_segments_kernel(const pgr_mm128 *__restrict__ src,
                                                              const uint64_t *__restrict__ seg_off,
                                                              const uint32_t *__restrict__ seg_cnt,
                                                              const uint64_t *__restrict__ seg_dst, uint32_t n_segs,
                                                              pgr_mm128 *__restrict__ dst, uint64_t dst_cap) {
    const uint32_t seg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= n_segs) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t cnt = seg_cnt[seg];
    if (seg_dst[seg] + cnt > dst_cap) return;  // result buffer sized by estimate: the host retries with the true size
    const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(src) + seg_off[seg];
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(dst) + seg_dst[seg];
    for (uint32_t i = lane; i < cnt; i += 64) d[i] = s[i];
}

void launch_gather_segments(hipStream_t st, const pgr_mm128 *src, const uint64_t *seg_off, const uint32_t *seg_cnt,
                            const uint64_t *seg_dst, uint32_t n_segs, pgr_mm128 *dst, uint64_t dst_cap) {
    if (n_segs == 0) return;
    hipLaunchKernelGGL(gather_segments_kernel, dim3((n_segs + 3) / 4), dim3(256), 0, st, src, seg_off, seg_cnt, seg_dst,
                       n_segs, dst, dst_cap);
}

namespace {
__device__ __forceinline__ uint32_t find_seg(const uint64_t *__restrict__ off, uint32_t n, uint64_t i) {
    uint32_t lo = 0, hi = n;  // largest c with off[c] <= i
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid;
        else hi = mid;
    }
    return lo;
}
}  // namespace

// ------------------------------------------------------------------ shimmer-pair records
// off: list offsets per contig; rec_off: record offsets per contig (count-1 per non-empty contig)
__global__ __launch_bounds__(256) void frag_recs_kernel(const pgr_mm128 *__restrict__ mm,
                                                        const uint64_t *__restrict__ off,
                                                        const uint64_t *__restrict__ rec_off, uint32_t n, uint64_t total,
                                                        const uint32_t *__restrict__ sids, int query_side,
                                                        int rid_is_index, pgr_frag_rec *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // contig of element i: the rid field itself when the caller did not override rids, else a search
    const uint32_t c = rid_is_index ? (uint32_t)(mm[i].y >> 32) : find_seg(off, n, i);
    if (i + 1 >= off[c + 1]) return;  // last shimmer of the contig starts no pair
    const pgr_mm128 s0 = mm[i], s1 = mm[i + 1];
    const uint64_t h0 = s0.x >> 8, h1 = s1.x >> 8;
    const bool keep = query_side ? (h0 < h1) : (h0 <= h1);  // seq_db.rs:1213 vs :391
    pgr_frag_rec r;
    r.h0 = keep ? h0 : h1;
    r.h1 = keep ? h1 : h0;
    r.frg_id = (uint32_t)(i - off[c]);
    r.sid = sids ? sids[c] : c;
    r.bgn = (uint32_t)((s0.y & 0xFFFFFFFFull) >> 1) + 1;
    r.end = (uint32_t)((s1.y & 0xFFFFFFFFull) >> 1) + 1;
    r.orient = keep ? 0u : 1u;
    r._pad = 0;
    out[rec_off[c] + (i - off[c])] = r;
}

void launch_frag_recs(hipStream_t st, const pgr_mm128 *mm, const uint64_t *off, const uint64_t *rec_off,
                      uint32_t n_contigs, uint64_t n, const uint32_t *sids, int query_side, int rid_is_index,
                      pgr_frag_rec *out) {
    if (n == 0) return;
    hipLaunchKernelGGL(frag_recs_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, mm, off, rec_off,
                       n_contigs, n, sids, query_side, rid_is_index, out);
}

// ---- the same on the device only, for a consumer that is enqueued BEHIND the shimmer pipeline before the host has seen its
// counts (query path, index.hip): rec_off from the list offsets by one workgroup (sixteen wavefronts each add up and then scan a
// contiguous sixteenth of the contigs), the records with the element count read from device memory.  The list may hold garbage
// when the pipeline is about to repeat a stage (result buffer smaller than the result): every index is checked.
__global__ __launch_bounds__(1024) void pair_offsets_kernel(const uint64_t *__restrict__ off, uint32_t n, uint64_t *__restrict__ rec_off,
                                                            uint32_t *__restrict__ clear3) {
    constexpr uint32_t NW = 16;
    constexpr int U = 8;  // chunks of 64 contigs whose loads are in flight together
    __shared__ unsigned long long tot[NW];
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (clear3 && t < 3) clear3[t] = 0;  // (the consumer's flag words: saves it a memset between two kernels)
    const uint32_t R = (((n + NW - 1) / NW) + 63) & ~63u;
    const uint32_t lo = w * R < n ? w * R : n, hi = lo + R < n ? lo + R : n;
    auto pairs_of = [&](uint32_t c) -> uint32_t {
        if (c >= hi) return 0u;
        const uint64_t a = off[c], b = off[c + 1];
        return b > a + 1 ? (uint32_t)(b - a - 1) : 0u;  // (b < a: garbage offsets of a pass that will be repeated)
    };
    unsigned long long s = 0;
    for (uint32_t c0 = lo; c0 < hi; c0 += 64 * U) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = pairs_of(c0 + 64 * u + lane);
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u];
    }
    for (int d = 32; d >= 1; d >>= 1) s += shfl_xor64(s, d);
    if (lane == 0) tot[w] = s;
    __syncthreads();
    unsigned long long base = 0;
    for (uint32_t x = 0; x < w; ++x) base += tot[x];
    for (uint32_t c0 = lo; c0 < hi; c0 += 64 * U) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = pairs_of(c0 + 64 * u + lane);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t c = c0 + 64 * u + lane;
            const uint32_t incl = wave_incl_sum(v[u]);
            if (c < hi) rec_off[c] = base + incl - v[u];
            base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    if (t == 0) {
        unsigned long long all = 0;
        for (uint32_t x = 0; x < NW; ++x) all += tot[x];
        rec_off[n] = all;
    }
}

__global__ __launch_bounds__(256) void frag_recs_dev_kernel(const pgr_mm128 *__restrict__ mm, const uint64_t *__restrict__ off,
                                                            const uint64_t *__restrict__ rec_off, uint32_t n, uint64_t cap,
                                                            const uint64_t *__restrict__ total_ptr, int query_side,
                                                            const uint32_t *__restrict__ sids, pgr_frag_rec *__restrict__ out,
                                                            uint64_t out_cap, const uint64_t *__restrict__ base_ptr) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = *total_ptr < cap ? *total_ptr : cap;
    if (i + 1 >= total) return;
    const uint32_t c = (uint32_t)(mm[i].y >> 32);  // (the rid field is the contig index: the caller did not override rids)
    if (c >= n || i < off[c] || i + 1 >= off[c + 1]) return;  // last shimmer of the contig starts no pair
    const uint64_t o = (base_ptr ? *base_ptr : 0ull) + rec_off[c] + (i - off[c]);  // (base_ptr: the cursor of the records' destination)
    if (o >= out_cap) return;
    const pgr_mm128 s0 = mm[i], s1 = mm[i + 1];
    const uint64_t h0 = s0.x >> 8, h1 = s1.x >> 8;
    const bool keep = query_side ? (h0 < h1) : (h0 <= h1);  // seq_db.rs:1213 vs :391
    pgr_frag_rec r;
    r.h0 = keep ? h0 : h1;
    r.h1 = keep ? h1 : h0;
    r.frg_id = (uint32_t)(i - off[c]);
    r.sid = sids ? sids[c] : c;
    r.bgn = (uint32_t)((s0.y & 0xFFFFFFFFull) >> 1) + 1;
    r.end = (uint32_t)((s1.y & 0xFFFFFFFFull) >> 1) + 1;
    r.orient = keep ? 0u : 1u;
    r._pad = 0;
    out[o] = r;
}

void launch_frag_recs_dev(hipStream_t st, const pgr_mm128 *mm, const uint64_t *off, uint32_t n_contigs, uint64_t cap,
                          const uint64_t *total_ptr, int query_side, uint64_t *rec_off, pgr_frag_rec *out, uint64_t out_cap,
                          uint32_t *clear3, const uint32_t *sids, const uint64_t *base_ptr, uint32_t lds_match) {
    uint32_t pad = 0;
    hipFuncAttributes at;
    if (lds_match && hipFuncGetAttributes(&at, (const void *)pair_offsets_kernel) == hipSuccess && at.sharedSizeBytes < lds_match)
        pad = lds_match - (uint32_t)at.sharedSizeBytes;
    hipLaunchKernelGGL(pair_offsets_kernel, dim3(1), dim3(1024), pad, st, off, n_contigs, rec_off, clear3);
    if (cap < 2) return;
    hipLaunchKernelGGL(frag_recs_dev_kernel, dim3((uint32_t)((cap + 255) / 256)), dim3(256), 0, st, mm, off, rec_off, n_contigs, cap,
                       total_ptr, query_side, sids, out, out_cap, base_ptr);
}

// cursor += *count; the value before goes to *before (a consumer that places its output behind its predecessor's without the host)
__global__ void cursor_bump_kernel(uint64_t *cursor, const uint64_t *count, uint64_t *before) {
    const uint64_t c = *cursor;
    *before = c;
    *cursor = c + *count;
}
void launch_cursor_bump(hipStream_t st, uint64_t *cursor, const uint64_t *count, uint64_t *before) {
    hipLaunchKernelGGL(cursor_bump_kernel, dim3(1), dim3(1), 0, st, cursor, count, before);
}

}  // namespace pgr

namespace pgr {
// per-contig offsets of the ordered level-1 list from the segment scan
__global__ void contig_offsets_kernel(const uint64_t *__restrict__ seg_dst, const uint32_t *__restrict__ tile_first,
                                      uint32_t n, uint32_t n_segs, uint64_t *__restrict__ off) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n) return;
    off[c] = (c == n) ? seg_dst[n_segs] : seg_dst[tile_first[c] + c];
}
void launch_contig_offsets(hipStream_t st, const uint64_t *seg_dst, const uint32_t *tile_first, uint32_t n,
                           uint32_t n_segs, uint64_t *off) {
    hipLaunchKernelGGL(contig_offsets_kernel, dim3((n + 1 + 255) / 256), dim3(256), 0, st, seg_dst, tile_first, n,
                       n_segs, off);
}
// final list with the reference's padding artefact: contigs whose level-1 list is empty yield two
// {MAX,MAX} sentinels when padding is on and r > 1 (shmmrutils.rs:367-380 on an empty input)
__global__ void copy_or_sentinel_kernel(const pgr_mm128 *__restrict__ in, const uint64_t *__restrict__ off_in,
                                        const uint64_t *__restrict__ off_out, uint32_t n,
                                        pgr_mm128 *__restrict__ out) {
    const uint32_t c = blockIdx.x;
    const uint64_t si = off_in[c], cnt_in = off_in[c + 1] - si;
    const uint64_t so = off_out[c], cnt_out = off_out[c + 1] - so;
    if (cnt_in == cnt_out) {
        for (uint64_t i = threadIdx.x; i < cnt_in; i += blockDim.x) out[so + i] = in[si + i];
    } else {
        for (uint64_t i = threadIdx.x; i < cnt_out; i += blockDim.x) {
            pgr_mm128 m;
            m.x = U64MAX;
            m.y = U64MAX;
            out[so + i] = m;
        }
    }
}
void launch_copy_or_sentinel(hipStream_t st, const pgr_mm128 *in, const uint64_t *off_in, const uint64_t *off_out,
                             uint32_t n, pgr_mm128 *out) {
    if (n == 0) return;
    hipLaunchKernelGGL(copy_or_sentinel_kernel, dim3(n), dim3(256), 0, st, in, off_in, off_out, n, out);
}
}  // namespace pgr

// ================================================================================================
// Fused list stage: reduce_shmmr x2 + min_span stencil in ONE pass over the level-1 segments.
//
// A workgroup owns FUSED_B consecutive elements of the (logically ordered) level-1 stream plus a halo of
// H = 2 r^2 elements on each side.  H bounds the dependency radius of the final decision:
//   - consecutive survivors of a reduction are at most r list elements apart (every full r-window
//     contains its own minimum), so the r-1 reduced neighbours an element needs for the second
//     reduction lie within r(r-1) level-1 elements, and the previous / next twice-reduced element the
//     min_span stencil needs lies within r^2; each of those needs its own +-(r-1) context.
// The stream is read straight from the unordered per-tile segments (seg_off / seg_cnt / seg_dst = scan of
// seg_cnt): no ordered copy of the level-1 list is ever materialised.  Survivors of the core range go to a
// cursor-allocated block segment; a scan + gather_segments_kernel orders them afterwards.
namespace pgr {

namespace {

constexpr int FUSED_T = 256;
constexpr int FUSED_B = 1024;
constexpr int FUSED_EMAX_BIG = FUSED_B + 2 * 288;   // r <= 12
constexpr int FUSED_EMAX_SMALL = FUSED_B + 2 * 32;  // r <= 4 (the common spec): 22 KB of LDS, 7 workgroups / CU

constexpr int FUSED_SEGS = 128;  // segment descriptors staged per round (a block of a read batch spans ~90 segments)

template <int EMAX>
struct FusedLdsT {
    static constexpr int CMAX = (EMAX + FUSED_T - 1) / FUSED_T;  // items per lane, lane-strided (k = j*256 + t)
    uint64_t x[EMAX];
    uint64_t y[EMAX];
    uint16_t s1[EMAX];
    uint16_t s2[EMAX];
    uint32_t cnt[CMAX][FUSED_T / 64];   // per (j, wave) survivor counts
    uint32_t base[CMAX][FUSED_T / 64];  // their exclusive prefix in (j, wave) order
    uint32_t total;
    uint64_t sdst[FUSED_SEGS + 1];  // logical start of the staged segments (+ sentinel)
    uint64_t soff[FUSED_SEGS];      // physical offset
    uint32_t scid[FUSED_SEGS];      // contig of the segment's records
    uint32_t n_seg;
    uint32_t wide;    // the block spans more than 254 contigs: the packed keys cannot tell them apart
    uint32_t wide64;  // ... more than 125: the packed keys are not all positive normal doubles (reduce_round_wave)
    uint32_t ccnt[(EMAX + 57) / 58 + 1];  // survivors per chunk of 58 list places (reduce_round_wave)
    unsigned long long base_out;
};

// Ordered compaction for lane-strided items k = j*256 + t (consecutive lanes <-> consecutive elements, so
// the neighbour reads of the predicates are LDS-conflict free).  keep[j] per lane -> rank of every kept item
// in k order.  One barrier per call.
template <int CM, class FusedLds>
__device__ __forceinline__ void strided_ranks(FusedLds &L, const bool (&keep)[CM], int C, uint32_t (&rank)[CM],
                                              uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t lt = (lane == 0) ? 0ull : (U64MAX >> (64 - lane));
    uint64_t bal[CM];
#pragma unroll
    for (int j = 0; j < CM; ++j) {
        bal[j] = (j < C) ? __ballot(keep[j]) : 0ull;
        if (j < C && lane == 0) L.cnt[j][wv] = (uint32_t)__popcll(bal[j]);
    }
    __syncthreads();
    // every lane adds up the (j, wave) counts that precede its own in (j, wave) order: 4 C broadcast LDS reads
    // instead of a serial prefix by one thread between two more barriers
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < CM; ++j) {
        rank[j] = 0u;
        if (j >= C) continue;
#pragma unroll
        for (int v = 0; v < FUSED_T / 64; ++v) {
            if (v == (int)wv) rank[j] = acc + (uint32_t)__popcll(bal[j] & lt);
            acc += L.cnt[j][v];
        }
    }
    *total = acc;  // (L.cnt is only rewritten after the barrier every caller places behind its list writes)
}

// reduce predicate on an indexed LDS list: list[k] -> element index e; neighbours must share the contig id.
// Branch-free over the 2(r-1) neighbours: all LDS reads are issued up front (a loop with an early exit is a
// chain of dependent ~100-cycle LDS round trips and made this kernel latency bound).  TR = compile-time r
// (0: runtime).
template <int TR, class FusedLds>
__device__ __forceinline__ bool reduce_keep_lds(const FusedLds &L, const uint16_t *list, int n, int k, uint32_t r_rt,
                                                uint32_t padding, bool lo_is_start, bool hi_is_end) {
    const uint32_t r = TR ? (uint32_t)TR : r_rt;
    const int e = list ? list[k] : k;
    const uint64_t xi = L.x[e];
    const uint32_t cid = (uint32_t)(L.y[e] >> 32);
    uint32_t a = 0, b = 0;
    bool run_a = true, run_b = true;
#pragma unroll
    for (uint32_t d = 1; d < (TR ? (uint32_t)TR : 12u); ++d) {
        if (!TR && d >= r) break;
        // left neighbour
        {
            const int kk = k - (int)d;
            const bool inside = kk >= 0;
            const int ee = inside ? (list ? list[kk] : kk) : e;
            const bool same = inside && (uint32_t)(L.y[ee] >> 32) == cid;
            // a true list / contig boundary: with padding the virtual sentinels are >= everything
            // (kk < 0 with !lo_is_start is a loading edge: only reached by far-halo elements)
            const bool virt = !same && padding && (inside || lo_is_start);
            const bool ge = same ? (L.x[ee] >= xi) : virt;
            run_a = run_a && ge;
            a += run_a ? 1u : 0u;
        }
        {
            const int kk = k + (int)d;
            const bool inside = kk < n;
            const int ee = inside ? (list ? list[kk] : kk) : e;
            const bool same = inside && (uint32_t)(L.y[ee] >> 32) == cid;
            const bool virt = !same && padding && (inside || hi_is_end);
            const bool ge = same ? (L.x[ee] >= xi) : virt;
            run_b = run_b && ge;
            b += run_b ? 1u : 0u;
        }
    }
    return a + b + 1 >= r;
}

// The same predicate on PACKED keys: L.x[e] = ordinal << 56 | hash key, ordinal = contig of the element minus the contig of
// the block's first element (non-decreasing along the list, at most 254).  A left neighbour of another contig has a
// smaller ordinal, so its packed key is smaller and "P_n >= P_i" is false by itself; a right neighbour of another
// contig has a larger ordinal: one compare of the high words against (ordinal_i + 1) << 24 rules it out.  No contig-id
// reads (6 of the 12 LDS reads per element of the first reduction).  Only without padding.
template <int TR, class FusedLds>
__device__ __forceinline__ bool reduce_keep_packed(const FusedLds &L, const uint16_t *list, int n, int k, uint32_t r_rt) {
    const uint32_t r = TR ? (uint32_t)TR : r_rt;
    const int e = list ? list[k] : k;
    const uint64_t pi = L.x[e];
    const uint32_t lim = (((uint32_t)(pi >> 56)) + 1u) << 24;
    uint32_t a = 0, b = 0;
    bool run_a = true, run_b = true;
#pragma unroll
    for (uint32_t d = 1; d < (TR ? (uint32_t)TR : 12u); ++d) {
        if (!TR && d >= r) break;
        {
            const int kk = k - (int)d;
            const bool inside = kk >= 0;
            const int ee = inside ? (list ? list[kk] : kk) : e;
            run_a = run_a && inside && L.x[ee] >= pi;
            a += run_a ? 1u : 0u;
        }
        {
            const int kk = k + (int)d;
            const bool inside = kk < n;
            const int ee = inside ? (list ? list[kk] : kk) : e;
            const uint64_t pn = L.x[ee];
            run_b = run_b && inside && pn >= pi && (uint32_t)(pn >> 32) < lim;
            b += run_b ? 1u : 0u;
        }
    }
    return a + b + 1 >= r;
}

// ---- One reduction (reduce_shmmr with r = 4, shmmrutils.rs:359-415) over an LDS list, neighbours from REGISTERS.
// The closed form of the machine -- an element survives iff it is a minimum (ties included) of some full r-window of its contig's
// list -- is the level-1 selection again with w = 4: M[j] = min(P[j-3 .. j]), E[i] = max(M[i .. i+3]), survive iff E[i] == P[i].
// A wavefront takes 58 consecutive list places (+ 3 on both sides) with place p in lane p - first + 3; neighbours come by DPP
// wave shifts, the two minima / maxima by doubling (2 + 2 instructions), and the packed keys (ordinal + 1) << 56 | hash -- ordinal
// <= 124 -- are positive normal doubles ordered like the integers, so they are v_min_f64 / v_max_f64 (0 = "no element" /
// "no window" is +0.0, below every key).  A window that crosses a contig boundary, or the list's end, has a minimum whose
// ordinal differs from the ordinal of its last element (ordinals do not decrease along the list; an empty place is 0): it counts
// as no window.  reduce_keep_packed did the same with 6 dependent LDS reads, 12 64-bit compares and their selects per element.
__device__ __forceinline__ double dpp_shr1(double v) {  // lane l <- lane l - 1 (lane 0: +0.0)
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x138, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double dpp_shl1(double v) {  // lane l <- lane l + 1 (lane 63: +0.0)
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x130, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double f64min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double f64max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

constexpr int RW_CORE = 58;  // list places a wavefront decides per step (64 lanes - 2 x 3 neighbours)
template <int ITMAX, class FusedLds>
__device__ __forceinline__ void reduce_round_wave(FusedLds &L, const uint16_t *cur, int n, uint16_t *dstl, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: scalar chunk indices
    const uint64_t lt = (lane == 0) ? 0ull : (U64MAX >> (64 - lane));
    const int n_chunks = (n + RW_CORE - 1) / RW_CORE;
    uint64_t bal[ITMAX];
    uint16_t elem[ITMAX];
#pragma unroll
    for (int it = 0; it < ITMAX; ++it) {
        const int c = it * (FUSED_T / 64) + (int)wv;
        bal[it] = 0;
        elem[it] = 0;
        if (c >= n_chunks) continue;  // (wave-uniform)
        const int p = c * RW_CORE + (int)lane - 3;
        const bool inside = p >= 0 && p < n;
        const int e = inside ? (cur ? (int)cur[p] : p) : 0;
        elem[it] = (uint16_t)e;
        const double P = inside ? __longlong_as_double((long long)L.x[e]) : 0.0;
        const double m2 = f64min(P, dpp_shr1(P));
        const double m4 = f64min(m2, dpp_shr1(dpp_shr1(m2)));
        // window [l-3, l] lies in one contig (and holds four elements) iff its minimum carries this lane's ordinal
        const uint32_t hm = (uint32_t)((uint64_t)__double_as_longlong(m4) >> 32), hp = (uint32_t)((uint64_t)__double_as_longlong(P) >> 32);
        const double M = ((hm ^ hp) >> 24) == 0u ? m4 : 0.0;
        const double e2 = f64max(M, dpp_shl1(M));
        const double E = f64max(e2, dpp_shl1(dpp_shl1(e2)));
        const bool keep = inside && lane >= 3 && lane < 3 + RW_CORE &&
                          (uint64_t)__double_as_longlong(E) == (uint64_t)__double_as_longlong(P);
        bal[it] = __ballot(keep);
        if (lane == 0) L.ccnt[c] = (uint32_t)__popcll(bal[it]);
    }
    __syncthreads();
    // exclusive prefix of the chunk counts (lane c holds chunk c: one wave scan), then the survivors' list in order
    const uint32_t mine = (int)lane < n_chunks ? L.ccnt[lane] : 0u;
    const uint32_t incl = wave_incl_sum(mine);
#pragma unroll
    for (int it = 0; it < ITMAX; ++it) {
        const int c = it * (FUSED_T / 64) + (int)wv;
        if (c >= n_chunks) continue;
        const uint32_t base = c ? (uint32_t)__builtin_amdgcn_readlane((int)incl, c - 1) : 0u;
        if ((bal[it] >> lane) & 1ull) dstl[base + (uint32_t)__popcll(bal[it] & lt)] = elem[it];
    }
    *total = n_chunks ? (uint32_t)__builtin_amdgcn_readlane((int)incl, n_chunks - 1) : 0u;
}

}  // namespace

// FusedArgsPub (pgr_internal.h): l1 = unordered level-1 segments; seg_dst = exclusive scan of seg_cnt
// ([n_segs+1]); out = fixed block slots + overflow region; cursor[0] overflow allocated, [1] overflow flag.
using FusedArgs = FusedArgsPub;

// FB: list elements a workgroup owns (FUSED_B = 1024; 512 for the workgroups of a pipelined job: 14 KB of LDS, which fit BESIDE the
// four tile workgroups of a CU -- 4 x 36 352 of 163 840 bytes leave 18 432)
template <int EMAX, int FB>
__device__ __forceinline__ void fused_select_block(const FusedArgs &a, const uint32_t blk) {
    using FusedLds = FusedLdsT<EMAX>;
    constexpr int FUSED_CMAX = FusedLds::CMAX;
    __shared__ FusedLds L;
    const uint32_t t = threadIdx.x;
    const uint64_t total = *a.total;
    const uint64_t core_lo = (uint64_t)blk * FB;
    if (core_lo >= total) {  // the grid is an upper bound (sized before the level-1 count is known): nothing to do here
        if (t == 0) {
            a.blk_off[blk] = 0;
            a.blk_cnt[blk] = 0;
        }
        return;
    }
    uint64_t core_hi = core_lo + FB;
    if (core_hi > total) core_hi = total;
    const uint64_t lo = core_lo >= a.halo ? core_lo - a.halo : 0;
    uint64_t hi = core_hi + a.halo;
    if (hi > total) hi = total;
    const int ne = (int)(hi - lo);
    const bool lo_is_start = (lo == 0), hi_is_end = (hi == total);

    // ---- stream [lo, hi) of the logical level-1 list into LDS.  Segment descriptors are staged 64 at a time
    // (the segment holding `lo` was located by block_first_seg_kernel); then every lane fetches its own
    // elements (k = j*256 + t) independently: ~5 outstanding 16-byte loads per lane.
    uint32_t seg0 = a.blk_first_seg[blk];
    uint64_t done = lo;  // logical elements below `done` are loaded
    uint32_t cid0 = 0;   // contig of the block's first element (the first staged segment holds it)
    bool first_batch = true;
    if (t == 0) {
        L.wide = 0;
        L.wide64 = 0;
    }
    while (done < hi) {
        if (t < FUSED_SEGS + 1) {
            const uint32_t sg = seg0 + t;
            const uint64_t d = (sg <= a.n_segs) ? a.seg_dst[sg] : total;  // seg_dst[n_segs] = total
            L.sdst[t] = d;
            if (t < FUSED_SEGS) {
                L.soff[t] = (sg < a.n_segs) ? a.seg_off[sg] : 0;
                L.scid[t] = (sg < a.n_segs) ? a.seg_cid[sg] : 0;
            }
        }
        __syncthreads();
        if (first_batch) {
            cid0 = L.scid[0];
            first_batch = false;
        }
        // elements covered by the staged descriptors: [sdst[0], sdst[64]) intersected with [done, hi)
        const uint64_t cover_hi = L.sdst[FUSED_SEGS] < hi ? L.sdst[FUSED_SEGS] : hi;
        // One wavefront per staged segment, lane i its element i, i + 64, ...: a segment is a tile's ~100 minimizers, contiguous in
        // the level-1 buffer, so the copy needs no search (round 3: every lane looked its elements up in the descriptors -- a
        // binary search + 64-bit offset arithmetic per element, a third of the kernel's VALU instructions).
        {
            const uint32_t lane = t & 63;
            const int wv = __builtin_amdgcn_readfirstlane((int)(t >> 6));
            // LOAD_U segments at a time: their first 64 records are in flight together before any goes to LDS (a batch of reads
            // is ~25 minimizers per segment: one dependent global load per segment left the wavefront waiting ~1 us for each)
            constexpr int LOAD_U = 4;
            auto put = [&](uint32_t at, const L1Rec &m, uint32_t cid, uint32_t top) {
                // 12-byte record -> MM128: x = key << 8 | k, y = contig << 32 | pos << 1 | strand
#if PGR_L2_PACKED
                L.x[at] = ((uint64_t)(top | m.key_hi) << 32) | m.key_lo;
#else
                L.x[at] = ((((uint64_t)m.key_hi << 32) | m.key_lo) << 8) | (uint64_t)a.k;
#endif
                L.y[at] = ((uint64_t)cid << 32) | m.ypos;
            };
            for (int sgb = wv; sgb < FUSED_SEGS; sgb += LOAD_U * (FUSED_T / 64)) {
                if (L.sdst[sgb] >= cover_hi) break;  // (logical starts do not decrease)
                const L1Rec *src[LOAD_U];
                uint32_t cnt[LOAD_U], d0[LOAD_U], cid[LOAD_U], top[LOAD_U];
                L1Rec m[LOAD_U];
#pragma unroll
                for (int u = 0; u < LOAD_U; ++u) {
                    const int sg = sgb + u * (FUSED_T / 64);
                    cnt[u] = 0;
                    src[u] = a.l1;
                    d0[u] = cid[u] = top[u] = 0;
                    if (sg < FUSED_SEGS) {
                        const uint64_t s_lo = L.sdst[sg], s_hi = L.sdst[sg + 1];
                        const uint64_t g0 = s_lo > done ? s_lo : done, g1 = s_hi < cover_hi ? s_hi : cover_hi;
                        if (g1 > g0) {  // (not an empty segment, one below `done` or one behind the block)
                            src[u] = a.l1 + (L.soff[sg] + (g0 - s_lo));
                            cid[u] = L.scid[sg];
                            cnt[u] = (uint32_t)(g1 - g0);
                            d0[u] = (uint32_t)(g0 - lo);
#if PGR_L2_PACKED
                            const uint32_t ord = cid[u] - cid0;
                            if (ord > 253u) L.wide = 1;    // benign race: every writer stores 1
                            if (ord > 124u) L.wide64 = 1;  // (the same)
                            top[u] = ((ord + 1u) & 0xFFu) << 24;
#endif
                        }
                    }
                    if (lane < cnt[u]) m[u] = src[u][lane];
                }
#pragma unroll
                for (int u = 0; u < LOAD_U; ++u) {
                    if (lane < cnt[u]) put(d0[u] + lane, m[u], cid[u], top[u]);
                    for (uint32_t i = lane + 64; i < cnt[u]; i += 64) put(d0[u] + i, src[u][i], cid[u], top[u]);
                }
            }
        }
        const bool stalled = cover_hi <= done;  // 64 segments without an element of [done, hi): the inside of a gap (emptied tiles)
        done = cover_hi > done ? cover_hi : done;
        seg0 += FUSED_SEGS;
        if (stalled && seg0 < a.n_segs) {
            // the workgroup at the end of an 18 Mbp gap used to stage its 4600 empty segments 64 at a time (~2 us a round, the
            // whole list stage of a chromosome waiting for it): the last segment that starts at or below `done`, by bisection
            uint32_t s_lo = seg0, s_hi = a.n_segs;  // seg_dst[seg0] <= done < total = seg_dst[n_segs]
            while (s_hi - s_lo > 1) {
                const uint32_t mid = (s_lo + s_hi) >> 1;
                if (a.seg_dst[mid] <= done) s_lo = mid;
                else s_hi = mid;
            }
            seg0 = s_lo;
        }
        __syncthreads();
        if (seg0 >= a.n_segs && done < hi) break;  // cannot happen: seg_dst[n_segs] == total >= hi
    }

#if PGR_ABLATE_L2 == 1
    if (L.x[t] != 0x1234567ull) {  // load only
        if (t == 0) { a.blk_off[blk] = 0; a.blk_cnt[blk] = 0; }
        return;
    }
#endif
#if PGR_L2_PACKED
    // the generic predicate compares contig ids explicitly and the key order is what it needs: it also works on packed
    // keys EXCEPT that keys of different contigs must not be told apart by the ordinal when the block is wide (ordinals
    // wrapped) -- there the ordinal byte is cleared again
    const bool packed = !a.padding && L.wide == 0;
    if (!packed) {
        for (int e = (int)t; e < ne; e += FUSED_T) L.x[e] &= 0x00FFFFFFFFFFFFFFull;
        __syncthreads();
    }
#endif
    // ---- reduce x2 (shmmrutils.rs:533-535) on index lists, then the min_span stencil (:536-555)
    const uint16_t *cur = nullptr;  // nullptr = identity list over [0, ne)
    int n_cur = ne;
#if PGR_L2_PACKED
    const bool wave_rounds = a.do_reduce && a.r == 4 && packed && L.wide64 == 0;
#else
    const bool wave_rounds = false;
#endif
    if (wave_rounds) {
        constexpr int ITMAX = ((EMAX + RW_CORE - 1) / RW_CORE + FUSED_T / 64 - 1) / (FUSED_T / 64);
        static_assert((EMAX + RW_CORE - 1) / RW_CORE <= 64, "one wave scan over the chunk counts");
        for (int round = 0; round < 2; ++round) {
            uint16_t *dstl = round == 0 ? L.s1 : L.s2;
            uint32_t tot;
            reduce_round_wave<ITMAX, FusedLds>(L, cur, n_cur, dstl, &tot);
            __syncthreads();
            cur = dstl;
            n_cur = (int)tot;
        }
    } else if (a.do_reduce) {
        for (int round = 0; round < 2; ++round) {
            uint16_t *dstl = round == 0 ? L.s1 : L.s2;
            const int C = (n_cur + FUSED_T - 1) / FUSED_T;
            bool keep[FUSED_CMAX];
            uint32_t rank[FUSED_CMAX];
#pragma unroll
            for (int j = 0; j < FUSED_CMAX; ++j) {
                const int k = j * FUSED_T + (int)t;
#if PGR_L2_PACKED
                if (packed)
                    keep[j] = (j < C) && k < n_cur &&
                              (a.r == 4 ? reduce_keep_packed<4, FusedLds>(L, cur, n_cur, k, 4) : reduce_keep_packed<0, FusedLds>(L, cur, n_cur, k, a.r));
                else
#endif
                keep[j] = (j < C) && k < n_cur &&
                          (a.r == 4 ? reduce_keep_lds<4, FusedLds>(L, cur, n_cur, k, 4, a.padding, lo_is_start, hi_is_end)
                                    : reduce_keep_lds<0, FusedLds>(L, cur, n_cur, k, a.r, a.padding, lo_is_start, hi_is_end));
            }
            uint32_t tot;
            strided_ranks<FUSED_CMAX, FusedLds>(L, keep, C, rank, &tot);
#pragma unroll
            for (int j = 0; j < FUSED_CMAX; ++j)
                if (keep[j]) {
                    const int k = j * FUSED_T + (int)t;
                    dstl[rank[j]] = (uint16_t)(cur ? cur[k] : k);
                }
            __syncthreads();
            cur = dstl;
            n_cur = (int)tot;
        }
    }
    // span filter (shmmrutils.rs:536-555) over `cur`; survivors that are core elements are emitted in order.  The same
    // wave-per-chunk scheme as the reductions: 62 list places per step, the two neighbours' key, position and contig by DPP.
    const int c_lo = (int)(core_lo - lo), c_hi = (int)(core_hi - lo);
    constexpr int SF_CORE = 62;
    constexpr int SF_ITMAX = ((EMAX + SF_CORE - 1) / SF_CORE + FUSED_T / 64 - 1) / (FUSED_T / 64);
    const uint32_t lane = t & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
    const uint64_t lt = (lane == 0) ? 0ull : (U64MAX >> (64 - lane));
    const int n_chunks = (n_cur + SF_CORE - 1) / SF_CORE;
    uint64_t bal[SF_ITMAX], kx[SF_ITMAX], ky[SF_ITMAX];
#pragma unroll
    for (int it = 0; it < SF_ITMAX; ++it) {
        const int c = it * (FUSED_T / 64) + (int)wv;
        bal[it] = 0;
        kx[it] = ky[it] = 0;
        if (c >= n_chunks) continue;  // (wave-uniform)
        const int p = c * SF_CORE + (int)lane - 1;
        const bool inside = p >= 0 && p < n_cur;
        const int e = inside ? (cur ? (int)cur[p] : p) : 0;
        const uint64_t px = inside ? L.x[e] : 0ull, py = inside ? L.y[e] : ~0ull;  // (contig ~0: no neighbour shares it)
        kx[it] = px;
        ky[it] = py;
        const uint32_t xl = (uint32_t)px, xh = (uint32_t)(px >> 32), yl = (uint32_t)py, cid = (uint32_t)(py >> 32);
        // previous / next list place (lane 0 / 63 of the step: not core lanes)
        const uint32_t p_xl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xl, 0x138, 0xf, 0xf, false);
        const uint32_t p_xh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xh, 0x138, 0xf, 0xf, false);
        const uint32_t p_yl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)yl, 0x138, 0xf, 0xf, false);
        const uint32_t p_cid = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)cid, 0x138, 0xf, 0xf, false);
        const uint32_t n_xl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xl, 0x130, 0xf, 0xf, false);
        const uint32_t n_xh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xh, 0x130, 0xf, 0xf, false);
        const uint32_t n_yl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)yl, 0x130, 0xf, 0xf, false);
        const uint32_t n_cid = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)cid, 0x130, 0xf, 0xf, false);
        bool kp = inside && lane >= 1 && lane <= SF_CORE && e >= c_lo && e < c_hi;
        // (a place outside the list carries contig ~0, so "no previous / next element" and "another contig" are one test)
        const bool first = p_cid != cid, last = n_cid != cid;
        if (!first && !last) {
            const uint32_t pp = p_yl >> 1, mp = yl >> 1, np = n_yl >> 1;
            kp = kp && (uint32_t)(mp - pp) > a.min_span && (uint32_t)(np - mp) > a.min_span && (p_xl != xl || p_xh != xh) &&
                 (n_xl != xl || n_xh != xh);
        }
        bal[it] = __ballot(kp);
        if (lane == 0) L.ccnt[c] = (uint32_t)__popcll(bal[it]);
    }
    __syncthreads();
    const uint32_t mine = (int)lane < n_chunks ? L.ccnt[lane] : 0u;
    const uint32_t incl = wave_incl_sum(mine);
    const uint32_t tot = n_chunks ? (uint32_t)__builtin_amdgcn_readlane((int)incl, n_chunks - 1) : 0u;
    if (t == 0) {
        // fixed slot per workgroup; only blocks with more survivors than the slot use the shared cursor
        // (same-address atomics saturate at ~88/us on gfx950)
        unsigned long long base = (unsigned long long)blk * a.slot;
        bool ok = true;
        if (tot > a.slot) {
            const unsigned long long ob = atomicAdd(a.cursor, (unsigned long long)tot);
            base = a.ovf_base + ob;
            ok = ob + tot <= a.cap;
            if (!ok) atomicExch(a.cursor + 1, 1ull);
        }
        L.base_out = ok ? base : ~0ull;
        a.blk_off[blk] = base;
        a.blk_cnt[blk] = ok ? tot : 0u;
    }
    __syncthreads();
    const unsigned long long base = L.base_out;
    if (base != ~0ull) {
#pragma unroll
        for (int it = 0; it < SF_ITMAX; ++it) {
            const int c = it * (FUSED_T / 64) + (int)wv;
            if (c >= n_chunks) continue;
            const uint32_t cb = c ? (uint32_t)__builtin_amdgcn_readlane((int)incl, c - 1) : 0u;  // (all lanes: uniform)
            if ((bal[it] >> lane) & 1ull) {
                pgr_mm128 m;
#if PGR_L2_PACKED
                m.x = (kx[it] << 8) | (uint64_t)a.k;  // the ordinal byte falls off the top
#else
                m.x = kx[it];
#endif
                m.y = ky[it];
                a.out[base + cb + (uint32_t)__popcll(bal[it] & lt)] = m;
            }
        }
    }
}

template <int EMAX, int FB>
__global__ __launch_bounds__(FUSED_T) void fused_select_kernel(FusedArgs a) {
    fused_select_block<EMAX, FB>(a, blockIdx.x);
}
// The same as a PERSISTENT grid (a pipelined job's list stage, context option pipe_persistent_list): gridDim.x workgroups of the
// 14 KB variant loop over the blocks blk, blk + gridDim.x, ...  Launched with about one workgroup per CU it lives in the 18 KB of
// LDS that a CU's four tile workgroups leave free and never asks the dispatcher for a tile's slot again -- a list workgroup that
// takes a tile's 36 KB range spends most of its time waiting for memory there.  No prefetch of the next block (the
// form of round 5 that cost 54 VGPRs): the loop is around the body as it is.
template <int EMAX, int FB>
__global__ __launch_bounds__(FUSED_T) void fused_select_persistent_kernel(FusedArgs a, uint32_t n_blocks) {
    for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        fused_select_block<EMAX, FB>(a, blk);
        __syncthreads();  // (the next block's first writes to LDS)
    }
}

// first segment of every fused workgroup: largest s with seg_dst[s] <= max(0, b*FUSED_B - halo)
__global__ void block_first_seg_kernel(const uint64_t *__restrict__ seg_dst, uint32_t n_segs, uint32_t n_blocks,
                                       uint32_t halo, uint32_t *__restrict__ blk_first_seg, uint32_t *__restrict__ blk_cnt, uint32_t fb) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) blk_cnt[n_blocks] = 0;  // sentinel of the scan over the block counts (saves a memset per call)
    if (b >= n_blocks) return;
    const uint64_t core_lo = (uint64_t)b * fb;
    const uint64_t lo = core_lo >= halo ? core_lo - halo : 0;
    uint32_t s_lo = 0, s_hi = n_segs;
    while (s_hi - s_lo > 1) {
        const uint32_t mid = (s_lo + s_hi) >> 1;
        if (seg_dst[mid] <= lo) s_lo = mid;
        else s_hi = mid;
    }
    blk_first_seg[b] = s_lo;
}

// offsets of the final ordered list: off[c] = first element whose (internal) rid >= c.  The first ten threads also collect the
// pipeline's status words (cursors, level-1 total, final count) in front of the offsets: one copy brings both to the host.
__global__ void offsets_by_rid_kernel(const pgr_mm128 *__restrict__ mm, const uint64_t *__restrict__ n_ptr, uint64_t cap,
                                      uint32_t n_contigs, uint64_t *__restrict__ off,
                                      const unsigned long long *__restrict__ cursor, const uint64_t *__restrict__ total1,
                                      uint64_t *__restrict__ status) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (status) {
        if (c < 8) status[c] = cursor[c];
        if (c == 8) status[8] = *total1;
        if (c == 9) status[9] = *n_ptr;
    }
    if (c > n_contigs) return;
    const uint64_t n = *n_ptr < cap ? *n_ptr : cap;
    uint64_t lo = 0, hi = n;  // first j with rid(j) >= c
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint32_t)(mm[mid].y >> 32) < c) lo = mid + 1;
        else hi = mid;
    }
    off[c] = lo;
}

__global__ void patch_rid_kernel(pgr_mm128 *__restrict__ mm, const uint64_t *__restrict__ n_ptr, uint64_t cap,
                                 const uint32_t *__restrict__ rids, uint32_t n_contigs) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap || i >= *n_ptr) return;
    const uint64_t y = mm[i].y;
    const uint32_t c = (uint32_t)(y >> 32);
    // when the result buffer turned out too small the gather left holes (uninitialised elements): their "contig" is
    // garbage and must not index rids[] -- the host sees the true count and repeats the stage with a bigger buffer
    if (c >= n_contigs) return;
    mm[i].y = ((uint64_t)rids[c] << 32) | (y & 0xFFFFFFFFull);
}

void launch_fused_select_pub(hipStream_t st, const FusedArgsPub &a, uint32_t n_blocks) {
    if (n_blocks == 0) return;
    hipLaunchKernelGGL(block_first_seg_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, a.seg_dst, a.n_segs,
                       n_blocks, a.halo, a.blk_first_seg, a.blk_cnt, a.block_elems == 512u && a.halo <= 32 ? 512u : (uint32_t)FUSED_B);
    // (dynamic LDS on top of the static block: the workgroup then occupies exactly a.lds_match bytes of its CU)
    auto pad_for = [&](const void *f) -> uint32_t {
        hipFuncAttributes at;
        if (!a.lds_match || hipFuncGetAttributes(&at, f) != hipSuccess || at.sharedSizeBytes >= a.lds_match) return 0u;
        return a.lds_match - (uint32_t)at.sharedSizeBytes;
    };
    if (a.halo <= 32 && a.block_elems == 512u && a.persistent_grid)
        hipLaunchKernelGGL((fused_select_persistent_kernel<512 + 2 * 32, 512>), dim3(std::min(a.persistent_grid, n_blocks)), dim3(FUSED_T), 0, st, a,
                           n_blocks);
    else if (a.halo <= 32 && a.block_elems == 512u)
        hipLaunchKernelGGL((fused_select_kernel<512 + 2 * 32, 512>), dim3(n_blocks), dim3(FUSED_T), 0, st, a);
    else if (a.halo <= 32)
        hipLaunchKernelGGL((fused_select_kernel<FUSED_EMAX_SMALL, FUSED_B>), dim3(n_blocks), dim3(FUSED_T),
                           pad_for((const void *)fused_select_kernel<FUSED_EMAX_SMALL, FUSED_B>), st, a);
    else
        hipLaunchKernelGGL((fused_select_kernel<FUSED_EMAX_BIG, FUSED_B>), dim3(n_blocks), dim3(FUSED_T),
                           pad_for((const void *)fused_select_kernel<FUSED_EMAX_BIG, FUSED_B>), st, a);
}
void launch_offsets_by_rid(hipStream_t st, const pgr_mm128 *mm, const uint64_t *n_ptr, uint64_t cap, uint32_t n_contigs,
                           uint64_t *off, const unsigned long long *cursor, const uint64_t *total1, uint64_t *status) {
    hipLaunchKernelGGL(offsets_by_rid_kernel, dim3((n_contigs + 1 + 255) / 256), dim3(256), 0, st, mm, n_ptr, cap, n_contigs,
                       off, cursor, total1, status);
}
__global__ void collect_status_kernel(const unsigned long long *__restrict__ cursor, const uint64_t *__restrict__ total1,
                                      const uint64_t *__restrict__ n_final, uint64_t *__restrict__ status) {
    const uint32_t t = threadIdx.x;
    if (t < 8) status[t] = cursor[t];
    if (t == 8) status[8] = *total1;
    if (t == 9) status[9] = *n_final;
}
void launch_collect_status(hipStream_t st, const unsigned long long *cursor, const uint64_t *total1, const uint64_t *n_final,
                           uint64_t *status) {
    hipLaunchKernelGGL(collect_status_kernel, dim3(1), dim3(64), 0, st, cursor, total1, n_final, status);
}

// ------------------------------------------------------------------ content checksum (bench.py / tests: full-size parity)
// sums[2c] += splitmix64(x ^ K1 (i+1)), sums[2c+1] += splitmix64((ylo | i << 32) + K2 x), i = ordinal of the element in
// contig c (splitmix64 = the usual 3-step finaliser, additive constant 0x9E3779B97F4A7C15; K1 = that constant,
// K2 = 0xD1B54A32D192ED03; sums wrap mod 2^64).  blockIdx.x = contig, blockIdx.y = slice of 4096.
__device__ __forceinline__ uint64_t splitmix64_dev(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void shmmr_checksum_kernel(const pgr_mm128 *__restrict__ mm, const uint64_t *__restrict__ off,
                                                             unsigned long long *__restrict__ sums) {
    __shared__ unsigned long long s_a[4], s_b[4];
    const uint32_t c = blockIdx.x;
    const uint64_t o = off[c], cnt = off[c + 1] - o;
    const uint64_t i0 = (uint64_t)blockIdx.y * 4096;
    if (i0 >= cnt) return;
    uint64_t a = 0, b = 0;
    for (uint64_t i = i0 + threadIdx.x; i < cnt && i < i0 + 4096; i += 256) {
        const pgr_mm128 m = mm[o + i];
        a += splitmix64_dev(m.x ^ (0x9E3779B97F4A7C15ull * (i + 1)));
        b += splitmix64_dev(((m.y & 0xFFFFFFFFull) | (i << 32)) + 0xD1B54A32D192ED03ull * m.x);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a += shfl_xor64(a, m);
        b += shfl_xor64(b, m);
    }
    if ((threadIdx.x & 63) == 0) {
        s_a[threadIdx.x >> 6] = a;
        s_b[threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sums + 2 * (size_t)c, s_a[0] + s_a[1] + s_a[2] + s_a[3]);
        atomicAdd(sums + 2 * (size_t)c + 1, s_b[0] + s_b[1] + s_b[2] + s_b[3]);
    }
}
void launch_shmmr_checksum(hipStream_t st, const pgr_mm128 *mm, const uint64_t *off, uint32_t n_contigs, uint64_t max_cnt,
                           uint64_t *sums) {
    if (n_contigs == 0 || max_cnt == 0) return;
    hipLaunchKernelGGL(shmmr_checksum_kernel, dim3(n_contigs, (uint32_t)((max_cnt + 4095) / 4096)), dim3(256), 0, st, mm, off,
                       (unsigned long long *)sums);
}
__global__ void copy_add_rid_kernel(const pgr_mm128 *__restrict__ in, uint64_t n, uint32_t rid_add,
                                    pgr_mm128 *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pgr_mm128 m = in[i];
    m.y += (uint64_t)rid_add << 32;
    out[i] = m;
}
__global__ void copy_map_rid_kernel(const pgr_mm128 *__restrict__ in, uint64_t n, const uint32_t *__restrict__ rids,
                                    pgr_mm128 *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pgr_mm128 m = in[i];
    m.y = ((uint64_t)rids[(uint32_t)(m.y >> 32)] << 32) | (m.y & 0xFFFFFFFFull);
    out[i] = m;
}
void launch_copy_map_rid(hipStream_t st, const pgr_mm128 *in, uint64_t n, const uint32_t *rids, pgr_mm128 *out) {
    if (n == 0) return;
    hipLaunchKernelGGL(copy_map_rid_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, in, n, rids, out);
}
void launch_copy_add_rid(hipStream_t st, const pgr_mm128 *in, uint64_t n, uint32_t rid_add, pgr_mm128 *out) {
    if (n == 0) return;
    hipLaunchKernelGGL(copy_add_rid_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, in, n, rid_add, out);
}
void launch_patch_rid(hipStream_t st, pgr_mm128 *mm, const uint64_t *n_ptr, uint64_t cap, const uint32_t *rids,
                      uint32_t n_contigs) {
    if (cap == 0) return;
    hipLaunchKernelGGL(patch_rid_kernel, dim3((uint32_t)((cap + 255) / 256)), dim3(256), 0, st, mm, n_ptr, cap, rids, n_contigs);
}

}  // namespace pgr
