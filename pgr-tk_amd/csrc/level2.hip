// level2.hip -- list-level stages of sequence_to_shmmrs: ordered gather of the level-1 segments,
// the two hierarchical reductions (reduce_shmmr, pgr-db/src/shmmrutils.rs:359-415), the min_span
// stencil (:536-555) and the shimmer-pair records (pgr-db/src/seq_db.rs:381-400, 1205-1217).
//
// The level-1 list is ~2.5 % of the positions (16 B each).  It is read ONCE, straight from the unordered
// per-tile segments, by fused_select_kernel (reduce x2 + min_span in LDS); only the survivors (~12 %) are
// written, ordered by a scan + gather_segments_kernel.  Plain HBM streaming code, no MFMA.
#include "pgr_device.h"
#include "pgr_internal.h"

namespace pgr {

// ------------------------------------------------------------------ ordered gather
// one wavefront per segment
__global__ __launch_bounds__(256) void gather_segments_kernel(const pgr_mm128 *__restrict__ src,
                                                              const uint64_t *__restrict__ seg_off,
                                                              const uint32_t *__restrict__ seg_cnt,
                                                              const uint64_t *__restrict__ seg_dst, uint32_t n_segs,
                                                              pgr_mm128 *__restrict__ dst) {
    const uint32_t seg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= n_segs) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t cnt = seg_cnt[seg];
    const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(src) + seg_off[seg];
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(dst) + seg_dst[seg];
    for (uint32_t i = lane; i < cnt; i += 64) d[i] = s[i];
}

void launch_gather_segments(hipStream_t st, const pgr_mm128 *src, const uint64_t *seg_off, const uint32_t *seg_cnt,
                            const uint64_t *seg_dst, uint32_t n_segs, pgr_mm128 *dst) {
    if (n_segs == 0) return;
    hipLaunchKernelGGL(gather_segments_kernel, dim3((n_segs + 3) / 4), dim3(256), 0, st, src, seg_off, seg_cnt, seg_dst,
                       n_segs, dst);
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
                                                        pgr_frag_rec *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t c = find_seg(off, n, i);
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
                      uint32_t n_contigs, uint64_t n, const uint32_t *sids, int query_side, pgr_frag_rec *out) {
    if (n == 0) return;
    hipLaunchKernelGGL(frag_recs_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, mm, off, rec_off,
                       n_contigs, n, sids, query_side, out);
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
constexpr int FUSED_EMAX = FUSED_B + 2 * 288;  // r = 12

struct FusedLds {
    uint64_t x[FUSED_EMAX];
    uint64_t y[FUSED_EMAX];
    uint16_t s1[FUSED_EMAX];
    uint16_t s2[FUSED_EMAX];
    uint32_t wsum[FUSED_T / 64];
    unsigned long long base;
};

// ordered compaction helper: thread t owns items [t*C, (t+1)*C); returns exclusive rank of its first item,
// total through *total.  All threads must call.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t cnt, uint32_t *wsum, uint32_t *total) {
    const uint32_t incl = wave_incl_sum(cnt);
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();  // wsum reuse
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < FUSED_T / 64; ++i) {
        const uint32_t v = wsum[i];
        if (i < (int)wv) base += v;
        tot += v;
    }
    *total = tot;
    return base + incl - cnt;
}

// reduce predicate on an indexed LDS list: list[k] -> element index e; neighbours must share the contig id
__device__ __forceinline__ bool reduce_keep_lds(const FusedLds &L, const uint16_t *list, int n, int k, uint32_t r,
                                                uint32_t padding, bool lo_is_start, bool hi_is_end) {
    const int e = list ? list[k] : k;
    const uint64_t xi = L.x[e];
    const uint32_t cid = (uint32_t)(L.y[e] >> 32);
    uint32_t a = 0, b = 0;
    for (uint32_t d = 1; d < r; ++d) {
        const int kk = k - (int)d;
        bool boundary = kk < 0;
        int ee = 0;
        if (!boundary) {
            ee = list ? list[kk] : kk;
            boundary = (uint32_t)(L.y[ee] >> 32) != cid;
        }
        if (boundary) {
            // a true list / contig boundary: with padding the virtual sentinels are >= everything.
            // (kk < 0 with !lo_is_start is a loading edge: only reached by far-halo elements)
            if (padding && (kk >= 0 || lo_is_start)) a = r - 1;
            break;
        }
        if (L.x[ee] >= xi) ++a;
        else break;
    }
    for (uint32_t d = 1; d < r; ++d) {
        const int kk = k + (int)d;
        bool boundary = kk >= n;
        int ee = 0;
        if (!boundary) {
            ee = list ? list[kk] : kk;
            boundary = (uint32_t)(L.y[ee] >> 32) != cid;
        }
        if (boundary) {
            if (padding && (kk < n || hi_is_end)) b = r - 1;
            break;
        }
        if (L.x[ee] >= xi) ++b;
        else break;
    }
    return a + b + 1 >= r;
}

}  // namespace

// FusedArgsPub (pgr_internal.h): l1 = unordered level-1 segments; seg_dst = exclusive scan of seg_cnt
// ([n_segs+1]); out = cursor-allocated block segments; cursor[0] allocated, [1] overflow.
using FusedArgs = FusedArgsPub;

__global__ __launch_bounds__(FUSED_T) void fused_select_kernel(FusedArgs a) {
    __shared__ FusedLds L;
    const uint32_t t = threadIdx.x;
    const uint64_t core_lo = (uint64_t)blockIdx.x * FUSED_B;
    uint64_t core_hi = core_lo + FUSED_B;
    if (core_hi > a.total) core_hi = a.total;
    const uint64_t lo = core_lo >= a.halo ? core_lo - a.halo : 0;
    uint64_t hi = core_hi + a.halo;
    if (hi > a.total) hi = a.total;
    const int ne = (int)(hi - lo);
    const bool lo_is_start = (lo == 0), hi_is_end = (hi == a.total);

    // ---- stream the segments of [lo, hi) into LDS; the segment holding `lo` was located by
    // block_first_seg_kernel (a 22-step dependent binary search per workgroup here would dominate)
    {
        const uint32_t wv = t >> 6, lane = t & 63;
        const uint32_t first_seg = a.blk_first_seg[blockIdx.x];
        for (uint32_t s = first_seg + wv; s < a.n_segs; s += FUSED_T / 64) {
            const uint64_t d0 = a.seg_dst[s];
            if (d0 >= hi) break;
            const uint32_t cnt = a.seg_cnt[s];
            if (cnt == 0) continue;
            const uint64_t b0 = d0 > lo ? d0 : lo;                  // logical range of this segment inside [lo, hi)
            const uint64_t b1 = (d0 + cnt) < hi ? (d0 + cnt) : hi;
            const pgr_mm128 *src = a.l1 + a.seg_off[s] + (b0 - d0);
            const int dst = (int)(b0 - lo);
            for (int i = lane; i < (int)(b1 - b0); i += 64) {
                const pgr_mm128 m = src[i];
                L.x[dst + i] = m.x;
                L.y[dst + i] = m.y;
            }
        }
    }
    __syncthreads();

    // ---- reduce x2 (shmmrutils.rs:533-535) on index lists, then the min_span stencil (:536-555)
    const uint16_t *cur = nullptr;  // nullptr = identity list over [0, ne)
    int n_cur = ne;
    if (a.do_reduce) {
        for (int round = 0; round < 2; ++round) {
            uint16_t *dstl = round == 0 ? L.s1 : L.s2;
            const int C = (n_cur + FUSED_T - 1) / FUSED_T;
            const int k0 = (int)t * C;
            uint32_t flags = 0, cnt = 0;  // C <= 7
            for (int j = 0; j < C; ++j) {
                const int k = k0 + j;
                if (k < n_cur && reduce_keep_lds(L, cur, n_cur, k, a.r, a.padding, lo_is_start, hi_is_end)) {
                    flags |= 1u << j;
                    ++cnt;
                }
            }
            uint32_t tot;
            uint32_t o = block_excl_scan(cnt, L.wsum, &tot);
            for (int j = 0; j < C; ++j)
                if (flags & (1u << j)) dstl[o++] = (uint16_t)(cur ? cur[k0 + j] : (k0 + j));
            __syncthreads();
            cur = dstl;
            n_cur = (int)tot;
        }
    }
    // span filter over `cur`; survivors that are core elements are emitted in order
    const int c_lo = (int)(core_lo - lo), c_hi = (int)(core_hi - lo);
    uint32_t flags = 0, cnt = 0;
    const int C = (n_cur + FUSED_T - 1) / FUSED_T;
    const int k0 = (int)t * C;
    for (int j = 0; j < C; ++j) {
        const int k = k0 + j;
        if (k >= n_cur) break;
        const int e = cur ? cur[k] : k;
        if (e < c_lo || e >= c_hi) continue;
        const uint64_t ye = L.y[e];
        const uint32_t cid = (uint32_t)(ye >> 32);
        bool keep = true;
        const bool has_p = k > 0, has_n = k + 1 < n_cur;
        const int ep = has_p ? (cur ? cur[k - 1] : k - 1) : 0;
        const int en = has_n ? (cur ? cur[k + 1] : k + 1) : 0;
        const bool first = !has_p || (uint32_t)(L.y[ep] >> 32) != cid;
        const bool last = !has_n || (uint32_t)(L.y[en] >> 32) != cid;
        if (!first && !last) {
            const uint32_t pp = (uint32_t)((L.y[ep] & 0xFFFFFFFFull) >> 1), mp = (uint32_t)((ye & 0xFFFFFFFFull) >> 1),
                           np = (uint32_t)((L.y[en] & 0xFFFFFFFFull) >> 1);
            keep = (uint32_t)(mp - pp) > a.min_span && (uint32_t)(np - mp) > a.min_span && L.x[ep] != L.x[e] &&
                   L.x[e] != L.x[en];
        }
        if (keep) {
            flags |= 1u << j;
            ++cnt;
        }
    }
    uint32_t tot;
    const uint32_t o0 = block_excl_scan(cnt, L.wsum, &tot);
    if (t == 0) {
        // fixed slot per workgroup; only blocks with more survivors than the slot use the shared cursor
        // (same-address atomics saturate at ~88/us on gfx950)
        unsigned long long base = (unsigned long long)blockIdx.x * a.slot;
        bool ok = true;
        if (tot > a.slot) {
            const unsigned long long ob = atomicAdd(a.cursor, (unsigned long long)tot);
            base = a.ovf_base + ob;
            ok = ob + tot <= a.cap;
            if (!ok) atomicExch(a.cursor + 1, 1ull);
        }
        L.base = ok ? base : ~0ull;
        a.blk_off[blockIdx.x] = base;
        a.blk_cnt[blockIdx.x] = ok ? tot : 0u;
    }
    __syncthreads();
    const unsigned long long base = L.base;
    if (cnt && base != ~0ull) {
        uint64_t o = base + o0;
        for (int j = 0; j < C; ++j)
            if (flags & (1u << j)) {
                const int e = cur ? cur[k0 + j] : (k0 + j);
                pgr_mm128 m;
                m.x = L.x[e];
                m.y = L.y[e];
                a.out[o++] = m;
            }
    }
}

// first segment of every fused workgroup: largest s with seg_dst[s] <= max(0, b*FUSED_B - halo)
__global__ void block_first_seg_kernel(const uint64_t *__restrict__ seg_dst, uint32_t n_segs, uint32_t n_blocks,
                                       uint32_t halo, uint32_t *__restrict__ blk_first_seg) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint64_t core_lo = (uint64_t)b * FUSED_B;
    const uint64_t lo = core_lo >= halo ? core_lo - halo : 0;
    uint32_t s_lo = 0, s_hi = n_segs;
    while (s_hi - s_lo > 1) {
        const uint32_t mid = (s_lo + s_hi) >> 1;
        if (seg_dst[mid] <= lo) s_lo = mid;
        else s_hi = mid;
    }
    blk_first_seg[b] = s_lo;
}

// offsets of the final ordered list: off[c] = first element whose (internal) rid >= c
__global__ void offsets_by_rid_kernel(const pgr_mm128 *__restrict__ mm, uint64_t n, uint32_t n_contigs,
                                      uint64_t *__restrict__ off) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_contigs) return;
    uint64_t lo = 0, hi = n;  // first j with rid(j) >= c
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint32_t)(mm[mid].y >> 32) < c) lo = mid + 1;
        else hi = mid;
    }
    off[c] = lo;
}

__global__ void patch_rid_kernel(pgr_mm128 *__restrict__ mm, uint64_t n, const uint32_t *__restrict__ rids) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t y = mm[i].y;
    mm[i].y = ((uint64_t)rids[(uint32_t)(y >> 32)] << 32) | (y & 0xFFFFFFFFull);
}

void launch_fused_select_pub(hipStream_t st, const FusedArgsPub &a, uint32_t n_blocks) {
    if (n_blocks == 0) return;
    hipLaunchKernelGGL(block_first_seg_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, a.seg_dst, a.n_segs,
                       n_blocks, a.halo, a.blk_first_seg);
    hipLaunchKernelGGL(fused_select_kernel, dim3(n_blocks), dim3(FUSED_T), 0, st, a);
}
void launch_offsets_by_rid(hipStream_t st, const pgr_mm128 *mm, uint64_t n, uint32_t n_contigs, uint64_t *off) {
    hipLaunchKernelGGL(offsets_by_rid_kernel, dim3((n_contigs + 1 + 255) / 256), dim3(256), 0, st, mm, n, n_contigs, off);
}
void launch_patch_rid(hipStream_t st, pgr_mm128 *mm, uint64_t n, const uint32_t *rids) {
    if (n == 0) return;
    hipLaunchKernelGGL(patch_rid_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, mm, n, rids);
}

}  // namespace pgr
