// level2.hip -- list-level stages of sequence_to_shmmrs: ordered gather of the level-1 segments,
// the two hierarchical reductions (reduce_shmmr, pgr-db/src/shmmrutils.rs:359-415), the min_span
// stencil (:536-555) and the shimmer-pair records (pgr-db/src/seq_db.rs:381-400, 1205-1217).
//
// Every stage is an order-preserving segmented select over a concatenated per-contig list:
//   count kernel -> exclusive scan of per-block counts -> scatter kernel.
// The lists are ~2.5 % of the positions, so all of this is a few % of the level-1 kernel's time; the
// kernels are plain HBM streaming code (16 B per element, coalesced).
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

// ------------------------------------------------------------------ select predicates
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

// reduce_shmmr in closed form: element i survives iff it is a minimum (ties included) of some full
// window of r consecutive list elements; with `padding` the list is virtually extended by r-1
// {MAX,MAX} sentinels on both sides (shmmrutils.rs:367-380).
__device__ __forceinline__ bool reduce_keep(const pgr_mm128 *__restrict__ in, uint64_t i, uint64_t S, uint64_t E,
                                            uint32_t r, uint32_t padding) {
    const uint64_t xi = in[i].x;
    uint32_t a = 0, b = 0;
    for (uint32_t d = 1; d < r; ++d) {
        if (i < S + d) {
            if (padding) a = r - 1;
            break;
        }
        if (in[i - d].x >= xi) ++a;
        else break;
    }
    for (uint32_t d = 1; d < r; ++d) {
        if (i + d >= E) {
            if (padding) b = r - 1;
            break;
        }
        if (in[i + d].x >= xi) ++b;
        else break;
    }
    return a + b + 1 >= r;
}

// min_span stencil on the unfiltered neighbours (shmmrutils.rs:541-553)
__device__ __forceinline__ bool span_keep(const pgr_mm128 *__restrict__ in, uint64_t i, uint64_t S, uint64_t E,
                                          uint32_t min_span) {
    if (i == S || i + 1 == E) return true;
    const pgr_mm128 p = in[i - 1], m = in[i], n = in[i + 1];
    const uint32_t pp = (uint32_t)((p.y & 0xFFFFFFFFull) >> 1), mp = (uint32_t)((m.y & 0xFFFFFFFFull) >> 1),
                   np = (uint32_t)((n.y & 0xFFFFFFFFull) >> 1);
    return (uint32_t)(mp - pp) > min_span && (uint32_t)(np - mp) > min_span && p.x != m.x && m.x != n.x;
}

__device__ __forceinline__ bool sel_keep(const SelArgs &a, uint64_t i, uint32_t &cid, uint64_t &S) {
    cid = (uint32_t)(a.in[i].y >> 32);  // internal rid = contig index until the last stage
    S = a.off_in[cid];
    const uint64_t E = a.off_in[cid + 1];
    return a.mode == 0 ? reduce_keep(a.in, i, S, E, a.r, a.padding) : span_keep(a.in, i, S, E, a.min_span);
}

}  // namespace

__global__ __launch_bounds__(256) void select_count_kernel(SelArgs a, uint32_t *__restrict__ blk_cnt) {
    __shared__ uint32_t s_w[4];
    const uint64_t base = (uint64_t)blockIdx.x * SEL_BLOCK_ELEMS + threadIdx.x * 4;
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t i = base + j;
        if (i < a.n) {
            uint32_t cid;
            uint64_t S;
            cnt += sel_keep(a, i, cid, S) ? 1u : 0u;
        }
    }
    const uint32_t incl = wave_incl_sum(cnt);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void select_scatter_kernel(SelArgs a, const uint64_t *__restrict__ blk_base,
                                                             pgr_mm128 *__restrict__ out,
                                                             uint64_t *__restrict__ start_rank) {
    __shared__ uint32_t s_w[4];
    const uint64_t base = (uint64_t)blockIdx.x * SEL_BLOCK_ELEMS + threadIdx.x * 4;
    uint32_t flags = 0, firsts = 0;
    uint32_t cids[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t i = base + j;
        cids[j] = 0;
        if (i < a.n) {
            uint64_t S;
            if (sel_keep(a, i, cids[j], S)) flags |= 1u << j;
            if (i == S) firsts |= 1u << j;
        }
    }
    const uint32_t cnt = __popc(flags);
    const uint32_t incl = wave_incl_sum(cnt);
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) s_w[wv] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t i = 0; i < wv; ++i) wbase += s_w[i];
    uint64_t o = blk_base[blockIdx.x] + wbase + (incl - cnt);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (firsts & (1u << j)) start_rank[cids[j]] = o;
        if (flags & (1u << j)) {
            pgr_mm128 m = a.in[base + j];
            if (a.rids) m.y = ((uint64_t)a.rids[cids[j]] << 32) | (m.y & 0xFFFFFFFFull);
            out[o++] = m;
        }
    }
}

// offsets of the selected list: contigs with an empty input list inherit the rank of the next element
__global__ void fill_offsets_kernel(const uint64_t *__restrict__ off_in, const uint64_t *__restrict__ start_rank,
                                    uint32_t n, const uint64_t *__restrict__ d_total, uint64_t *__restrict__ off_out) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n) return;
    const uint64_t total = *d_total;
    if (c == n) {
        off_out[n] = total;
        return;
    }
    const uint64_t me = off_in[c];
    if (off_in[c + 1] > me) {
        off_out[c] = start_rank[c];
        return;
    }
    // smallest m in (c, n] with off_in[m] > me
    uint32_t lo = c, hi = n + 1;  // off_in[lo] <= me ; hi = n+1 means "none"
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off_in[mid] > me) hi = mid;
        else lo = mid;
    }
    off_out[c] = (hi == n + 1) ? total : start_rank[hi - 1];
}

void launch_select_count(hipStream_t st, const SelArgs &a, uint32_t *blk_cnt, uint32_t n_blocks) {
    if (n_blocks == 0) return;
    hipLaunchKernelGGL(select_count_kernel, dim3(n_blocks), dim3(256), 0, st, a, blk_cnt);
}
void launch_select_scatter(hipStream_t st, const SelArgs &a, const uint64_t *blk_base, uint32_t n_blocks,
                           pgr_mm128 *out, uint64_t *start_rank) {
    if (n_blocks == 0) return;
    hipLaunchKernelGGL(select_scatter_kernel, dim3(n_blocks), dim3(256), 0, st, a, blk_base, out, start_rank);
}
void launch_fill_offsets(hipStream_t st, const uint64_t *off_in, const uint64_t *start_rank, uint32_t n_contigs,
                         const uint64_t *d_total, uint64_t *off_out) {
    hipLaunchKernelGGL(fill_offsets_kernel, dim3((n_contigs + 1 + 255) / 256), dim3(256), 0, st, off_in, start_rank,
                       n_contigs, d_total, off_out);
}

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
