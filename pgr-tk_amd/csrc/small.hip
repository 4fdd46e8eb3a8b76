// small.hip -- the whole sequence_to_shmmrs pipeline of a SHORT contig in one workgroup, one launch for the batch.
//
// The reference's real callers feed small batches: load_index_from_reader hands over <= 129 contigs at a time
// (pgr-db/src/seq_db.rs:549-564), a query is one sequence of a few kbp (pgr-db/src/ext.rs:252-282, pgr-query.rs:135-165).
// The general path (level1.hip + level2.hip) is a chain of ~15 dependent device operations built for 10 Gbp batches: for a
// 10 kbp contig its latency is the launches, not the work.  Here one workgroup of 256 lanes owns one contig:
//   level 1   tile after tile with the same closed form as level1_tile_kernel (tile_select, level1_select.h), the selected
//             minimizers appended in position order to a list in LDS; the rescan-only tail (shmmrutils.rs:503-515 with
//             :516-520 false) by the first wavefront, as level1_tail_kernel does it
//   level 2   reduce_shmmr twice (shmmrutils.rs:359-415, 533-535; "an element survives iff it is an arg-min of some full
//             r-window": >= r consecutive neighbours including itself are >= it) and the min_span stencil (:536-555) on index
//             lists in LDS
//   output    final MM128s into the contig's slot, count per contig
// What it does not do is handed back to the general path by a flag (count = SMALL_FALLBACK): a palindromic k-mer (skipped
// push, shmmrutils.rs:477-480 -- needs the exact state machine), more level-1 minimizers than the LDS list holds
// (low-complexity sequence), a slot too small.  Non-ACGT bytes, sketch specs, padding and w < 17 never get here (host).
// Input and output pointers may be HBM or pinned host memory: the host entry points of small calls let the kernel read the
// host-packed planes and write the result over PCIe directly -- one launch and one synchronization per call.
#include "level1_select.h"
#include "pgr_small.h"

namespace pgr {

namespace {

// The level-1 list (hash key 8 B + pos/strand 4 B per minimizer) lives in DYNAMIC LDS sized by the longest contig of the batch
// (SmallArgs::l1_cap, at most SMALL_L1_CAP_MAX): a batch of 10 kbp queries needs 320 entries = 3.75 KB next to the 35 KB of the
// tile stage, so that four workgroups share a CU exactly as with level1_tile_kernel.
struct SmallLds {
    double suf[L1_G][L1_BLOCK];  // tile stage: window rows; later: scratch of the tail and the level-2 index lists
    double row[L1_BLOCK];
    uint2 words[L1_WORDS];
    uint32_t wsum[L1_BLOCK / 64];
    uint32_t base;
    int skip;
    uint32_t n1;        // level-1 minimizers so far
    uint32_t overflow;  // the list would not hold them
    uint32_t invalid;   // the contig holds a non-ACGT byte
};

// ordered compaction of the k in [0, n) with pred(k): out[j] = src ? src[k] : k.  Returns the number kept (uniform).
// Two barriers per 256 elements; L.wsum is the scratch.
template <class Pred>
__device__ __forceinline__ uint32_t compact_indices(SmallLds &L, uint32_t n, const uint16_t *src, uint16_t *out, Pred pred) {
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint64_t lt = (lane == 0) ? 0ull : (U64MAX >> (64 - lane));
    uint32_t total = 0;
    for (uint32_t b = 0; b < n; b += L1_BLOCK) {
        const uint32_t k = b + t;
        const bool keep = k < n && pred(k);
        const uint64_t bal = __ballot(keep);
        if (lane == 0) L.wsum[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int v = 0; v < L1_BLOCK / 64; ++v) {
            const uint32_t c = L.wsum[v];
            if (v < (int)wv) before += c;
            all += c;
        }
        if (keep) out[total + before + (uint32_t)__popcll(bal & lt)] = src ? src[k] : (uint16_t)k;
        total += all;
        __syncthreads();
    }
    return total;
}

// reduce_shmmr without padding on an index list: list[k] (or k itself) -> element of the level-1 arrays
template <int TR>
__device__ __forceinline__ bool reduce_keep(const uint64_t *key, const uint16_t *list, int n, int k, uint32_t r_rt) {
    const uint32_t r = TR ? (uint32_t)TR : r_rt;
    const uint64_t xi = key[list ? list[k] : k];
    uint32_t a = 0, b = 0;
    bool run_a = true, run_b = true;
#pragma unroll
    for (uint32_t d = 1; d < (TR ? (uint32_t)TR : 12u); ++d) {
        if (!TR && d >= r) break;
        {
            const int kk = k - (int)d;
            const bool inside = kk >= 0;
            const uint64_t xn = key[inside ? (list ? list[kk] : kk) : 0];
            run_a = run_a && inside && xn >= xi;
            a += run_a ? 1u : 0u;
        }
        {
            const int kk = k + (int)d;
            const bool inside = kk < n;
            const uint64_t xn = key[inside ? (list ? list[kk] : kk) : 0];
            run_b = run_b && inside && xn >= xi;
            b += run_b ? 1u : 0u;
        }
    }
    return a + b + 1 >= r;
}

}  // namespace

template <int TW, int TK>
__global__ __launch_bounds__(L1_BLOCK) void small_shmmr_kernel(SmallArgs a) {
    __shared__ SmallLds L;
    extern __shared__ uint64_t small_dyn[];
    const uint32_t t = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const SmallContig cd = a.desc[c];
    const uint32_t l1_cap = a.l1_cap;
    uint64_t *key = small_dyn;                                            // [l1_cap] 56-bit hash keys, position order
    uint32_t *ypos = reinterpret_cast<uint32_t *>(small_dyn + l1_cap);    // [l1_cap] pos << 1 | strand
    const uint32_t w = TW ? (uint32_t)TW : a.w, k = TK ? (uint32_t)TK : a.k;
    if (cd.len == 0) {
        if (t == 0) a.counts[c] = 0;
        return;
    }
    const ContigGeom g = contig_geom(cd.len, w, k);
    const uint2 *__restrict__ planes = a.planes + cd.word_off;
    const long long nwords = (g.L + 31) >> 5;
    const uint32_t nt = (uint32_t)((g.L + a.tc - 1) / a.tc);
    L1Args la;  // what tile_select reads of it: r (sketch threshold; unused here)
    la.r = a.r;
    if (t == 0) {
        L.n1 = 0;
        L.overflow = 0;
        L.skip = 0;
        L.invalid = 0;
    }
    if (a.valid) {  // resident batches: nobody has looked at the validity plane yet (the host entry points pack and check)
        __syncthreads();
        const uint32_t *__restrict__ v = a.valid + cd.word_off;
        bool bad = false;
        for (long long j = t; j < nwords; j += L1_BLOCK) {
            uint32_t want = 0xFFFFFFFFu;
            if (j == nwords - 1 && (g.L & 31)) want = ~(0xFFFFFFFFu >> (uint32_t)(g.L & 31));
            bad = bad || (v[j] & want) != want;
        }
        if (bad) L.invalid = 1;  // (benign race: all writers store 1)
        __syncthreads();
        if (L.invalid) {
            if (t == 0) {
                a.counts[c] = SMALL_FALLBACK;
                atomicOr(a.flags, 1u);
            }
            return;
        }
    }
    // ---------------------------------------------------------------- level 1: the tiles of the contig, in order
    for (uint32_t tile_local = 0; tile_local < nt; ++tile_local) {
        const long long c0 = (long long)tile_local * a.tc;
        long long c1 = c0 + a.tc;
        if (c1 > g.L) c1 = g.L;
        const long long e0 = c0 - (long long)(w - 1);
        const long long wbase = (e0 - 96) >> 5;
        if (t < L1_WORDS) {
            const long long wi = wbase + t;
            uint2 v = make_uint2(0u, 0u);
            if (wi >= 0 && wi < nwords) v = planes[wi];
            L.words[t] = v;
        }
        __syncthreads();
        const int t16 = (int)t * L1_G;
        const uint32_t core_mask = lane_range_mask(t16, clamp_rel<L1_EXT>(c0 - e0), clamp_rel<L1_EXT>(c1 - e0));
        const bool interior = e0 >= (long long)k && e0 + L1_EXT <= g.L && e0 >= g.jstart && e0 + L1_EXT - 1 <= g.jend;
        uint32_t valid_mask = 0xFFFFu, mwin_mask = 0xFFFFu;
        if (!interior) {
            valid_mask = lane_range_mask(t16, clamp_rel<L1_EXT>((long long)k - e0), clamp_rel<L1_EXT>(g.L - e0));
            mwin_mask = lane_range_mask(t16, clamp_rel<L1_EXT>(g.jstart - e0), clamp_rel<L1_EXT>(g.jend + 1 - e0));
        }
        double x[L1_G];
        uint32_t strand_bits = 0, emit = 0;
        const long long q = e0 + (long long)t16;
        // (a wavefront that lies entirely outside the contig still runs the masked variant here: it meets the same barriers,
        // and a short contig's few tiles are latency, not throughput)
        const bool wave_full = interior || __all(valid_mask == 0xFFFFu && mwin_mask == 0xFFFFu);
        if (wave_full)
            tile_select<TW, TK, false, false, L1_BLOCK>(la, w, k, t, q, wbase, L.words, L.suf, L.row, &L.skip, valid_mask, mwin_mask, core_mask, x,
                                             strand_bits, emit);
        else
            tile_select<TW, TK, false, true, L1_BLOCK>(la, w, k, t, q, wbase, L.words, L.suf, L.row, &L.skip, valid_mask, mwin_mask, core_mask, x,
                                            strand_bits, emit);
        // ordered append to the level-1 list
        const uint32_t cnt = __popc(emit);
        const uint32_t incl = wave_incl_sum(cnt);
        const uint32_t lane = t & 63, wv = t >> 6;
        if (lane == 63) L.wsum[wv] = incl;
        __syncthreads();
        uint32_t wave_base = 0, total = 0;
#pragma unroll
        for (int i = 0; i < L1_BLOCK / 64; ++i) {
            if ((uint32_t)i < wv) wave_base += L.wsum[i];
            total += L.wsum[i];
        }
        const uint32_t base = L.n1;  // (written behind the barrier at the end of the previous tile)
        const bool fits = base + total <= l1_cap;
        // the selected keys go through the lane's own column of the (now free) window rows, as in level1_tile_kernel: a loop
        // over the ~0.4 set bits per lane instead of 16 predicated stores; no barrier needed for a lane's own data
#pragma unroll
        for (int u = 0; u < L1_G; ++u) L.suf[u][t] = x[u];
        if (fits && cnt) {
            uint32_t o = base + wave_base + (incl - cnt);
            const uint32_t q32 = (uint32_t)q;
            uint32_t em = emit;
            while (em) {
                const uint32_t u = (uint32_t)__builtin_ctz(em);
                em &= em - 1;
                const uint64_t kb = (uint64_t)__double_as_longlong(L.suf[u][t]);
                key[o] = kb & 0x00FFFFFFFFFFFFFFull;
                ypos[o] = ((q32 + u) << 1) | ((strand_bits >> u) & 1u);
                ++o;
            }
        }
        __syncthreads();
        if (t == 0) {
            if (fits) L.n1 = base + total;
            else L.overflow = 1;
        }
        __syncthreads();
    }
    // ---------------------------------------------------------------- level 1: the rescan-only tail (first wavefront)
    const long long n_tail = (g.jend < g.jstart) ? 0 : (g.L - 1 - g.jend);
    if (n_tail > 0 && !L.overflow) {
        // scratch in the (now free) window rows: x[256] u64, strand[256] u32, emitted indices[256] u32
        uint64_t *s_x = reinterpret_cast<uint64_t *>(&L.suf[0][0]);
        uint32_t *s_st = reinterpret_cast<uint32_t *>(s_x + 256);
        uint32_t *s_emit = s_st + 256;
        const long long lo = g.jend - (long long)w + 1;
        const int n = (int)(g.L - lo);  // <= w + (w - k) <= 256
        for (int i = (int)t; i < n; i += L1_BLOCK) {
            uint64_t f0, f1;
            kmer_at(planes, nwords, lo + i, k, f0, f1);
            const uint64_t r0 = rc_plane(f0, k), r1 = rc_plane(f1, k);
            uint32_t st;
            uint64_t h;
            const uint64_t xv = kmer_x(f0, f1, r0, r1, k, st, h);
            const bool pal = (f0 == r0 && f1 == r1);
            if (pal) L.skip = 1;  // (benign race: all writers store 1) -> the contig goes to the general path
            s_x[i] = pal ? U64MAX : xv;
            s_st[i] = st;
        }
        __syncthreads();
        if (t < 64) {  // the tiny machine, wave-parallel exactly as in level1_tail_kernel (level1.hip)
            int n_emit = 0;
            {
                uint64_t xq[4];
#pragma unroll
                for (int q2 = 0; q2 < 4; ++q2) {
                    const int i = (int)t + 64 * q2;
                    xq[q2] = i < n ? s_x[i] : U64MAX;
                }
                const uint64_t lt = (t == 0) ? 0ull : (U64MAX >> (64 - t));
                auto window_min = [&](int lo2, int hi2) {
                    uint64_t v = U64MAX;
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        const int i = (int)t + 64 * q2;
                        if (i >= lo2 && i <= hi2) v = umin64(v, xq[q2]);
                    }
                    return wave_min64(v);
                };
                auto equal_to = [&](int lo2, int hi2, uint64_t m2, bool record) {
                    int last = lo2;
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        const int i = (int)t + 64 * q2;
                        const bool e = i >= lo2 && i <= hi2 && xq[q2] == m2;
                        const uint64_t bal = __ballot(e);
                        if (bal == 0) continue;
                        if (record) {
                            const int pos = n_emit + (int)__popcll(bal & lt);
                            if (e && pos < 256) s_emit[pos] = (uint32_t)i;
                            n_emit = n_emit + (int)__popcll(bal) > 256 ? 256 : n_emit + (int)__popcll(bal);
                        }
                        last = 64 * q2 + 63 - (int)__clzll((long long)bal);
                    }
                    return last;
                };
                const uint64_t mn = window_min(0, (int)w - 1);
                int mdist = (int)w - 1 - equal_to(0, (int)w - 1, mn, false);
                for (int j = (int)w; j < n; ++j) {
                    if (mdist == (int)w - 1) {
                        const int wl = j - (int)w + 1;
                        const uint64_t m2 = window_min(wl, j);
                        mdist = j - equal_to(wl, j, m2, true);
                    } else {
                        ++mdist;
                    }
                }
            }
            const uint32_t base = L.n1;
            const bool fits = base + (uint32_t)n_emit <= l1_cap;
            __builtin_amdgcn_wave_barrier();
            if (fits) {
                for (int i = (int)t; i < n_emit; i += 64) {
                    const uint32_t idx = s_emit[i];
                    key[base + i] = s_x[idx] >> 8;
                    ypos[base + i] = ((uint32_t)(lo + idx) << 1) | (s_st[idx] & 1u);
                }
            }
            if (t == 0) {
                if (fits) L.n1 = base + (uint32_t)n_emit;
                else L.overflow = 1;
            }
        }
    }
    __syncthreads();
    if (L.skip || L.overflow) {
        if (t == 0) {
            a.counts[c] = SMALL_FALLBACK;
            atomicOr(a.flags, 1u);
        }
        return;
    }
    // ---------------------------------------------------------------- level 2 on index lists (in the free window rows)
    const uint32_t n1 = L.n1;
    uint16_t *idx_a = reinterpret_cast<uint16_t *>(&L.suf[0][0]);
    uint16_t *idx_b = idx_a + l1_cap;
    uint16_t *idx_c = idx_b + l1_cap;
    __syncthreads();  // (the tail's scratch lives in the same memory)
    const uint16_t *fin = nullptr;  // nullptr: the identity list
    uint32_t n3 = n1;
    if (a.r > 1) {
        const uint32_t n2 = (a.r == 4) ? compact_indices(L, n1, nullptr, idx_a, [&](uint32_t kk) { return reduce_keep<4>(key, nullptr, (int)n1, (int)kk, 4); })
                                       : compact_indices(L, n1, nullptr, idx_a, [&](uint32_t kk) { return reduce_keep<0>(key, nullptr, (int)n1, (int)kk, a.r); });
        n3 = (a.r == 4) ? compact_indices(L, n2, idx_a, idx_b, [&](uint32_t kk) { return reduce_keep<4>(key, idx_a, (int)n2, (int)kk, 4); })
                        : compact_indices(L, n2, idx_a, idx_b, [&](uint32_t kk) { return reduce_keep<0>(key, idx_a, (int)n2, (int)kk, a.r); });
        fin = idx_b;
    }
    // min_span stencil on the unfiltered neighbours (shmmrutils.rs:536-555): first and last always stay
    const uint32_t ms = a.min_span;
    const uint32_t n4 = compact_indices(L, n3, fin, idx_c, [&](uint32_t i) {
        if (i == 0 || i + 1 == n3) return true;
        const uint32_t e = fin ? fin[i] : i, ep = fin ? fin[i - 1] : i - 1, en = fin ? fin[i + 1] : i + 1;
        const uint32_t p = ypos[e] >> 1, pp = ypos[ep] >> 1, pn = ypos[en] >> 1;
        const uint64_t xk = key[e];
        return (p - pp > ms) && (pn - p > ms) && key[ep] != xk && key[en] != xk;
    });
    if (n4 > cd.out_cap) {
        if (t == 0) {
            a.counts[c] = SMALL_FALLBACK;
            atomicOr(a.flags, 1u);
        }
        return;
    }
    pgr_mm128 *__restrict__ out = a.out + cd.out_off;
    for (uint32_t j = t; j < n4; j += L1_BLOCK) {
        const uint32_t e = idx_c[j];
        pgr_mm128 m;
        m.x = (key[e] << 8) | (uint64_t)k;
        m.y = ((uint64_t)cd.rid << 32) | ypos[e];
        out[j] = m;
    }
    if (t == 0) a.counts[c] = n4;
}

uint32_t small_l1_cap(uint32_t max_len, uint32_t w) {
    const double expect = 2.0 * (double)max_len / (double)(w + 1);  // level-1 density 2 / (w + 1)
    const uint32_t cap = (((uint32_t)(expect * 1.15) + 32 + 63) / 64) * 64;
    return cap > SMALL_L1_CAP_MAX ? SMALL_L1_CAP_MAX : cap;
}

void launch_small_shmmr(hipStream_t st, const SmallArgs &a) {
    if (a.n == 0) return;
    const size_t dyn = (size_t)a.l1_cap * 12;
    if (a.w == 80 && a.k == 56)
        hipLaunchKernelGGL((small_shmmr_kernel<80, 56>), dim3(a.n), dim3(L1_BLOCK), dyn, st, a);
    else if (a.w == 48 && a.k == 56)
        hipLaunchKernelGGL((small_shmmr_kernel<48, 56>), dim3(a.n), dim3(L1_BLOCK), dyn, st, a);
    else
        hipLaunchKernelGGL((small_shmmr_kernel<0, 0>), dim3(a.n), dim3(L1_BLOCK), dyn, st, a);
}

// ------------------------------------------------------------------ resident results: slots -> one contiguous list
// one wavefront per contig copies its slot behind the contigs in front of it (off = exclusive scan of the counts)
__global__ __launch_bounds__(256) void small_gather_kernel(const pgr_mm128 *__restrict__ slots, const SmallContig *__restrict__ desc,
                                                           const uint32_t *__restrict__ counts, const uint64_t *__restrict__ off, uint32_t n,
                                                           pgr_mm128 *__restrict__ dst, uint64_t dst_cap) {
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n) return;
    const uint32_t cnt = counts[c];
    if (cnt == SMALL_FALLBACK || off[c] + cnt > dst_cap) return;  // (the host sees the flag / the true total and repeats)
    const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(slots) + desc[c].out_off;
    ulonglong2 *d = reinterpret_cast<ulonglong2 *>(dst) + off[c];
    for (uint32_t i = threadIdx.x & 63; i < cnt; i += 64) d[i] = s[i];
}
// counts with the fallback marker read as 0 (input of the scan); counts[n] (the flag word's place) = 0 as the scan's sentinel
__global__ void small_counts_kernel(const uint32_t *__restrict__ counts, uint32_t n, uint32_t *__restrict__ clean) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n) return;
    clean[c] = (c == n || counts[c] == SMALL_FALLBACK) ? 0u : counts[c];
}
void launch_small_counts(hipStream_t st, const uint32_t *counts, uint32_t n, uint32_t *clean) {
    hipLaunchKernelGGL(small_counts_kernel, dim3((n + 1 + 255) / 256), dim3(256), 0, st, counts, n, clean);
}
void launch_small_gather(hipStream_t st, const pgr_mm128 *slots, const SmallContig *desc, const uint32_t *counts, const uint64_t *off,
                         uint32_t n, pgr_mm128 *dst, uint64_t dst_cap) {
    if (n == 0) return;
    hipLaunchKernelGGL(small_gather_kernel, dim3((n + 3) / 4), dim3(256), 0, st, slots, desc, counts, off, n, dst, dst_cap);
}

}  // namespace pgr
