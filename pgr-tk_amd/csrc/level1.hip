// level1.hip -- gfx950 kernels for the level-1 SHIMMER selection (k-mer hash + windowed minimizers).
//
// Replaces the per-base loop of shmmrutils::sequence_to_shmmrs1 (pgr-db/src/shmmrutils.rs:454-530)
// and the sketch loop of sequence_to_shmmrs2 (:580-630).
//
// Kernels, in launch order:
//   tile_desc_kernel           one thread per tile: its descriptor; clears the call's cursors and flags.
//   mark_invalid_tiles_kernel  flags tiles with a non-ACGT byte in reach (in the descriptor: the tile kernel leaves them to the
//                              islands) and records every tile's last valid position.
//   level1_tile_kernel<W,K,SKETCH,BLK>  position-parallel closed form: one workgroup of BLK = 256 lanes per
//                              tile of 4096 positions (core + 2*(w-1) halo) -- or BLK = 64, one wavefront per tile of 1024, for
//                              batches of short contigs --; every lane owns 16 consecutive positions in registers.  The last
//                              tile of a contig also runs the contig's tail (tail_of_contig): the last (w-k) positions, where
//                              the reference only rescans (branch 2 disabled, shmmrutils.rs:516-520).
//   level1_tail_kernel         the same tails, one wavefront per contig, for specs without a tile path (w < 17).
//   level1_chunk_kernel        the exact ring-buffer state machine, event driven, one wavefront per chunk (1-32 kbp) with verified
//                              seams.  Runs on islands around what the closed form does not cover: non-ACGT bytes,
//                              reverse-complement-palindromic k-mers (skipped pushes, shmmrutils.rs:477-480); whole contigs for w < 17.
//   assemble_chunks_kernel, set_segs_kernel   the lists of the chunks that start in one tile become that tile's segment.
//   splice_segs_kernel         a tile an island begins or ends INSIDE (round 6): its own elements in front of the island's begin / from
//                              its end on, spliced with the chunks' lists into one contiguous segment.
//
// Integer / byte work only: no MFMA.  The tile kernel is VALU bound (two 64-bit mix hashes per position).
#include "pgr_device.h"
#include "pgr_internal.h"

#include "level1_select.h"

namespace pgr {

// one wavefront per contig: positions after jend, rescans only (shmmrutils.rs:503-515 with :516-520 false)
__device__ __forceinline__ void tail_wave_sync() {  // orders the LDS accesses of ONE wavefront (its part of the workgroup's LDS is its own)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The tail of contig c on ONE wavefront (s_x[256], s_st[256], s_emit[256], s_base: LDS of that wavefront alone); sidx: the
// contig's tail segment; planes: the contig's plane words; lds_words: the words from lds_wbase on that the caller holds in
// LDS (nullptr: none); tk: the tile's own keys (nullptr: none).  Called by the tile kernel for the contig's last tile -- the
// tail's w + (w - k) positions are positions of that tile unless it is a sliver of the contig (then its staged words still cover
// the tail's k-mers unless w > 96) --, and by level1_tail_kernel for specs without a tile path.
struct TileKeys {           // what the tile kernel has left in LDS: the keys of its extended positions (position e0 + 16 t + u at
    const double *keys;     // keys[u * blk + t], 56 bits) and the strand bits of lane t's 16 positions (strands[2 t]).  May overlap
    const uint32_t *strands;  // s_x / s_st / s_emit: everything is read before anything is written.
    int blk;
    long long e0;
};
__device__ __forceinline__ void tail_of_contig(const L1Args &a, uint32_t c, uint32_t len, const uint2 *__restrict__ planes,
                                               const uint2 *lds_words, long long lds_wbase, const TileKeys *tk, uint32_t lane,
                                               uint32_t sidx, uint64_t *s_x, uint32_t *s_st, uint32_t *s_emit,
                                               unsigned long long *s_base_p) {
    unsigned long long &s_base = *s_base_p;
    const uint32_t w = a.w, k = a.k;
    const ContigGeom g = contig_geom(len, w, k);
    const long long n_tail = (a.sketch || g.jend < g.jstart) ? 0 : (g.L - 1 - g.jend);
    if (n_tail <= 0) {
        if (lane == 0) {
            a.seg_off[sidx] = 0;
            a.seg_cnt[sidx] = 0;
            a.seg_cid[sidx] = c;
        }
        return;
    }
    const long long lo = g.jend - (long long)w + 1;  // first position of the window ending at jend (>= k)
    const int n = (int)(g.L - lo);                   // <= w + (w-k) <= 256
    const long long nwords = (g.L + 31) >> 5;
    const bool from_tile = tk != nullptr && lo >= tk->e0;                                          // (uniform)
    const bool from_lds = !from_tile && lds_words != nullptr && (lo >> 5) - 2 >= lds_wbase;  // (uniform)
    uint64_t xq[4];
    {
        uint32_t sq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            xq[q] = U64MAX;
            sq[q] = 0;
            if (64 * q >= n) continue;  // (uniform)
            const int i = (int)lane + 64 * q;
            if (i >= n) continue;
            const long long p = lo + i;
            if (from_tile) {  // hashed by the tile already.  (A palindromic k-mer there has flagged the tile: an island replaces this tail)
                const int rel = (int)(p - tk->e0);
                const uint64_t kb = (uint64_t)__double_as_longlong(tk->keys[(rel & 15) * tk->blk + (rel >> 4)]);
                xq[q] = ((kb & 0x00FFFFFFFFFFFFFFull) << 8) | (uint64_t)k;
                sq[q] = (tk->strands[2 * (rel >> 4)] >> (rel & 15)) & 1u;
                continue;
            }
            uint64_t f0, f1;
            if (from_lds) {  // kmer_at on the staged words (words outside the contig are staged as zeros)
                const int j = (int)((p >> 5) - lds_wbase);
                const uint32_t sh = 31u - (uint32_t)(p & 31);
                const uint2 w0 = lds_words[j], w1 = lds_words[j - 1], w2 = lds_words[j - 2];
                const uint64_t kmask = U64MAX >> (64 - k);
                f0 = (((uint64_t)funnel(w2.x, w1.x, sh) << 32) | funnel(w1.x, w0.x, sh)) & kmask;
                f1 = (((uint64_t)funnel(w2.y, w1.y, sh) << 32) | funnel(w1.y, w0.y, sh)) & kmask;
            } else {
                kmer_at(planes, nwords, p, k, f0, f1);
            }
            const uint64_t r0 = rc_plane(f0, k), r1 = rc_plane(f1, k);
            uint64_t h;
            const uint64_t xv = kmer_x(f0, f1, r0, r1, k, sq[q], h);
            // a palindromic k-mer here means the contig is re-done by the serial kernel anyway
            xq[q] = (f0 == r0 && f1 == r1) ? U64MAX : xv;
        }
        tail_wave_sync();  // (the tile's keys have been read: their rows become s_x / s_st / s_emit)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = (int)lane + 64 * q;
            if (i < n) {
                s_x[i] = xq[q];
                s_st[i] = sq[q];
            }
        }
    }
    tail_wave_sync();
    // The tiny machine, wave-parallel: every lane keeps its (at most 4) elements i = lane + 64 q in registers, the state
    // (mdist, n_emit) is wave-uniform, a rescan is one min-reduction + one ballot per slice instead of two serial walks over the
    // window in LDS (~20 us of dependent LDS latency per contig -- 10 000 queries or 10^6 reads feel that).
    int n_emit = 0;
    {
        const uint64_t lt = (lane == 0) ? 0ull : (U64MAX >> (64 - lane));
        auto window_min = [&](int lo2, int hi2) {
            uint64_t v = U64MAX;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = (int)lane + 64 * q;
                if (i >= lo2 && i <= hi2) v = umin64(v, xq[q]);
            }
            return wave_min64(v);
        };
        // elements of [lo2, hi2] equal to m2, in index order: recorded when `record`; returns the index of the last one
        auto equal_to = [&](int lo2, int hi2, uint64_t m2, bool record) {
            int last = lo2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = (int)lane + 64 * q;
                const bool e = i >= lo2 && i <= hi2 && xq[q] == m2;
                const uint64_t bal = __ballot(e);
                if (bal == 0) continue;
                if (record) {
                    const int pos = n_emit + (int)__popcll(bal & lt);
                    if (e && pos < 256) s_emit[pos] = (uint32_t)i;
                    n_emit = n_emit + (int)__popcll(bal) > 256 ? 256 : n_emit + (int)__popcll(bal);
                }
                last = 64 * q + 63 - (int)__clzll((long long)bal);
            }
            return last;
        };
        // right-most arg-min of the window ending at jend
        const uint64_t mn = window_min(0, (int)w - 1);
        const int mi = equal_to(0, (int)w - 1, mn, false);
        int mdist = (int)w - 1 - mi;
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
    if (lane == 0) {
        // a tail emits 0-3 minimizers: they go to the contig's own slot; the shared cursor (a same-address atomic WITH return:
        // ~280 per us on this chip, 3.6 ms for 10^6 reads) only when there are more
        unsigned long long ob = 0, at = a.tail_base + (unsigned long long)c * L1_TAIL_SLOT;
        bool ok = true;
        if (n_emit > (int)L1_TAIL_SLOT) {
            ob = atomicAdd(a.cursor, (unsigned long long)n_emit);
            ok = ob + n_emit <= a.cap;
            at = a.ovf_base + ob;
            if (!ok) atomicExch(a.cursor + 1, 1ull);
        }
        s_base = ok ? at : ~0ull;
        a.seg_off[sidx] = at;
        a.seg_cnt[sidx] = ok ? (uint32_t)n_emit : 0u;
        a.seg_cid[sidx] = c;
    }
    tail_wave_sync();
    const unsigned long long base = s_base;
    if (base != ~0ull) {
        for (int i = lane; i < n_emit; i += 64) {
            const uint32_t idx = s_emit[i];
            a.out[base + i] = l1rec_from_xy(s_x[idx], ((uint64_t)(lo + idx) << 1) | (s_st[idx] & 1u));
        }
    }
}


// TW / TK: compile-time window and k-mer size (0 = take them from the arguments).  The common specs are
// instantiated with constants so that every row offset, shift and mask is an immediate.
template <int TW, int TK, bool SKETCH, int BLK>
__global__ __launch_bounds__(BLK) PGR_TILE_ATTR void level1_tile_kernel(L1Args a) {
    constexpr int EXT = BLK * L1_G;                 // positions per extended tile
    constexpr int WORDS = (EXT + 96) / 32 + 5;      // plane words staged per tile (tile + k-mer look-back)
    static_assert(WORDS <= BLK, "one lane per staged word");
    __shared__ double s_suf[L1_G][BLK];  // suffix-min per row, later prefix-max
    __shared__ double s_row[BLK];        // row min, later row max
    __shared__ uint2 s_words[WORDS];
    __shared__ uint32_t s_wsum[BLK / 64];
    __shared__ unsigned long long s_base;
    __shared__ int s_skip;
    __shared__ uint32_t s_pal[2];  // first / last extended position of the tile with a palindromic k-mer (when s_skip)
#ifdef PGR_LDS_PAD
    __shared__ uint32_t s_pad[PGR_LDS_PAD / 4];  // occupancy experiment only
    if (a.w == 0xdead) s_pad[threadIdx.x] = 1;
#endif

    const uint32_t t = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    const TileDesc td = a.desc[tile];
    const uint32_t c = td.contig;
    // A tile with a non-ACGT byte in reach (mark_invalid_tiles_kernel, which runs in front of this kernel, says so in the tile's
    // descriptor) is replaced by an island of the exact machine or, deep inside a gap, emptied (api.hip: run_islands): nothing of
    // it is computed here.  (The 40 Mbp of gaps of a chromosome-like contig were 10 000 tiles of "poly-A": every position a tie,
    // 3904 records each.)
    if (td.skip) {  // (uniform)
        asm volatile("s_nop 15");  // (cold for tools/isa_histogram.py)
        if (t == 0) {
            asm volatile("s_nop 15");
            const uint32_t sidx = tile + c;
            a.seg_off[sidx] = 0;
            a.seg_cnt[sidx] = 0;
            a.seg_cid[sidx] = c;
            if (l1_core_end((long long)td.tile_local * a.tc, (long long)td.len, a.tc, (uint32_t)EXT) == (long long)td.len) {
                asm volatile("s_nop 15");
                a.seg_off[sidx + 1] = 0;  // the contig's tail segment: the island that replaces this tile reaches the contig's end
                a.seg_cnt[sidx + 1] = 0;
                a.seg_cid[sidx + 1] = c;
            }
        }
        return;
    }
    const uint32_t tile_local = td.tile_local;
    const uint32_t w = TW ? (uint32_t)TW : a.w, k = TK ? (uint32_t)TK : a.k;
    const ContigGeom g = contig_geom(td.len, w, k);
    const long long c0 = (long long)tile_local * a.tc;
    const long long c1 = l1_core_end(c0, g.L, a.tc, (uint32_t)EXT);
    // first extended position: the core's look-back, except in front of a contig's first tile (pgr_internal.h: a contig of up
    // to EXT positions is one tile whatever its core -- a 1 kbp read is one wavefront's 1024 positions)
    const long long e0 = tile_local ? c0 - (long long)(w - 1) : 0;

    // ---- stage the 2-bit planes of the tile (+ k-mer look-back) in LDS
    const long long wbase = (e0 - 96) >> 5;  // floor
    const long long nwords = (g.L + 31) >> 5;
    const uint2 *__restrict__ planes = a.b.planes + td.word_off;
    if (t < WORDS) {
        const long long wi = wbase + t;
        uint2 v = make_uint2(0u, 0u);
        if (wi >= 0 && wi < nwords) v = planes[wi];
        s_words[t] = v;
    }
    if (t == 0) {
        s_skip = 0;
        s_pal[0] = 0xFFFFFFFFu;
        s_pal[1] = 0u;
    }
    __syncthreads();

    // ---- per-lane masks over this lane's 16 positions (tile-extended coordinates 16t .. 16t+15)
    const int t16 = (int)t * L1_G;
    const uint32_t core_mask = lane_range_mask(t16, clamp_rel<EXT>(c0 - e0), clamp_rel<EXT>(c1 - e0));
    // interior tile (uniform): all 4096 extended positions are real k-mers and every window end is in range
    const bool interior = e0 >= (long long)k && e0 + EXT <= g.L && e0 >= g.jstart && e0 + EXT - 1 <= g.jend;
    uint32_t valid_mask = 0xFFFFu, mwin_mask = 0xFFFFu;
    if (!interior) {
        valid_mask = lane_range_mask(t16, clamp_rel<EXT>((long long)k - e0), clamp_rel<EXT>(g.L - e0));
        mwin_mask = lane_range_mask(t16, clamp_rel<EXT>(g.jstart - e0), clamp_rel<EXT>(g.jend + 1 - e0));
    }

    // ---- hash + select.  Waves whose 64x16 positions are all inside the contig and inside the window-end
    // range skip every masking instruction (uniform branch; both variants hit the same barriers).
    double x[L1_G];
    uint32_t strand_bits = 0, emit = 0;
    const long long q = e0 + (long long)t16;
    // A wavefront whose 64 x 16 positions all lie outside the contig (the last tile of a contig: a 10 kbp query fills 2.5
    // tiles, a 1 kbp read a quarter of one) computes nothing: it leaves in the shared rows what the masked variant would --
    // the "not a k-mer" sentinel for pass 1, "no window" for pass 2 --, meets the barriers of the live path (three in
    // tile_select unless SKETCH, two in the compaction below) and is done.  Never wavefront 0: thread 0 writes the tile's
    // segment record even when the contig is shorter than k and no position of the tile is a k-mer.  A separate exit: the live path's registers are
    // not shared with it.
    if (BLK > 64 && !interior && t >= 64 && __all(valid_mask == 0u && mwin_mask == 0u)) {
        if (!SKETCH) {
            const double big = mk_double(0u, KEY_INF);
#pragma unroll
            for (int u = 0; u < L1_G; ++u) s_suf[u][t] = big;
            s_row[t] = big;
            __syncthreads();
            __syncthreads();
            const double none = __longlong_as_double((long long)NO_WINDOW);
#pragma unroll
            for (int u = 0; u < L1_G; ++u) s_suf[u][t] = none;
            s_row[t] = none;
            __syncthreads();
        }
        if ((t & 63) == 63) s_wsum[t >> 6] = 0;
        __syncthreads();
        __syncthreads();
        if (c1 == g.L) __syncthreads();  // (the barrier in front of the contig's tail, below)
        return;
    }
    const bool wave_full = interior || __all(valid_mask == 0xFFFFu && mwin_mask == 0xFFFFu);
    if (wave_full)
        tile_select<TW, TK, SKETCH, false, BLK>(a, w, k, t, q, wbase, s_words, s_suf, s_row, &s_skip, valid_mask, mwin_mask,
                                          core_mask, x, strand_bits, emit, s_pal);
    else
        tile_select<TW, TK, SKETCH, true, BLK>(a, w, k, t, q, wbase, s_words, s_suf, s_row, &s_skip, valid_mask, mwin_mask,
                                         core_mask, x, strand_bits, emit, s_pal);

    // ---- ordered compaction: block scan of per-lane counts, one cursor bump per tile
    const uint32_t cnt = __popc(emit);
    const uint32_t incl = wave_incl_sum(cnt);
    const uint32_t lane = t & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));  // wave-uniform: scalar compares below
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
#pragma unroll
    for (int i = 0; i < BLK / 64 - 1; ++i)
        if ((uint32_t)i < wv) wave_base += s_wsum[i];
    if (t == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int i = 0; i < BLK / 64; ++i) total += s_wsum[i];
        // Every tile owns a fixed slot of a.slot elements (no atomics: one shared cursor saturates at ~88
        // same-address atomics/us, which would cap the kernel at ~29 ms for 2.5 M tiles).  Only tiles denser
        // than the slot (low-complexity sequence: ties emit every position) allocate from the overflow cursor.
        unsigned long long base = (unsigned long long)tile * a.slot;
        bool ok = true;
        if (total > a.slot) {
            const unsigned long long ob = atomicAdd(a.cursor, (unsigned long long)total);
            base = a.ovf_base + ob;
            ok = ob + total <= a.cap;
            if (!ok) atomicExch(a.cursor + 1, 1ull);
        }
        s_base = ok ? base : ~0ull;
        const uint32_t sidx = tile + c;  // one tail segment per preceding contig
        a.seg_off[sidx] = base;
        a.seg_cnt[sidx] = ok ? total : 0u;
        a.seg_cid[sidx] = c;
        if (s_skip) {
            atomicOr(a.contig_flags + c, 1u);
            a.tile_flags[tile] |= 1;
            atomicOr(a.cursor + 2, 1ull);  // batch-wide "some tile needs the exact path" word (read once by the host)
            if (a.tile_pal) {
                // where in the tile's core, in blocks of 64 positions (in front of the core: 0; behind it: the last block)
                const long long lo = (e0 + (long long)s_pal[0] - c0) >> 6, hi = (e0 + (long long)s_pal[1] - c0) >> 6;
                const uint32_t lo_b = (uint32_t)(lo < 0 ? 0 : lo > 63 ? 63 : lo), hi_b = (uint32_t)(hi < 0 ? 0 : hi > 63 ? 63 : hi);
                a.tile_pal[tile] = (uint16_t)(lo_b | (hi_b << 8));
            }
        }
    }
    // selected keys go through LDS (s_suf is free now; each lane re-reads only its own column, so no barrier
    // is needed for the data): a loop over the ~0.4 set bits per lane instead of 16 predicated stores
#pragma unroll
    for (int u = 0; u < L1_G; ++u) s_suf[u][t] = x[u];
    __syncthreads();
    {
        const unsigned long long base = s_base;
        if (cnt && base != ~0ull) {
            L1Rec *__restrict__ o = a.out + (base + wave_base + (incl - cnt));
            const uint32_t q32 = (uint32_t)q;  // core positions are >= 0 and < 2^31
            uint32_t em = emit;
            while (em) {
                const uint32_t u = (uint32_t)__builtin_ctz(em);
                em &= em - 1;
                const uint64_t kb = (uint64_t)__double_as_longlong(s_suf[u][t]);
                L1Rec m;
                m.key_lo = (uint32_t)kb;
                m.key_hi = (uint32_t)(kb >> 32) & 0x00FFFFFFu;  // drops bit 62, keeps the 56 hash bits
                m.ypos = ((q32 + u) << 1) | ((strand_bits >> u) & 1u);
                *o++ = m;
            }
        }
    }
    // ---- the contig's last tile also runs its tail (the positions behind jend: rescans only) on its first wavefront, in the
    // rows the output has just been read from.  (A kernel of its own -- one latency-bound wavefront per contig -- was 0.87 ms
    // for 10^6 reads, a third of this kernel's time there; here its loads hide behind the other workgroups' arithmetic.)
    if (c1 == g.L) {  // (uniform)
        asm volatile("s_nop 14");  // (marks everything from here on as cold for tools/isa_histogram.py: one tile in 2562 of a 10 Mbp contig)
        ((uint32_t *)s_row)[2 * t] = strand_bits;  // (s_row is free since the max pass)
        if (BLK > 64) __syncthreads();  // the other wavefronts have read their columns
        else tail_wave_sync();
        if (t < 64) {
            double *flat = &s_suf[0][0];
            const TileKeys tk{flat, (const uint32_t *)s_row, BLK, e0};
            tail_of_contig(a, c, td.len, planes, s_words, wbase, &tk, t, tile + 1 + c, (uint64_t *)flat, (uint32_t *)(flat + 256),
                           (uint32_t *)(flat + 384), &s_base);
        }
    }
}

// TAIL_WAVES contigs per workgroup, one wavefront each (nothing is shared between them: the barriers are wave barriers).  With
// one single-wave workgroup per contig the chip held ~7 of these latency-bound wavefronts per CU (counters: 90 % of their
// cycles waiting, 1.8 wavefronts per SIMD): 10^6 reads spent 3.6 ms here.
constexpr int TAIL_WAVES = 4;
__global__ __launch_bounds__(64 * TAIL_WAVES) void level1_tail_kernel(L1Args a) {
    __shared__ uint64_t s_x_all[TAIL_WAVES][256];
    __shared__ uint32_t s_st_all[TAIL_WAVES][256];
    __shared__ uint32_t s_emit_all[TAIL_WAVES][256];
    __shared__ unsigned long long s_base_all[TAIL_WAVES];
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x * TAIL_WAVES + wv;
    if (c >= a.n_contigs) return;
    const uint32_t lane = threadIdx.x & 63;
    if (c == 0 && lane == 0) a.seg_cnt[a.n_tiles + a.n_contigs] = 0;  // sentinel of the scan over the segment counts
    tail_of_contig(a, c, a.b.len[c], a.b.planes + a.b.word_off[c], nullptr, 0, nullptr, lane, a.tile_first[c + 1] + c, s_x_all[wv], s_st_all[wv],
                   s_emit_all[wv], &s_base_all[wv]);
}

// ------------------------------------------------------------------------------------------------
// Exact state machine (shmmrutils.rs:438-530), event driven, one wavefront per CHUNK of a contig:
// positions are hashed 64 at a time, the machine jumps from event to event (rescan R when
// mdist == w-1, or branch-2 emission B when x <= min_mer.x); between events mdist just counts pushes.
//
// Chunking: a chunk [cs, ce) that does not start at 0 first rebuilds the rolling
// k-mer (looking back until k valid bases have been seen) and then runs the machine from an empty ring
// over a warm-up of `warm` positions WITHOUT emitting.  After w pushes the warmed-up state equals the
// true state whenever the true machine is in its regular regime (min_mer = right-most minimum of the last
// w pushes); the state at cs is recorded next to the previous chunk's state at its ce so that the host can
// verify every seam and re-run the rare chunk whose assumption failed with the true state (`override`).
// Emissions are attributed by the step (position) at which the reference emits them.
// x mod w for ring-buffer slots: x < w + 128 always (a slot < w plus a lane or push rank), so for w >= 64 it is at most two
// conditional subtractions -- an integer division by a run-time w is ~30 dependent instructions, and the exact machine is one
// latency-bound wavefront that does several per event
__device__ __forceinline__ uint32_t ring_mod(uint32_t x, uint32_t w) {
    if (w >= 64u) {  // (uniform)
        x -= x >= w ? w : 0u;
        x -= x >= w ? w : 0u;
        return x;
    }
    return x % w;
}
__device__ __forceinline__ uint64_t ring_signature(const uint64_t *s_rx, uint32_t rstart, uint32_t rlen, uint32_t w,
                                                   uint32_t lane) {
    uint64_t sig = 0;
    for (uint32_t q = lane; q < w; q += 64) {
        const uint64_t v = s_rx[ring_mod(rstart + q, w)];
        sig ^= (v + 0x9E3779B97F4A7C15ull * (q + 1)) * (2ull * q + 1);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sig ^= shfl_xor64(sig, m);
    return sig + rlen;
}

__global__ __launch_bounds__(64) void level1_chunk_kernel(L1Args a, const ChunkDesc *__restrict__ descs,
                                                          ChunkState *__restrict__ st_in,
                                                          ChunkState *__restrict__ st_out,
                                                          uint32_t *__restrict__ status, uint64_t *__restrict__ rings,
                                                          uint64_t *__restrict__ info) {
    __shared__ uint64_t s_rx[128], s_ry[128];  // ring buffer (storage order)
    const uint64_t t_begin = wall_clock64();
    const uint32_t lane = threadIdx.x;
    const ChunkDesc cd = descs[blockIdx.x];
    if (lane == 0) {  // (a chunk that never reaches its seam reports an empty state; descs / st_in / st_out / status / info: pinned host memory)
        ChunkState z;
        memset(&z, 0, sizeof(z));
        st_in[blockIdx.x] = z;
    }
    const uint32_t c = cd.contig;
    const uint32_t w = a.w, k = a.k;
    const long long L = a.b.len[c];
    const uint2 *__restrict__ planes = a.b.planes + a.b.word_off[c];
    const uint32_t *__restrict__ vplane = a.b.valid + a.b.word_off[c];
    const long long nwords = (L + 31) >> 5;
    L1Rec *__restrict__ out = a.out + cd.region_off;
    const uint64_t cap = cd.region_cap;
    const uint64_t kmask = U64MAX >> (64 - k);
    const uint32_t shift = k - 1;
    const uint64_t sketch_thr = (U64MAX >> 4) >> a.r;
    // branch 2 enabled for w+k <= pos < Lb; Rust usize arithmetic wraps in release builds
    const uint64_t Lb = (uint64_t)L - (uint64_t)w + (uint64_t)k;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (U64MAX >> (64 - lane));
    const long long cs = (long long)cd.cs;
    long long ce = (long long)cd.ce;  // (moves on behind a stuck machine: cd.ext_limit)

    s_rx[lane] = U64MAX;
    s_rx[lane + 64] = U64MAX;
    s_ry[lane] = U64MAX;
    s_ry[lane + 64] = U64MAX;
    __syncthreads();

    // ---- where to start: pm = first position the machine sees, pk = first position the k-mer roll sees
    long long pm = cs - (long long)cd.warm;
    if (pm < 0) pm = 0;
    long long pk = pm;
    // A chunk that runs again from the true state with the ring of the chunk in front (ChunkDesc::ring_in) and NO warm-up takes the
    // rolling k-mer from that state as well: nothing in front of cs is looked at (a re-run of 1024 positions was 33 steps of 64 with
    // its 1024 warm-up positions, 16 now)
    const bool install_kmer = cd.override_state && cd.ring_in != 0xFFFFFFFFu && !a.sketch && cd.warm == 0 && cs > 0;
    // last valid position below x (x > 0; -1: none; -2: no table).  The scanned entry of the tile that holds x - 1 is the last
    // valid position up to the END of that tile: below x it is the answer.  Otherwise the tile has a valid base at or behind x
    // (the chunk starts in the tile a gap ends in): the wavefront looks at the 4096 positions below x itself, and if they hold
    // none the tile of x - 4097 ends below x, so its entry answers.  (Round 3 returned "unknown" there and the caller walked
    // the gap 4096 positions per step: 1.7 ms for the 18 Mbp gap of a chromosome, the whole chunk kernel's time.)
    const uint64_t *__restrict__ tile_lv = a.tile_lv ? a.tile_lv + a.tile_first[c] : nullptr;
    const long long n_tiles_c = a.tile_lv ? (long long)(a.tile_first[c + 1] - a.tile_first[c]) : 0;
    auto table_lv = [&](long long x) -> long long {  // entry of the tile holding x - 1, as a position (-1: none up to its end)
        long long ti = (x - 1) / (long long)a.tc;
        if (ti > n_tiles_c - 1) ti = n_tiles_c - 1;  // (a contig that is one tile may be longer than a tile core)
        const uint64_t e = tile_lv[ti];
        return ((uint32_t)(e >> 32) == c + 1) ? (long long)(uint32_t)e - 1 : -1;
    };
    auto last_valid_below = [&](long long x) -> long long {
        if (!tile_lv) return -2;
        const long long lv = table_lv(x);
        if (lv < x) return lv;
        // lane l: block of 64 positions ending at or below x, l blocks back
        const long long b0 = ((x - 1) >> 6) - (long long)lane;
        long long best = -1;
        if (b0 >= 0) {
            const long long w2 = b0 << 1;
            uint64_t m = ((uint64_t)vplane[w2] << 32) | (w2 + 1 < nwords ? (uint64_t)vplane[w2 + 1] : 0ull);  // position 64 b0 + i at bit 63 - i
            const long long first = b0 << 6;
            if (first + 64 > x) m &= ~(U64MAX >> (uint32_t)(x - first));  // keep positions below x (1 <= x - first <= 63 here)
            if (m) best = first + 63 - (long long)__builtin_ctzll(m);
        }
        const uint64_t any = __ballot(best >= 0);
        if (any) return shfl64((uint64_t)best, (int)__ffsll((unsigned long long)any) - 1);  // the nearest block wins
        const long long y = (((x - 1) >> 6) - 63) << 6;  // nothing valid in [y, x)
        return y > 0 ? table_lv(y) : -1;
    };
    if (pm > 0 && !install_kmer) {
        // look back (64 blocks of 64 positions per step) until k valid bases precede pm
        uint32_t have = 0;
        long long hi = pm;  // blocks [hi - 64(lane+1), hi - 64 lane)
        {   // a chunk deep inside a run of N starts right behind the last valid base in front of the run
            const long long lv = last_valid_below(hi);
            if (lv >= -1) hi = ((lv + 64) / 64) * 64;  // smallest multiple of 64 above lv (0 when there is none)
        }
        while (have < k && hi > 0) {
            const long long b0 = hi - 64ll * (lane + 1);
            uint32_t cnt = 0;
            if (b0 >= 0) cnt = __popc(vplane[b0 >> 5]) + __popc(vplane[(b0 >> 5) + 1]);
            const uint32_t incl = wave_incl_sum(cnt);
            const uint64_t enough = __ballot(have + incl >= k);
            if (enough) {
                const int l0 = (int)__ffsll((unsigned long long)enough) - 1;
                pk = hi - 64ll * (l0 + 1);
                have = k;
            } else {
                const uint32_t got = __shfl(incl, 63, 64);
                have += got;
                hi -= 64ll * 64;
                if (got == 0 && hi > 0) {  // nothing valid in these 4096 positions: skip the rest of the run in one step
                    const long long lv = last_valid_below(hi);
                    if (lv >= -1) hi = ((lv + 64) / 64) * 64;
                }
                pk = hi > 0 ? hi : 0;
            }
        }
        if (pk < 0) pk = 0;
    }

    const uint64_t t_lookback = wall_clock64() - t_begin;
    uint64_t F0 = 0, F1 = 0, R0 = 0, R1 = 0;  // rolling k-mer planes (uniform)
    uint32_t rlen = 0, rstart = 0, rend = 0;  // ring state (uniform)
    uint64_t min_x = U64MAX, min_y = U64MAX;
    uint64_t mdist = 0;
    uint64_t n_out = 0;
    uint32_t stat = 0;

    long long drain_end = (long long)cd.drain_end > ce ? (long long)cd.drain_end : ce;
    const long long ext_limit = a.sketch ? 0 : (long long)cd.ext_limit;
    uint64_t n_ext = 0;  // blocks of 64 positions added behind cd.ce
    const uint64_t emit_lo = cd.emit_lo_pos;
    bool out_captured = false;
    bool any_push = false;  // a position of the by-step range [cs, ce) was pushed
    uint64_t sig_out = 0;
    ChunkState o_out;
    auto seam = [&]() {
            // seam: record the warmed-up state, or install the true state handed over by the host
            if (cd.override_state && cd.ring_in != 0xFFFFFFFFu && !a.sketch) {
                // the ring the chunk in front left at its end (push order): inside a stretch of skipped pushes the ring
                // still holds what was pushed in front of the stretch, which no warm-up inside it can rebuild
                const uint64_t *rg = rings + (size_t)cd.ring_in * CHUNK_RING_WORDS;
                __syncthreads();
                s_rx[lane] = rg[lane];
                s_rx[lane + 64] = rg[lane + 64];
                s_ry[lane] = rg[128 + lane];
                s_ry[lane + 64] = rg[128 + lane + 64];
                rlen = (uint32_t)rg[256];
                rstart = 0;
                rend = rlen % w;
                __syncthreads();
            }
            const uint64_t sig = ring_signature(s_rx, rstart, rlen, w, lane);
            if (cd.override_state) {
                const ChunkState t = cd.in_state;
                if (install_kmer) {
                    F0 = t.F0;
                    F1 = t.F1;
                    R0 = t.R0;
                    R1 = t.R1;
                }
                if (t.F0 != F0 || t.F1 != F1 || t.R0 != R0 || t.R1 != R1 || (!a.sketch && t.ring_sig != sig))
                    stat |= 2u;  // warm-up could not even rebuild the k-mer / ring: whole-contig re-run
                min_x = t.min_x;
                min_y = t.min_y;
                mdist = t.mdist;
            }
            if (lane == 0) {
                ChunkState o;
                o.min_x = a.sketch ? 0 : min_x;
                o.min_y = a.sketch ? 0 : min_y;
                o.mdist = a.sketch ? 0 : mdist;
                o.F0 = F0;
                o.F1 = F1;
                o.R0 = R0;
                o.R1 = R1;
                o.ring_sig = a.sketch ? 0 : sig;
                st_in[blockIdx.x] = o;
            }
        };
    auto leave_ring = [&]() {  // the ring at the end of the by-step range, in push order, for the chunk behind this one
        if (cd.ring_out == 0xFFFFFFFFu || a.sketch) return;
        uint64_t *rg = rings + (size_t)cd.ring_out * CHUNK_RING_WORDS;
        const uint32_t q0 = lane, q1 = lane + 64;
        rg[q0] = q0 < w ? s_rx[ring_mod(rstart + q0, w)] : U64MAX;
        rg[q1] = q1 < w ? s_rx[ring_mod(rstart + q1, w)] : U64MAX;
        rg[128 + q0] = q0 < w ? s_ry[ring_mod(rstart + q0, w)] : U64MAX;
        rg[128 + q1] = q1 < w ? s_ry[ring_mod(rstart + q1, w)] : U64MAX;
        if (lane == 0) rg[256] = rlen;
    };
    long long cblk = -1;  // first block (64 positions) of the 64 blocks held in the lanes' registers
    uint32_t c_vh = 0, c_vl = 0;
    uint2 c_ph = make_uint2(0, 0), c_pl = make_uint2(0, 0);
    auto fetch_blocks = [&](long long blk) {
        cblk = blk;
        const long long wj = (blk + lane) << 1;
        c_vh = wj < nwords ? vplane[wj] : 0u;
        c_vl = wj + 1 < nwords ? vplane[wj + 1] : 0u;
        c_ph = wj < nwords ? planes[wj] : make_uint2(0, 0);
        c_pl = wj + 1 < nwords ? planes[wj + 1] : make_uint2(0, 0);
    };
    uint64_t n_push = 0;        // pushes at the steps [cs, ce)
    uint64_t lane_bmin = U64MAX;  // per lane: smallest x of those pushes with branch 2 enabled (shmmrutils.rs:516-520)
    uint32_t n_steps = 0;  // loop iterations (diagnostics: status bits 8..31)
    for (long long base = pk; base < drain_end; base += 64) {
        ++n_steps;
        if (base == ce && ce + 64 <= ext_limit && mdist > (uint64_t)(w - 1)) {
            // The island was to end here, inside a tile, and the machine arrives STUCK: it emits nothing until a push reaches down
            // to min_mer (shmmrutils.rs:516-520) -- the tile's closed form knows nothing of that.  The island goes on, block by
            // block; the push that frees the machine is smaller than every push since it got stuck, i.e. the minimum of its
            // window, and from there on the machine is the regular one again (the probe at the new end verifies it).
            ce += 64;
            drain_end += 64;
            ++n_ext;
        }
        if (base >= ce && !out_captured) {
            // end of the by-step range: this is the state the next chunk / the island-end probe must match
            leave_ring();
            sig_out = ring_signature(s_rx, rstart, rlen, w, lane);
            o_out.min_x = a.sketch ? 0 : min_x;
            o_out.min_y = a.sketch ? 0 : min_y;
            o_out.mdist = a.sketch ? 0 : mdist;
            o_out.F0 = F0;
            o_out.F1 = F1;
            o_out.R0 = R0;
            o_out.R1 = R1;
            o_out.ring_sig = a.sketch ? 0 : sig_out;
            out_captured = true;
        }
        const bool draining = base >= ce;  // island end: only elements below ce are still ours
        if (base == cs && cs > 0) seam();
        // the step's four words come from the lanes' registers: lane j holds block cblk + j (one coalesced load per 4096
        // positions instead of a dependent trip to memory per step -- a lone wavefront spent ~2 us per step on those)
        {
            const long long blk = base >> 6;
            if (cblk < 0 || blk < cblk || blk >= cblk + 64) fetch_blocks(blk);
        }
        const int li = __builtin_amdgcn_readfirstlane((int)((base >> 6) - cblk));
        const uint32_t v_hi = (uint32_t)__builtin_amdgcn_readlane((int)c_vh, li), v_lo = (uint32_t)__builtin_amdgcn_readlane((int)c_vl, li);
        const uint64_t V = ((uint64_t)v_hi << 32) | v_lo;
        const bool kmer_only = base < pm;
        if (kmer_only && V == 0) {
            // nothing touches the k-mer here (shmmrutils.rs:461-476).  Inside a long run of non-ACGT bytes (a chunk deep in
            // an 18 Mbp gap of a reference chromosome rolls from the last k valid bases in FRONT of the gap): jump to the
            // next block of 64 positions that holds a valid base, 64 blocks per step, instead of visiting every block
            long long nb = base + 64;
            {   // the rest of the way to pm holds no valid base at all (the usual case inside a long run of N)
                const long long lv = pm > 0 ? last_valid_below(pm) : -2;
                if (lv >= -1 && lv < nb) nb = pm;
            }
            while (nb < pm) {
                const long long bl = nb + 64ll * lane;
                uint32_t any = 0;
                if (bl < pm) {
                    const long long w2 = bl >> 5;
                    any = vplane[w2];
                    if (w2 + 1 < nwords) any |= vplane[w2 + 1];
                }
                const uint64_t m = __ballot(any != 0u);
                if (m) {
                    nb += 64ll * ((long long)__ffsll((unsigned long long)m) - 1);
                    break;
                }
                nb += 64ll * 64;
            }
            if (nb > pm) nb = pm;  // (pm and base are congruent mod 64)
            base = nb - 64;
            continue;
        }
        const uint64_t P0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)c_ph.x, li) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)c_pl.x, li);  // position base+i at bit 63-i
        const uint64_t P1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)c_ph.y, li) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)c_pl.y, li);
        const long long pos = base + lane;
        uint64_t f0, f1, r0, r1;
        if (V == U64MAX) {
            // all 64 bases valid: lane i's state = carried state advanced by i+1 bases, in closed form
            f0 = (((F0 << lane) << 1) | (P0 >> (63 - lane))) & kmask;
            f1 = (((F1 << lane) << 1) | (P1 >> (63 - lane))) & kmask;
            r0 = (((R0 >> lane) >> 1) | (__brevll((~P0) >> (63 - lane)) >> (64 - k))) & kmask;
            r1 = (((R1 >> lane) >> 1) | (__brevll((~P1) >> (63 - lane)) >> (64 - k))) & kmask;
            F0 = readlane64(f0, 63);
            F1 = readlane64(f1, 63);
            R0 = readlane64(r0, 63);
            R1 = readlane64(r1, 63);
        } else if (V == 0) {
            // no valid base in the block (inside a run of N): the k-mer stays what it was for all 64 positions
            f0 = F0;
            f1 = F1;
            r0 = R0;
            r1 = R1;
        } else {
            // bytes outside ACGT do not touch the k-mer (shmmrutils.rs:461-476): roll uniformly
            f0 = f1 = r0 = r1 = 0;
            for (uint32_t i = 0; i < 64; ++i) {
                const uint32_t bit = 63 - i;
                if ((V >> bit) & 1) {
                    const uint64_t c0b = (P0 >> bit) & 1, c1b = (P1 >> bit) & 1;
                    F0 = ((F0 << 1) | c0b) & kmask;
                    F1 = ((F1 << 1) | c1b) & kmask;
                    R0 = ((R0 >> 1) | ((c0b ^ 1) << shift)) & kmask;
                    R1 = ((R1 >> 1) | ((c1b ^ 1) << shift)) & kmask;
                }
                if (lane == i) {
                    f0 = F0;
                    f1 = F1;
                    r0 = R0;
                    r1 = R1;
                }
            }
        }
        if (kmer_only) continue;
        const bool emit_on = base >= cs;
        const uint64_t pos_lo = emit_lo, pos_hi = draining ? (uint64_t)ce : ~0ull;  // position filter
        const bool skip = (f0 == r0) && (f1 == r1);
        const bool pushed = !skip && pos >= (long long)k && pos < L;
        // a block without a single push (inside an array of palindromic k-mers, shmmrutils.rs:477-480) touches nothing
        if (!a.sketch && __ballot(pushed) == 0) continue;
        uint32_t st;
        uint64_t h;
        const uint64_t x = kmer_x(f0, f1, r0, r1, k, st, h);
        const uint64_t y = ((uint64_t)c << 32) | ((uint64_t)pos << 1) | st;

        if (a.sketch) {  // shmmrutils.rs:621-628
            const bool em = emit_on && !draining && pushed && h < sketch_thr;
            const uint64_t m = __ballot(em);
            if (em) {
                const uint64_t o = n_out + __popcll(m & lt_mask);
                if (o < cap) {
                    out[o] = l1rec_from_xy(x, y);
                }
            }
            n_out += __popcll(m);
            continue;
        }

        const uint64_t pmask = __ballot(pushed);
        const bool b_en = (uint64_t)pos >= (uint64_t)(w + k) && (uint64_t)pos < Lb && pos < L;
        if (emit_on && !draining) {
            any_push = true;
            n_push += __popcll(pmask);
            if (pushed && b_en) lane_bmin = umin64(lane_bmin, x);
        }
        uint32_t cur = 0;  // first unprocessed lane of this step
        // All branch-2 events of the step at once.  A push is a branch-2 event (shmmrutils.rs:516-527) iff it is enabled and its
        // x is <= the x of the last event -- and every event lowers that bound to its own x, so the bound a lane sees is
        // min(min_mer.x, the enabled pushes in front of it): ONE exclusive prefix minimum decides all 64 lanes.  That is the
        // whole story of the step when no rescan falls into it: a rescan fires at the push that finds mdist == w - 1, i.e. after
        // w - 1 pushes without an event.  Tie runs (runs of N, homopolymers: every push is an event) and microsatellites (an
        // event every period) used to cost one iteration of the loop below per event -- 16 us for a step of (AC)n.
        if (pmask) {
            uint64_t thr;  // exclusive prefix minimum over the enabled pushes, and min_mer.x
            {
                const uint64_t incl = wave_incl_min64((pushed && b_en) ? x : U64MAX);
                thr = umin64(wave_shr1_ones(incl), min_x);
            }
            const bool is_b = pushed && b_en && x <= thr;
            const uint64_t bm = __ballot(is_b);
            const uint32_t tot = __popcll(pmask);
            const uint32_t rk = __popcll(pmask & lt_mask);  // rank of this lane among the pushes of the step
            bool safe;
            if (mdist > (uint64_t)(w - 1)) {
                safe = bm != 0;  // stuck beyond w - 1: only an event moves the machine (none: the loop below adds the pushes)
            } else {
                const uint32_t t_r = (uint32_t)((uint64_t)(w - 1) - mdist);  // the push of rank t_r finds mdist == w - 1 unless an event came first
                if (bm) {
                    const int fb = (int)__ffsll((unsigned long long)bm) - 1;
                    safe = (uint32_t)__popcll(pmask & ((1ull << fb) - 1ull)) < t_r;
                } else {
                    safe = false;  // no event: one iteration of the loop below
                }
            }
            if (safe && w < 65) {  // a rescan between two events of the step, or behind the last one, needs w - 1 pushes in between
                const uint64_t below = bm & lt_mask;
                const int pb = below ? 63 - (int)__clzll((long long)below) : -1;
                const uint32_t prk = pb >= 0 ? (uint32_t)__popcll(pmask & ((1ull << pb) - 1ull)) : 0u;
                const bool viol = is_b && pb >= 0 && rk - prk >= w;
                const int lb = 63 - (int)__clzll((long long)bm);
                const uint32_t after = (uint32_t)__popcll(pmask & ~((2ull << lb) - 1ull));  // pushes behind the last event
                safe = __ballot(viol) == 0 && after < w;
            }
            if (safe) {
                if (pushed && tot - rk <= w) {
                    const uint32_t slot = ring_mod(rend + rk, w);
                    s_rx[slot] = x;
                    s_ry[slot] = y;
                }
                if (emit_on && !draining) {
                    if (is_b) {
                        const uint64_t o = n_out + __popcll(bm & lt_mask);
                        if (o < cap) out[o] = l1rec_from_xy(x, y);
                    }
                    n_out += __popcll(bm);
                }
                rend = ring_mod(rend + tot, w);
                if (rlen + tot >= w) {
                    rlen = w;
                    rstart = rend;
                } else {
                    rlen += tot;
                }
                const int last = 63 - (int)__clzll((long long)bm);
                min_x = readlane64(x, last);
                min_y = readlane64(y, last);
                mdist = (uint64_t)__popcll(pmask & ~((2ull << last) - 1ull));
                cur = 64;
                __syncthreads();
            }
        }
        for (;;) {
            const uint64_t rest = (cur >= 64) ? 0ull : (pmask & (U64MAX << cur));
            if (rest == 0) break;
            // R: the (w-1-mdist+1)-th remaining push, if mdist <= w-1
            uint64_t mR = 0;
            if (mdist <= (uint64_t)(w - 1)) {
                const uint32_t tsteps = (uint32_t)((uint64_t)(w - 1) - mdist);
                const uint32_t rank = __popcll(rest & lt_mask);
                mR = __ballot(pushed && lane >= cur && rank == tsteps);
            }
            const uint64_t mB = __ballot(pushed && lane >= cur && b_en && x <= min_x);
            const int iR = mR ? (int)__ffsll((unsigned long long)mR) - 1 : 64;
            const int iB = mB ? (int)__ffsll((unsigned long long)mB) - 1 : 64;
            const int iE = (iB < iR) ? iB : iR;  // B only when strictly before R (R is tested first)
            const uint64_t range = (iE >= 63) ? rest : (rest & ((2ull << iE) - 1));
            // push every pushed position in [cur, iE] (or to the end of the step) into the ring
            const uint32_t tot = __popcll(range);
            if ((range >> lane) & 1) {
                const uint32_t rk = __popcll(range & lt_mask);
                if (tot - rk <= w) {
                    const uint32_t slot = ring_mod(rend + rk, w);
                    s_rx[slot] = x;
                    s_ry[slot] = y;
                }
            }
            rend = ring_mod(rend + tot, w);
            if (rlen + tot >= w) {
                rlen = w;
                rstart = rend;
            } else {
                rlen += tot;
            }
            __syncthreads();
            if (iE == 64) {  // no event in the rest of the step
                mdist += tot;
                break;
            }
            if (iB < iR) {  // branch 2 (shmmrutils.rs:516-527)
                const uint64_t ex = readlane64(x, iB), ey = readlane64(y, iB);
                if (emit_on && !draining) {  // the element of a B event sits at the step itself
                    if (lane == 0 && n_out < cap) {
                        out[n_out] = l1rec_from_xy(ex, ey);
                    }
                    n_out += 1;
                }
                min_x = ex;
                min_y = ey;
                mdist = 0;
            } else {  // rescan (shmmrutils.rs:503-515)
                const uint32_t q0 = lane, q1 = lane + 64;
                const uint32_t s0 = ring_mod(rstart + q0, w), s1 = ring_mod(rstart + q1, w);
                const uint64_t x0 = (q0 < w) ? s_rx[s0] : U64MAX;
                const uint64_t x1 = (q1 < w) ? s_rx[s1] : U64MAX;
                const uint64_t mn = wave_min64(umin64(x0, x1));
                const bool e0 = (q0 < w) && x0 == mn, e1 = (q1 < w) && x1 == mn;
                const uint64_t m0 = __ballot(e0), m1 = __ballot(e1);
                if (emit_on) {
                    const uint64_t y0 = e0 ? s_ry[s0] : 0, y1 = e1 ? s_ry[s1] : 0;
                    const uint64_t p0 = (y0 & 0xFFFFFFFFull) >> 1, p1 = (y1 & 0xFFFFFFFFull) >> 1;
                    const bool w0 = e0 && p0 >= pos_lo && p0 < pos_hi, w1 = e1 && p1 >= pos_lo && p1 < pos_hi;
                    const uint64_t wm0 = __ballot(w0), wm1 = __ballot(w1);
                    const uint32_t n0 = __popcll(wm0), n1 = __popcll(wm1);
                    if (w0) {
                        const uint64_t o = n_out + __popcll(wm0 & lt_mask);
                        if (o < cap) {
                            out[o] = l1rec_from_xy(x0, y0);
                        }
                    }
                    if (w1) {
                        const uint64_t o = n_out + n0 + __popcll(wm1 & lt_mask);
                        if (o < cap) {
                            out[o] = l1rec_from_xy(x1, y1);
                        }
                    }
                    n_out += n0 + n1;
                }
                const uint32_t qlast = m1 ? (64 + 63 - (uint32_t)__clzll((long long)m1)) : (63 - (uint32_t)__clzll((long long)m0));
                min_x = mn;
                min_y = s_ry[ring_mod(rstart + qlast, w)];
                const uint64_t ppos = (uint64_t)(base + iR);
                mdist = ppos - ((min_y & 0xFFFFFFFFull) >> 1);
            }
            cur = (uint32_t)iE + 1;
            __syncthreads();
        }
    }
    if (cs > 0 && cs >= drain_end) seam();  // probe: nothing to emit, only the warmed-up state at cs
    // ---- state at the end of the by-step range (the next chunk's seam) and the segment entry
    if (!out_captured) {
        leave_ring();
        sig_out = ring_signature(s_rx, rstart, rlen, w, lane);
        o_out.min_x = a.sketch ? 0 : min_x;
        o_out.min_y = a.sketch ? 0 : min_y;
        o_out.mdist = a.sketch ? 0 : mdist;
        o_out.F0 = F0;
        o_out.F1 = F1;
        o_out.R0 = R0;
        o_out.R1 = R1;
        o_out.ring_sig = a.sketch ? 0 : sig_out;
    }
    if (lane == 0) {
        st_out[blockIdx.x] = o_out;
        // (the chunk's list stays in its region: the host puts the lists of the chunks that start in one tile together behind
        // the last round -- api.hip: run_exact_islands -- so chunks may be shorter than a tile)
        if (cd.seg != 0xFFFFFFFFu && (n_out > cap || n_out > 0xFFFFFFFFull)) {
            stat |= 1u;  // region too small: the host re-runs this chunk with a full-size region
            n_out = 0;
        }
        // bit 2: no position of [cs, ce) was pushed and nothing is drained behind ce -- the machine's state at ce IS its state at
        // cs and the chunk emits nothing, whatever that state was: the host hands the state of the chunk in front straight
        // to the chunk behind (a long array of palindromic k-mers costs no round of seam correction per chunk)
        if (!a.sketch && !any_push && drain_end <= ce && cs < ce) stat |= 4u;
        status[blockIdx.x] = stat | ((n_steps > 0xFFFFFFu ? 0xFFFFFFu : n_steps) << 8);
    }
    // what the host needs to pass a state THROUGH this chunk without running it again (api.hip: run_exact_islands): the pushes
    // of [cs, ce) and the smallest x among those that could be a branch-2 event
    {
        const uint64_t bmin = wave_min64(lane_bmin);
        if (lane == 0) {
            info[4 * (size_t)blockIdx.x] = n_push;
            info[4 * (size_t)blockIdx.x + 1] = bmin;
            info[4 * (size_t)blockIdx.x + 2] = ((wall_clock64() - t_begin) & 0xFFFFFFFFull) | (t_lookback << 32);  // 100 MHz ticks (diagnostics)
            info[4 * (size_t)blockIdx.x + 3] = (cd.seg != 0xFFFFFFFFu ? n_out : 0ull) | (n_ext << 40);  // elements in the chunk's region; blocks added
        }
    }
}

// every segment (tiles + tail) of the listed contigs becomes empty: their lists come from the chunk kernel
__global__ void zero_contig_segs_kernel(L1Args a, const uint32_t *__restrict__ list, uint32_t n_list) {
    const uint32_t c = list[blockIdx.x];
    const uint32_t s0 = a.tile_first[c] + c, s1 = a.tile_first[c + 1] + c + 1;  // incl. the tail segment
    for (uint32_t s = s0 + threadIdx.x; s < s1; s += blockDim.x) a.seg_cnt[s] = 0;
}

__global__ void zero_seg_ranges_kernel(L1Args a, const uint32_t *__restrict__ ranges, uint32_t n_ranges) {
    const uint32_t s0 = ranges[2 * blockIdx.x], s1 = ranges[2 * blockIdx.x + 1];
    for (uint32_t s = s0 + threadIdx.x; s < s1; s += blockDim.x) a.seg_cnt[s] = 0;
}

// one lane per tile: does the tile's extended range (core +- (w-1), k-mer look-back, one 64-position step of slack) contain a
// byte that is not a base, and where is the last valid position of its core?  Tiles of clean contigs (the usual case) are done
// with one look at the contig's count.  The tiles of a contig WITH such bytes are scanned by the whole wavefront, one tile after
// the other, every lane two or three words of the tile's ~135 (a lane walking its own tile word by word made every load of
// the wavefront a gather of 64 cache lines); eight tiles per wavefront keep that serial part short.  Runs IN FRONT of the tile
// kernel, which leaves flagged tiles alone.
constexpr uint32_t MARK_TPW = 8;  // tiles per wavefront: a wavefront's dirty tiles are scanned one after the other (~1.5 us each)
__global__ __launch_bounds__(256) void mark_invalid_tiles_kernel(L1Args a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t tile = (blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * MARK_TPW + lane;
    const bool live = lane < MARK_TPW && tile < a.n_tiles;
    TileDesc td;
    td.word_off = 0;
    td.len = td.contig = td.tile_local = 0;
    if (live) td = a.desc[tile];
    bool dirty = false;
    if (live) {
        const long long c1 = l1_core_end((long long)td.tile_local * a.tc, (long long)td.len, a.tc, a.ext);
        if (a.b.n_invalid[td.contig] == 0) a.tile_lv[tile] = ((uint64_t)(td.contig + 1) << 32) | (uint64_t)c1;  // every base valid
        else dirty = true;
    }
    uint64_t todo = __ballot(dirty);
    while (todo) {  // (uniform)
        const int j = __builtin_amdgcn_readfirstlane((int)__ffsll((unsigned long long)todo) - 1);
        todo &= todo - 1;
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)td.contig, j);
        const long long L = (long long)(uint32_t)__builtin_amdgcn_readlane((int)td.len, j);
        const long long tl = (long long)(uint32_t)__builtin_amdgcn_readlane((int)td.tile_local, j);
        const uint64_t woff = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(td.word_off >> 32), j) << 32) |
                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)td.word_off, j);
        const uint32_t *__restrict__ v = a.b.valid + woff;
        const long long c0 = tl * a.tc, c1 = l1_core_end(c0, L, a.tc, a.ext);
        long long lo = c0 - (long long)(a.w - 1) - (long long)(a.k - 1) - 64;
        long long hi = c1 + (long long)(a.w - 1) + 64;
        if (lo < 0) lo = 0;
        if (hi > L) hi = L;
        bool bad = false, any_valid = false;
        long long last = -1;  // last valid position of the core [c0, c1) among this lane's words
        for (long long wj = (lo >> 5) + lane; wj <= (hi - 1) >> 5; wj += 64) {
            const long long w0 = wj << 5;
            uint32_t m = 0xFFFFFFFFu;  // bits of this word inside [lo, hi)  (position w0 + i at bit 31 - i)
            if (w0 < lo) m &= 0xFFFFFFFFu >> (uint32_t)(lo - w0);
            if (w0 + 32 > hi) m &= 0xFFFFFFFFu << (uint32_t)(w0 + 32 - hi);
            const uint32_t word = v[wj];
            const uint32_t x = word & m;
            bad = bad || x != m;
            any_valid = any_valid || x != 0u;
            uint32_t mc = 0u;  // bits inside the core
            if (w0 + 32 > c0 && w0 < c1) {
                mc = 0xFFFFFFFFu;
                if (w0 < c0) mc &= 0xFFFFFFFFu >> (uint32_t)(c0 - w0);
                if (w0 + 32 > c1) mc &= 0xFFFFFFFFu << (uint32_t)(w0 + 32 - c1);
            }
            const uint32_t bits = word & mc;
            if (bits) last = w0 + 31 - (long long)__builtin_ctz(bits);  // (this lane's words ascend)
        }
        const bool w_bad = __ballot(bad) != 0, w_any = __ballot(any_valid) != 0;
        // the largest `last` of the wave: all ones minus the value through the wave minimum
        const uint64_t inv = wave_min64(~(uint64_t)(last + 1));
        const long long w_last = (long long)(~inv) - 1;
        if ((int)lane == j) {
            a.tile_lv[tile] = ((uint64_t)(c + 1) << 32) | (uint64_t)(w_last + 1);
            if (w_bad) {
                // bit 0: the tile kernel, which has finished, saw a palindromic k-mer; bit 1: a non-ACGT byte in the extended range;
                // bit 2: NOTHING but non-ACGT bytes there -- the inside of a gap (api.hip: run_islands leaves such tiles out)
                a.tile_flags[tile] |= w_any ? 2 : 6;
                a.desc[tile].skip = 1u;  // (the tile kernel leaves this tile alone)
                atomicOr(a.cursor + 2, 2ull);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// one thread per tile: its descriptor.  The same threads clear the call's cursors, contig flags and tile flags (one block of
// memory, api.hip) -- a memset in front of this kernel was a launch and a bubble of its own in every call.
__global__ void tile_desc_kernel(L1Args a) {
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile < 8) a.cursor[tile] = 0ull;
    if (tile < a.n_contigs) {
        a.contig_flags[tile] = 0u;
        if (a.b.len[tile] == 0) {  // no tile, hence nobody to run the tail: its (empty) segment
            const uint32_t sidx = a.tile_first[tile + 1] + tile;
            a.seg_off[sidx] = 0;
            a.seg_cnt[sidx] = 0;
            a.seg_cid[sidx] = tile;
        }
    }
    if (tile == 0) a.seg_cnt[a.n_tiles + a.n_contigs] = 0;  // sentinel of the scan over the segment counts
    if (tile >= a.n_tiles) return;
    a.tile_flags[tile] = 0;
    const uint32_t c = find_contig(a.tile_first, a.n_contigs, tile);
    TileDesc d;
    d.word_off = a.b.word_off[c];
    d.len = a.b.len[c];
    d.contig = c;
    d.tile_local = tile - a.tile_first[c];
    d.skip = 0u;  // contigs with non-ACGT bytes run too: only the tiles near such bytes are replaced (islands)
    d._pad[0] = d._pad[1] = 0;
    a.desc[tile] = d;
}

void launch_mark_invalid_tiles(hipStream_t st, const L1Args &a);
void launch_level1_pre(hipStream_t st, const L1Args &a, uint64_t *tile_lv, bool known_clean) {
    if (a.n_tiles == 0) return;
    const uint32_t n_desc = a.n_tiles > a.n_contigs ? a.n_tiles : a.n_contigs;  // (>= 8: a tile per non-empty contig ... or not)
    hipLaunchKernelGGL(tile_desc_kernel, dim3(((n_desc > 8 ? n_desc : 8) + 255) / 256), dim3(256), 0, st, a);
    // known_clean: the host packer has counted the batch's non-ACGT bytes and found none, AND the caller never runs islands on this
    // pass (the query path's level-1 form: what the tile kernel flags goes back to the shimmer pipeline, which plans again) -- no tile
    // can be flagged here and nobody reads the last-valid table
    if (!known_clean) {   // flags tiles with a non-ACGT byte in reach (the tile kernel skips them), records every tile's last valid position
        L1Args am = a;
        am.tile_lv = tile_lv;
        launch_mark_invalid_tiles(st, am);
    }
}
// (behind launch_level1_pre on the same stream)
void launch_level1_tiles(hipStream_t st, const L1Args &a) {
    if (a.n_tiles == 0) return;
    if (a.ext == (uint32_t)L1_EXT_SHORT) {  // batches of short contigs: one wavefront per tile
        const dim3 g(a.n_tiles), bl(L1_BLOCK_SHORT);
        if (a.sketch)
            hipLaunchKernelGGL((level1_tile_kernel<0, 0, true, L1_BLOCK_SHORT>), g, bl, 0, st, a);
        else if (a.w == 80 && a.k == 56)
            hipLaunchKernelGGL((level1_tile_kernel<80, 56, false, L1_BLOCK_SHORT>), g, bl, 0, st, a);
        else
            hipLaunchKernelGGL((level1_tile_kernel<0, 0, false, L1_BLOCK_SHORT>), g, bl, 0, st, a);
        return;
    }
    if (a.sketch)
        hipLaunchKernelGGL((level1_tile_kernel<0, 0, true, L1_BLOCK>), dim3(a.n_tiles), dim3(L1_BLOCK), 0, st, a);
    else if (a.w == 80 && a.k == 56)
        hipLaunchKernelGGL((level1_tile_kernel<80, 56, false, L1_BLOCK>), dim3(a.n_tiles), dim3(L1_BLOCK), 0, st, a);
    else if (a.w == 48 && a.k == 56)
        hipLaunchKernelGGL((level1_tile_kernel<48, 56, false, L1_BLOCK>), dim3(a.n_tiles), dim3(L1_BLOCK), 0, st, a);
    else
        hipLaunchKernelGGL((level1_tile_kernel<0, 0, false, L1_BLOCK>), dim3(a.n_tiles), dim3(L1_BLOCK), 0, st, a);
}
// LDS one workgroup of the tile kernel launch_level1_tiles would launch for `a` occupies on its CU (static size rounded up to the
// allocation granule).  A kernel that is to run BESIDE the tile kernel asks for exactly as much: LDS is handed out as contiguous
// ranges, and a workgroup of another size that comes and goes between the tiles' 36 KB ranges leaves them shifted -- four ranges'
// worth of free LDS in two pieces, three tile workgroups per CU instead of four for the rest of the launch (measured: 18.4 -> 28 ms).
uint32_t level1_tile_lds_bytes(const L1Args &a) {
    const void *f;
    if (a.ext == (uint32_t)L1_EXT_SHORT)
        f = a.sketch ? (const void *)level1_tile_kernel<0, 0, true, L1_BLOCK_SHORT>
                     : (a.w == 80 && a.k == 56) ? (const void *)level1_tile_kernel<80, 56, false, L1_BLOCK_SHORT>
                                                : (const void *)level1_tile_kernel<0, 0, false, L1_BLOCK_SHORT>;
    else
        f = a.sketch ? (const void *)level1_tile_kernel<0, 0, true, L1_BLOCK>
                     : (a.w == 80 && a.k == 56) ? (const void *)level1_tile_kernel<80, 56, false, L1_BLOCK>
                     : (a.w == 48 && a.k == 56) ? (const void *)level1_tile_kernel<48, 56, false, L1_BLOCK>
                                                : (const void *)level1_tile_kernel<0, 0, false, L1_BLOCK>;
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, f) != hipSuccess) return 0;
    return (uint32_t)((at.sharedSizeBytes + LDS_GRANULE - 1) / LDS_GRANULE * LDS_GRANULE);
}
void launch_level1_tails(hipStream_t st, const L1Args &a) {
    if (a.n_contigs == 0) return;
    hipLaunchKernelGGL(level1_tail_kernel, dim3((a.n_contigs + TAIL_WAVES - 1) / TAIL_WAVES), dim3(64 * TAIL_WAVES), 0, st, a);
}
void launch_level1_chunks(hipStream_t st, const L1Args &a, const ChunkDesc *d_descs, uint32_t n_chunks,
                          ChunkState *d_in, ChunkState *d_out, uint32_t *d_status, uint64_t *d_rings, uint64_t *d_info) {
    if (n_chunks == 0) return;
    hipLaunchKernelGGL(level1_chunk_kernel, dim3(n_chunks), dim3(64), 0, st, a, d_descs, d_in, d_out, d_status, d_rings, d_info);
}
// the lists of the chunks that start in one tile, put together: copy i moves list[3i + 2] records from element list[3i] to
// element list[3i + 1] of the level-1 buffer (a fresh region: source and destination never overlap); one wavefront per copy
__global__ __launch_bounds__(64) void assemble_chunks_kernel(L1Rec *__restrict__ buf, const uint64_t *__restrict__ list) {
    const uint64_t src = list[3 * (size_t)blockIdx.x], dst = list[3 * (size_t)blockIdx.x + 1], cnt = list[3 * (size_t)blockIdx.x + 2];
    for (uint64_t i = threadIdx.x; i < cnt; i += 64) buf[dst + i] = buf[src + i];
}
// segment-table entries of the assembled tiles: segs[3i] = segment | contig << 32, segs[3i + 1] = first element, segs[3i + 2] = count
__global__ void set_segs_kernel(L1Args a, const uint64_t *__restrict__ segs, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t sg = (uint32_t)segs[3 * (size_t)i];
    a.seg_off[sg] = segs[3 * (size_t)i + 1];
    a.seg_cnt[sg] = (uint32_t)segs[3 * (size_t)i + 2];
    a.seg_cid[sg] = (uint32_t)(segs[3 * (size_t)i] >> 32);
}
// A tile an island of the exact machine begins or ends inside (pipeline.hip: IslandRun::finish): e[0] = segment | contig << 32, e[1] =
// first element of the chunks' lists put together for this tile, e[2] = their count, e[3] = lo (the tile's own elements at positions
// below lo stay in front; ~0: none do), e[4] = hi (those at positions >= hi stay behind; ~0: none), e[5] = room behind the lists.  The
// tile's own segment is in position order (the tile kernel's ordered compaction): the kept prefix is copied right-aligned against
// e[1], the kept suffix behind e[1] + e[2], and the tile's entry points at the contiguous result.  One wavefront per tile.
__global__ __launch_bounds__(64) void splice_segs_kernel(L1Args a, const uint64_t *__restrict__ ents) {
    const uint64_t *e = ents + 6 * (size_t)blockIdx.x;
    const uint32_t sg = (uint32_t)e[0];
    const uint64_t mid = e[1], mid_cnt = e[2], lo = e[3], hi = e[4], room = e[5];
    const uint64_t own = a.seg_off[sg];
    const uint32_t cnt = a.seg_cnt[sg];
    const L1Rec *__restrict__ src = a.out + own;
    uint32_t n_lo = 0, n_hi = 0;  // own elements below lo / at or above hi
    for (uint32_t i = threadIdx.x; i < cnt; i += 64) {
        const uint64_t pos = (uint64_t)(src[i].ypos >> 1);
        n_lo += (lo != ~0ull && pos < lo) ? 1u : 0u;
        n_hi += (hi != ~0ull && pos >= hi) ? 1u : 0u;
    }
    n_lo = wave_incl_sum(n_lo);
    n_hi = wave_incl_sum(n_hi);
    n_lo = (uint32_t)__builtin_amdgcn_readlane((int)n_lo, 63);
    n_hi = (uint32_t)__builtin_amdgcn_readlane((int)n_hi, 63);
    if ((uint64_t)n_hi > room) n_hi = (uint32_t)room;  // (cannot happen: one element per position at most; never write beyond the room)
    L1Rec *__restrict__ dst_lo = a.out + (mid - n_lo), *__restrict__ dst_hi = a.out + (mid + mid_cnt);
    for (uint32_t i = threadIdx.x; i < n_lo; i += 64) dst_lo[i] = src[i];
    for (uint32_t i = threadIdx.x; i < n_hi; i += 64) dst_hi[i] = src[cnt - n_hi + i];
    __syncthreads();
    if (threadIdx.x == 0) {
        a.seg_off[sg] = mid - n_lo;
        a.seg_cnt[sg] = (uint32_t)(n_lo + mid_cnt + n_hi);
        a.seg_cid[sg] = (uint32_t)(e[0] >> 32);
    }
}
void launch_splice_segs(hipStream_t st, const L1Args &a, const uint64_t *d_ents, uint32_t n_ents) {
    if (n_ents) hipLaunchKernelGGL(splice_segs_kernel, dim3(n_ents), dim3(64), 0, st, a, d_ents);
}
void launch_assemble_chunks(hipStream_t st, const L1Args &a, const uint64_t *d_copies, uint32_t n_copies, const uint64_t *d_segs, uint32_t n_segs) {
    if (n_copies) hipLaunchKernelGGL(assemble_chunks_kernel, dim3(n_copies), dim3(64), 0, st, a.out, d_copies);
    if (n_segs) hipLaunchKernelGGL(set_segs_kernel, dim3((n_segs + 255) / 256), dim3(256), 0, st, a, d_segs, n_segs);
}
void launch_zero_seg_ranges(hipStream_t st, const L1Args &a, const uint32_t *d_ranges, uint32_t n_ranges) {
    if (n_ranges == 0) return;
    hipLaunchKernelGGL(zero_seg_ranges_kernel, dim3(n_ranges), dim3(256), 0, st, a, d_ranges, n_ranges);
}
void launch_mark_invalid_tiles(hipStream_t st, const L1Args &a) {
    if (a.n_tiles == 0) return;
    hipLaunchKernelGGL(mark_invalid_tiles_kernel, dim3((a.n_tiles + 4 * MARK_TPW - 1) / (4 * MARK_TPW)), dim3(256), 0, st, a);
}
void launch_zero_contig_segs(hipStream_t st, const L1Args &a, const uint32_t *d_list, uint32_t n_list) {
    if (n_list == 0) return;
    hipLaunchKernelGGL(zero_contig_segs_kernel, dim3(n_list), dim3(256), 0, st, a, d_list, n_list);
}

}  // namespace pgr
