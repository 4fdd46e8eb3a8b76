// level1_select.h -- the position-parallel closed form of the level-1 SHIMMER selection for ONE tile (16 positions per lane):
// k-mer planes, canonical strand, u64hash, the two van Herk window passes.  Shared by level1_tile_kernel (level1.hip: the
// 10 Gbp batches) and small_shmmr_kernel (small.hip: batches of short contigs, one workgroup per contig).  Everything here is
// internal-linkage device code: including it in two translation units produces two identical copies.
#pragma once
#include "pgr_device.h"
#include "pgr_internal.h"

#ifndef PGR_TILE_ATTR
#define PGR_TILE_ATTR  // experiment hook: e.g. -DPGR_TILE_ATTR='__attribute__((amdgpu_waves_per_eu(4,4)))'
#endif

namespace pgr {

namespace {

__device__ __forceinline__ L1Rec l1rec_from_xy(uint64_t x, uint64_t y) {  // x = key << 8 | k
    L1Rec r;
    r.key_lo = (uint32_t)(x >> 8);
    r.key_hi = (uint32_t)(x >> 40);
    r.ypos = (uint32_t)y;
    return r;
}

__device__ __forceinline__ uint32_t find_contig(const uint32_t *__restrict__ tile_first, uint32_t n, uint32_t tile) {
    // largest c with tile_first[c] <= tile
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tile_first[mid] <= tile) lo = mid;
        else hi = mid;
    }
    return lo;
}

struct ContigGeom {
    long long L;       // contig length
    long long jstart;  // first window end (position) handled by the closed form
    long long jend;    // last window end handled by the closed form (jend < jstart: none)
};

// gap-free contig whose first pushed position is k (all bases valid)
__device__ __forceinline__ ContigGeom contig_geom(uint32_t len, uint32_t w, uint32_t k) {
    ContigGeom g;
    g.L = len;
    g.jstart = (long long)k + w - 1;
    if (g.L - (long long)k < (long long)w) {  // fewer than w pushed k-mers: no rescan ever happens
        g.jend = g.jstart - 1;
    } else {
        // branch 2 enabled for w+k <= pos < L-w+k (shmmrutils.rs:516-519)
        const long long lb = g.L - (long long)w + (long long)k;
        long long je = lb - 1;
        if (je > g.L - 1) je = g.L - 1;
        if (je < g.jstart) je = g.jstart;
        g.jend = je;
    }
    return g;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// level1_tile_kernel.  Written against the measured issue cost of every opcode it uses (profiles/r03_ubench/
// valu_cycles.txt): ~2.4 cycles per wave64 instruction for add / sub / and / or / xor / not / right
// shifts / mov, ~4.2 for everything else, more for a v_cndmask through vcc.  Hence:
//   * window minima / maxima are single v_min_f64 / v_max_f64: a 56-bit hash read as a non-negative (possibly denormal)
//     double orders like the integer;
//   * validity / core / window-range tests are 16-bit per-lane masks applied with v_bfe_i32;
//   * the canonical strand is ONE v_cmp_lt_u64 into an SGPR lane mask: selects read it, v_addc shifts it into a word.
// (The round-1 / round-2 instruction selections that used to live here behind PGR_TILE_V2 / _V3 / PGR_KEY_NOEXP / PGR_ABLATE
// are in the history: git show a409c78:pgr-tk_amd/csrc/level1_select.h.)
namespace {

__device__ __forceinline__ double dmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double dmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// bit u of bits -> 0 / 0xffffffff.  Inline asm on purpose: written with __builtin_amdgcn_sbfe the optimiser
// turns "value & mask" back into v_cmp + v_cndmask, the slow pattern this kernel avoids.
__device__ __forceinline__ uint32_t bit_to_mask(uint32_t bits, uint32_t u) {
    uint32_t r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(bits), "s"(u));
    return r;
}
// acc = 2*acc + (a < b): v_cmp_lt_f64 feeds the carry-in of v_addc_co_u32 (8.4 cycles for both)
__device__ __forceinline__ void shift_in_lt(uint32_t &acc, double a, double b) {
    asm("v_cmp_lt_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ double mk_double(uint32_t lo, uint32_t hi) {
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// bits u (0..15) with lo <= 16*t + u < hi  (all in tile-extended coordinates)
__device__ __forceinline__ uint32_t lane_range_mask(int t16, int lo, int hi) {
    int a = lo - t16, b = hi - t16;
    a = a < 0 ? 0 : (a > 16 ? 16 : a);
    b = b < 0 ? 0 : (b > 16 ? 16 : b);
    return (b > a) ? (((1u << b) - 1u) & ~((1u << a) - 1u)) : 0u;
}
template <int EXT>
__device__ __forceinline__ int clamp_rel(long long v) { return v < 0 ? 0 : (v > EXT ? EXT : (int)v); }

// low 64 bits of the 96-bit value (w2:w1:w0) >> S, S a compile-time constant in 0..63
template <int S>
__device__ __forceinline__ uint64_t shr96_lo64(uint32_t w2, uint32_t w1, uint32_t w0) {
    if (S == 0) return ((uint64_t)w1 << 32) | w0;
    if (S < 32) return ((uint64_t)funnel(w2, w1, S) << 32) | funnel(w1, w0, S);
    if (S == 32) return ((uint64_t)w2 << 32) | w1;
    return ((uint64_t)(w2 >> ((S - 32) & 31)) << 32) | funnel(w2, w1, (S - 32) & 31);
}

// (v >> sh) & mask with the shift as ONE v_lshrrev_b64 (opaque: left alone the compiler lowers about a third of
// them to v_alignbit_b32 + v_bfe_u32, 8.4 instead of 6.9 cycles)
__device__ __forceinline__ uint64_t shr_mask(uint64_t v, uint32_t sh, uint64_t mask) {
    if (sh != 0) {  // sh is a constant after unrolling; an SGPR operand keeps the asm generic
        uint64_t r;
        asm("v_lshrrev_b64 %0, %1, %2" : "=v"(r) : "s"(sh), "v"(v));
        return r & mask;
    }
    return (v >> sh) & mask;
}

// --- comparisons into SGPR pairs (wave64 lane masks) and selects from them.  Measured (profiles/r02_ubench): a
// v_cmp that writes an SGPR pair + v_cndmask_b32 reading it cost 4.1 + 3.5 cycles per wave64 instruction; the same
// through vcc stalls (5.7 per v_cndmask), and the sign-mask + v_bfi_b32 form costs 4.2 per select plus 3 instructions
// for the mask.
__device__ __forceinline__ uint64_t cmp_lt_u64(uint64_t a, uint64_t b) {  // lane mask of a < b
    uint64_t m;
    asm("v_cmp_lt_u64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ uint64_t cmp_eq_u64(uint64_t a, uint64_t b) {
    uint64_t m;
    asm("v_cmp_eq_u64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ uint64_t cmp_eq_u32(uint32_t a, uint32_t b) {
    uint64_t m;
    asm("v_cmp_eq_u32 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ uint32_t sel(uint64_t mask, uint32_t if_set, uint32_t if_clear) {
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}
// acc = 2 * acc + (lane's bit of mask)
__device__ __forceinline__ void shift_in_mask(uint32_t &acc, uint64_t mask) {
    uint64_t cy;
    asm("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(acc), "=s"(cy) : "s"(mask));
}

constexpr uint32_t KEY_INF = 0x7FE00000u;   // hi word of the "not a k-mer" sentinel (finite, above every key)
// what a window end outside [jstart, jend] contributes to the max pass: below every key
constexpr uint64_t NO_WINDOW = 0xFFF0000000000000ull /* -inf */;

}  // namespace

// Hash and select this lane's 16 positions.  MASKED = false: every position of the wave is a real k-mer
// and every window end is inside [jstart, jend] (interior of a contig) -> no masking instructions.
template <int TW, int TK, bool SKETCH, bool MASKED, int BLK>
__device__ __forceinline__ void tile_select(const L1Args &a, uint32_t w, uint32_t k, uint32_t t, long long q,
                                            long long wbase, const uint2 *s_words, double (*s_suf)[BLK],
                                            double *s_row, int *s_skip, uint32_t valid_mask, uint32_t mwin_mask,
                                            uint32_t core_mask, double (&x)[L1_G], uint32_t &strand_bits,
                                            uint32_t &emit, uint32_t *s_pal = nullptr) {
    // ---- per-lane 96-bit windows of both planes ending at this lane's last position
    const long long e = q + (L1_G - 1);
    const int jl = (int)((e >> 5) - wbase);
    const uint32_t s = 31u - (uint32_t)(e & 31);
    const uint2 W0 = s_words[jl], W1 = s_words[jl - 1], W2 = s_words[jl - 2], W3 = s_words[jl - 3];
    const uint32_t a0 = funnel(W1.x, W0.x, s), a1 = funnel(W2.x, W1.x, s), a2 = funnel(W3.x, W2.x, s);
    const uint32_t b0 = funnel(W1.y, W0.y, s), b1 = funnel(W2.y, W1.y, s), b2 = funnel(W3.y, W2.y, s);
    const uint64_t kmask = U64MAX >> (64 - k);
    const uint64_t sketch_thr = (U64MAX >> 4) >> a.r;  // shmmrutils.rs:621
    // Two 64-bit anchors per plane so that every position's k-mer plane is ONE 64-bit shift (by 0..7) + mask:
    //   FA = window ending at position q+15 (positions u = 8..15), FB = window ending at q+7 (u = 0..7).
    const uint64_t fa0 = ((uint64_t)a1 << 32) | a0, fb0 = ((uint64_t)funnel(a2, a1, 8) << 32) | funnel(a1, a0, 8);
    const uint64_t fa1 = ((uint64_t)b1 << 32) | b0, fb1 = ((uint64_t)funnel(b2, b1, 8) << 32) | funnel(b1, b0, 8);
    // bit-reversed complement windows: Rv bit i = ~base[q + 15 - 95 + i]; with a compile-time k the reverse-
    // complement planes (shmmrutils.rs:469-475) are r = (Rv >> (81 + u - k)) & kmask: anchors at shift
    // 81-k (u = 0..7) and 89-k (u = 8..15)
    const uint32_t ra0 = __brev(~a2), ra1 = __brev(~a1), ra2 = __brev(~a0);
    const uint32_t rb0 = __brev(~b2), rb1 = __brev(~b1), rb2 = __brev(~b0);
    constexpr int RS1 = 81 - (TK ? TK : 56), RS0 = RS1 + 8;
    const uint64_t rA0 = shr96_lo64<RS1>(ra2, ra1, ra0), rB0 = shr96_lo64<RS0>(ra2, ra1, ra0);
    const uint64_t rA1 = shr96_lo64<RS1>(rb2, rb1, rb0), rB1 = shr96_lo64<RS0>(rb2, rb1, rb0);

    // x[]: ordered keys: hash & (2^56-1) read as a double; sentinel for "no k-mer here"
    uint32_t strand_rev = 0;  // strand bits in reversed order (bit 15 - u)
    uint64_t eq0[L1_G];       // per position: lanes whose low planes are equal (f0 == r0), SGPR pairs
    uint64_t pal0_any = 0;
#pragma unroll
    for (int u = 0; u < L1_G; ++u) {
        const uint64_t f0 = shr_mask(u >= 8 ? fa0 : fb0, (uint32_t)((L1_G - 1 - u) & 7), kmask);
        const uint64_t f1 = shr_mask(u >= 8 ? fa1 : fb1, (uint32_t)((L1_G - 1 - u) & 7), kmask);
        uint64_t r0, r1;
        if (TK) {
            r0 = shr_mask(u >= 8 ? rB0 : rA0, (uint32_t)(u & 7), kmask);
            r1 = shr_mask(u >= 8 ? rB1 : rA1, (uint32_t)(u & 7), kmask);
        } else {
            r0 = rc_plane(f0, k);
            r1 = rc_plane(f1, k);
        }
        // canonical strand: reverse iff r0 < f0 (low plane only, shmmrutils.rs:485-488): one compare into an SGPR lane
        // mask, four selects from it, and the strand bit shifted into the per-lane word by an add-with-carry
        const uint64_t rev = cmp_lt_u64(r0, f0);
        // the canonical low plane is min(f0, r0) (reverse iff r0 < f0, and equal planes are the same either way): both are k <= 56
        // bit patterns, i.e. non-negative finite doubles ordered like the integers (denormals preserved) -- ONE v_min_f64
        // (4.2 cycles) instead of two selects (8.3)
        const uint64_t m0 = (uint64_t)__double_as_longlong(dmin(__longlong_as_double((long long)f0), __longlong_as_double((long long)r0)));
        const uint32_t m0l = (uint32_t)m0, m0h = (uint32_t)(m0 >> 32);
        const uint32_t m1l = sel(rev, (uint32_t)r1, (uint32_t)f1), m1h = sel(rev, (uint32_t)(r1 >> 32), (uint32_t)(f1 >> 32));
        uint32_t m1x = m1l ^ 0xAD12CF59u;
        asm("" : "+v"(m1x));  // keep the constant out of the hash's first step (the optimiser would distribute it)
        // h = A ^ B is only needed as the 56-bit key (the sketch threshold aside): low word one v_xor, high word
        // (Ahi ^ Bhi) & 0x00FFFFFF as ONE v_bitop3_b32 (3.65 cycles; v_xor + v_and: 4.9)
        const uint64_t hA = u64hash_mad(((uint64_t)m0h << 32) | m0l), hB = u64hash_mad(((uint64_t)m1h << 32) | m1x);
        const uint64_t h = SKETCH ? (hA ^ hB) : 0ull;
        uint32_t key_hi;
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x28" : "=v"(key_hi) : "v"((uint32_t)(hA >> 32)), "v"((uint32_t)(hB >> 32)), "s"(0x00FFFFFFu));
        shift_in_mask(strand_rev, rev);  // position u ends up at bit 15 - u
        const uint64_t key = ((uint64_t)key_hi << 32) | ((uint32_t)hA ^ (uint32_t)hB);
        uint64_t ok_mask = ~0ull;  // lanes whose position u holds a k-mer
        if (MASKED) {
            const uint32_t inval = bit_to_mask(~valid_mask, u);
            const uint64_t iv = (uint64_t)inval << 32;  // sentinel: only the high word decides
            x[u] = __longlong_as_double((long long)((key & ~iv) | (((uint64_t)KEY_INF << 32) & iv)));
            ok_mask = cmp_eq_u32(inval, 0u);
        } else {
            x[u] = __longlong_as_double((long long)key);
        }
        if (SKETCH) {
            // exact skip test (shmmrutils.rs:603-606) and the sketch threshold on the full 64-bit hash (:621)
            const bool skip = (f0 == r0) && (f1 == r1);
            if (!skip && h < sketch_thr) emit |= 1u << u;
        } else {
            // palindromic k-mer (fmmer == rmmer, shmmrutils.rs:477-480): the low planes are equal with probability
            // 2^-(k/2) per position on random sequence, so ONE compare into an SGPR mask per position is kept and the high
            // planes are only looked at in the (practically never taken) scalar branch behind the loop
            eq0[u] = cmp_eq_u64(f0, r0) & ok_mask;
            pal0_any |= eq0[u];
        }
    }
    strand_bits = __brev(strand_rev) >> 16;
    uint64_t pal_any = 0;
    if (!SKETCH && pal0_any) {  // wave-uniform: some lane has f0 == r0 somewhere; now the exact test on the high planes
        // (s_pal: the tile's first and last extended position with a palindromic k-mer -- scalar arithmetic on the lane masks, in
        // this cold block only; the islands of the exact machine end a tile early when the last one lies well inside its tile)
        uint32_t p_lo = 0xFFFFFFFFu, p_hi = 0u;
        const uint32_t wave_pos0 = (t & ~63u) * (uint32_t)L1_G;
#pragma unroll
        for (int u = 0; u < L1_G; ++u) {
            if (eq0[u] == 0) continue;
            asm volatile("s_nop 15");  // (marks the cold block for tools/isa_histogram.py: weight 0)
            const uint64_t f1 = shr_mask(u >= 8 ? fa1 : fb1, (uint32_t)((L1_G - 1 - u) & 7), kmask);
            const uint64_t r1 = TK ? shr_mask(u >= 8 ? rB1 : rA1, (uint32_t)(u & 7), kmask) : rc_plane(f1, k);
            const uint64_t pm = eq0[u] & cmp_eq_u64(f1, r1);
            pal_any |= pm;
            if (s_pal && pm) {
                p_lo = min(p_lo, wave_pos0 + (uint32_t)__builtin_ctzll(pm) * (uint32_t)L1_G + (uint32_t)u);
                p_hi = max(p_hi, wave_pos0 + (63u - (uint32_t)__builtin_clzll(pm)) * (uint32_t)L1_G + (uint32_t)u);
            }
        }
        if (s_pal && pal_any && (t & 63u) == 0) {
            atomicMin(&s_pal[0], p_lo);
            atomicMax(&s_pal[1], p_hi);
        }
    }
    const uint32_t pal_min = pal_any ? 0u : 1u;

    if (SKETCH) {
        emit &= valid_mask & core_mask;
    } else {
        if (pal_min == 0) *s_skip = 1;  // benign race: all writers store 1

        // ---- pass 1: M[j] = min(x[j-w+1 .. j])  (van Herk / Gil-Werman with 16-wide rows in registers)
        {
            double run = x[L1_G - 1];
            s_suf[L1_G - 1][t] = run;
#pragma unroll
            for (int u = L1_G - 2; u >= 0; --u) {
                run = dmin(run, x[u]);
                s_suf[u][t] = run;
            }
            s_row[t] = run;
        }
        __syncthreads();
        const int wm1 = (int)w - 1;
        const double big = mk_double(0u, KEY_INF);
        double M[L1_G];
        // w a multiple of 16 (the instantiated specs: 80, 48): the window of position u is the suffix of row t - w/16
        // from offset u + 1, then w/16 - 1 WHOLE rows, then this lane's prefix 0..u -- the same whole rows for every u,
        // so their minimum seeds the prefix chain and a window minimum is ONE v_min_f64 on top of the chain
        constexpr bool FOLD = TW != 0 && (TW % 16) == 0;
        if (FOLD) {
            constexpr int NB = (TW ? TW : 16) / 16 - 1;  // whole rows inside every window of the lane
            double pre = big;
#pragma unroll
            for (int i = 1; i <= NB; ++i) {
                const int ti = (int)t - i;
                pre = dmin(pre, s_row[ti < 0 ? 0 : ti]);
            }
            int ts = (int)t - NB - 1;
            ts = ts < 0 ? 0 : ts;
#pragma unroll
            for (int u = 0; u < L1_G; ++u) {
                pre = dmin(pre, x[u]);
                const double m = (u < L1_G - 1) ? dmin(pre, s_suf[u + 1][ts]) : pre;
                if (MASKED) {  // window ends outside [jstart, jend] do not select anything: M = +0 (below every key)
                    const uint64_t mb = (uint64_t)__double_as_longlong(m);
                    const uint64_t keep = (uint64_t)(int64_t)(int32_t)bit_to_mask(mwin_mask, u);
                    M[u] = __longlong_as_double((long long)((mb & keep) | (~keep & NO_WINDOW)));
                } else {
                    M[u] = m;
                }
            }
        } else
        {
            const int rs_lo = (-wm1) >> 4;  // floor(-(w-1)/16)
            const int nb = -rs_lo - 1;      // whole rows between the window start row and this row (u small)
            double acc = big, qlo = big;
            for (int i = 1; i <= nb; ++i) {
                if (i == nb) qlo = acc;
                const int ti = (int)t - i;
                acc = dmin(acc, s_row[ti < 0 ? 0 : ti]);
            }
            double pre = big;
#pragma unroll
            for (int u = 0; u < L1_G; ++u) {
                pre = dmin(pre, x[u]);
                const int d = u - wm1;
                const int rs = d >> 4;
                const int off = d & 15;
                int ts = (int)t + rs;
                ts = ts < 0 ? 0 : ts;
                double m = dmin(pre, s_suf[off][ts]);
                m = dmin(m, rs == rs_lo ? acc : qlo);
                // window ends outside [jstart, jend] do not select anything: M = +0 (below every key)
                if (MASKED) {
                    const uint64_t mb = (uint64_t)__double_as_longlong(m);
                    const uint64_t keep = (uint64_t)(int64_t)(int32_t)bit_to_mask(mwin_mask, u);
                    M[u] = __longlong_as_double((long long)((mb & keep) | (~keep & NO_WINDOW)));
                } else {
                    M[u] = m;
                }
            }
        }
        __syncthreads();
        // ---- pass 2: E[i] = max(M[i .. i+w-1]); i is selected iff x[i] == E[i]
        {
            double pm = M[0];
            s_suf[0][t] = pm;
#pragma unroll
            for (int u = 1; u < L1_G; ++u) {
                pm = dmax(pm, M[u]);
                s_suf[u][t] = pm;
            }
            s_row[t] = pm;
        }
        __syncthreads();
        constexpr bool FOLD2 = TW != 0 && (TW % 16) == 0;
        if (FOLD2) {
            // E[u] = max(M[u .. u + w - 1]): this lane's suffix from u, w/16 - 1 whole rows (they seed the suffix chain),
            // and the prefix of row t + w/16 through offset u - 1
            constexpr int NB = (TW ? TW : 16) / 16 - 1;
            double sm = __longlong_as_double((long long)NO_WINDOW);
#pragma unroll
            for (int i = 1; i <= NB; ++i) {
                const int ti = (int)t + i;
                sm = dmax(sm, s_row[ti > BLK - 1 ? BLK - 1 : ti]);
            }
            int te = (int)t + NB + 1;
            te = te > BLK - 1 ? BLK - 1 : te;
            uint32_t neq = 0;  // bit u set iff E[u] < x[u]  (E <= x always: every window minimum is <= x)
#pragma unroll
            for (int u = L1_G - 1; u >= 0; --u) {
                sm = dmax(sm, M[u]);
                const double ev = (u > 0) ? dmax(sm, s_suf[u - 1][te]) : sm;
                shift_in_lt(neq, ev, x[u]);  // u runs 15..0, so bit u ends up at position u
            }
            emit = ~neq & valid_mask & core_mask;
        } else {
            const int re_lo = wm1 >> 4;
            const double none = __longlong_as_double((long long)NO_WINDOW);
            double acc = none, qlo = none;
            for (int i = 1; i <= re_lo; ++i) {
                if (i == re_lo) qlo = acc;
                const int ti = (int)t + i;
                acc = dmax(acc, s_row[ti > BLK - 1 ? BLK - 1 : ti]);
            }
            double sm = none;
            uint32_t neq = 0;  // bit u set iff E[u] < x[u]  (E <= x always: every window minimum is <= x)
#pragma unroll
            for (int u = L1_G - 1; u >= 0; --u) {
                sm = dmax(sm, M[u]);
                const int d = u + wm1;
                const int re = d >> 4;
                const int off = d & 15;
                int te = (int)t + re;
                te = te > BLK - 1 ? BLK - 1 : te;
                double ev = dmax(sm, s_suf[off][te]);
                ev = dmax(ev, re == re_lo ? qlo : acc);
                shift_in_lt(neq, ev, x[u]);  // u runs 15..0, so bit u ends up at position u
            }
            emit = ~neq & valid_mask & core_mask;
        }
    }

}


// ------------------------------------------------------------------------------------------------
// k-mer planes at an arbitrary position straight from global memory (all k bases valid, p >= k-1)
__device__ __forceinline__ void kmer_at(const uint2 *__restrict__ planes, long long nwords, long long p, uint32_t k,
                                        uint64_t &f0, uint64_t &f1) {
    const long long j = p >> 5;
    const uint32_t s = 31u - (uint32_t)(p & 31);
    uint2 w0 = make_uint2(0, 0), w1 = w0, w2 = w0;
    if (j >= 0 && j < nwords) w0 = planes[j];
    if (j - 1 >= 0 && j - 1 < nwords) w1 = planes[j - 1];
    if (j - 2 >= 0 && j - 2 < nwords) w2 = planes[j - 2];
    const uint64_t kmask = U64MAX >> (64 - k);
    f0 = (((uint64_t)funnel(w2.x, w1.x, s) << 32) | funnel(w1.x, w0.x, s)) & kmask;
    f1 = (((uint64_t)funnel(w2.y, w1.y, s) << 32) | funnel(w1.y, w0.y, s)) & kmask;
}


}  // namespace pgr
