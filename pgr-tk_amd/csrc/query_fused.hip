// query_fused.hip -- query_fragment_to_hps for batches of SHORT queries: every stage behind the queries' shimmer-pair records in
// ONE kernel, one wavefront per query.
//
// The general path (index.hip) runs one kernel per stage over the whole batch: lookup, pair multiplicities, hit counts, scan,
// [host reads the number of hits], hit expansion, grouping sort, scan, group starts, chaining (three kernels), counts, two
// scans, [host reads the totals], packing, download -- ~25 launches and three host round trips behind the shimmers.  For the
// reference's own use (pgr-query: a gene-sized query against the index; BASELINE.json configs[2]: 10 000 x 10 kbp) a query
// has ~30 shimmer pairs and ~30 hits: every one of those kernels is a few microseconds of work and the batch costs the sum
// of their latencies.  Here a wavefront keeps ITS query's pairs, hits and chains in LDS and registers:
//
//   aln.rs:147-242 query_fragment_to_hps   lookup of every pair (seq_db.rs:1200-1228), per-query key multiplicities
//                                          (aln.rs:180-181), count filters (aln.rs:197-228), hits in query order
//   aln.rs:230-240                         grouping by target sid (stable: the hits of a group stay in query order)
//   aln.rs:12-142  sparse_aln              per group: lane j holds hit j; look-back of hit i = one step of all lanes
//                                          (filters -> ballot, "stop after max_span distinct query intervals" -> popcounts of
//                                          the ballot, best predecessor -> wave maximum), chain extraction by readlane walks
//
// and leaves chains in a fixed-size slot of its query.  Two small kernels turn the slots into the flat result (exclusive scans
// of the per-query counts; packing), which the DMA engine downloads into a pinned host block.  The host looks at the totals
// ONCE, after everything -- and when the index has seen a batch before, not even the shimmer pipeline in front of this stage
// waits for the host (QueryFusedRun::enqueue_from_shimmers is called through pgr_ctx::post_enqueue).
//
// f32 arithmetic in the reference's operation order, no contraction (-ffp-contract=off), the same tie rules as
// sparse_aln_kernel / sparse_aln_wave_kernel (index.hip): tests/test_gpu_02_query_fused.py compares the two paths with each other
// and with the CPU restatement of the reference.
//
// The path DECLINES a batch it cannot hold (the flag is read with the totals): a query with more than P <= QF_MAX_PAIRS pairs or
// QF_H_MAX hits, a key with more than HITS_HEAVY records, a (query, target) group of more than 64 hits in an LDS image without
// the long-group arrays.  The caller then takes the general path and the index remembers the refusal for its next calls; a
// query that only needs a larger slot (H) makes the stage run once more with it.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pgr_aln.h"
#include "pgr_device.h"
#include "pgr_host.h"
#include "pgr_index.h"
#include "pgr_internal.h"

#ifndef PGR_QF_ABLATE
#define PGR_QF_ABLATE 0  // timing experiments only (tools/build_variant.sh): 1 no chaining DP, 2 nothing behind the level-1 list, 4 nothing behind the pairs,
                         // 8 ... the lookups, 16 ... the count filters, 32 ... the hits, 64 ... the sort by target; 128 the DP's look-back loop is
                         // empty, 256 no chain is extracted
#endif

namespace pgr {

namespace {

// P = pairs a query may have (64 .. QF_MAX_PAIRS), H = hits a query may have = entries of its slot (64 .. QF_H_MAX): powers of
// two chosen per call from the longest query of the batch and the index's records per key; the kernel's LDS image is sized by
// them (44 B per pair + 36 B per hit, + 12 B per hit and 1.8 KB for the look-back of groups of more than 64 hits, which
// batches with queries of more than 64 pairs can have)
constexpr uint32_t QF_P_MIN = 64, QF_H_MIN = 64, QF_H_MAX = 512;
constexpr uint32_t QF_DECLINE = 1u, QF_MORE_HITS = 2u;  // flags[0]: the batch does not fit at all / fits with a larger H
constexpr uint32_t QF_LOOKBACK_TIMEOUT = 4u;            // (with QF_DECLINE) the single-pass form gave up waiting for a predecessor
// the level-1 form: (with QF_DECLINE) the level-1 kernels asked for islands / overflowed (the batch is for the shimmer pipeline);
// a query has more pairs than P / more level-1 minimizers than C1 (flags[3] / flags[4] say how many: once more with room)
constexpr uint32_t QF_L1_FLAGGED = 8u, QF_MORE_PAIRS = 16u, QF_MORE_L1 = 32u;

struct QfArgs {
    const pgr_frag_rec *qrec;
    const uint64_t *pair_off;
    uint32_t n_queries;
    const pgr_frag_rec *recs;
    const uint64_t *key_off;
    uint64_t n_keys;
    const uint32_t *lut;
    uint32_t lut_bits, lut_shift;
    const ulonglong2 *keys;
    const ulonglong4 *qkeys;  // pgr_index.h: the key with its record when it has exactly one (nullptr: none)
    QParams qp;
    AlnParams ap;
    uint32_t P, H;  // pairs / hits a query may have (H = entries of a slot)
    uint32_t long_groups;  // the LDS image has the work arrays of groups of more than 64 hits
    // per-query slots of H entries: chains in their final order
    pgr_hitpair *s_hp;   // hit pairs of the chains
    float *s_cscore;     // score per chain
    uint32_t *s_choff;   // first hit pair of the chain, relative to the slot
    uint32_t *s_tsid;    // target sid per kept (>= 2 hits) group
    uint32_t *s_tcoff;   // first chain of the target, relative to the slot
    // per-query counts: targets, chains, hit pairs of chains, hits after the filters, looked-up signatures
    uint32_t *q_nt, *q_nc, *q_nh, *q_nhit;
    unsigned long long *q_nsig;
    uint32_t *flags;  // [0] QF_DECLINE | QF_MORE_HITS, [1] groups the reference never finishes, [2] most hits of one query
    // the single-pass form (query_fused_kernel<true>): every query's wavefront finds where its targets / chains / hit pairs start
    // by looking back over its predecessors' descriptors and writes them straight into the host's pinned block
    uint64_t *desc;        // 3 words per query, 3 per tile of 64 queries (status in the top two bits), a counter per tile: zero before the launch
    uint8_t *host;         // the pinned block, laid out by qf_layout(n_queries, cap_t, cap_c, cap_h)
    uint64_t cap_t, cap_c, cap_h;
    uint64_t *words;       // the pinned mailbox (the totals, written by the last query's wavefront)
    // the level-1 form (qf_one_query<true>): the query's level-1 minimizers come straight from the tile kernel's segments -- contig q
    // owns the segments [tile_first[q] + q, tile_first[q + 1] + q] (its tiles, then its tail) -- and the wavefront runs the list
    // stage itself (both reductions, min_span, the pairs) on them in LDS
    const L1Rec *l1;
    const uint64_t *seg_off;
    const uint32_t *seg_cnt;
    const uint32_t *tile_first;
    const unsigned long long *l1_status;  // the level-1 kernels' cursor words: [0] overflow elements taken, [1] overflow region too small, [2] islands needed
    uint64_t l1_ovf_cap;
    uint32_t C1;  // level-1 minimizers a query may have
    uint32_t r, min_span;
};

// dynamic LDS: h0[P] h1[P] lo[P] (u64) | nrec[P] pc[P] hoff[P] sid[P] fl[P] (u32) [ | bgn[P] end_or[P] (u32): level-1 form ]
//              | hit[H] (24 B) | hsid[H] ssid[H] perm[H] (u32) [ | vs[H] (f32) sl[H] pv[H] (i32) | span_q[64][3] cand[4][64] (u32) ]
// level-1 form (C1 > 0): the level-1 list -- key1[C1] (u64) ypos1[C1] (u32), + the segment table sbase[65] soff[64] -- shares its
// bytes with the hits (the pairs are formed before the first hit is written), its index lists idx_a[C1] idx_b[C1] (u16) with the pairs
constexpr uint32_t QF_L1_SEGS = 64;  // segments (tiles + tail) of one query the level-1 form can take
inline __host__ __device__ size_t qf_l1_bytes(uint32_t C1) { return C1 ? (size_t)C1 * 12 + (QF_L1_SEGS + 2) * 4 + QF_L1_SEGS * 8 + 8 : 0; }
inline size_t qf_lds_bytes(uint32_t P, uint32_t H, bool long_groups, uint32_t C1 = 0) {
    const size_t hits = (size_t)H * (sizeof(pgr_hitpair) + 12) + (long_groups ? (size_t)H * 12 + (64 * 3 + 4 * 64) * 4 : 0);
    return (size_t)P * (C1 ? 52 : 44) + std::max(hits, qf_l1_bytes(C1));
}
// (the level-1 form keeps its two index lists, 4 B per level-1 minimizer, in the pairs' 52 P bytes: P at least C1 / 13)

// sparse_aln for one group of n <= 64 hits (hit[perm[0..n)], ascending query bgn), the whole wavefront, hit j in lane j.
// Appends the group's chains to the slot: hit pairs at o_hp[nh..], scores / first-hit offsets at [nc..]; returns true when
// the reference would never finish the group (aln.rs:129-131).
__device__ __forceinline__ bool chain_group_regs(const pgr_hitpair *hit, const uint32_t *perm, const int n, const AlnParams &prm,
                                                 const int lane,
                                                 pgr_hitpair *__restrict__ o_hp, float *__restrict__ o_cscore,
                                                 uint32_t *__restrict__ o_choff, uint32_t &nc, uint32_t &nh) {
    const bool in = lane < n;
    auto hs = [&](int x) -> pgr_hitpair { return hit[perm[x]]; };
    const pgr_hitpair h = hs(in ? lane : 0);
    // value slot (v_s / best_pre_v are keyed by the HitPair VALUE, aln.rs:24): the earliest identical hit pair -- identical
    // pairs share their query bgn, so they sit in the run of equal bgn around the lane.  eqm: the LATER hits with this lane's
    // query interval (qb, qe, qo), what the span set of a look-back (aln.rs:70) sees before it reaches this lane.
    int sl = lane;
    uint64_t eqm = 0;
    for (int d = 1;; ++d) {
        const int o = lane - d;
        const pgr_hitpair x = hs(o >= 0 ? o : 0);
        const bool v = in && o >= 0 && x.qb == h.qb;
        if (!__ballot(v)) break;
        if (v && same_hp(x, h)) sl = o;
    }
    for (int d = 1;; ++d) {
        const int o = lane + d;
        const pgr_hitpair x = hs(o < n ? o : 0);
        const bool v = in && o < n && x.qb == h.qb;
        if (!__ballot(v)) break;
        if (v && same_q(x, h)) eqm |= 1ull << o;
    }
    const bool has_dup = __ballot(in && sl != lane) != 0;
    const bool any_eq = __ballot(in && eqm != 0) != 0;  // some query interval occurs more than once in the group
    const uint64_t above = lane == 63 ? 0ull : (U64MAX << (lane + 1));
    // The look-back of hit i is ONE step of all lanes, and the kernel is bound by instruction issue (profiles/r06_query: 5 700
    // instructions per query, more than half of them in this loop): whatever does not depend on i is computed once per lane -- the
    // conversions of aln.rs:31-33 / :52-84 ((float)qb, qe, tb, te of a hit, its own length: the same f32 operations on the same
    // values, whoever performs them) --, what is the same for all lanes stays in scalar registers (the candidates of hit i as a
    // 64-bit mask: the lanes below i, minus the filters of aln.rs:43-67 where a filter is on at all), and the span set's rule
    // (aln.rs:70, :91) is a mask operation whenever the candidates are dense.
    const float f_qb = (float)h.qb, f_qe = (float)h.qe, f_tb = (float)h.tb, f_te = (float)h.te;
    const float len_f = f_qe - f_qb;
    const int oo = (int)(h.qo | (h.to << 1));  // (orients are 0 / 1)
    const uint32_t eq_lo = (uint32_t)eqm, eq_hi = (uint32_t)(eqm >> 32);
    auto bcast = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    float vs = 0.0f;
    int pv = -1;
    if (lane == 0) vs = len_f;  // aln.rs:25-27 (sl[0] == 0)
    uint64_t lt = 1ull;         // the lanes below i
    // the usual group -- no orientation or gap filter, no query interval twice (hence no hit pair twice) --: every lane below i is a
    // candidate and opens an interval, the scored ones are the max_span nearest; nothing of that depends on the lanes' data
    const bool plain = !prm.oriented && !prm.has_max_gap && !any_eq && !has_dup;
    if (plain) {
        for (int i = 1; i < ((PGR_QF_ABLATE & 128) ? 0 : n); ++i, lt = (lt << 1) | 1ull) {
            const float c_fqb = bcast(f_qb, i), cur_len = bcast(len_f, i), c_ftb = bcast(f_tb, i), c_fte = bcast(f_te, i);
            const int coo = __builtin_amdgcn_readlane(oo, i);
            const bool c_same = ((coo ^ (coo >> 1)) & 1) == 0;
            const float a = __builtin_fabsf(c_fqb - f_qe);
            const float b = __builtin_fabsf(c_same ? (c_ftb - f_te) : (c_fte - f_tb));
            const uint64_t pm = (uint32_t)i <= prm.max_span ? lt : lt & ~(lt >> prm.max_span);
            float sc = vs + cur_len;           // :71-72
            const float sum = a + b;           // :74-84
            const float pen = prm.penalty * sum;
            sc = sc - pen;
            if (!__builtin_amdgcn_inverse_ballot_w64(pm)) sc = -INFINITY;
            const uint32_t mb = wave_max_pos_bits(sc);
            float nv = cur_len;  // :96-102
            int np = -1;
            if (mb) {  // :86-89
                nv = __int_as_float((int)mb);
                np = 63 - __builtin_clzll(__ballot((uint32_t)__float_as_int(sc) == mb));
            }
            if (lane == i) {
                vs = nv;
                pv = np;
            }
        }
    }
    for (int i = 1; i < ((PGR_QF_ABLATE & 128) || plain ? 0 : n); ++i, lt = (lt << 1) | 1ull) {  // aln.rs:29-103
        const float c_fqb = bcast(f_qb, i), cur_len = bcast(len_f, i), c_ftb = bcast(f_tb, i), c_fte = bcast(f_te, i);
        const int coo = __builtin_amdgcn_readlane(oo, i);
        const bool c_same = ((coo ^ (coo >> 1)) & 1) == 0;  // cqo == cto
        uint64_t cm = lt;
        if (prm.oriented) cm &= __ballot((((oo ^ (oo >> 1)) & 1) == 0) == c_same);  // :43-50
        float a = c_fqb - f_qe;
        float b = c_same ? (c_ftb - f_te) : (c_fte - f_tb);
        a = __builtin_fabsf(a);  // (a source modifier; the select of pgr_aln.h: absf differs for -0 only, which no comparison and no sum below can tell from +0)
        b = __builtin_fabsf(b);
        if (prm.has_max_gap) {  // :52-65
            const float mg = (float)prm.max_gap;
            cm &= ~__ballot(a > mg || b > mg);
        }
        if (any_eq) cm &= ~__ballot((((i < 32 ? eq_lo : eq_hi) >> (i & 31)) & 1u) != 0u);  // :67 (bit i of eqm: hit i has this lane's interval)
        float best_s = 0.0f;
        int best_v = -1;
        if (cm) {
            // candidates are taken from lane i-1 downwards.  A candidate opens a new query interval of the span set (:70)
            // unless a considered candidate above it has the same interval; the look-back stops behind the candidate that
            // completes max_span intervals (:91), i.e. a candidate is scored iff fewer than max_span intervals were opened
            // above it.
            const bool cons = __builtin_amdgcn_inverse_ballot_w64(cm);
            const uint64_t fm = any_eq ? __ballot(cons && (eqm & cm) == 0) : cm;
            const uint32_t n_fresh = (uint32_t)__popcll(fm);
            uint64_t pm;
            if (any_eq ? n_fresh < prm.max_span : n_fresh <= prm.max_span) pm = cm;  // the span set never fills: every candidate is scored
            else if (fm == lt) pm = lt & ~(lt >> prm.max_span);                          // all lanes below i, all fresh: the max_span nearest
            else pm = __ballot(cons && (uint32_t)__popcll(fm & above) < prm.max_span);
            const bool proc = __builtin_amdgcn_inverse_ballot_w64(pm);
            const float p_s = has_dup ? __shfl(vs, sl, 64) : vs;  // :71
            float s = p_s + cur_len;                              // :72
            const float sum = a + b;                              // :74-84
            const float pen = prm.penalty * sum;
            s = s - pen;
            if (!proc) s = -INFINITY;
            const uint32_t mb = wave_max_pos_bits(s);
            if (mb) {  // :86-89: strict > from 0, the nearest candidate among equal maxima
                const uint64_t mm = __ballot((uint32_t)__float_as_int(s) == mb);  // (a lane that is not scored holds -inf)
                const int top = 63 - __builtin_clzll(mm);
                best_s = __int_as_float((int)mb);
                best_v = has_dup ? __builtin_amdgcn_readlane(sl, top) : top;
            }
        }
        const int si = has_dup ? __builtin_amdgcn_readlane(sl, i) : i;
        if (lane == si) {  // :96-102
            vs = best_s > 0.0f ? best_s : cur_len;
            pv = best_s > 0.0f ? best_v : -1;
        }
    }
    // chain extraction (aln.rs:105-140): best unvisited value slot, walk its predecessors until a visited one
    uint64_t unv = __ballot(in && sl == lane);
    bool stuck = false;
    if (PGR_QF_ABLATE & 256) unv = 0;
    while (unv) {
        const bool cand = __builtin_amdgcn_inverse_ballot_w64(unv);
        const uint32_t mb = wave_max_pos_bits(cand ? vs : -INFINITY);
        if (!mb) {  // only non-positive scores left: aln.rs:129-131 would spin forever
            stuck = true;
            break;
        }
        const float m = __int_as_float((int)mb);
        const int bv = __builtin_ctzll(__ballot(cand && (uint32_t)__float_as_int(vs) == mb));  // strict >: the lowest sorted index among equal maxima
        int len = 0, first_v = bv, v = bv, ord = -1;
        while (v >= 0 && ((unv >> v) & 1ull)) {  // :121-128
            if (lane == v) ord = len;
            ++len;
            first_v = v;
            unv &= ~(1ull << v);  // :133-137
            v = __builtin_amdgcn_readlane(pv, v);
        }
        if (!(PGR_QF_ABLATE & 512) && ord >= 0) o_hp[nh + (uint32_t)(len - 1 - ord)] = h;  // :132 reversed
        const float first_s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vs), first_v));
        if (!(PGR_QF_ABLATE & 1024) && lane == 0) {
            o_cscore[nc] = m - first_s;  // :138-139
            o_choff[nc] = nh;
        }
        ++nc;
        nh += (uint32_t)len;
    }
    return stuck;
}

// sparse_aln for one group of more than 64 hits with its work arrays in LDS: the scheme of sparse_aln_wave_kernel (index.hip)
// -- the look-back of a hit evaluates 64 candidates per step, the span set (aln.rs:70, :91) is carried between steps in
// span_q -- on hits reached through the permutation.  Same outputs as chain_group_regs.
__device__ __forceinline__ bool chain_group_lds(const pgr_hitpair *hit, const uint32_t *perm, const int n, const AlnParams &prm,
                                                const int lane, float *vs, int *sl, int *pv, uint32_t (*span_q)[3],
                                                uint32_t (*cand)[64], pgr_hitpair *__restrict__ o_hp,
                                                float *__restrict__ o_cscore, uint32_t *__restrict__ o_choff, uint32_t &nc,
                                                uint32_t &nh) {
    auto hs = [&](int x) -> pgr_hitpair { return hit[perm[x]]; };
    for (int i = lane; i < n; i += 64) {  // value slots: the earliest identical hit pair of the run of equal query bgn
        int s = i;
        const pgr_hitpair hi = hs(i);
        for (int j = i - 1; j >= 0; --j) {
            const pgr_hitpair hj = hs(j);
            if (hj.qb != hi.qb) break;
            if (same_hp(hj, hi)) s = j;
        }
        sl[i] = s;
    }
    wave_sync();
    if (lane == 0) {  // aln.rs:25-27
        const pgr_hitpair h0 = hs(0);
        vs[sl[0]] = (float)h0.qe - (float)h0.qb;
        pv[sl[0]] = -1;
    }
    wave_sync();
    for (int i = 1; i < n; ++i) {  // aln.rs:29-103
        const pgr_hitpair cur = hs(i);
        const float cur_len = (float)cur.qe - (float)cur.qb;
        float best_s = 0.0f;
        int best_v = -1;
        uint32_t span_n = 0;
        bool stop = false;
        for (int jb = i - 1; jb >= 0 && !stop; jb -= 64) {
            const int j = jb - lane;
            bool cons = j >= 0;
            const pgr_hitpair p = cons ? hs(j) : cur;
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
            // span set in candidate order, lane parallel: a candidate opens a new distinct query interval unless an earlier
            // step (span_q) or an earlier considered lane of this step has the same interval (equal intervals share their
            // qb, and the candidates are sorted by qb: the lanes to look at are the run of equal qb right before this lane)
            cand[0][lane] = p.qb;
            cand[1][lane] = p.qe;
            cand[2][lane] = p.qo;
            cand[3][lane] = cons ? 1u : 0u;
            wave_sync();
            bool fresh = cons;
            for (uint32_t t = 0; t < span_n; ++t)
                if (p.qb == span_q[t][0] && p.qe == span_q[t][1] && p.qo == span_q[t][2]) fresh = false;
            if (fresh)
                for (int l = lane - 1; l >= 0 && cand[0][l] == p.qb; --l)
                    if (cand[3][l] && cand[1][l] == p.qe && cand[2][l] == p.qo) {
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
                span_q[span_n + before][0] = p.qb;
                span_q[span_n + before][1] = p.qe;
                span_q[span_n + before][2] = p.qo;
            }
            span_n += (uint32_t)__popcll(stop_lane < 63 ? nm & (U64MAX >> (63 - stop_lane)) : nm);
            wave_sync();  // span_q is read back in the next step
            const bool proc = cons && lane <= stop_lane;
            float s = -INFINITY;
            int slj = 0;
            if (proc) {
                slj = sl[j];
                const float p_s = vs[slj];  // :71
                s = p_s + cur_len;          // :72
                const float sum = a + b;    // :74-84
                const float pen = prm.penalty * sum;
                s = s - pen;
            }
            const float m = wave_max_f32(s);  // :86-89: strict >, the nearest candidate among equal maxima
            if (m > best_s) {
                const int l = __builtin_ctzll(__ballot(proc && s == m));
                best_s = m;
                best_v = __builtin_amdgcn_readlane(slj, l);
            }
            if (stop_lane < 64) stop = true;
        }
        if (lane == 0) {  // :96-102
            const int si = sl[i];
            vs[si] = best_s > 0.0f ? best_s : cur_len;
            pv[si] = best_s > 0.0f ? best_v : -1;
        }
        wave_sync();
    }
    // extraction (aln.rs:105-140); a visited value slot is marked by sl[v] = -1 - v
    int n_unvisited = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        n_unvisited += (int)__popcll(__ballot(i < n && sl[i] == i));
    }
    while (n_unvisited > 0) {
        float bs = 0.0f;
        int bv = -1;
        for (int i = lane; i < n; i += 64)
            if (sl[i] == i && vs[i] > bs) {  // strict >: the lowest index of this lane's maxima
                bs = vs[i];
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
        if (bv < 0) return true;  // aln.rs:129-131 would spin forever (only non-positive scores left)
        int len = 0, first_v = bv;
        if (lane == 0) {
            int v = bv;
            while (v >= 0 && sl[v] == v) {  // :121-128
                o_hp[nh + len] = hs(v);
                ++len;
                first_v = v;
                const int nv = pv[v];
                sl[v] = -1 - v;  // :133-137
                v = nv;
            }
        }
        len = __builtin_amdgcn_readfirstlane(len);
        first_v = __builtin_amdgcn_readfirstlane(first_v);
        wave_sync();
        for (int x = lane; x < len / 2; x += 64) {  // :132 reverse
            const pgr_hitpair t = o_hp[nh + x];
            o_hp[nh + x] = o_hp[nh + len - 1 - x];
            o_hp[nh + len - 1 - x] = t;
        }
        if (lane == 0) {
            o_cscore[nc] = bs - vs[first_v];  // :138-139
            o_choff[nc] = nh;
        }
        ++nc;
        nh += (uint32_t)len;
        n_unvisited -= len;
        wave_sync();
    }
    return false;
}

// lookup_range with the bucket's keys loaded at once when they are few (independent loads: one memory round trip instead of
// one per step of the binary search)
__device__ __forceinline__ void lookup_range_short(uint64_t h0, uint64_t h1, const QfArgs &a, uint64_t &lo_out, uint64_t &hi_out) {
    if (a.lut && a.keys) {
        const uint64_t top = (1ull << a.lut_bits) - 1;
        const uint64_t bk = (h0 >> a.lut_shift) < top ? (h0 >> a.lut_shift) : top;
        const uint64_t lo = a.lut[bk], hi = a.lut[bk + 1];
        if (hi - lo <= 8) {
            uint64_t found = U64MAX;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint64_t k = lo + (uint64_t)j;
                if (k < hi) {
                    const ulonglong2 kk = a.keys[k];
                    if (kk.x == h0 && kk.y == h1) found = k;
                }
            }
            lo_out = hi_out = 0;
            if (found != U64MAX) {
                lo_out = a.key_off[found];
                hi_out = a.key_off[found + 1];
            }
            return;
        }
    }
    lookup_range(h0, h1, a.recs, a.key_off, a.n_keys, a.lut, a.lut_bits, a.lut_shift, a.keys, lo_out, hi_out);
}

// ---- the list stage of ONE query on one wavefront (the level-1 form).  reduce_shmmr without padding (shmmrutils.rs:359-415): an
// element survives iff it is an arg-min of some full r-window of its list = at least r consecutive elements including itself are
// >= it (small.hip has the 256-lane form of the same predicate).  list: indices into key (nullptr: the identity).
template <int TR>
__device__ __forceinline__ bool qf_reduce_keep(const uint64_t *key, const uint16_t *list, int n, int k, uint32_t r_rt) {
    const uint32_t r = TR ? (uint32_t)TR : r_rt;
    const uint64_t xi = key[list ? list[k] : k];
    uint32_t run = 1;
    bool left = true, right = true;
#pragma unroll
    for (uint32_t d = 1; d < (TR ? (uint32_t)TR : 12u); ++d) {
        if (!TR && d >= r) break;
        const int kl = k - (int)d, kr = k + (int)d;
        const bool in_l = kl >= 0, in_r = kr < n;
        const uint64_t xl = key[in_l ? (list ? list[kl] : kl) : 0], xr = key[in_r ? (list ? list[kr] : kr) : 0];
        left = left && in_l && xl >= xi;
        right = right && in_r && xr >= xi;
        run += (left ? 1u : 0u) + (right ? 1u : 0u);
    }
    return run >= r;
}
// ordered compaction by one wavefront: out[j] = src(k) for the k in [0, n) with pred(k), in order; returns how many
template <class Pred, class Src>
__device__ __forceinline__ uint32_t qf_compact(uint32_t n, int lane, uint16_t *out, Pred pred, Src src) {
    const uint64_t lt = lane ? (U64MAX >> (64 - lane)) : 0ull;
    uint32_t total = 0;
    for (uint32_t b = 0; b < n; b += 64) {  // (uniform)
        const uint32_t k = b + (uint32_t)lane;
        const bool keep = k < n && pred(k);
        const uint64_t bal = __ballot(keep);
        if (keep) out[total + (uint32_t)__popcll(bal & lt)] = src(k);
        total += (uint32_t)__popcll(bal);
    }
    return total;
}

// The same through the per-query kernel's own table (pgr_index.h: qkeys): the bucket's entries' keys at once, then -- the same
// cache line -- the second half of the entry that matched: a key with ONE record is answered here (single: tbte = bgn | end << 32,
// sid, to), any other key by key_off as before.
__device__ __forceinline__ void lookup_qkey(uint64_t h0, uint64_t h1, const QfArgs &a, uint64_t &lo_out, uint32_t &nrec, bool &single,
                                            uint64_t &tbte, uint32_t &sid, uint32_t &to) {
    constexpr uint64_t KM = ~(1ull << 63);
    const uint64_t top = (1ull << a.lut_bits) - 1;
    const uint64_t bk = (h0 >> a.lut_shift) < top ? (h0 >> a.lut_shift) : top;
    const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(a.qkeys);  // entry k: k2[2 k] = (x, y), k2[2 k + 1] = (z, w)
    uint64_t lo = a.lut[bk], hi = a.lut[bk + 1];
    uint64_t found = U64MAX, fx = 0, fy = 0;
    if (hi - lo <= 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t k = lo + (uint64_t)j;
            if (k < hi) {
                const ulonglong2 kk = k2[2 * k];
                if ((kk.x & KM) == h0 && (kk.y & KM) == h1) {
                    found = k;
                    fx = kk.x;
                    fy = kk.y;
                }
            }
        }
    } else {
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            const ulonglong2 kk = k2[2 * mid];
            const uint64_t kx = kk.x & KM, ky = kk.y & KM;
            if (kx < h0 || (kx == h0 && ky < h1)) lo = mid + 1;
            else hi = mid;
        }
        if (lo < a.n_keys) {
            const ulonglong2 kk = k2[2 * lo];
            if ((kk.x & KM) == h0 && (kk.y & KM) == h1) {
                found = lo;
                fx = kk.x;
                fy = kk.y;
            }
        }
    }
    lo_out = 0;
    nrec = 0;
    single = false;
    tbte = 0;
    sid = to = 0;
    if (found == U64MAX) return;
    if (fx >> 63) {
        const ulonglong2 zw = k2[2 * found + 1];
        single = true;
        nrec = 1;
        sid = (uint32_t)zw.x;
        tbte = (zw.x >> 32) | (zw.y << 32);
        to = (uint32_t)(fy >> 63);
    } else {
        lo_out = a.key_off[found];
        nrec = (uint32_t)(a.key_off[found + 1] - lo_out);  // (one key holds fewer than 2^32 records)
    }
}

// One query, the whole wavefront: chains into the query's slot; the counts come back wave-uniform (all zero for a query the
// path cannot hold: the batch's flags say why).  L1: the level-1 form (QfArgs).
template <bool L1>
__device__ __forceinline__ void qf_one_query(const QfArgs &a, uint8_t *qf_dyn, const uint32_t q, const int lane, uint32_t &nt,
                                             uint32_t &nc, uint32_t &nh, uint32_t &n_hits_out, unsigned long long &nsig,
                                             uint32_t &n_pairs_out) {
    const uint32_t P = a.P, H = a.H;
    n_pairs_out = 0;
    uint64_t *L_h0 = reinterpret_cast<uint64_t *>(qf_dyn), *L_h1 = L_h0 + P, *L_lo = L_h1 + P;
    uint32_t *L_nrec = reinterpret_cast<uint32_t *>(L_lo + P), *L_pc = L_nrec + P, *L_hoff = L_pc + P;
    uint32_t *L_sid = L_hoff + P, *L_fl = L_sid + P;  // a pair whose key has ONE record (qkeys): its sid; bit 0 single, bit 1 the record's orient
    uint32_t *L_bgn = L_fl + P, *L_eo = L_bgn + P;    // (level-1 form only: begin, end | orient << 31 of the pair)
    pgr_hitpair *hit = reinterpret_cast<pgr_hitpair *>(L_fl + (L1 ? 3 * P : P));
    uint32_t *hsid = reinterpret_cast<uint32_t *>(hit + H), *ssid = hsid + H, *perm = ssid + H;
    uint32_t m = 0;
    nt = nc = nh = n_hits_out = 0;
    nsig = 0;
    bool decline = false;
    int np = 0;
    uint64_t p0 = 0;
    if (L1) {
        // ---- the level-1 kernels' own verdict first: a batch that needs islands or overflowed belongs to the shimmer pipeline
        // (looked at below, when the segment table's loads are on their way: one trip to memory less in front of everything)
        const unsigned long long s0 = a.l1_status[0], s1 = a.l1_status[1], s2 = a.l1_status[2];
        const uint32_t C1 = a.C1;
        uint64_t *key1 = reinterpret_cast<uint64_t *>(hit);
        uint32_t *ypos1 = reinterpret_cast<uint32_t *>(key1 + C1);
        uint32_t *sbase = ypos1 + C1;                                           // [QF_L1_SEGS + 1] first list place of a segment
        uint64_t *soff = reinterpret_cast<uint64_t *>(sbase + QF_L1_SEGS + 2);  // [QF_L1_SEGS] its first record
        // (the two index lists live where the pairs will be: nothing of a pair is written before the final list has been read)
        uint16_t *idx_a = reinterpret_cast<uint16_t *>(qf_dyn), *idx_b = idx_a + C1;
        // ---- the query's segments: its tiles in order, then its tail
        const uint32_t tf0 = a.tile_first[q], tf1 = a.tile_first[q + 1];
        const uint32_t nseg = tf1 - tf0 + 1, seg0 = tf0 + q;
        if (nseg > QF_L1_SEGS) {
            if (lane == 0) {
                const uint32_t was = atomicOr(a.flags, QF_DECLINE);
                asm volatile("" ::"v"(was));
            }
            return;
        }
        const bool sl = (uint32_t)lane < nseg;
        const uint32_t scnt = sl ? a.seg_cnt[seg0 + lane] : 0u;
        const uint64_t sof = sl ? a.seg_off[seg0 + lane] : 0ull;
        if (s1 || s2 || s0 > a.l1_ovf_cap) {
            if (q == 0 && lane == 0) {
                const uint32_t was = atomicOr(a.flags, QF_DECLINE | QF_L1_FLAGGED);
                asm volatile("" ::"v"(was));
            }
            return;
        }
        const uint32_t sincl = wave_incl_sum(scnt);
        const uint32_t n1 = (uint32_t)__builtin_amdgcn_readlane((int)sincl, 63);
        if (n1 > C1) {  // (low-complexity sequence: denser than the estimate)
            if (lane == 0) {
                const uint32_t was = atomicMax(a.flags + 4, n1) | atomicOr(a.flags, QF_MORE_L1);
                asm volatile("" ::"v"(was));
            }
            return;
        }
        if (sl) {
            sbase[lane] = sincl - scnt;
            soff[lane] = sof;
        }
        if (lane == 0) sbase[nseg] = n1;
        wave_sync();
        // (six records per lane in flight at once: one trip to memory for a 10 kbp query's ~250 instead of one per 64)
        constexpr int QF_L1_U = 6;
        for (uint32_t e0 = 0; e0 < n1; e0 += 64 * QF_L1_U) {  // (uniform)
            L1Rec rec[QF_L1_U];
#pragma unroll
            for (int u = 0; u < QF_L1_U; ++u) {
                const uint32_t e = e0 + 64 * (uint32_t)u + (uint32_t)lane;
                rec[u].key_lo = rec[u].key_hi = rec[u].ypos = 0;
                if (e < n1) {
                    uint32_t j = 0;
                    while (sbase[j + 1] <= e) ++j;  // (a handful of segments: empty ones are stepped over)
                    rec[u] = a.l1[soff[j] + (e - sbase[j])];
                }
            }
#pragma unroll
            for (int u = 0; u < QF_L1_U; ++u) {
                const uint32_t e = e0 + 64 * (uint32_t)u + (uint32_t)lane;
                if (e < n1) {
                    key1[e] = ((uint64_t)rec[u].key_hi << 32) | rec[u].key_lo;
                    ypos1[e] = rec[u].ypos;
                }
            }
        }
        wave_sync();
        if (PGR_QF_ABLATE & 2) return;
        // ---- reduce_shmmr twice (shmmrutils.rs:533-535), then the min_span stencil on the unfiltered neighbours (:536-555: the
        // first and the last element always stay)
        const uint16_t *fin = nullptr;  // nullptr: the identity list
        uint32_t n3 = n1;
        auto ident = [](uint32_t k) { return (uint16_t)k; };
        if (a.r > 1) {
            const int n1i = (int)n1;
            const uint32_t n2 = a.r == 4 ? qf_compact(n1, lane, idx_a, [&](uint32_t k) { return qf_reduce_keep<4>(key1, nullptr, n1i, (int)k, 4); }, ident)
                                         : qf_compact(n1, lane, idx_a, [&](uint32_t k) { return qf_reduce_keep<0>(key1, nullptr, n1i, (int)k, a.r); }, ident);
            wave_sync();
            const int n2i = (int)n2;
            auto from_a = [&](uint32_t k) { return idx_a[k]; };
            n3 = a.r == 4 ? qf_compact(n2, lane, idx_b, [&](uint32_t k) { return qf_reduce_keep<4>(key1, idx_a, n2i, (int)k, 4); }, from_a)
                          : qf_compact(n2, lane, idx_b, [&](uint32_t k) { return qf_reduce_keep<0>(key1, idx_a, n2i, (int)k, a.r); }, from_a);
            wave_sync();
            fin = idx_b;
        }
        const uint32_t ms = a.min_span;
        const uint32_t n4 = qf_compact(
            n3, lane, idx_a,
            [&](uint32_t i) {
                if (i == 0 || i + 1 == n3) return true;
                const uint32_t e = fin ? fin[i] : i, ep = fin ? fin[i - 1] : i - 1, en = fin ? fin[i + 1] : i + 1;
                const uint32_t p = ypos1[e] >> 1, pp = ypos1[ep] >> 1, pn = ypos1[en] >> 1;
                const uint64_t xk = key1[e];
                return (p - pp > ms) && (pn - p > ms) && key1[ep] != xk && key1[en] != xk;
            },
            [&](uint32_t i) { return fin ? fin[i] : (uint16_t)i; });
        wave_sync();
        // ---- the query's shimmer pairs (seq_db.rs:1205-1217: the smaller hash first, strict <)
        const uint32_t npair = n4 ? n4 - 1 : 0u;
        if (npair > P) {
            if (lane == 0) {
                const uint32_t was = atomicMax(a.flags + 3, npair) | atomicOr(a.flags, QF_MORE_PAIRS);
                asm volatile("" ::"v"(was));
            }
            return;
        }
        np = (int)npair;
        n_pairs_out = npair;
        {   // (P <= 256: four pairs per lane; everything is read -- the final list lies where the pairs go -- before anything is written)
            constexpr int QF_PU = 4;
            static_assert(QF_MAX_PAIRS <= 64 * QF_PU && QF_C1_MAX * 4 <= 64 * QF_PU * 52, "pairs per lane");
            uint64_t k0[QF_PU], k1[QF_PU];
            uint32_t y0[QF_PU], y1[QF_PU];
#pragma unroll
            for (int u = 0; u < QF_PU; ++u) {
                const int i = lane + 64 * u;
                k0[u] = k1[u] = 0;
                y0[u] = y1[u] = 0;
                if (i < np) {
                    const uint32_t e0 = idx_a[i], e1 = idx_a[i + 1];
                    k0[u] = key1[e0];
                    k1[u] = key1[e1];
                    y0[u] = ypos1[e0];
                    y1[u] = ypos1[e1];
                }
            }
            wave_sync();
#pragma unroll
            for (int u = 0; u < QF_PU; ++u) {
                const int i = lane + 64 * u;
                if (i < np) {
                    const bool keep = k0[u] < k1[u];
                    L_h0[i] = keep ? k0[u] : k1[u];
                    L_h1[i] = keep ? k1[u] : k0[u];
                    L_bgn[i] = (y0[u] >> 1) + 1;
                    L_eo[i] = ((y1[u] >> 1) + 1) | (keep ? 0u : 0x80000000u);
                }
            }
        }
        wave_sync();
    } else {
        p0 = a.pair_off[q];
        const uint64_t np64 = a.pair_off[q + 1] - p0;
        decline = np64 > (uint64_t)P;
        np = decline ? 0 : (int)np64;
    }
    if (PGR_QF_ABLATE & 4) return;
    // ---- lookup of every pair
    for (int i = lane; i < np; i += 64) {
        const uint64_t h0 = L1 ? L_h0[i] : a.qrec[p0 + i].h0, h1 = L1 ? L_h1[i] : a.qrec[p0 + i].h1;
        L_h0[i] = h0;
        L_h1[i] = h1;
        if (a.qkeys) {  // (uniform)
            uint64_t lo, tbte;
            uint32_t nrec, sid, to;
            bool single;
            lookup_qkey(h0, h1, a, lo, nrec, single, tbte, sid, to);
            L_lo[i] = single ? tbte : lo;
            L_nrec[i] = nrec;
            L_sid[i] = sid;
            L_fl[i] = (single ? 1u : 0u) | (to << 1);
            nsig += nrec;
        } else {
            uint64_t lo, hi;
            lookup_range_short(h0, h1, a, lo, hi);
            L_lo[i] = lo;
            L_nrec[i] = (uint32_t)(hi - lo);  // (one key holds fewer than 2^32 records)
            L_fl[i] = 0u;
            nsig += hi - lo;
        }
    }
    wave_sync();
    if (PGR_QF_ABLATE & 8) return;
    // ---- multiplicity of the pair's key inside the query (aln.rs:180-181), count filters (aln.rs:197-228), hit counts
    for (int i0 = 0; i0 < np; i0 += 64) {
        const int i = i0 + lane;
        const bool live = i < np;
        uint32_t c = 0, n = 0;
        if (live) {
            const uint64_t h0 = L_h0[i], h1 = L_h1[i];
            // (four pairs per step: their LDS reads are in flight together -- the loop is a chain of LDS round trips otherwise)
            int j = 0;
            for (; j + 4 <= np; j += 4) {
                const uint64_t x0 = L_h0[j], x1 = L_h0[j + 1], x2 = L_h0[j + 2], x3 = L_h0[j + 3];
                const uint64_t y0 = L_h1[j], y1 = L_h1[j + 1], y2 = L_h1[j + 2], y3 = L_h1[j + 3];
                c += ((x0 == h0 && y0 == h1) ? 1u : 0u) + ((x1 == h0 && y1 == h1) ? 1u : 0u) + ((x2 == h0 && y2 == h1) ? 1u : 0u) +
                     ((x3 == h0 && y3 == h1) ? 1u : 0u);
            }
            for (; j < np; ++j) c += (L_h0[j] == h0 && L_h1[j] == h1) ? 1u : 0u;
        }
        const bool pass = live && c <= a.qp.max_count && c <= a.qp.max_count_query;
        const uint32_t nr = pass ? L_nrec[i] : 0u;
        if (nr > (uint32_t)HITS_HEAVY) decline = true;
        else if (nr && (L_fl[i] & 1u)) {  // the key's one record: a run of length 1
            if ((uint64_t)c <= a.qp.max_count_target) n = 1;
        } else if (nr) {
            const uint64_t s0 = L_lo[i], e0 = s0 + nr;
            uint64_t s = s0;
            while (s < e0) {  // records of one key are sorted by sid: target_shmer_pair_count[(key, sid)] = c * run length
                const uint32_t sid = a.recs[s].sid;
                uint64_t t = s + 1;
                while (t < e0 && a.recs[t].sid == sid) ++t;
                if ((uint64_t)(t - s) * c <= a.qp.max_count_target) n += (uint32_t)(t - s);
                s = t;
            }
        }
        if (live) L_pc[i] = pass ? c : 0u;
        const uint32_t incl = wave_incl_sum(n);
        if (live) L_hoff[i] = m + incl - n;
        m += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    const uint32_t n_hits = m;
    if (PGR_QF_ABLATE & 16) return;
    const bool any_decline = __ballot(decline) != 0;
    if (any_decline || m > H) {
        if (lane == 0) {
            uint32_t was;
            if (any_decline) was = atomicOr(a.flags, QF_DECLINE);
            else was = atomicMax(a.flags + 2, m) | atomicOr(a.flags, QF_MORE_HITS);
            asm volatile("" ::"v"(was));  // returned = done in memory before anything this wavefront publishes later
        }
        nsig = 0;
        return;
    }
    wave_sync();
    // ---- the hits, pair by pair = in query position order (aln.rs:21 wants a group's hits in ascending query bgn)
    for (int i = lane; i < np; i += 64) {
        const uint32_t c = L_pc[i];
        if (c == 0) continue;
        uint32_t q_bgn, q_end, q_or;
        if (L1) {
            q_bgn = L_bgn[i];
            q_end = L_eo[i] & 0x7FFFFFFFu;
            q_or = L_eo[i] >> 31;
        } else {
            const pgr_frag_rec qr = a.qrec[p0 + i];
            q_bgn = qr.bgn;
            q_end = qr.end;
            q_or = qr.orient;
        }
        uint32_t o = L_hoff[i];
        if (L_fl[i] & 1u) {  // the key's one record came with the lookup
            if ((uint64_t)c <= a.qp.max_count_target) {
                const uint64_t tbte = L_lo[i];
                pgr_hitpair hp;
                hp.qb = q_bgn;
                hp.qe = q_end;
                hp.qo = q_or;
                hp.tb = (uint32_t)tbte;
                hp.te = (uint32_t)(tbte >> 32);
                hp.to = L_fl[i] >> 1;
                hit[o] = hp;
                hsid[o] = L_sid[i];
            }
            continue;
        }
        const uint64_t s0 = L_lo[i], e0 = s0 + L_nrec[i];
        uint64_t s = s0;
        while (s < e0) {
            const uint32_t sid = a.recs[s].sid;
            uint64_t t = s + 1;
            while (t < e0 && a.recs[t].sid == sid) ++t;
            if ((uint64_t)(t - s) * c <= a.qp.max_count_target)
                for (uint64_t u = s; u < t; ++u) {
                    const pgr_frag_rec r = a.recs[u];
                    pgr_hitpair hp;
                    hp.qb = q_bgn;
                    hp.qe = q_end;
                    hp.qo = q_or;
                    hp.tb = r.bgn;
                    hp.te = r.end;
                    hp.to = r.orient;
                    hit[o] = hp;
                    hsid[o] = sid;
                    ++o;
                }
            s = t;
        }
    }
    wave_sync();
    if (PGR_QF_ABLATE & 32) return;
    // ---- stable sort by target sid: rank by counting; the hits stay where they are, perm[rank] = position
    for (uint32_t i0 = 0; i0 < m; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        if (i < m) {
            const uint32_t mine = hsid[i];
            uint32_t rank = 0;
            uint32_t j = 0;
            for (; j + 4 <= m; j += 4) {  // (as above: four reads in flight)
                const uint32_t o0 = hsid[j], o1 = hsid[j + 1], o2 = hsid[j + 2], o3 = hsid[j + 3];
                rank += ((o0 < mine || (o0 == mine && j < i)) ? 1u : 0u) + ((o1 < mine || (o1 == mine && j + 1 < i)) ? 1u : 0u) +
                        ((o2 < mine || (o2 == mine && j + 2 < i)) ? 1u : 0u) + ((o3 < mine || (o3 == mine && j + 3 < i)) ? 1u : 0u);
            }
            for (; j < m; ++j) {
                const uint32_t o = hsid[j];
                rank += (o < mine || (o == mine && j < i)) ? 1u : 0u;
            }
            perm[rank] = i;
            ssid[rank] = mine;
        }
    }
    wave_sync();
    if (PGR_QF_ABLATE & 64) return;
    // ---- groups = runs of equal sid; targets with a single hit are dropped (aln.rs:234)
    const size_t sb = (size_t)q * H;
    pgr_hitpair *o_hp = a.s_hp + sb;
    float *o_cscore = a.s_cscore + sb;
    uint32_t *o_choff = a.s_choff + sb;
    uint32_t n_stuck = 0;
    uint32_t gs = 0;
    while (gs < m) {
        const uint32_t sid = ssid[gs];
        uint32_t ge = gs + 1;
        for (;;) {  // first index behind gs with another sid, 64 at a time
            const uint32_t i = ge + (uint32_t)lane;
            const uint64_t diff = __ballot(i >= m || ssid[i < m ? i : 0] != sid);
            if (diff) {
                ge += (uint32_t)__builtin_ctzll(diff);
                break;
            }
            ge += 64;
        }
        const uint32_t n = ge - gs;
        if (n > 64 && !a.long_groups) {
            decline = true;
            break;
        }
        if (n >= 2) {
            if (lane == 0) {
                a.s_tsid[sb + nt] = sid;
                a.s_tcoff[sb + nt] = nc;
            }
            ++nt;
            bool stuck;
            if (PGR_QF_ABLATE & 1) {
                stuck = false;
            } else if (n <= 64) {
                stuck = chain_group_regs(hit, perm + gs, (int)n, a.ap, lane, o_hp, o_cscore, o_choff, nc, nh);
            } else {
                float *w_vs = reinterpret_cast<float *>(perm + H);
                int *w_sl = reinterpret_cast<int *>(w_vs + H), *w_pv = w_sl + H;
                uint32_t(*w_span)[3] = reinterpret_cast<uint32_t(*)[3]>(w_pv + H);
                uint32_t(*w_cand)[64] = reinterpret_cast<uint32_t(*)[64]>(w_pv + H + 64 * 3);
                stuck = chain_group_lds(hit, perm + gs, (int)n, a.ap, lane, w_vs, w_sl, w_pv, w_span, w_cand, o_hp, o_cscore,
                                        o_choff, nc, nh);
            }
            if (stuck) ++n_stuck;
        }
        gs = ge;
    }
    for (int d = 32; d >= 1; d >>= 1) nsig += shfl_xor64(nsig, d);
    if (lane == 0) {
        uint32_t was = 0;
        if (decline) was = atomicOr(a.flags, QF_DECLINE);
        if (n_stuck) was |= atomicAdd(a.flags + 1, n_stuck);
        asm volatile("" ::"v"(was));  // (as above)
    }
    if (decline) nt = nc = nh = 0;
    n_hits_out = n_hits;
}

inline __host__ __device__ size_t qf_up8(size_t v) { return (v + 7) & ~(size_t)7; }

// flat result = q_off | t_off | c_off | hps | c_score | t_sid (every section 8-byte aligned), the layout pgr_hps_result points into
struct QfLayout {
    size_t o_toff, o_coff, o_hps, o_cscore, o_tsid, bytes;
};
inline __host__ __device__ QfLayout qf_layout(uint64_t nq, uint64_t nt, uint64_t nc, uint64_t nh) {
    QfLayout l;
    l.o_toff = (nq + 1) * 8;
    l.o_coff = l.o_toff + (nt + 1) * 8;
    l.o_hps = l.o_coff + (nc + 1) * 8;
    l.o_cscore = l.o_hps + nh * sizeof(pgr_hitpair);
    l.o_tsid = qf_up8(l.o_cscore + nc * 4);
    l.bytes = qf_up8(l.o_tsid + nt * 4);
    return l;
}

// descriptor words of the look-back: status in bits 63:62 (0 not there yet, 1 this query's own counts, 2 the counts of all
// queries up to and including this one); [0] targets | chains << 31, [1] hit pairs | hits << 31, [2] looked-up signatures
constexpr uint64_t QF_ST_OWN = 1ull << 62, QF_ST_INCL = 2ull << 62, QF_VAL = (1ull << 62) - 1;
constexpr uint32_t QF_LOOKBACK_SPINS = 1u << 15;  // x (three loads + a sleep): tens of milliseconds, then the batch is declined

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
    return v;
}

struct QfSums {
    uint32_t t, c, h, hit;
    unsigned long long sig;
};
__device__ __forceinline__ QfSums qf_add(const QfSums &x, const QfSums &y) {
    return QfSums{x.t + y.t, x.c + y.c, x.h + y.h, x.hit + y.hit, x.sig + y.sig};
}
// Everything the descriptors carry is IN their words, and every reader checks the status of what it read: relaxed atomics
// (device scope: they pass the XCDs' L2s).  Release / acquire at device scope would write back / invalidate the XCD's whole L2
// per operation -- the first version of this kernel did, and took 1.15 ms instead of 0.1.
__device__ __forceinline__ void qf_desc_store(uint64_t *D, uint64_t status, const QfSums &v) {
    __hip_atomic_store(D + 0, status | (uint64_t)v.t | ((uint64_t)v.c << 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(D + 1, status | (uint64_t)v.h | ((uint64_t)v.hit << 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(D + 2, status | (v.sig & QF_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void qf_desc_load(const uint64_t *S, uint64_t &w0, uint64_t &w1, uint64_t &w2) {
    w0 = __hip_atomic_load(S + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w1 = __hip_atomic_load(S + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w2 = __hip_atomic_load(S + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (a writer between its three stores shows mixed states: 0 = look again)
__device__ __forceinline__ uint32_t qf_desc_status(uint64_t w0, uint64_t w1, uint64_t w2) {
    const uint32_t s0 = (uint32_t)(w0 >> 62), s1 = (uint32_t)(w1 >> 62), s2 = (uint32_t)(w2 >> 62);
    return (s0 == s1 && s1 == s2) ? s0 : 0u;
}
// the fields of the lanes with `take`, added up over the wavefront
__device__ __forceinline__ QfSums qf_wave_sums(bool take, uint64_t w0, uint64_t w1, uint64_t w2) {
    QfSums r;
    r.t = wave_sum_u32(take ? (uint32_t)(w0 & 0x7FFFFFFFull) : 0u);
    r.c = wave_sum_u32(take ? (uint32_t)((w0 >> 31) & 0x7FFFFFFFull) : 0u);
    r.h = wave_sum_u32(take ? (uint32_t)(w1 & 0x7FFFFFFFull) : 0u);
    r.hit = wave_sum_u32(take ? (uint32_t)((w1 >> 31) & 0x7FFFFFFFull) : 0u);
    unsigned long long sv = take ? (w2 & QF_VAL) : 0ull;
    for (int d = 32; d >= 1; d >>= 1) sv += shfl_xor64(sv, d);
    r.sig = sv;
    return r;
}

// DIRECT = false: counts per query; query_offsets_kernel + query_pack_kernel + a download make the flat result.
// DIRECT = true: single pass.  The wavefront publishes its counts, learns the counts of all queries in front of it (below:
// workgroups start in index order, so every predecessor is running or done) and copies its slot to where it belongs in the
// HOST's block -- the writes cross PCIe while other queries still chain (tools/probe/host_write_probe.hip: 10 000 wavefronts
// x 768 B contiguous reach the link's 50 GB/s and hide behind arithmetic; this kernel's small pieces do not: see enqueue()).
template <bool DIRECT, bool L1 = false>
__global__ __launch_bounds__(64) void query_fused_kernel(const QfArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t qf_dyn[];
    const uint32_t q = blockIdx.x;
    const int lane = (int)threadIdx.x;
    uint32_t nt, nc, nh, nhit, npairs;
    unsigned long long nsig;
    qf_one_query<L1>(a, qf_dyn, q, lane, nt, nc, nh, nhit, nsig, npairs);
    if (!DIRECT) {
        if (lane == 0) {
            a.q_nt[q] = nt;
            a.q_nc[q] = nc;
            a.q_nh[q] = nh;
            a.q_nhit[q] = nhit | (npairs << 16);  // (hits of a query <= QF_H_MAX; the level-1 form's pairs ride along: the host has not seen them)
            a.q_nsig[q] = nsig;
        }
        return;
    }
    // ---- where this query's entries start: counts of all queries in front of it.  Queries are taken in TILES of 64: the
    // wavefront that finishes last in a tile adds the tile's counts up and chains them to the tiles in front (a look-back over
    // tile descriptors, 64 tiles = 4096 queries per step); every wavefront then needs the running total of the tile in front
    // and the counts of the earlier queries of its own tile.  (A look-back over the queries themselves moved 64 queries per
    // memory round trip: 1.15 ms for 10 000 queries.)  Every wait is for a workgroup with a smaller index.
    const uint32_t n = a.n_queries, n_tiles = (n + 63) >> 6, tile = q >> 6, tile_lo = tile << 6;
    const uint32_t tile_n = n - tile_lo < 64u ? n - tile_lo : 64u;
    uint64_t *D = a.desc + 3 * (size_t)q;
    uint64_t *TD = a.desc + 3 * (size_t)n;
    uint32_t *TC = (uint32_t *)(TD + 3 * (size_t)n_tiles);
    QfSums own{nt, nc, nh, nhit, nsig};
    uint32_t spins = 0;
    bool gave_up = false;
    uint32_t last = 0;
    if (lane == 0) {  // (a flag of this query was set with a returning atomic: it is in memory)
        qf_desc_store(D, QF_ST_OWN, own);
        last = __hip_atomic_fetch_add(TC + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tile_n - 1 ? 1u : 0u;
    }
    last = (uint32_t)__builtin_amdgcn_readfirstlane((int)last);
    QfSums front{0, 0, 0, 0, 0};  // the tiles in front
    if (last) {  // every query of the tile has published its counts (or is about to: the counter is not ordered behind them)
        uint64_t w0 = 0, w1 = 0, w2 = 0;
        for (;;) {
            if ((uint32_t)lane < tile_n) qf_desc_load(a.desc + 3 * (size_t)(tile_lo + lane), w0, w1, w2);
            if (!__ballot((uint32_t)lane < tile_n && qf_desc_status(w0, w1, w2) == 0u)) break;
            if (++spins > QF_LOOKBACK_SPINS) {
                gave_up = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        QfSums agg = qf_wave_sums((uint32_t)lane < tile_n, w0, w1, w2);
        if (lane == 0 && tile + 1 < n_tiles) qf_desc_store(TD + 3 * (size_t)tile, QF_ST_OWN, agg);
        for (long long base = tile; base > 0;) {  // tiles base-1, base-2, ... 0 are not in the sums yet
            const long long idx = base - 1 - lane;
            const bool valid = idx >= 0;
            w0 = w1 = w2 = 0;
            if (valid) qf_desc_load(TD + 3 * (size_t)idx, w0, w1, w2);
            const uint32_t st = qf_desc_status(w0, w1, w2);
            const uint64_t incl = __ballot(valid && st == 2u), none = __ballot(valid && st == 0u);
            const int n_valid = base < 64 ? (int)base : 64;
            const int lim = incl ? __builtin_ctzll(incl) + 1 : n_valid;  // lanes [0, lim): own counts up to the nearest running total
            const uint64_t need = lim >= 64 ? U64MAX : ((1ull << lim) - 1);
            if (none & need) {
                if (++spins > QF_LOOKBACK_SPINS) {
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
            front = qf_add(front, qf_wave_sums(lane < lim, w0, w1, w2));
            if (incl) break;
            base -= 64;
        }
        if (lane == 0 && tile + 1 < n_tiles)  // (also after giving up: the tiles behind must not wait as well)
            qf_desc_store(TD + 3 * (size_t)tile, QF_ST_INCL, qf_add(front, agg));
    } else if (tile > 0) {
        for (;;) {
            uint64_t w0, w1, w2;
            qf_desc_load(TD + 3 * (size_t)(tile - 1), w0, w1, w2);
            if (qf_desc_status(w0, w1, w2) == 2u) {
                front = qf_wave_sums(lane == 0, w0, w1, w2);
                break;
            }
            if (++spins > QF_LOOKBACK_SPINS) {
                gave_up = true;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    QfSums before = front;  // + the earlier queries of this tile
    if (q > tile_lo) {
        const bool valid = (uint32_t)lane < q - tile_lo;
        for (;;) {
            uint64_t w0 = 0, w1 = 0, w2 = 0;
            if (valid) qf_desc_load(a.desc + 3 * (size_t)(tile_lo + lane), w0, w1, w2);
            if (!__ballot(valid && qf_desc_status(w0, w1, w2) == 0u)) {
                before = qf_add(before, qf_wave_sums(valid, w0, w1, w2));
                break;
            }
            if (++spins > QF_LOOKBACK_SPINS) {
                gave_up = true;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (gave_up && lane == 0) atomicOr(a.flags, QF_DECLINE | QF_LOOKBACK_TIMEOUT);
    const uint32_t eT = before.t, eC = before.c, eH = before.h, eHit = before.hit;
    const unsigned long long eS = before.sig;
    const QfLayout l = qf_layout(a.n_queries, a.cap_t, a.cap_c, a.cap_h);
    uint64_t *q_off = (uint64_t *)a.host, *t_off = (uint64_t *)(a.host + l.o_toff), *c_off = (uint64_t *)(a.host + l.o_coff);
    const uint64_t T0 = eT, C0 = eC, H0 = eH;
    const bool fits = T0 + nt <= a.cap_t && C0 + nc <= a.cap_c && H0 + nh <= a.cap_h;  // (if not, the totals tell the host)
    if (lane == 0) q_off[q] = T0;
    if (fits) {
        pgr_hitpair *hps = (pgr_hitpair *)(a.host + l.o_hps);
        float *c_score = (float *)(a.host + l.o_cscore);
        uint32_t *t_sid = (uint32_t *)(a.host + l.o_tsid);
        const size_t sb = (size_t)q * a.H;
        wave_sync();  // the slot's entries were written by other lanes of this wavefront
        for (uint32_t i = lane; i < nt; i += 64) {
            t_sid[T0 + i] = a.s_tsid[sb + i];
            t_off[T0 + i] = C0 + a.s_tcoff[sb + i];
        }
        for (uint32_t i = lane; i < nc; i += 64) {
            c_score[C0 + i] = a.s_cscore[sb + i];
            c_off[C0 + i] = H0 + a.s_choff[sb + i];
        }
        const uint64_t *src = (const uint64_t *)(a.s_hp + sb);
        uint64_t *dst = (uint64_t *)(hps + H0);
        for (uint32_t i = lane; i < nh * 3; i += 64) dst[i] = src[i];
    }
    if (q + 1 == a.n_queries && lane == 0) {  // every query's counts are in: the totals
        const uint64_t NT = T0 + nt, NC = C0 + nc, NH = H0 + nh;
        q_off[a.n_queries] = NT;
        if (NT <= a.cap_t) t_off[NT] = NC;
        if (NC <= a.cap_c) c_off[NC] = NH;
        a.words[0] = NT;
        a.words[1] = NC;
        a.words[2] = NH;
        a.words[3] = eS + nsig;
        a.words[4] = (uint64_t)eHit + nhit;
        a.words[5] = __hip_atomic_load(a.flags + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.words[6] = __hip_atomic_load(a.flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.words[8] = __hip_atomic_load(a.flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// exclusive scans of the per-query counts (one workgroup: a batch has 1 .. 2^17 queries) -> where every query's targets, chains
// and hit pairs start in the flat result; q_off of the result; the totals (straight into the host's pinned mailbox).
// words: [0] targets [1] chains [2] hit pairs [3] signatures [4] hits [5] QF_DECLINE | QF_MORE_HITS [6] non-terminating groups
//        [8] most hits of one query that overflowed its slot; level-1 form: [9] most pairs, [10] most level-1 minimizers of a query
//        that overflowed P / C1
// Every wavefront takes one contiguous sixteenth of the queries: it adds its range up, the sixteen sums are exchanged once, then
// it scans its range 64 queries per step with the running total in a register -- no barrier inside the loops.  The counts of
// one query are < 2^10: 32-bit sums.
constexpr int QF_SCAN_T = 1024;
__global__ __launch_bounds__(QF_SCAN_T) void query_offsets_kernel(const uint32_t *__restrict__ q_nt, const uint32_t *__restrict__ q_nc,
                                                                 const uint32_t *__restrict__ q_nh, const uint32_t *__restrict__ q_nhit,
                                                                 const unsigned long long *__restrict__ q_nsig,
                                                                 const uint32_t *__restrict__ flags, uint32_t n,
                                                                 uint64_t *__restrict__ t0, uint64_t *__restrict__ c0,
                                                                 uint64_t *__restrict__ h0, uint64_t *__restrict__ img_q_off,
                                                                 uint64_t *__restrict__ words, uint32_t l1_form) {
    constexpr uint32_t NW = QF_SCAN_T / 64;
    __shared__ uint32_t tot[3][NW];
    __shared__ unsigned long long red[2][NW];
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t R = (((n + NW - 1) / NW) + 63) & ~63u;  // queries per wavefront, a multiple of 64
    const uint32_t lo = w * R < n ? w * R : n, hi = lo + R < n ? lo + R : n;
    constexpr int U = 8;  // chunks of 64 queries whose loads are in flight together
    uint32_t st = 0, sc = 0, sh = 0;
    unsigned long long sig = 0, hits = 0;  // (hits: low 32 bits the hits, high 32 bits the pairs of the level-1 form)
    for (uint32_t q0 = lo; q0 < hi; q0 += 64 * U) {
        uint32_t a[U], b[U], c[U], d[U];
        unsigned long long e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = q0 + 64 * u + lane;
            const bool live = q < hi;
            a[u] = live ? q_nt[q] : 0u;
            b[u] = live ? q_nc[q] : 0u;
            c[u] = live ? q_nh[q] : 0u;
            d[u] = live ? q_nhit[q] : 0u;
            e[u] = live ? q_nsig[q] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            st += a[u];
            sc += b[u];
            sh += c[u];
            hits += (unsigned long long)(d[u] & 0xFFFFu) | ((unsigned long long)(d[u] >> 16) << 32);
            sig += e[u];
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        st += (uint32_t)__shfl_xor((int)st, d, 64);
        sc += (uint32_t)__shfl_xor((int)sc, d, 64);
        sh += (uint32_t)__shfl_xor((int)sh, d, 64);
        sig += shfl_xor64(sig, d);
        hits += shfl_xor64(hits, d);
    }
    if (lane == 0) {
        tot[0][w] = st;
        tot[1][w] = sc;
        tot[2][w] = sh;
        red[0][w] = sig;
        red[1][w] = hits;
    }
    __syncthreads();
    uint32_t bt = 0, bc = 0, bh = 0;
    for (uint32_t x = 0; x < w; ++x) {
        bt += tot[0][x];
        bc += tot[1][x];
        bh += tot[2][x];
    }
    for (uint32_t q0 = lo; q0 < hi; q0 += 64 * U) {
        uint32_t a[U], b[U], c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = q0 + 64 * u + lane;
            const bool live = q < hi;
            a[u] = live ? q_nt[q] : 0u;
            b[u] = live ? q_nc[q] : 0u;
            c[u] = live ? q_nh[q] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = q0 + 64 * u + lane;
            const uint32_t it = wave_incl_sum(a[u]), ic = wave_incl_sum(b[u]), ih = wave_incl_sum(c[u]);
            if (q < hi) {
                const uint64_t T = (uint64_t)bt + it - a[u];
                t0[q] = T;
                c0[q] = (uint64_t)bc + ic - b[u];
                h0[q] = (uint64_t)bh + ih - c[u];
                img_q_off[q] = T;
            }
            bt += (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
            bc += (uint32_t)__builtin_amdgcn_readlane((int)ic, 63);
            bh += (uint32_t)__builtin_amdgcn_readlane((int)ih, 63);
        }
    }
    if (t == 0) {
        uint32_t T = 0, Cn = 0, Hn = 0;
        unsigned long long s = 0, h = 0;
        for (uint32_t x = 0; x < NW; ++x) {
            T += tot[0][x];
            Cn += tot[1][x];
            Hn += tot[2][x];
            s += red[0][x];
            h += red[1][x];
        }
        img_q_off[n] = T;
        words[0] = T;
        words[1] = Cn;
        words[2] = Hn;
        words[3] = s;
        words[4] = h & 0xFFFFFFFFull;
        words[11] = h >> 32;  // pairs of all queries (level-1 form)
        words[5] = flags[0];
        words[6] = flags[1];
        words[8] = flags[2];
        words[9] = l1_form ? flags[3] : 0u;   // most pairs of a query that has more than P
        words[10] = l1_form ? flags[4] : 0u;  // most level-1 minimizers of a query that has more than C1
    }
}

// slots -> the flat result (a device image sized for full slots; the host downloads the part that is used)
__global__ __launch_bounds__(64) void query_pack_kernel(const QfArgs a, const uint64_t *__restrict__ t0, const uint64_t *__restrict__ c0,
                                                        const uint64_t *__restrict__ h0, const uint64_t *__restrict__ words,
                                                        uint8_t *__restrict__ img) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    if (words[5]) return;  // declined or asked for larger slots: there is no result
    const uint64_t NT = words[0], NC = words[1], NH = words[2];
    const QfLayout l = qf_layout(a.n_queries, NT, NC, NH);
    uint64_t *t_off = (uint64_t *)(img + l.o_toff), *c_off = (uint64_t *)(img + l.o_coff);
    pgr_hitpair *hps = (pgr_hitpair *)(img + l.o_hps);
    float *c_score = (float *)(img + l.o_cscore);
    uint32_t *t_sid = (uint32_t *)(img + l.o_tsid);
    if (q == 0 && lane == 0) {
        t_off[NT] = NC;
        c_off[NC] = NH;
    }
    const uint32_t nt = a.q_nt[q], nc = a.q_nc[q], nh = a.q_nh[q];
    const uint64_t T0 = t0[q], C0 = c0[q], H0 = h0[q];
    const size_t sb = (size_t)q * a.H;
    for (uint32_t i = lane; i < nt; i += 64) {
        t_sid[T0 + i] = a.s_tsid[sb + i];
        t_off[T0 + i] = C0 + a.s_tcoff[sb + i];
    }
    for (uint32_t i = lane; i < nc; i += 64) {
        c_score[C0 + i] = a.s_cscore[sb + i];
        c_off[C0 + i] = H0 + a.s_choff[sb + i];
    }
    // hit pairs: 24 bytes each, copied as 8-byte words (slot and destination are 8-byte aligned)
    const uint64_t *src = (const uint64_t *)(a.s_hp + sb);
    uint64_t *dst = (uint64_t *)(hps + H0);
    for (uint32_t i = lane; i < nh * 3; i += 64) dst[i] = src[i];
}

}  // namespace

bool query_fused_eligible(const pgr_ctx *ctx, uint32_t n_queries, uint64_t max_pairs, uint32_t max_aln_span) {
    if (ctx->opt.no_fused_query) return false;
    // slots and the result image are sized by the number of queries (up to QF_H_MAX x ~90 B each)
    return n_queries >= 1 && n_queries <= (1u << 17) && max_pairs <= QF_MAX_PAIRS && max_aln_span >= 1 && max_aln_span <= 64;
}

QueryFusedRun::QueryFusedRun(pgr_ctx *c, const pgr_index *i, uint32_t nq, uint64_t max_pairs, const QParams &q, const AlnParams &p)
    : ctx(c), ix(i), n_queries(nq), qp(q), ap(p) {
    // P: the longest query's pairs.  H: its pairs x the index's records per key with room, or what the last batch on this
    // index needed; a query with more hits makes the kernel ask for a larger H (once per call)
    P = QF_P_MIN;
    while (P < max_pairs) P <<= 1;
    const double per_key = ix->n_keys ? (double)ix->n / (double)ix->n_keys : 1.0;
    const uint64_t want_h =
        std::max<uint64_t>((uint64_t)((double)max_pairs * per_key * 1.5) + 8, ix->fused_hits.load(std::memory_order_relaxed));
    H = QF_H_MIN;
    while (H < QF_H_MAX && H < want_h) H <<= 1;
    if (ctx->opt.fused_query_hits > 0)
        H = (uint32_t)std::min<int64_t>(QF_H_MAX, std::max<int64_t>(QF_H_MIN, ctx->opt.fused_query_hits)) & ~63u;
}

QueryFusedRun::~QueryFusedRun() {
    // an error between enqueue and finish: the DMA engine may still be writing the pinned block -- wait before it goes back to the pool
    if (enqueued && !finish_called && block) {
        (void)hipStreamSynchronize(stream ? stream : ctx->stream);
        if (copy_stream) (void)hipStreamSynchronize(copy_stream);
    }
    for (void *p : {d_cnt, d_offs, d_shp, d_sf, d_img, d_qrec, d_rec_off, d_desc}) ctx->dfree(p);
    if (block) result_block_release(block);
}

namespace {
int grow(pgr_ctx *ctx, void *&p, size_t &have, size_t want) {
    if (want <= have && p) return PGR_OK;
    ctx->dfree(p);
    p = nullptr;
    have = 0;
    const int rc = ctx->dmalloc(&p, std::max<size_t>(want, 16));
    if (rc == PGR_OK) have = want;
    return rc;
}
}  // namespace

// pair records of the queries from the shimmer pipeline's device result, without the host: then enqueue()
int QueryFusedRun::enqueue_from_shimmers(const pgr_mm128 *d_mm, const uint64_t *d_off, uint64_t cap, const uint64_t *d_count) {
    int rc;
    const size_t nq = n_queries;
    if ((rc = grow(ctx, d_qrec, qrec_bytes, (size_t)cap * sizeof(pgr_frag_rec))) ||
        (rc = grow(ctx, d_rec_off, rec_off_bytes, (nq + 1) * 8)) || (rc = grow(ctx, d_cnt, cnt_bytes, nq * 24 + 16)))
        return rc;
    uint32_t *flags = (uint32_t *)d_cnt + 2 * nq + 4 * nq;  // (behind q_nsig | q_nt | q_nc | q_nh | q_nhit, as in enqueue)
    launch_frag_recs_dev(stream ? stream : ctx->stream, d_mm, d_off, n_queries, cap, d_count, 1, (uint64_t *)d_rec_off, (pgr_frag_rec *)d_qrec, cap, flags);
    return enqueue((const pgr_frag_rec *)d_qrec, (const uint64_t *)d_rec_off, true);
}

uint32_t query_fused_level1_cap(uint32_t max_len, uint32_t w) {
    const double expect = 2.0 * (double)max_len / (double)(w + 1);
    const uint64_t cap = (((uint64_t)(expect * 1.15) + 36 + 31) / 32) * 32;
    return cap > QF_C1_MAX ? 0u : (uint32_t)cap;
}

int QueryFusedRun::enqueue_from_level1(const QfLevel1View &v, uint32_t c1) {
    from_l1 = true;
    l1v = v;
    C1 = c1;
    while ((size_t)P * 52 < (size_t)C1 * 4) P <<= 1;  // (the index lists of the level-1 stage live in the pairs' bytes)
    return enqueue(nullptr, nullptr, true);
}

// the per-query kernel, the offsets, the packing and the first download, all stream ordered; finish() after a synchronization
int QueryFusedRun::enqueue(const pgr_frag_rec *qrec, const uint64_t *pair_off, bool flags_cleared) {
    hipStream_t st = stream ? stream : ctx->stream;
    int rc;
    const size_t nq = n_queries, slots = nq * H;
    if (!mail && (rc = ctx->ensure_qmail())) return rc;
    uint64_t *mb = mail ? mail : (uint64_t *)ctx->qmail;  // pinned: the kernels write the totals here
    const QfLayout lmax = qf_layout(nq, slots / 2, slots, slots);
    // EXPERIMENT (context option direct_query_result, off by default: measured SLOWER).  Single pass (query_fused_kernel<true>)
    // when an earlier batch on this index has shown how many targets, chains and hit pairs a query has: the sections of the
    // host's block are placed for that + 10 %, the kernel writes them itself.  A batch that needs more says so in its totals
    // and is done again in the two-pass form (finish()).  10 000 x 10 kbp queries (profiles/r05_query/single_pass.txt): the
    // kernel takes 141 us with the block in device memory (two-pass: 95 + 13 + 10), 300 us writing the host's block (7.5 MB in
    // 4-24 byte pieces of five sections per query: the link takes them at ~30 GB/s, and nearly all queries are resident at once
    // and finish together, so the writes do not hide behind the chaining) -- against 95 + 13 + 10 + 136 us for two passes and
    // the download.  Fewer queries resident at once (direct_query_lds_kb) only makes it longer.
    const float h_t = ix->fused_per_q[0].load(std::memory_order_relaxed), h_c = ix->fused_per_q[1].load(std::memory_order_relaxed),
                h_h = ix->fused_per_q[2].load(std::memory_order_relaxed);
    direct = !from_l1 && !direct_failed && ctx->opt.direct_query_result && h_h >= 0.0f && h_t >= 0.0f && h_c >= 0.0f;
    if (direct) {
        auto room = [&](float per_q, uint64_t most) { return std::min<uint64_t>(most, (uint64_t)((double)nq * per_q * 1.10) + 512); };
        cap_t = room(h_t, slots / 2);
        cap_c = room(h_c, slots);
        cap_h = room(h_h, slots);
    }
    // d_cnt: q_nsig | q_nt | q_nc | q_nh | q_nhit | flags ; d_offs: t0 | c0 | h0 ; d_sf: s_cscore | s_choff | s_tsid | s_tcoff
    if ((rc = grow(ctx, d_cnt, cnt_bytes, nq * 24 + 16)) || (rc = grow(ctx, d_shp, shp_bytes, slots * sizeof(pgr_hitpair))) ||
        (rc = grow(ctx, d_sf, sf_bytes, slots * 16)))
        return rc;
    const size_t n_qtiles = (nq + 63) / 64, desc_need = (nq + n_qtiles) * 24 + n_qtiles * 4;
    if (direct ? (rc = grow(ctx, d_desc, desc_bytes, desc_need))
               : ((rc = grow(ctx, d_offs, offs_bytes, nq * 24)) || (rc = grow(ctx, d_img, img_bytes, lmax.bytes))))
        return rc;
    QfArgs a;
    a.qrec = qrec;
    a.pair_off = pair_off;
    a.n_queries = n_queries;
    a.recs = ix->recs;
    a.key_off = ix->key_off;
    a.n_keys = ix->n_keys;
    a.lut = ix->lut;
    a.lut_bits = ix->lut_bits;
    a.lut_shift = ix->lut_shift;
    a.keys = ix->keys;
    a.qkeys = ix->lut ? ix->qkeys : nullptr;
    a.qp = qp;
    a.ap = ap;
    a.P = P;
    a.H = H;
    a.long_groups = P > 64 ? 1u : 0u;
    a.q_nsig = (unsigned long long *)d_cnt;
    a.q_nt = (uint32_t *)d_cnt + 2 * nq;
    a.q_nc = a.q_nt + nq;
    a.q_nh = a.q_nc + nq;
    a.q_nhit = a.q_nh + nq;
    a.flags = from_l1 ? l1v.flags : a.q_nhit + nq;
    a.l1 = l1v.l1;
    a.seg_off = l1v.seg_off;
    a.seg_cnt = l1v.seg_cnt;
    a.tile_first = l1v.tile_first;
    a.l1_status = l1v.status;
    a.l1_ovf_cap = l1v.ovf_cap;
    a.C1 = from_l1 ? C1 : 0u;
    a.r = l1v.r;
    a.min_span = l1v.min_span;
    a.desc = nullptr;
    a.host = nullptr;
    a.cap_t = a.cap_c = a.cap_h = 0;
    a.words = nullptr;
    a.s_hp = (pgr_hitpair *)d_shp;
    a.s_cscore = (float *)d_sf;
    a.s_choff = (uint32_t *)d_sf + slots;
    a.s_tsid = (uint32_t *)d_sf + 2 * slots;
    a.s_tcoff = (uint32_t *)d_sf + 3 * slots;
    // The host block of the result is pinned: the DMA engine writes it, the caller reads it, no staging copy.  How much to
    // download is known on the device only: the copy is enqueued for an estimate (what the last batch on this index needed
    // per query + 5 %; first time: a quarter of the slots) and the rest follows when the totals say there is more.
    const float bytes_hint = ix->fused_bytes_per_q.load(std::memory_order_relaxed);
    const size_t est = direct ? qf_layout(nq, cap_t, cap_c, cap_h).bytes
                              : std::min(lmax.bytes, bytes_hint > 0 ? (size_t)(nq * (double)bytes_hint * 1.05) + 32768
                                                                     : (nq + 1) * 8 + slots * 10 + 4096);
    if (block && direct && cap < est) {
        result_block_release(block);
        block = nullptr;
    }
    if (!block && !(block = (uint8_t *)pinned_result_acquire(est, &cap))) {
        no_pinned = true;  // the host cannot pin more memory: the stage-by-stage path needs none
        return PGR_OK;
    }
    first = std::min(est, cap);
    hipError_t e = flags_cleared ? hipSuccess : hipMemsetAsync(a.flags, 0, from_l1 ? 24 : 12, st);
    if (e == hipSuccess && direct) {
        a.desc = (uint64_t *)d_desc;
        a.host = block;
        if (ctx->opt.direct_query_lds_kb < 0) {  // EXPERIMENT (timing only, the result is not delivered): the block in device memory
            if ((rc = grow(ctx, d_img, img_bytes, est))) return rc;
            a.host = (uint8_t *)d_img;
        }
        a.cap_t = cap_t;
        a.cap_c = cap_c;
        a.cap_h = cap_h;
        a.words = mb;
        e = hipMemsetAsync(d_desc, 0, desc_need, st);
        if (e == hipSuccess)
            hipLaunchKernelGGL((query_fused_kernel<true, false>), dim3(n_queries), dim3(64),
                               std::max<size_t>(qf_lds_bytes(P, H, a.long_groups != 0), (size_t)std::max<int64_t>(0, ctx->opt.direct_query_lds_kb) << 10), st, a);
    } else if (e == hipSuccess) {
        uint64_t *t0 = (uint64_t *)d_offs, *c0 = t0 + nq, *h0 = c0 + nq;
        if (from_l1)
            hipLaunchKernelGGL((query_fused_kernel<false, true>), dim3(n_queries), dim3(64), qf_lds_bytes(P, H, a.long_groups != 0, C1), st, a);
        else
            hipLaunchKernelGGL((query_fused_kernel<false, false>), dim3(n_queries), dim3(64), qf_lds_bytes(P, H, a.long_groups != 0), st, a);
        hipLaunchKernelGGL(query_offsets_kernel, dim3(1), dim3(QF_SCAN_T), 0, st, a.q_nt, a.q_nc, a.q_nh, a.q_nhit, a.q_nsig, a.flags,
                           n_queries, t0, c0, h0, (uint64_t *)d_img, mb, from_l1 ? 1u : 0u);
        hipLaunchKernelGGL(query_pack_kernel, dim3(n_queries), dim3(64), 0, st, a, t0, c0, h0, mb, (uint8_t *)d_img);
        if (copy_stream && ev_packed && ev_copied) {
            e = hipEventRecord(ev_packed, st);
            if (e == hipSuccess) e = hipStreamWaitEvent(copy_stream, ev_packed, 0);
            if (e == hipSuccess) e = hipMemcpyAsync(block, d_img, first, hipMemcpyDeviceToHost, copy_stream);
            if (e == hipSuccess) e = hipEventRecord(ev_copied, copy_stream);
        } else {
            e = hipMemcpyAsync(block, d_img, first, hipMemcpyDeviceToHost, st);
        }
    }
    if (e != hipSuccess) return ctx->fail(PGR_ERR_DEVICE, std::string("query kernels: ") + hipGetErrorString(e));
    qrec_used = qrec;
    pair_off_used = pair_off;
    enqueued = true;
    return PGR_OK;
}

// behind a synchronization of the stream: the totals are in the mailbox.  declined: the batch is for the stage-by-stage path.
int QueryFusedRun::finish(pgr_hps_result *out, QueryFusedCounts *counts, bool *declined) {
    *declined = false;
    finish_called = true;
    if (no_pinned || !enqueued) {
        *declined = true;
        return PGR_OK;
    }
    hipStream_t st = stream ? stream : ctx->stream;
    const size_t nq = n_queries;
    uint64_t *mb = mail ? mail : (uint64_t *)ctx->qmail;
    bool grew = false, grew_p = false, grew_c1 = false;
    for (;;) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return ctx->fail(PGR_ERR_DEVICE, std::string("query kernels: ") + hipGetErrorString(e));
        const bool lookback_gave_up = direct && (mb[5] & QF_LOOKBACK_TIMEOUT);
        l1_flagged = from_l1 && (mb[5] & QF_L1_FLAGGED);
        // the level-1 form: a query with more pairs than P / more level-1 minimizers than C1 (low-complexity sequence, a first
        // guess that was too small) -- once more with room, when there is such a form
        const bool more_pairs = from_l1 && (mb[5] & QF_MORE_PAIRS), more_l1 = from_l1 && (mb[5] & QF_MORE_L1);
        if (!lookback_gave_up &&
            ((mb[5] & QF_DECLINE) || ((mb[5] & QF_MORE_HITS) && (H >= QF_H_MAX || mb[8] > QF_H_MAX || grew)) ||
             (more_pairs && (mb[9] > QF_MAX_PAIRS || grew_p)) || (more_l1 && (mb[10] > QF_C1_MAX || grew_c1)))) {
            *declined = true;
            return PGR_OK;
        }
        if (more_pairs || more_l1) {
            if (more_pairs) {
                grew_p = true;
                while (P < mb[9]) P <<= 1;
                ix->fused_pairs.store(P, std::memory_order_relaxed);
            }
            if (more_l1) {
                grew_c1 = true;
                C1 = (uint32_t)((mb[10] + 31) / 32 * 32);
                while ((size_t)P * 52 < (size_t)C1 * 4) P <<= 1;
            }
            // (a larger P may come with more hits than the slot holds: QF_MORE_HITS of this pass is looked at in the next one)
            int rc = enqueue(nullptr, nullptr);
            if (rc) return rc;
            if (no_pinned) {
                *declined = true;
                return PGR_OK;
            }
            if (hipStreamSynchronize(st) != hipSuccess || (copy_stream && hipStreamSynchronize(copy_stream) != hipSuccess))
                return ctx->fail(PGR_ERR_DEVICE, "query kernels failed on the device");
            continue;
        }
        const bool more_hits = !lookback_gave_up && (mb[5] & QF_MORE_HITS);
        // the single-pass form placed the sections for fewer targets / chains / hit pairs than the batch has (or gave up
        // waiting): the two-pass form takes it
        const bool redo_two_pass = direct && !more_hits && (lookback_gave_up || mb[0] > cap_t || mb[1] > cap_c || mb[2] > cap_h);
        if (redo_two_pass && ctx->opt.debug)
            fprintf(stderr, "[pgr] query batch: single-pass result %s (%llu / %llu / %llu of %llu / %llu / %llu): two passes\n",
                    lookback_gave_up ? "gave up in its look-back" : "needs more room", (unsigned long long)mb[0], (unsigned long long)mb[1],
                    (unsigned long long)mb[2], (unsigned long long)cap_t, (unsigned long long)cap_c, (unsigned long long)cap_h);
        if (more_hits || redo_two_pass) {  // every query fits a larger slot: once more with it
            if (more_hits) {
                grew = true;
                while (H < mb[8]) H <<= 1;
            } else {
                direct_failed = true;
            }
            int rc = enqueue(qrec_used, pair_off_used);
            if (rc) return rc;
            if (no_pinned) {
                *declined = true;
                return PGR_OK;
            }
            if (hipStreamSynchronize(st) != hipSuccess || (copy_stream && hipStreamSynchronize(copy_stream) != hipSuccess))
                return ctx->fail(PGR_ERR_DEVICE, "query kernels failed on the device");
            continue;
        }
        break;
    }
    const uint64_t NT = mb[0], NC = mb[1], NH = mb[2];
    const QfLayout l = direct ? qf_layout(nq, cap_t, cap_c, cap_h) : qf_layout(nq, NT, NC, NH);
    const size_t need = l.bytes;
    if (!direct && need > first) {  // more than the estimate
        hipError_t e;
        if (need > cap) {
            result_block_release(block);
            if (!(block = (uint8_t *)pinned_result_acquire(need, &cap))) {
                *declined = true;
                return PGR_OK;
            }
            e = hipMemcpyAsync(block, d_img, need, hipMemcpyDeviceToHost, st);
        } else {
            e = hipMemcpyAsync(block + first, (const uint8_t *)d_img + first, need - first, hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return ctx->fail(PGR_ERR_DEVICE, std::string("query result download: ") + hipGetErrorString(e));
    }
    if (direct) ++ctx->opt.direct_query_results_delivered;
    if (!direct) ix->fused_bytes_per_q.store((float)((double)need / (double)nq), std::memory_order_relaxed);
    ix->fused_per_q[0].store((float)((double)NT / (double)nq), std::memory_order_relaxed);
    ix->fused_per_q[1].store((float)((double)NC / (double)nq), std::memory_order_relaxed);
    ix->fused_per_q[2].store((float)((double)NH / (double)nq), std::memory_order_relaxed);
    ix->fused_hits.store(H > QF_H_MIN ? H : 0, std::memory_order_relaxed);
    counts->n_signatures = mb[3];
    counts->n_hits = mb[4];
    counts->n_pairs = from_l1 ? mb[11] : 0;
    out->n_queries = n_queries;
    out->q_off = (uint64_t *)block;
    out->n_targets = NT;
    out->t_off = (uint64_t *)(block + l.o_toff);
    out->t_sid = (uint32_t *)(block + l.o_tsid);
    out->n_chains = NC;
    out->c_off = (uint64_t *)(block + l.o_coff);
    out->c_score = (float *)(block + l.o_cscore);
    out->n_hps = NH;
    out->hps = (pgr_hitpair *)(block + l.o_hps);
    out->n_nonterminating = mb[6];
    out->_owner = block;
    block = nullptr;  // the caller's now
    return PGR_OK;
}

}  // namespace pgr
