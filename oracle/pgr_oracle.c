/*
 * pgr_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see pgr_oracle.h).
 *
 * Sequential restatement of the reference's SHIMMER path.  Deliberately keeps the
 * reference's structure (ring buffer with O(w) rescan, per-base loop, quirks) so that it
 * can be read side by side with the Rust source; it is also what bench.py times as the
 * "reference-equivalent CPU path" (cpu_baseline.kind = "port").
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off).
 */
#include "pgr_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* small growable vector of mm128                                                        */
typedef struct {
    orc_mm128 *v;
    size_t n, cap;
} mmvec;

static void mmvec_push(mmvec *a, orc_mm128 m) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 64;
        a->v = (orc_mm128 *)realloc(a->v, a->cap * sizeof(orc_mm128));
    }
    a->v[a->n++] = m;
}

void orc_free(void *p) { free(p); }

/* pgr-db/src/shmmrutils.rs:271-280  (Thomas Wang / minimap2 hash64 with a full mask) */
uint64_t orc_u64hash(uint64_t key) {
    key = (~key) + (key << 21);
    key = key ^ key >> 24;
    key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14;
    key = (key + (key << 2)) + (key << 4);
    key = key ^ key >> 28;
    key = key + (key << 31);
    return key;
}

/* pgr-db/src/shmmrutils.rs:293-357 */
typedef struct {
    orc_mm128 *v;
    size_t size, start_pos, end_pos, len;
} ringbuf;

static const orc_mm128 MM_MAX = {UINT64_MAX, UINT64_MAX};

static void rb_init(ringbuf *rb, size_t size) {
    rb->v = (orc_mm128 *)malloc(sizeof(orc_mm128) * (size ? size : 1));
    for (size_t i = 0; i < size; i++) rb->v[i] = MM_MAX;
    rb->size = size;
    rb->start_pos = rb->end_pos = rb->len = 0;
}
/* :319-332 */
static void rb_push(ringbuf *rb, orc_mm128 m) {
    if (rb->len < rb->size) {
        rb->v[rb->end_pos] = m;
        rb->end_pos = (rb->end_pos + 1) % rb->size;
        rb->len += 1;
    } else {
        rb->v[rb->end_pos] = m;
        rb->end_pos = (rb->end_pos + 1) % rb->size;
        rb->start_pos = (rb->start_pos + 1) % rb->size;
    }
}
/* :341-352  (scans storage order, strict <) */
static orc_mm128 rb_get_min(const ringbuf *rb) {
    orc_mm128 min = MM_MAX;
    for (size_t i = 0; i < rb->len; i++)
        if (rb->v[i].x < min.x) min = rb->v[i];
    return min;
}
/* :354-356 */
static orc_mm128 rb_get(const ringbuf *rb, size_t i) { return rb->v[(rb->start_pos + i) % rb->size]; }

/* pgr-db/src/shmmrutils.rs:359-415 */
size_t orc_reduce_shmmr(const orc_mm128 *mers_in, size_t n_in, uint32_t r, int padding,
                        orc_mm128 **out) {
    mmvec shmmrs = {0};
    ringbuf rbuf;
    rb_init(&rbuf, r);
    orc_mm128 min_mer = MM_MAX;

    const orc_mm128 *mers = mers_in;
    size_t n = n_in;
    orc_mm128 *mers2 = NULL;
    if (padding) { /* :371-380 */
        n = n_in + 2 * (size_t)(r - 1);
        mers2 = (orc_mm128 *)malloc(sizeof(orc_mm128) * (n ? n : 1));
        size_t q = 0;
        for (uint32_t i = 0; i + 1 < r; i++) mers2[q++] = min_mer;
        for (size_t i = 0; i < n_in; i++) mers2[q++] = mers_in[i];
        for (uint32_t i = 0; i + 1 < r; i++) mers2[q++] = min_mer;
        mers = mers2;
    }

    size_t pos = 0, mdist = 0;
    for (;;) {
        if (pos >= n) break;
        orc_mm128 m = mers[pos];
        rb_push(&rbuf, m);
        if (mdist == (size_t)(r - 1)) { /* :390-403 */
            min_mer = rb_get_min(&rbuf);
            size_t last_i = 0;
            for (size_t i = 0; i < rbuf.size; i++) {
                orc_mm128 mm = rb_get(&rbuf, i);
                if (mm.x == min_mer.x) {
                    mmvec_push(&shmmrs, mm);
                    min_mer = mm;
                    last_i = i;
                }
            }
            mdist = (size_t)r - 1 - last_i;
            pos += 1;
            continue;
        } else if (m.x <= min_mer.x && pos >= (size_t)r) { /* :404-410 */
            mmvec_push(&shmmrs, m);
            min_mer = m;
            mdist = 0;
            pos += 1;
            continue;
        }
        mdist += 1;
        pos += 1;
    }
    free(rbuf.v);
    free(mers2);
    *out = shmmrs.v;
    return shmmrs.n;
}

/* pgr-db/src/shmmrutils.rs:426-436 */
static uint64_t base2bits(uint8_t c) {
    switch (c) {
    case 0: case 'A': case 'a': return 0;
    case 1: case 'C': case 'c': return 1;
    case 2: case 'G': case 'g': return 2;
    case 3: case 'T': case 't': return 3;
    default: return 4;
    }
}

/* per-base k-mer roll: shmmrutils.rs:459-476 */
typedef struct {
    uint64_t f0, f1, r0, r1;
} kroll;

static inline void kroll_step(kroll *s, uint64_t c, uint64_t mask, uint32_t shift) {
    if (c < 4) {
        s->f0 = ((s->f0 << 1) | (c & 1)) & mask;
        s->f1 = ((s->f1 << 1) | ((c & 2) >> 1)) & mask;
        uint64_t rc = 3 ^ c;
        s->r0 = ((s->r0 >> 1) | ((rc & 1) << shift)) & mask;
        s->r1 = ((s->r1 >> 1) | (((rc & 2) >> 1) << shift)) & mask;
    }
}

/* level-1 windowed minimizers: shmmrutils.rs:438-530 */
static void level1(uint32_t rid, const uint8_t *seq, size_t len, uint32_t w, uint32_t k,
                   mmvec *shmmrs) {
    size_t pos = 0, mdist = 0;
    const uint32_t shift = k - 1;
    kroll s = {0, 0, 0, 0};
    const uint64_t mask = UINT64_MAX >> (64 - k);
    ringbuf rbuf;
    rb_init(&rbuf, w);
    orc_mm128 min_mer = MM_MAX;
    for (;;) {
        if (pos >= len) break;
        uint64_t c = base2bits(seq[pos]);
        kroll_step(&s, c, mask, shift);
        if (s.f0 == s.r0 && s.f1 == s.r1) { /* :477-480 */
            pos += 1;
            continue;
        }
        if (pos < (size_t)k) { /* :481-484 */
            pos += 1;
            continue;
        }
        int forward = 1;
        if (s.r0 < s.f0) forward = 0; /* :485-488: low plane only */
        uint64_t mmer_hash = forward ? (orc_u64hash(s.f0) ^ orc_u64hash(s.f1 ^ 0xAD12CF59ULL))
                                     : (orc_u64hash(s.r0) ^ orc_u64hash(s.r1 ^ 0xAD12CF59ULL));
        uint64_t strand = forward ? 0 : 1;
        orc_mm128 m;
        m.x = (mmer_hash << 8) | (uint64_t)k;
        m.y = ((uint64_t)rid << 32) | ((uint64_t)pos << 1) | strand;
        rb_push(&rbuf, m);
        if (mdist == (size_t)(w - 1)) { /* :503-515 */
            min_mer = rb_get_min(&rbuf);
            for (size_t i = 0; i < rbuf.size; i++) {
                orc_mm128 mm = rb_get(&rbuf, i);
                if (mm.x == min_mer.x) {
                    mmvec_push(shmmrs, mm);
                    min_mer = mm;
                }
            }
            mdist = pos - (size_t)((min_mer.y & 0xFFFFFFFFULL) >> 1);
            pos += 1;
            continue;
        } else if (m.x <= min_mer.x && pos >= (size_t)(w + k) &&
                   /* Rust: pos < seq.len() - w as usize + k as usize, evaluated left to right in
                    * usize.  Release builds wrap, so the sum is exact modulo 2^64 (== len + k - w
                    * whenever len + k >= w); the same modular arithmetic is used here. */
                   pos < (size_t)(len - (size_t)w + (size_t)k) && pos < len) { /* :516-520 */
            mmvec_push(shmmrs, m);
            min_mer = m;
            mdist = 0;
            pos += 1;
            continue;
        }
        mdist += 1;
        pos += 1;
    }
    free(rbuf.v);
}

size_t orc_level1_minimizers(uint32_t rid, const uint8_t *seq, size_t len, uint32_t w, uint32_t k,
                             orc_mm128 **out) {
    mmvec v = {0};
    level1(rid, seq, len, w, k, &v);
    *out = v.v;
    return v.n;
}

static inline uint32_t mm_pos(const orc_mm128 *m) { return (uint32_t)((m->y & 0xFFFFFFFFULL) >> 1); }

/* min_span 3-point stencil: shmmrutils.rs:536-555 (u32 wrapping subtraction as in release) */
static size_t span_filter(const orc_mm128 *s, size_t n, uint32_t min_span, orc_mm128 **out) {
    mmvec o = {0};
    for (size_t i = 0; i < n; i++) {
        if (i != 0 && i != n - 1) {
            uint32_t p_pos = mm_pos(&s[i - 1]), pos = mm_pos(&s[i]), n_pos = mm_pos(&s[i + 1]);
            uint64_t px = s[i - 1].x, x = s[i].x, nx = s[i + 1].x;
            if ((uint32_t)(pos - p_pos) > min_span && (uint32_t)(n_pos - pos) > min_span && px != x &&
                x != nx)
                mmvec_push(&o, s[i]);
        } else {
            mmvec_push(&o, s[i]);
        }
    }
    *out = o.v;
    return o.n;
}

/* shmmrutils.rs:417-556 */
static size_t sequence_to_shmmrs1(uint32_t rid, const uint8_t *seq, size_t len, uint32_t w,
                                  uint32_t k, uint32_t r, uint32_t min_span, int padding,
                                  orc_mm128 **out) {
    mmvec l1 = {0};
    level1(rid, seq, len, w, k, &l1);
    orc_mm128 *cur = l1.v;
    size_t n = l1.n;
    if (r > 1) { /* :533-535 */
        orc_mm128 *a = NULL, *b = NULL;
        size_t na = orc_reduce_shmmr(cur, n, r, padding, &a);
        size_t nb = orc_reduce_shmmr(a, na, r, padding, &b);
        free(cur);
        free(a);
        cur = b;
        n = nb;
    }
    size_t no = span_filter(cur, n, min_span, out);
    free(cur);
    return no;
}

/* shmmrutils.rs:558-655 (sketch variant) */
static size_t sequence_to_shmmrs2(uint32_t rid, const uint8_t *seq, size_t len, uint32_t k,
                                  uint32_t r, uint32_t min_span, orc_mm128 **out) {
    mmvec v = {0};
    const uint32_t shift = k - 1;
    kroll s = {0, 0, 0, 0};
    const uint64_t mask = UINT64_MAX >> (64 - k);
    for (size_t pos = 0; pos < len; pos++) {
        uint64_t c = base2bits(seq[pos]);
        kroll_step(&s, c, mask, shift);
        if (s.f0 == s.r0 && s.f1 == s.r1) continue;
        if (pos < (size_t)k) continue;
        int forward = 1;
        if (s.r0 < s.f0) forward = 0;
        uint64_t h = forward ? (orc_u64hash(s.f0) ^ orc_u64hash(s.f1 ^ 0xAD12CF59ULL))
                             : (orc_u64hash(s.r0) ^ orc_u64hash(s.r1 ^ 0xAD12CF59ULL));
        if (h < ((UINT64_MAX >> 4) >> r)) { /* :621 */
            orc_mm128 m;
            m.x = (h << 8) | (uint64_t)k;
            m.y = ((uint64_t)rid << 32) | ((uint64_t)pos << 1) | (forward ? 0ULL : 1ULL);
            mmvec_push(&v, m);
        }
    }
    size_t no = span_filter(v.v, v.n, min_span, out);
    free(v.v);
    return no;
}

/* shmmrutils.rs:657-669; asserts :443-445 / :575-576 */
size_t orc_sequence_to_shmmrs(uint32_t rid, const uint8_t *seq, size_t len, const orc_spec *spec,
                              int padding, orc_mm128 **out) {
    *out = NULL;
    if (spec->k > 56 || spec->k == 0) return (size_t)-1;
    if (!(spec->r > 0 && spec->r < 13)) return (size_t)-1;
    if (!spec->sketch) {
        if (spec->w > 128 || spec->w == 0) return (size_t)-1;
        return sequence_to_shmmrs1(rid, seq, len, spec->w, spec->k, spec->r, spec->min_span, padding,
                                   out);
    }
    return sequence_to_shmmrs2(rid, seq, len, spec->k, spec->r, spec->min_span, out);
}

/* seq_db.rs:381-400 (index: s0 <= s1) and seq_db.rs:1205-1217 (query: s0 < s1) */
size_t orc_shmmrs_to_frag_recs(const orc_mm128 *s, size_t n, uint32_t sid, int query_side,
                               orc_frag_rec **out) {
    if (n < 2) {
        *out = NULL;
        return 0;
    }
    orc_frag_rec *o = (orc_frag_rec *)malloc(sizeof(orc_frag_rec) * (n - 1));
    for (size_t i = 0; i + 1 < n; i++) {
        uint64_t s0 = s[i].x >> 8, s1 = s[i + 1].x >> 8;
        int keep = query_side ? (s0 < s1) : (s0 <= s1);
        o[i].h0 = keep ? s0 : s1;
        o[i].h1 = keep ? s1 : s0;
        o[i].orient = keep ? 0 : 1;
        o[i].bgn = mm_pos(&s[i]) + 1;
        o[i].end = mm_pos(&s[i + 1]) + 1;
        o[i].frg_id = (uint32_t)i;
        o[i].sid = sid;
    }
    *out = o;
    return n - 1;
}

/* ------------------------------------------------------------------------------------ */
/* index: frag_map as sorted CSR                                                         */
struct orc_index {
    orc_spec spec;
    orc_frag_rec *recs;
    size_t n, cap;
    uint32_t next_global_frag; /* FASTX numbering */
    int finalized;
    size_t n_keys;
    uint64_t *seqno; /* insertion sequence for stable sort */
};

orc_index *orc_index_new(const orc_spec *spec) {
    orc_index *ix = (orc_index *)calloc(1, sizeof(orc_index));
    ix->spec = *spec;
    return ix;
}
void orc_index_free(orc_index *ix) {
    if (!ix) return;
    free(ix->recs);
    free(ix->seqno);
    free(ix);
}
static void index_push(orc_index *ix, const orc_frag_rec *r) {
    if (ix->n == ix->cap) {
        ix->cap = ix->cap ? ix->cap * 2 : 1024;
        ix->recs = (orc_frag_rec *)realloc(ix->recs, ix->cap * sizeof(orc_frag_rec));
    }
    ix->recs[ix->n++] = *r;
    ix->finalized = 0;
}

/* seq_db.rs:573-615: per-contig 0-based frg_id */
int orc_index_add_seq(orc_index *ix, uint32_t sid, const uint8_t *seq, size_t len) {
    orc_mm128 *sh = NULL;
    size_t n = orc_sequence_to_shmmrs(sid, seq, len, &ix->spec, 0, &sh);
    if (n == (size_t)-1) return -1;
    orc_frag_rec *recs = NULL;
    size_t nr = orc_shmmrs_to_frag_recs(sh, n, sid, 0, &recs);
    for (size_t i = 0; i < nr; i++) index_push(ix, &recs[i]);
    free(recs);
    free(sh);
    return 0;
}

/* seq_db.rs:189-357 numbering: frg_id = frags.len() at the start of the sequence; Prefix +1,
 * each pair +1, Suffix +1; a sequence with no shmmrs pushes 2 fragments (:207-223). */
int orc_index_add_seq_fastx_ids(orc_index *ix, uint32_t sid, const uint8_t *seq, size_t len) {
    orc_mm128 *sh = NULL;
    size_t n = orc_sequence_to_shmmrs(sid, seq, len, &ix->spec, 0, &sh);
    if (n == (size_t)-1) return -1;
    if (n == 0) {
        ix->next_global_frag += 2;
        free(sh);
        return 0;
    }
    orc_frag_rec *recs = NULL;
    size_t nr = orc_shmmrs_to_frag_recs(sh, n, sid, 0, &recs);
    uint32_t frg_id = ix->next_global_frag + 1; /* after the Prefix */
    for (size_t i = 0; i < nr; i++) {
        recs[i].frg_id = frg_id++;
        index_push(ix, &recs[i]);
    }
    ix->next_global_frag = frg_id + 1; /* Suffix */
    free(recs);
    free(sh);
    return 0;
}

typedef struct {
    orc_frag_rec r;
    uint64_t seq;
} rec_sort;

static int rec_cmp(const void *a, const void *b) {
    const rec_sort *x = (const rec_sort *)a, *y = (const rec_sort *)b;
    if (x->r.h0 != y->r.h0) return x->r.h0 < y->r.h0 ? -1 : 1;
    if (x->r.h1 != y->r.h1) return x->r.h1 < y->r.h1 ? -1 : 1;
    if (x->seq != y->seq) return x->seq < y->seq ? -1 : 1;
    return 0;
}

void orc_index_finalize(orc_index *ix) {
    if (ix->finalized) return;
    rec_sort *t = (rec_sort *)malloc(sizeof(rec_sort) * (ix->n ? ix->n : 1));
    for (size_t i = 0; i < ix->n; i++) {
        t[i].r = ix->recs[i];
        t[i].seq = i;
    }
    qsort(t, ix->n, sizeof(rec_sort), rec_cmp);
    size_t nk = 0;
    for (size_t i = 0; i < ix->n; i++) {
        ix->recs[i] = t[i].r;
        if (i == 0 || t[i].r.h0 != t[i - 1].r.h0 || t[i].r.h1 != t[i - 1].r.h1) nk++;
    }
    free(t);
    ix->n_keys = nk;
    ix->finalized = 1;
}
size_t orc_index_n_keys(const orc_index *ix) { return ix->n_keys; }
size_t orc_index_n_recs(const orc_index *ix) { return ix->n; }
const orc_frag_rec *orc_index_recs(const orc_index *ix) { return ix->recs; }

/* lower bound of key in the sorted record array; returns [lo,hi) */
static void index_lookup(const orc_index *ix, uint64_t h0, uint64_t h1, size_t *lo_out,
                         size_t *hi_out) {
    size_t lo = 0, hi = ix->n;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        const orc_frag_rec *r = &ix->recs[mid];
        if (r->h0 < h0 || (r->h0 == h0 && r->h1 < h1)) lo = mid + 1;
        else hi = mid;
    }
    size_t e = lo;
    while (e < ix->n && ix->recs[e].h0 == h0 && ix->recs[e].h1 == h1) e++;
    *lo_out = lo;
    *hi_out = e;
}

/* ------------------------------------------------------------------------------------ */
/* chaining: aln.rs:12-142                                                               */
static void res_init(orc_hps_result *o) { memset(o, 0, sizeof(*o)); }

void orc_hps_result_free(orc_hps_result *o) {
    free(o->targets);
    free(o->chain_score);
    free(o->chain_first_hp);
    free(o->chain_n_hp);
    free(o->hps);
    memset(o, 0, sizeof(*o));
}

static int hp_cmp_full(const orc_hitpair *a, const orc_hitpair *b) {
    if (a->qb != b->qb) return a->qb < b->qb ? -1 : 1;
    if (a->qe != b->qe) return a->qe < b->qe ? -1 : 1;
    if (a->qo != b->qo) return a->qo < b->qo ? -1 : 1;
    if (a->tb != b->tb) return a->tb < b->tb ? -1 : 1;
    if (a->te != b->te) return a->te < b->te ? -1 : 1;
    if (a->to != b->to) return a->to < b->to ? -1 : 1;
    return 0;
}

typedef struct {
    orc_hitpair h;
    size_t idx;
} hp_sort;
static int hp_sort_qb(const void *a, const void *b) { /* stable by qb: aln.rs:21 */
    const hp_sort *x = (const hp_sort *)a, *y = (const hp_sort *)b;
    if (x->h.qb != y->h.qb) return x->h.qb < y->h.qb ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
static int hp_sort_full(const void *a, const void *b) {
    const hp_sort *x = (const hp_sort *)a, *y = (const hp_sort *)b;
    int c = hp_cmp_full(&x->h, &y->h);
    if (c) return c;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

static void res_push_chain(orc_hps_result *o, float score, const orc_hitpair *track, size_t n,
                           size_t *chain_cap, size_t *hp_cap) {
    if (o->n_chains == *chain_cap) {
        *chain_cap = *chain_cap ? *chain_cap * 2 : 16;
        o->chain_score = (float *)realloc(o->chain_score, *chain_cap * sizeof(float));
        o->chain_first_hp = (uint32_t *)realloc(o->chain_first_hp, *chain_cap * sizeof(uint32_t));
        o->chain_n_hp = (uint32_t *)realloc(o->chain_n_hp, *chain_cap * sizeof(uint32_t));
    }
    while (o->n_hps + n > *hp_cap) {
        *hp_cap = *hp_cap ? *hp_cap * 2 : 64;
        o->hps = (orc_hitpair *)realloc(o->hps, *hp_cap * sizeof(orc_hitpair));
    }
    o->chain_score[o->n_chains] = score;
    o->chain_first_hp[o->n_chains] = (uint32_t)o->n_hps;
    o->chain_n_hp[o->n_chains] = (uint32_t)n;
    memcpy(o->hps + o->n_hps, track, n * sizeof(orc_hitpair));
    o->n_hps += n;
    o->n_chains += 1;
}

typedef struct {
    size_t chain_cap, hp_cap;
} res_caps;

static int sparse_aln_impl(orc_hitpair *hits, size_t n, uint32_t max_span, float penalty,
                           int has_max_gap, uint32_t max_gap_u, int oriented, orc_hps_result *out,
                           res_caps *caps) {
    if (n < 2) return -1; /* aln.rs:24 assert */
    /* aln.rs:21 stable sort by query bgn */
    hp_sort *t = (hp_sort *)malloc(sizeof(hp_sort) * n);
    for (size_t i = 0; i < n; i++) {
        t[i].h = hits[i];
        t[i].idx = i;
    }
    qsort(t, n, sizeof(hp_sort), hp_sort_qb);
    for (size_t i = 0; i < n; i++) hits[i] = t[i].h;

    /* value-identity ids: v_s / best_pre_v are FxHashMaps keyed by the HitPair VALUE
     * (aln.rs:22-23), so identical hit pairs share one slot (later insert overwrites). */
    for (size_t i = 0; i < n; i++) {
        t[i].h = hits[i];
        t[i].idx = i;
    }
    qsort(t, n, sizeof(hp_sort), hp_sort_full);
    size_t *id = (size_t *)malloc(sizeof(size_t) * n);
    size_t n_ids = 0;
    for (size_t i = 0; i < n; i++) {
        if (i == 0 || hp_cmp_full(&t[i].h, &t[i - 1].h) != 0) n_ids++;
        id[t[i].idx] = n_ids - 1;
    }
    /* representative (first sorted index) of every id, for canonical iteration order */
    size_t *first_idx = (size_t *)malloc(sizeof(size_t) * n_ids);
    for (size_t q = 0; q < n_ids; q++) first_idx[q] = (size_t)-1;
    for (size_t i = 0; i < n; i++)
        if (first_idx[id[i]] == (size_t)-1) first_idx[id[i]] = i;
    free(t);

    float *v_s = (float *)calloc(n_ids, sizeof(float));
    char *has_s = (char *)calloc(n_ids, 1);
    long *best_pre = (long *)malloc(sizeof(long) * n_ids); /* id of predecessor or -1 */
    for (size_t q = 0; q < n_ids; q++) best_pre[q] = -1;

    /* aln.rs:25-27 */
    v_s[id[0]] = (float)hits[0].qe - (float)hits[0].qb;
    has_s[id[0]] = 1;
    best_pre[id[0]] = -1;

    /* span_set: distinct query coords (qb,qe,qo) seen; small linear set */
    uint32_t (*span)[3] = (uint32_t(*)[3])malloc(sizeof(uint32_t[3]) * (max_span + 1));

    for (size_t i = 1; i < n; i++) { /* aln.rs:29-103 */
        orc_hitpair hp = hits[i];
        long best_v = -1;
        float best_s = 0.0f;
        size_t j = i;
        size_t span_n = 0;
        for (;;) {
            if (j == 0) break;
            j -= 1;
            orc_hitpair pre = hits[j];
            if (oriented) { /* :43-50 */
                uint32_t p_o = pre.qo ^ pre.to;
                uint32_t o = hp.qo ^ hp.to;
                if (p_o != o) continue;
            }
            if (has_max_gap) { /* :52-65 */
                float max_gap = (float)max_gap_u;
                if (hp.qo == hp.to) {
                    float a = (float)hp.qb - (float)pre.qe;
                    float b = (float)hp.tb - (float)pre.te;
                    if ((a < 0 ? -a : a) > max_gap || (b < 0 ? -b : b) > max_gap) continue;
                } else {
                    float a = (float)hp.qb - (float)pre.qe;
                    float b = (float)hp.te - (float)pre.tb;
                    if ((a < 0 ? -a : a) > max_gap || (b < 0 ? -b : b) > max_gap) continue;
                }
            }
            if (pre.qb == hp.qb && pre.qe == hp.qe && pre.qo == hp.qo) continue; /* :67 */
            { /* :70 span_set.insert(pre_hp.0) */
                int found = 0;
                for (size_t q = 0; q < span_n; q++)
                    if (span[q][0] == pre.qb && span[q][1] == pre.qe && span[q][2] == pre.qo) {
                        found = 1;
                        break;
                    }
                if (!found) {
                    span[span_n][0] = pre.qb;
                    span[span_n][1] = pre.qe;
                    span[span_n][2] = pre.qo;
                    span_n++;
                }
            }
            float p_s = has_s[id[j]] ? v_s[id[j]] : 0.0f; /* :71 */
            float s = p_s + ((float)hp.qe - (float)hp.qb); /* :72 */
            if (hp.qo == hp.to) {                          /* :74-78 */
                float a = (float)hp.qb - (float)pre.qe;
                float b = (float)hp.tb - (float)pre.te;
                a = a < 0 ? -a : a;
                b = b < 0 ? -b : b;
                float sum = a + b;
                float pen = penalty * sum;
                s = s - pen;
            } else { /* :79-84 */
                float a = (float)hp.qb - (float)pre.qe;
                float b = (float)hp.te - (float)pre.tb;
                a = a < 0 ? -a : a;
                b = b < 0 ? -b : b;
                float sum = a + b;
                float pen = penalty * sum;
                s = s - pen;
            }
            if (s > best_s) { /* :86-89 */
                best_s = s;
                best_v = (long)id[j];
            }
            if (span_n >= (size_t)max_span) break; /* :91 */
        }
        if (best_s > 0.0f) { /* :96-102 */
            v_s[id[i]] = best_s;
            has_s[id[i]] = 1;
            best_pre[id[i]] = best_v;
        } else {
            v_s[id[i]] = (float)hp.qe - (float)hp.qb;
            has_s[id[i]] = 1;
            best_pre[id[i]] = -1;
        }
    }
    free(span);

    /* extraction: aln.rs:105-140.  unvisited_v is a set of VALUES -> ids. */
    char *unvisited = (char *)malloc(n_ids);
    memset(unvisited, 1, n_ids);
    size_t n_unvisited = n_ids;
    orc_hitpair *track = (orc_hitpair *)malloc(sizeof(orc_hitpair) * n_ids);
    long *track_id = (long *)malloc(sizeof(long) * n_ids);
    /* canonical iteration order: ids by ascending first sorted index */
    size_t *order = (size_t *)malloc(sizeof(size_t) * n_ids);
    {
        size_t q = 0;
        char *seen = (char *)calloc(n_ids, 1);
        for (size_t i = 0; i < n; i++)
            if (!seen[id[i]]) {
                seen[id[i]] = 1;
                order[q++] = id[i];
            }
        free(seen);
    }
    int rc = 0;
    while (n_unvisited > 0) {
        float best_s = 0.0f;
        long best_v = -1;
        for (size_t q = 0; q < n_ids; q++) { /* :112-118 strict >, first wins */
            size_t u = order[q];
            if (!unvisited[u]) continue;
            float s = has_s[u] ? v_s[u] : 0.0f;
            if (s > best_s) {
                best_s = s;
                best_v = (long)u;
            }
        }
        size_t tn = 0;
        long v = best_v;
        int cycle = 0;
        while (v >= 0) { /* :121-128 */
            if (!unvisited[v]) break;
            if (tn == n_ids) { /* best_pre has a cycle (only possible with exact duplicate hit pairs: a value
                                * slot re-scored after a later node chose it): the reference pushes to `track`
                                * forever here */
                cycle = 1;
                break;
            }
            track[tn] = hits[first_idx[v]];
            track_id[tn] = v;
            tn++;
            v = best_pre[v];
        }
        if (tn == 0 || cycle) { /* :129-131 `continue` -> the reference would spin forever */
            rc = -1;
            break;
        }
        for (size_t a = 0, b = tn - 1; a < b; a++, b--) { /* :132 reverse */
            orc_hitpair th = track[a];
            track[a] = track[b];
            track[b] = th;
            long ti = track_id[a];
            track_id[a] = track_id[b];
            track_id[b] = ti;
        }
        for (size_t q = 0; q < tn; q++) { /* :133-137 */
            if (unvisited[track_id[q]]) {
                unvisited[track_id[q]] = 0;
                n_unvisited--;
            }
        }
        float bgn_s = has_s[track_id[0]] ? v_s[track_id[0]] : 0.0f; /* :138 */
        res_push_chain(out, best_s - bgn_s, track, tn, &caps->chain_cap, &caps->hp_cap);
    }
    free(order);
    free(track);
    free(track_id);
    free(unvisited);
    free(v_s);
    free(has_s);
    free(best_pre);
    free(id);
    free(first_idx);
    return rc;
}

int orc_sparse_aln(orc_hitpair *hits, size_t n, uint32_t max_span, float penalty, int has_max_gap,
                   uint32_t max_gap, int oriented, orc_hps_result *out) {
    res_init(out);
    res_caps caps = {0, 0};
    return sparse_aln_impl(hits, n, max_span, penalty, has_max_gap, max_gap, oriented, out, &caps);
}

/* ------------------------------------------------------------------------------------ */
/* query: seq_db.rs:1200-1228 raw_query_fragment + aln.rs:147-242 query_fragment_to_hps  */
typedef struct {
    uint32_t sid;
    uint64_t order;
    orc_hitpair hp;
} sid_hit;

static int sid_hit_cmp(const void *a, const void *b) {
    const sid_hit *x = (const sid_hit *)a, *y = (const sid_hit *)b;
    if (x->sid != y->sid) return x->sid < y->sid ? -1 : 1;
    return x->order < y->order ? -1 : (x->order > y->order);
}

int orc_query_fragment_to_hps(const orc_index *ix, const uint8_t *seq, size_t len, float penalty,
                              uint32_t max_count, uint32_t query_max_count,
                              uint32_t target_max_count, uint32_t max_aln_span, int has_max_gap,
                              uint32_t max_gap, int oriented, orc_hps_result *out) {
    res_init(out);
    if (!ix->finalized) return -2;
    orc_mm128 *sh = NULL;
    size_t n = orc_sequence_to_shmmrs(0, seq, len, &ix->spec, 0, &sh); /* seq_db.rs:1205 */
    if (n == (size_t)-1) return -1;
    orc_frag_rec *q = NULL;
    size_t nq = orc_shmmrs_to_frag_recs(sh, n, 0, 1, &q); /* strict < : seq_db.rs:1213 */
    free(sh);

    /* aln.rs:172-193 counts.  shmmr_pair_hash_count[key] = number of query pairs with that key */
    uint32_t *count = (uint32_t *)calloc(nq ? nq : 1, sizeof(uint32_t));
    for (size_t i = 0; i < nq; i++)
        for (size_t j = 0; j < nq; j++)
            if (q[i].h0 == q[j].h0 && q[i].h1 == q[j].h1) count[i]++;

    sid_hit *sh_hits = NULL;
    size_t n_hits = 0, cap_hits = 0;
    uint64_t order = 0;
    for (size_t i = 0; i < nq; i++) { /* aln.rs:197-228 */
        if (count[i] > max_count) continue;       /* :203-207 */
        if (count[i] > query_max_count) continue; /* :208-211 */
        size_t lo, hi;
        index_lookup(ix, q[i].h0, q[i].h1, &lo, &hi);
        for (size_t s = lo; s < hi; s++) {
            /* target_shmer_pair_count[(key,sid)] = count[i] (query multiplicity) * number of
             * signatures of this key on sid (aln.rs:183-191: every raw hit adds one per sig) */
            uint32_t sid = ix->recs[s].sid;
            uint32_t per_sid = 0;
            for (size_t s2 = lo; s2 < hi; s2++)
                if (ix->recs[s2].sid == sid) per_sid++;
            uint64_t tcount = (uint64_t)per_sid * (uint64_t)count[i];
            if (tcount > (uint64_t)target_max_count) continue; /* :216-222 */
            if (n_hits == cap_hits) {
                cap_hits = cap_hits ? cap_hits * 2 : 256;
                sh_hits = (sid_hit *)realloc(sh_hits, cap_hits * sizeof(sid_hit));
            }
            sid_hit *h = &sh_hits[n_hits++];
            h->sid = sid;
            h->order = order++;
            h->hp.qb = q[i].bgn;
            h->hp.qe = q[i].end;
            h->hp.qo = q[i].orient;
            h->hp.tb = ix->recs[s].bgn;
            h->hp.te = ix->recs[s].end;
            h->hp.to = ix->recs[s].orient;
        }
    }
    free(count);
    free(q);

    qsort(sh_hits, n_hits, sizeof(sid_hit), sid_hit_cmp);
    res_caps caps = {0, 0};
    size_t tcap = 0;
    int rc = 0;
    for (size_t a = 0; a < n_hits;) {
        size_t b = a;
        while (b < n_hits && sh_hits[b].sid == sh_hits[a].sid) b++;
        if (b - a > 1) { /* aln.rs:234 */
            orc_hitpair *hp = (orc_hitpair *)malloc(sizeof(orc_hitpair) * (b - a));
            for (size_t i = a; i < b; i++) hp[i - a] = sh_hits[i].hp;
            size_t chains_before = out->n_chains;
            int r = sparse_aln_impl(hp, b - a, max_aln_span, penalty, has_max_gap, max_gap, oriented,
                                    out, &caps);
            if (r) rc = r;
            free(hp);
            if (out->n_targets == tcap) {
                tcap = tcap ? tcap * 2 : 16;
                out->targets = (orc_target_result *)realloc(out->targets, tcap * sizeof(orc_target_result));
            }
            out->targets[out->n_targets].sid = sh_hits[a].sid;
            out->targets[out->n_targets].chain_first = (uint32_t)chains_before;
            out->targets[out->n_targets].n_chains = (uint32_t)(out->n_chains - chains_before);
            out->n_targets++;
        }
        a = b;
    }
    free(sh_hits);
    return rc;
}

/* ------------------------------------------------------------------------------------ */
/* synthetic contigs: BASELINE.md section 4                                              */
static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void orc_synth_contig(uint64_t seed, uint64_t contig, size_t len, uint8_t *out) {
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
    uint64_t word = 0;
    for (size_t i = 0; i < len; i++) {
        if ((i & 31) == 0) word = splitmix64(seed ^ (contig * 0x9E3779B97F4A7C15ULL) ^ (uint64_t)(i >> 5));
        out[i] = (uint8_t)ACGT[(word >> (2 * (i & 31))) & 3];
    }
}

/* ------------------------------------------------------------------------------------ */
/* threaded batch: one task per contig (seq_db.rs:460-467 rayon par_iter)                */
typedef struct {
    const orc_spec *spec;
    uint32_t n_seqs;
    const uint8_t *const *seqs;
    const uint64_t *lens;
    uint64_t *counts;
    uint32_t next;
    pthread_mutex_t mu;
    uint64_t total;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *job = (batch_job *)arg;
    uint64_t local = 0;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        uint32_t i = job->next++;
        pthread_mutex_unlock(&job->mu);
        if (i >= job->n_seqs) break;
        orc_mm128 *o = NULL;
        size_t n = orc_sequence_to_shmmrs(i, job->seqs[i], (size_t)job->lens[i], job->spec, 0, &o);
        if (n == (size_t)-1) n = 0;
        free(o);
        if (job->counts) job->counts[i] = n;
        local += n;
    }
    pthread_mutex_lock(&job->mu);
    job->total += local;
    pthread_mutex_unlock(&job->mu);
    return NULL;
}

uint64_t orc_shmmr_batch_threads(const orc_spec *spec, uint32_t n_seqs, const uint8_t *const *seqs,
                                 const uint64_t *lens, int n_threads, uint64_t *out_counts) {
    batch_job job;
    job.spec = spec;
    job.n_seqs = n_seqs;
    job.seqs = seqs;
    job.lens = lens;
    job.counts = out_counts;
    job.next = 0;
    job.total = 0;
    pthread_mutex_init(&job.mu, NULL);
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, batch_worker, &job);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th);
    pthread_mutex_destroy(&job.mu);
    return job.total;
}

/* ------------------------------------------------------------------------------------ */
/* Content checksum of a shimmer list: 128 bits per contig, order sensitive (the ordinal of every element is mixed in),
 * additive (so that the GPU can accumulate it in any order).  The same formula is in libpgrhip's
 * shmmr_checksum_kernel (pgr-tk_amd/csrc/level2.hip); bench.py compares the two for ALL contigs of BASELINE.json
 * configs[1].  Only x and the low 32 bits of y (pos << 1 | strand) enter: the rid field is the caller's choice. */
void orc_shmmr_checksum(const orc_mm128 *mm, size_t n, uint64_t out[2]) {
    uint64_t a = 0, b = 0;
    for (size_t i = 0; i < n; i++) {
        const uint64_t x = mm[i].x, ylo = mm[i].y & 0xFFFFFFFFULL;
        a += splitmix64(x ^ (0x9E3779B97F4A7C15ULL * (uint64_t)(i + 1)));
        b += splitmix64((ylo | ((uint64_t)i << 32)) + 0xD1B54A32D192ED03ULL * x);
    }
    out[0] = a;
    out[1] = b;
}

/* one task per contig (= rayon par_iter, seq_db.rs:460-467); every worker GENERATES its contig (counter-based
 * generator), runs sequence_to_shmmrs and keeps only count + checksum, so host memory stays at
 * n_threads contigs.  busy_s[t] = seconds thread t spent inside orc_sequence_to_shmmrs (generation not included). */
typedef struct {
    const orc_spec *spec;
    uint32_t n;
    uint64_t seed, contig0;
    const uint64_t *ids; /* NULL: contig0 + i */
    size_t len;
    uint64_t *counts, *sums;
    double *busy;
    uint32_t next;
    pthread_mutex_t mu;
} synth_job;

typedef struct {
    synth_job *job;
    int tid;
} synth_arg;

#include <time.h>
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *synth_worker(void *argp) {
    synth_arg *sa = (synth_arg *)argp;
    synth_job *job = sa->job;
    uint8_t *buf = (uint8_t *)malloc(job->len ? job->len : 1);
    double busy = 0.0;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        uint32_t i = job->next++;
        pthread_mutex_unlock(&job->mu);
        if (i >= job->n) break;
        orc_synth_contig(job->seed, job->ids ? job->ids[i] : job->contig0 + i, job->len, buf);
        orc_mm128 *o = NULL;
        const double t0 = now_s();
        size_t n = orc_sequence_to_shmmrs(i, buf, job->len, job->spec, 0, &o);
        busy += now_s() - t0;
        if (n == (size_t)-1) n = 0;
        job->counts[i] = n;
        orc_shmmr_checksum(o, n, job->sums + 2 * (size_t)i);
        free(o);
    }
    free(buf);
    job->busy[sa->tid] = busy;
    return NULL;
}

int orc_synth_checksums_threads(const orc_spec *spec, uint32_t n, uint64_t seed, uint64_t contig0, size_t len,
                                int n_threads, uint64_t *counts, uint64_t *sums, double *busy_s) {
    return orc_synth_checksums_ids_threads(spec, n, seed, contig0, NULL, len, n_threads, counts, sums, busy_s);
}

/* the same for an arbitrary list of contig ids (one rank's shard of a partitioned contig set); ids == NULL: contig0 + i */
int orc_synth_checksums_ids_threads(const orc_spec *spec, uint32_t n, uint64_t seed, uint64_t contig0, const uint64_t *ids,
                                    size_t len, int n_threads, uint64_t *counts, uint64_t *sums, double *busy_s) {
    synth_job job;
    job.spec = spec;
    job.n = n;
    job.seed = seed;
    job.ids = ids;
    job.contig0 = contig0;
    job.len = len;
    job.counts = counts;
    job.sums = sums;
    job.busy = busy_s;
    job.next = 0;
    pthread_mutex_init(&job.mu, NULL);
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    synth_arg *args = (synth_arg *)malloc(sizeof(synth_arg) * (size_t)n_threads);
    for (int t = 0; t < n_threads; t++) {
        args[t].job = &job;
        args[t].tid = t;
        pthread_create(&th[t], NULL, synth_worker, &args[t]);
    }
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(args);
    pthread_mutex_destroy(&job.mu);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* query baseline: one task per query (= the rayon loop of pgr-query.rs:135-138) against one finalized index; the
 * results are kept (results[i]) so that the caller can compare chain CONTENT with the GPU's. */
typedef struct {
    const orc_index *ix;
    uint32_t n;
    const uint8_t *const *seqs;
    const uint64_t *lens;
    float penalty;
    uint32_t max_count, query_max_count, target_max_count, max_aln_span;
    int has_max_gap;
    uint32_t max_gap;
    int oriented;
    orc_hps_result *results;
    int *rcs;
    uint32_t next;
    pthread_mutex_t mu;
} query_job;

static void *query_worker(void *argp) {
    query_job *job = (query_job *)argp;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        uint32_t i = job->next++;
        pthread_mutex_unlock(&job->mu);
        if (i >= job->n) break;
        job->rcs[i] = orc_query_fragment_to_hps(job->ix, job->seqs[i], (size_t)job->lens[i], job->penalty, job->max_count,
                                                job->query_max_count, job->target_max_count, job->max_aln_span,
                                                job->has_max_gap, job->max_gap, job->oriented, &job->results[i]);
    }
    return NULL;
}

int orc_query_batch_threads(const orc_index *ix, uint32_t n, const uint8_t *const *seqs, const uint64_t *lens, float penalty,
                            uint32_t max_count, uint32_t query_max_count, uint32_t target_max_count,
                            uint32_t max_aln_span, int has_max_gap, uint32_t max_gap, int oriented, int n_threads,
                            orc_hps_result *results, int *rcs) {
    query_job job;
    job.ix = ix;
    job.n = n;
    job.seqs = seqs;
    job.lens = lens;
    job.penalty = penalty;
    job.max_count = max_count;
    job.query_max_count = query_max_count;
    job.target_max_count = target_max_count;
    job.max_aln_span = max_aln_span;
    job.has_max_gap = has_max_gap;
    job.max_gap = max_gap;
    job.oriented = oriented;
    job.results = results;
    job.rcs = rcs;
    job.next = 0;
    pthread_mutex_init(&job.mu, NULL);
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, query_worker, &job);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th);
    pthread_mutex_destroy(&job.mu);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* index over n synthetic contigs (sid = sid0 + i, contig id = contig0 + i): the pair records of every contig are
 * computed by a pool of threads (one task per contig, seq_db.rs:460-467), then inserted serially in sid order exactly as
 * load_index_from_seq_vec does (seq_db.rs:605-612).  For bench.py's query baseline: an index of a bounded sample. */
typedef struct {
    const orc_spec *spec;
    uint32_t n, sid0;
    uint64_t seed, contig0;
    size_t len;
    orc_frag_rec **recs;
    size_t *n_recs;
    uint32_t next;
    pthread_mutex_t mu;
} synth_index_job;

static void *synth_index_worker(void *argp) {
    synth_index_job *job = (synth_index_job *)argp;
    uint8_t *buf = (uint8_t *)malloc(job->len ? job->len : 1);
    for (;;) {
        pthread_mutex_lock(&job->mu);
        uint32_t i = job->next++;
        pthread_mutex_unlock(&job->mu);
        if (i >= job->n) break;
        orc_synth_contig(job->seed, job->contig0 + i, job->len, buf);
        orc_mm128 *sh = NULL;
        size_t ns = orc_sequence_to_shmmrs(job->sid0 + i, buf, job->len, job->spec, 0, &sh);
        if (ns == (size_t)-1) ns = 0;
        job->n_recs[i] = orc_shmmrs_to_frag_recs(sh, ns, job->sid0 + i, 0, &job->recs[i]);
        free(sh);
    }
    free(buf);
    return NULL;
}

int orc_index_add_synth_threads(orc_index *ix, uint32_t n, uint32_t sid0, uint64_t seed, uint64_t contig0, size_t len,
                                int n_threads) {
    synth_index_job job;
    job.spec = &ix->spec;
    job.n = n;
    job.sid0 = sid0;
    job.seed = seed;
    job.contig0 = contig0;
    job.len = len;
    job.recs = (orc_frag_rec **)calloc(n ? n : 1, sizeof(orc_frag_rec *));
    job.n_recs = (size_t *)calloc(n ? n : 1, sizeof(size_t));
    job.next = 0;
    pthread_mutex_init(&job.mu, NULL);
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, synth_index_worker, &job);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    for (uint32_t i = 0; i < n; i++) {
        for (size_t j = 0; j < job.n_recs[i]; j++) index_push(ix, &job.recs[i][j]);
        free(job.recs[i]);
    }
    free(th);
    free(job.recs);
    free(job.n_recs);
    pthread_mutex_destroy(&job.mu);
    return 0;
}
