/*
 * pgr_oracle.h -- CPU oracle for the SHIMMER index/query hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a sequential, line-faithful C restatement of the
 * reference algorithm (GeneDx/pgr-tk, Rust).  It exists so that the HIP path can be
 * checked bit-for-bit; it is never linked, imported or called from the product
 * (pgr-tk_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it.
 *
 * Pinning: the restatement is checked (tests/test_oracle_golden.py) against the
 * reference's own golden index fixture pgr-db/test/test_data/test_seqs_frag.mdb
 * (820 fragment signatures / 55 keys at w=80,k=56,r=4,min_span=64) and against the
 * reference's known-answer tests pgr-db/src/lib.rs:166-180 (rc_match) and :342-363
 * (reduction boundary condition).  The chaining part (sparse_aln) has NO expected
 * output anywhere in the reference (aln.rs:484 "TODO: Test the output properly"), so
 * chaining parity is "unpinned": oracle <-> GPU equality only.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference checkout root).
 */
#ifndef PGR_ORACLE_H
#define PGR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pgr-db/src/shmmrutils.rs:225-229 */
typedef struct {
    uint64_t x; /* (hash << 8) | k          */
    uint64_t y; /* rid << 32 | pos << 1 | strand */
} orc_mm128;

/* pgr-db/src/shmmrutils.rs:20-27 */
typedef struct {
    uint32_t w, k, r, min_span;
    uint32_t sketch; /* bool */
} orc_spec;

/* index record: pgr-db/src/seq_db.rs:381-400 + FragmentSignature seq_db.rs:75 */
typedef struct {
    uint64_t h0, h1;
    uint32_t frg_id, sid, bgn, end;
    uint32_t orient;
} orc_frag_rec;

/* HitPair: pgr-db/src/aln.rs:10 */
typedef struct {
    uint32_t qb, qe, qo;
    uint32_t tb, te, to;
} orc_hitpair;

uint64_t orc_u64hash(uint64_t key);

/* level-1 only (before reduce / min_span) -- for KATs. Returns count; *out malloc'ed. */
size_t orc_level1_minimizers(uint32_t rid, const uint8_t *seq, size_t len,
                             uint32_t w, uint32_t k, orc_mm128 **out);

size_t orc_reduce_shmmr(const orc_mm128 *mers, size_t n, uint32_t r, int padding,
                        orc_mm128 **out);

/* shmmrutils.rs:657-669. Returns count, *out malloc'ed (free with orc_free).
 * Returns (size_t)-1 when the reference would assert (k>56, w>128, r not in 1..12). */
size_t orc_sequence_to_shmmrs(uint32_t rid, const uint8_t *seq, size_t len,
                              const orc_spec *spec, int padding, orc_mm128 **out);

/* seq_db.rs:360-418 (index side, s0 <= s1) / seq_db.rs:1205-1217 (query side, s0 < s1) */
size_t orc_shmmrs_to_frag_recs(const orc_mm128 *shmmrs, size_t n, uint32_t sid,
                               int query_side, orc_frag_rec **out);

/* ---- index (frag_map as a sorted CSR; per-key order = insertion order) ---- */
typedef struct orc_index orc_index;
orc_index *orc_index_new(const orc_spec *spec);
void orc_index_free(orc_index *);
/* seq_db.rs:573-615 load_index_from_seq_vec semantics for one sequence (sid given) */
int orc_index_add_seq(orc_index *, uint32_t sid, const uint8_t *seq, size_t len);
/* FASTX/MEMORY backend numbering (seq_db.rs:189-357): global frag ids */
int orc_index_add_seq_fastx_ids(orc_index *, uint32_t sid, const uint8_t *seq, size_t len);
void orc_index_finalize(orc_index *);
size_t orc_index_n_keys(const orc_index *);
size_t orc_index_n_recs(const orc_index *);
/* sorted by (h0,h1), within key insertion order */
const orc_frag_rec *orc_index_recs(const orc_index *);

/* ---- query (seq_db.rs:1200-1228 + aln.rs:12-242) ---- */
typedef struct {
    uint32_t sid;
    uint32_t n_chains;
    uint32_t chain_first; /* index into chains arrays */
} orc_target_result;

typedef struct {
    size_t n_targets;
    orc_target_result *targets; /* sorted by sid (reference order = hash order, unspecified) */
    size_t n_chains;
    float *chain_score;
    uint32_t *chain_first_hp;
    uint32_t *chain_n_hp;
    size_t n_hps;
    orc_hitpair *hps;
} orc_hps_result;

/* has_max_gap==0 -> None.  Returns 0 on success. */
int orc_query_fragment_to_hps(const orc_index *, const uint8_t *seq, size_t len, float penalty,
                              uint32_t max_count, uint32_t query_max_count,
                              uint32_t target_max_count, uint32_t max_aln_span, int has_max_gap,
                              uint32_t max_gap, int oriented, orc_hps_result *out);
void orc_hps_result_free(orc_hps_result *);

/* aln.rs:12-142.  hits are sorted in place (stable by qb).  Chains appended to the result
 * arrays in extraction order.  Tie-break between equal best scores: lowest sorted index
 * (reference: FxHashSet iteration order -- unspecified). Returns 0 ok, -1 if the reference
 * would loop forever / assert. */
int orc_sparse_aln(orc_hitpair *hits, size_t n, uint32_t max_span, float penalty, int has_max_gap,
                   uint32_t max_gap, int oriented, orc_hps_result *out);

/* ---- synthetic contigs ---- */
void orc_synth_contig(uint64_t seed, uint64_t contig, size_t len, uint8_t *out_ascii);

/* threaded CPU baseline: one task per contig (= rayon par_iter, seq_db.rs:460-467).
 * Returns total number of final shmmrs; fills out_counts[n_seqs] if non-NULL. */
uint64_t orc_shmmr_batch_threads(const orc_spec *spec, uint32_t n_seqs, const uint8_t *const *seqs,
                                 const uint64_t *lens, int n_threads, uint64_t *out_counts);

void orc_free(void *);

/* ---- full-size content checks and threaded baselines (bench.py cpu_baseline legs) ---- */
/* 128-bit order-sensitive checksum of a shimmer list (x and the low 32 bits of y); same formula on the GPU */
void orc_shmmr_checksum(const orc_mm128 *mm, size_t n, uint64_t out[2]);
/* n synthetic contigs of `len` bases (orc_synth_contig(seed, contig0 + i)), generated inside the workers, one task per
 * contig: counts[n], sums[2n], busy_s[n_threads] = seconds each thread spent inside orc_sequence_to_shmmrs */
int orc_synth_checksums_threads(const orc_spec *spec, uint32_t n, uint64_t seed, uint64_t contig0, size_t len,
                                int n_threads, uint64_t *counts, uint64_t *sums, double *busy_s);
int orc_synth_checksums_ids_threads(const orc_spec *spec, uint32_t n, uint64_t seed, uint64_t contig0, const uint64_t *ids,
                                    size_t len, int n_threads, uint64_t *counts, uint64_t *sums, double *busy_s);
/* index over n synthetic contigs (sid0 + i <- contig0 + i): records computed by a thread pool, inserted in sid order */
int orc_index_add_synth_threads(orc_index *ix, uint32_t n, uint32_t sid0, uint64_t seed, uint64_t contig0, size_t len,
                                int n_threads);
/* one task per query on n_threads (the rayon loop of pgr-query.rs:135-138); results[n], rcs[n] */
int orc_query_batch_threads(const orc_index *ix, uint32_t n, const uint8_t *const *seqs, const uint64_t *lens, float penalty,
                            uint32_t max_count, uint32_t query_max_count, uint32_t target_max_count,
                            uint32_t max_aln_span, int has_max_gap, uint32_t max_gap, int oriented, int n_threads,
                            orc_hps_result *results, int *rcs);

#ifdef __cplusplus
}
#endif
#endif
