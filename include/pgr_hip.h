/*
 * pgr_hip.h -- C ABI of libpgrhip.so: MI355X (gfx950) SHIMMER minimizer indexing and
 * sparse hit chaining.  This is the drop-in boundary for ONE hot path of GeneDx/pgr-tk
 * (SURVEY.md section 8b).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns PGR_OK (0) or a negative pgr_status; the message of the last
 *     failure is available from pgr_last_error(ctx).  Nothing aborts or throws across the
 *     boundary (the reference panics / asserts instead: shmmrutils.rs:443-445).
 *   - inputs are borrowed for the duration of the call.  Host outputs are allocated by the
 *     library and released with pgr_free() (the reference's only existing C-ABI precedent,
 *     the AGC FFI, uses the same pattern: pgr-db/src/agc_io.rs:76-116, agc_list_destroy).
 *   - a context is bound to one GPU and is NOT thread safe: one context per (thread, GPU).
 *     The reference calls sequence_to_shmmrs once per contig from a rayon par_iter
 *     (pgr-db/src/seq_db.rs:456-469); the replacement takes the whole batch in one call.
 *   - there is no CPU fallback: without a usable gfx950 device pgr_ctx_create fails.
 *
 * All structs are POD, little endian, layout-identical to the Rust types they stand for.
 */
#ifndef PGR_HIP_H
#define PGR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PGR_OK = 0,
    PGR_ERR_INVALID_ARG = -1, /* NULL pointer, bad size ...                                   */
    PGR_ERR_BAD_SPEC = -2,    /* k>56, w>128, r not in 1..12 (reference: assert!)             */
    PGR_ERR_DEVICE = -3,      /* HIP runtime error / no gfx950 device                         */
    PGR_ERR_NOMEM = -4,
    PGR_ERR_TOO_LONG = -5,    /* contig length >= 2^31 (MM128.y holds pos in 31 bits)         */
    PGR_ERR_STATE = -6,       /* call order / handle misuse                                   */
    PGR_ERR_INTERNAL = -7
} pgr_status;

/* ShmmrSpec: pgr-db/src/shmmrutils.rs:20-27 (serialised as 5 x u32 in .mdb, seq_db.rs:1302-1306) */
typedef struct {
    uint32_t w, k, r, min_span;
    uint32_t sketch; /* bool */
} pgr_spec;

/* MM128: pgr-db/src/shmmrutils.rs:225-269.  x = hash<<8 | k ; y = rid<<32 | pos<<1 | strand */
typedef struct {
    uint64_t x, y;
} pgr_mm128;

/* one shimmer-pair record = key + FragmentSignature
 * ((u64,u64),u32,u32,u8) of seq_db.rs:381-400 + (frg_id, sid) of seq_db.rs:605-612 */
typedef struct {
    uint64_t h0, h1;  /* ShmmrPair key, h0 <= h1                         */
    uint32_t frg_id;  /* pair ordinal within the contig (index-only path) */
    uint32_t sid;     /* sequence id                                      */
    uint32_t bgn, end;
    uint32_t orient;  /* 0: hash(s0) <= hash(s1) (index side) / < (query side) */
    uint32_t _pad;
} pgr_frag_rec;

/* HitPair: pgr-db/src/aln.rs:10 */
typedef struct {
    uint32_t qb, qe, qo;
    uint32_t tb, te, to;
} pgr_hitpair;

typedef struct pgr_ctx pgr_ctx;
typedef struct pgr_batch pgr_batch;   /* a batch of contigs resident on the GPU, 2-bit packed */
typedef struct pgr_shmmrs pgr_shmmrs; /* device-resident result of one sequence_to_shmmrs pass */
typedef struct pgr_index pgr_index;   /* ShmmrToFrags as a GPU/host CSR                        */

/* ------------------------------------------------------------------ context */
int pgr_ctx_create(int device, pgr_ctx **out);
/* A second context on other's device whose work runs BESIDE other's (its stream is tried against other's until the two do not
 * share a hardware queue).  A context is not thread safe, a finalized index may be queried from several contexts at once: one
 * context per host thread and one index is the counterpart of the reference's rayon loop over the queries
 * (pgr-bin/src/bin/pgr-query.rs:135-165) -- two query batches in flight take 0.41 ms per batch instead of 0.69. */
int pgr_ctx_create_beside(pgr_ctx *other, pgr_ctx **out);
void pgr_ctx_destroy(pgr_ctx *ctx);
const char *pgr_last_error(const pgr_ctx *ctx); /* ctx may be NULL: last create error */
/* Every block the library hands out (*out_mm, *out_off, records, results ...) is released with pgr_free and ONLY with pgr_free:
 * large results are pinned blocks of a process-wide pool (the DMA engine wrote them directly), not malloc'd memory -- free()
 * on one of them is undefined behaviour. */
void pgr_free(void *p);
const char *pgr_version(void);
/* Tuning and A/B switches of a context.  Defaults come from the environment ONCE, at pgr_ctx_create (PGR_<NAME IN UPPER
 * CASE>=<integer>); afterwards only these two calls change or read them -- no entry point looks at the environment.
 * (Process wide, read once at first use by the host-side packer and its thread pool -- csrc/hostpack.cpp --, not per context:
 * PGR_HOST_THREADS=<n> threads of the pool instead of the CPUs the process may use; PGR_NO_AVX2 / PGR_NO_AVX512 force the
 * narrower packers.)
 *   debug, debug_times        progress lines / a host-side timeline on stderr
 *   no_small_path             never the one-workgroup-per-contig kernel for batches of short contigs
 *   no_pipeline               never cut a large host batch into staged sub-batches
 *   early_sync_bp             batches of at least this many bases read the level-1 flags before the list stage (2^30)
 *   gpu_pack                  ASCII over PCIe + pack kernel instead of the CPU packer (the round-2 host path)
 *   index_full_sort, index_two_key_sort    pgr_index_finalize: force the four-field / the two-key sort
 *   no_fused_query, no_query_chaining, query_global_sort, fused_query_hits   query path variants
 *   no_query_level1   query batches never take the level-1 form of the per-query kernel (tile kernel + per-query kernel on the tile
 *                     segments, no list stage of the batch: pgr_query_prof.path 3)
 *   no_query_keys     pgr_index_finalize builds no per-key table for that kernel (32 B per key: the key with its record when it has one)
 *   exchange_timeout_s        watchdog of pgr_exchange_*: bound on loading librccl.so.1, ncclGetUniqueId and ncclCommInitRank (the
 *                             rendezvous of the ranks) (300; 0 = wait for ever); on a timeout the call fails and its message
 *                             names the step that did not return
 *   debug_poison              debugging aid: every device block that is about to be used again is filled with 0xFF first (blocks leaving
 *                             the caching allocator, grown workspaces, a job's workspaces when it is planned, the status mailbox) and a
 *                             list stage checks the segment table it is about to read; a call that read what an earlier call left
 *                             behind fails with PGR_ERR_INTERNAL ("debug_poison: ...") instead of faulting or not, depending on the layout
 *   debug_inject_stale_segments   fault injection for the tests of debug_poison (re-creates the round-5 defect); never set it otherwise
 *   exchange_rccl_world1      an exchange of ONE rank uses a real RCCL communicator.  Default 0: every collective of one rank is
 *                             a device-to-device copy on the exchange's stream and RCCL is neither loaded nor initialised
 *                             (pgr_exchange_create then ignores `id`, which may be NULL)
 *   exchange_collective_timeout_s   the same for every wait for a collective -- which is also a wait for the slowest rank to get
 *                             there, so it is generous (1800; 0 = wait for ever).  A value that is not a number leaves a numeric
 *                             option at its default (a line on stderr says so).
 *   no_island_relay           exact islands: the round-3 seam correction (one seam per host round), for A/B timing
 *   island_settle             > 0: positions behind an array of palindromic k-mers at which its island ends (default 2 w + k + 64 rounded
 *                             up to 64; a machine that arrives there stuck moves the end on itself), for A/B
 *   no_sub_tile_islands       exact islands made of whole tiles only (rounds 3-5), for A/B: by default an island around palindromic
 *                             k-mers begins and ends inside the tiles the tile kernel reports them in
 *   no_short_tiles            batches of short contigs (mean length <= 2048): 4096-position tiles all the same, for A/B timing
 *   no_pre_islands            never list the islands around non-ACGT bytes while the tile kernel is still running, for A/B timing
 *   no_early_islands          ... never start their first round before the tile kernel's flags are seen, for A/B timing
 *   early_islands_in_stream   ... start it behind the tile kernel on the context's stream, not beside it on a stream of its own, for A/B
 *   no_early_merge            ... drop that early round when the tile kernel's flags add islands (tiles with a palindromic k-mer) instead
 *                             of keeping it and running only the added islands behind it, for A/B
 *   island_chunk_min          > 0: shortest chunk of the exact machine in positions (default 1024; 4096 = the round-3 minimum), for A/B
 *   back_priority             pgr_pipe: stream priority of the back stream (1 = highest, 0 = default, -1 = lowest), read when the
 *                             context's first pipe is created
 *   no_direct_h2d             packed input in pinned host memory is copied through the staging windows all the same, for A/B
 *   lds_match                 pgr_pipe: the back stream's kernels occupy exactly the tile kernel's LDS per workgroup, or none, for A/B
 *   pipe_small_list           pgr_pipe: the list kernel of a pipelined job runs 512-element workgroups (14 KB of LDS), for A/B
 *   front_priority            1: the context's stream gets the device's highest priority (environment only: read at pgr_ctx_create), for A/B
 *   pipe_staged_records       pgr_pipe: index jobs always stage their records and copy them in when collected, for A/B
 *   no_fix_stream             pgr_pipe: a job that needs a second pass (flagged tiles: every batch of a real assembly) is finished on the
 *                             context's stream, behind the next job's tiles, instead of on a stream of its own beside them, for A/B
 *   no_stage1_only            pgr_pipe: a job's first pass always includes its list stage, even when the job before it needed islands
 *                             (then that list stage is thrown away at collect), for A/B
 *   direct_query_result       batches of short queries: the per-query kernel writes the host's result block itself (single pass, sections
 *                             placed from the previous batch's counts); an experiment, measured slower than two passes + download
 *   direct_query_lds_kb       ... with this much LDS per workgroup (fewer queries resident at once); -1: its block in device memory
 *                             (timing only: no result is delivered).  direct_query_results_delivered: a counter (read only)
 * Unknown names: PGR_ERR_INVALID_ARG. */
/* Device memory of a context.  Results, batches and indexes come from a caching allocator (a released block is kept for the next
 * request of its size: the steady state of a loop over batches allocates nothing); pgr_ctx_trim gives the cached blocks (not
 * the live ones, not the workspaces) back to the device -- after a phase whose buffers the next one has no use for.
 * pgr_ctx_mem_stats: bytes the allocator holds now (live + cached) and the most it has held since the context was created or
 * the peak was last reset (reset_peak != 0 starts a new measurement). */
int pgr_ctx_trim(pgr_ctx *ctx);
int pgr_ctx_mem_stats(pgr_ctx *ctx, uint64_t *held_bytes, uint64_t *peak_bytes, int reset_peak);
/* pgr_ctx_reserve: ONE block of device memory, allocated and touched here, that every later device allocation of the context
 * (workspaces, batches, results, indexes, the caching allocator's blocks) is carved from and returns to.  For a host that runs
 * one build per process (pgr-mdb: one shot per file list, pgr-bin/src/bin/pgr-mdb.rs:53-111): a multi-GB hipMalloc in the middle of
 * a build can block for seconds, and device memory is slow at its first use (~26 ms per GiB) -- after a reserve neither happens
 * inside the build.  Sizing: pgr_ctx_mem_stats' peak of an earlier run of the same shape, or ~5.4 bytes per base of TWO batches in
 * flight + 40 bytes per expected pair record.  May be called again (grow-only: another block is added).  A request the arena
 * cannot serve goes to the runtime's allocator as before and is counted (pgr_ctx_arena_stats: fallback_bytes / fallback_calls).
 * Nothing is ever returned to the device before pgr_ctx_destroy.  Any out pointer of pgr_ctx_arena_stats may be NULL. */
int pgr_ctx_reserve(pgr_ctx *ctx, uint64_t bytes);
int pgr_ctx_arena_stats(pgr_ctx *ctx, uint64_t *reserved_bytes, uint64_t *used_bytes, uint64_t *peak_used_bytes,
                        uint64_t *fallback_bytes, uint64_t *fallback_calls);
int pgr_ctx_set_option(pgr_ctx *ctx, const char *name, int64_t value);
/* pgr_debug_take_hip_error: the HIP runtime keeps, per host thread, the last error any of its calls returned until somebody asks
 * for it -- and rocPRIM asks after every launch, so an error some earlier call (of this library, of PyTorch, of the host program)
 * left behind used to surface as the failure of an unrelated scan.  The library's rocPRIM wrappers now take stale state away before
 * they call rocPRIM; this entry point takes it (hipGetLastError) and returns it as an int (0: none) -- the GPU tests call it after
 * every test and fail the test that left one.  No reference counterpart (debug aid). */
int pgr_debug_take_hip_error(void);
int pgr_ctx_get_option(const pgr_ctx *ctx, const char *name, int64_t *value);

/* ------------------------------------------------------------------ B1: sequence_to_shmmrs
 * Replaces shmmrutils::sequence_to_shmmrs (pgr-db/src/shmmrutils.rs:657-669) as called
 * batched from CompactSeqDB::get_shmmrs_from_seqs (pgr-db/src/seq_db.rs:456-469):
 *   seqs[i]  ASCII bytes of contig i (same byte semantics as shmmrutils.rs:426-436)
 *   rids[i]  value placed in MM128.y >> 32 (NULL: rid = i)
 *   padding  as the reference's `padding` argument (only Python get_shmmr_pairs_from_seq
 *            passes true, pgr-tk/src/lib.rs:1597)
 * out:  *out_mm    concatenated MM128 of all contigs (pgr_free)
 *       *out_off   n_seqs+1 offsets into *out_mm    (pgr_free)                            */
int pgr_shmmr_batch(pgr_ctx *ctx, const pgr_spec *spec, uint32_t n_seqs,
                    const uint8_t *const *seqs, const uint64_t *lens, const uint32_t *rids,
                    int padding, pgr_mm128 **out_mm, uint64_t **out_off);

/* Fused B1 + pair_shmmrs/seq_to_index (pgr-db/src/seq_db.rs:102-111, 360-418): one record
 * per adjacent shimmer pair, frg_id = pair ordinal in the contig, sid = sids[i] (NULL: i).
 * query_side != 0 uses the strict `<` of raw_query_fragment (seq_db.rs:1213).             */
int pgr_frag_recs_batch(pgr_ctx *ctx, const pgr_spec *spec, uint32_t n_seqs,
                        const uint8_t *const *seqs, const uint64_t *lens, const uint32_t *sids,
                        int query_side, pgr_frag_rec **out_recs, uint64_t **out_off);

/* ------------------------------------------------------------------ packed host input
 * "pack ASCII -> 2-bit + N-mask (or accept pre-packed)".  A host that keeps -- or
 * produces -- its sequences 2-bit packed hands them over in the layout pgr_batch keeps in HBM, and 0.25 B per base
 * (0.375 with a validity plane) cross PCIe instead of 1 B:
 *   contig c owns words [woff[c], woff[c+1]),  woff[c] = sum_{i<c} ceil(lens[i] / 32)   (pgr_packed_words = woff[n])
 *   word j of a contig holds bases 32j .. 32j+31, base i at bit 31 - (i % 32)  (MSB first: a k-mer's fmmer.0 / fmmer.1,
 *   shmmrutils.rs:462-467, is a plain bit field of consecutive words)
 *   planes[w] = low bits of the 2-bit codes | high bits << 32    (A/a/0 -> 0, C/c/1 -> 1, G/g/2 -> 2, T/t/3 -> 3,
 *                                                                  shmmrutils.rs:426-436)
 *   valid[w]  = bit set: the byte was a base; NULL: every base of every contig is valid
 * Bits past a contig's end and plane bits of invalid positions are ignored (cleaned on the device).
 * pgr_pack_ascii is the library's own threaded CPU packer (AVX2 when the CPU has it; no GPU involved): ASCII contigs ->
 * planes / valid arrays of pgr_packed_words(n, lens) words each; n_threads <= 0: all CPUs the process may use;
 * *n_invalid (may be NULL) = number of non-ACGT bytes.  pgr_shmmr_batch / pgr_batch_from_ascii use the same packer
 * internally, writing straight into their pinned staging windows. */
uint64_t pgr_packed_words(uint32_t n_seqs, const uint64_t *lens);
int pgr_pack_ascii(uint32_t n_seqs, const uint8_t *const *seqs, const uint64_t *lens, int n_threads, uint64_t *planes,
                   uint32_t *valid, uint64_t *n_invalid);
/* A host that keeps its packed sequences in long-lived buffers pins them ONCE: pgr_host_register (hipHostRegister; the buffer
 * must stay where it is until pgr_host_unregister, which must come before the host frees it; memory from hipHostMalloc needs
 * neither).  The packed entry points recognise pinned planes (and validity plane) and let the DMA engine read them where they
 * lie: no copy into the library's staging windows, the link is the only stage left.  The validity plane is read on the host and
 * crosses the link only for stretches that hold a byte that is not a base.  (Errors: pgr_last_error(NULL).) */
int pgr_host_register(void *p, size_t bytes);
int pgr_host_unregister(void *p);
/* B1 with packed input: everything else as pgr_shmmr_batch */
int pgr_shmmr_batch_packed(pgr_ctx *ctx, const pgr_spec *spec, uint32_t n_seqs, const uint64_t *lens,
                           const uint64_t *planes, const uint32_t *valid, const uint32_t *rids, int padding,
                           pgr_mm128 **out_mm, uint64_t **out_off);

/* ------------------------------------------------------------------ device-resident path
 * (what bench.py times: inputs already in HBM, results left in HBM)                        */

/* ASCII -> 2-bit planes + validity plane (host threads, into pinned windows) + H2D */
int pgr_batch_from_ascii(pgr_ctx *ctx, uint32_t n_seqs, const uint8_t *const *seqs,
                         const uint64_t *lens, pgr_batch **out);
/* H2D of packed planes (layout above) */
int pgr_batch_from_packed(pgr_ctx *ctx, uint32_t n_seqs, const uint64_t *lens, const uint64_t *planes,
                          const uint32_t *valid, pgr_batch **out);
/* counter-based synthetic contigs generated on the device:
 * base(c,i) = (splitmix64(seed ^ c*0x9E3779B97F4A7C15 ^ (i>>5)) >> (2*(i&31))) & 3,
 * c = contig0 + index (the CPU checker generates the same bytes).         */
int pgr_batch_synthetic(pgr_ctx *ctx, uint32_t n_seqs, const uint64_t *lens, uint64_t seed,
                        uint64_t contig0, pgr_batch **out);
/* the same for an arbitrary list of global contig ids (one rank's shard of a partitioned contig set) */
int pgr_batch_synthetic_ids(pgr_ctx *ctx, uint32_t n_seqs, const uint64_t *lens, uint64_t seed,
                            const uint64_t *contig_ids, pgr_batch **out);
void pgr_batch_destroy(pgr_batch *b);
uint64_t pgr_batch_total_bases(const pgr_batch *b);

/* the hot path on a resident batch; result stays on the device.  rids may be NULL (rid=i). */
int pgr_shmmrs_compute(pgr_ctx *ctx, const pgr_batch *b, const pgr_spec *spec,
                       const uint32_t *rids, int padding, pgr_shmmrs **out);
uint64_t pgr_shmmrs_count(const pgr_shmmrs *s);
/* device pointers (valid until pgr_shmmrs_destroy): MM128[count], u64 offsets[n_seqs+1] */
const pgr_mm128 *pgr_shmmrs_device_ptr(const pgr_shmmrs *s);
const uint64_t *pgr_shmmrs_device_offsets(const pgr_shmmrs *s);
int pgr_shmmrs_download(pgr_ctx *ctx, const pgr_shmmrs *s, pgr_mm128 **out_mm, uint64_t **out_off);
void pgr_shmmrs_destroy(pgr_shmmrs *s);

/* 128-bit content checksum of every contig's shimmer list, computed on the GPU: out[2c], out[2c+1] (host, 2 * n_seqs
 * words).  Order sensitive (the ordinal of an element enters) and independent of the rid field; the formula is stated
 * at shmmr_checksum_kernel (csrc/level2.hip).  Lets a caller -- bench.py, the tests -- compare a full-size result with a
 * CPU run of the reference algorithm without moving the lists. */
int pgr_shmmrs_checksum(pgr_ctx *ctx, const pgr_shmmrs *s, uint64_t *out);

/* copy the MM128 list of a resident result into caller-owned DEVICE memory (e.g. a torch tensor that is then
 * all-gathered over RCCL: 16 B per shimmer instead of 40 B per pair record). capacity in elements; rid_add is
 * added to every rid (MM128.y >> 32), turning rank-local contig indices into global sequence ids. */
int pgr_shmmrs_copy_to_device(pgr_ctx *ctx, const pgr_shmmrs *s, pgr_mm128 *d_out, uint64_t capacity,
                              uint32_t rid_add);

/* the same with an explicit rid per contig (rids[i] replaces the contig index i): one rank's shard of a partitioned
 * contig set carries arbitrary global sequence ids */
int pgr_shmmrs_copy_to_device_rids(pgr_ctx *ctx, const pgr_shmmrs *s, pgr_mm128 *d_out, uint64_t capacity,
                                   const uint32_t *rids);
/* host copy of the n_seqs + 1 list offsets of a resident result (no device traffic) */
int pgr_shmmrs_offsets(const pgr_shmmrs *s, uint64_t *out);

/* device pair records from a resident result; d_out must hold count - n_nonempty records;
 * returns the number written in *n_out.  sids may be NULL.  d_out is a DEVICE pointer
 * (e.g. a torch tensor's data_ptr) so that per-GPU buffers can be all-gathered by RCCL.   */
uint64_t pgr_shmmrs_n_pairs(const pgr_shmmrs *s);
int pgr_shmmrs_to_frag_recs_device(pgr_ctx *ctx, const pgr_shmmrs *s, const uint32_t *sids,
                                   int query_side, pgr_frag_rec *d_out, uint64_t capacity,
                                   uint64_t *n_out);

/* pgr_shmmrs_compute + pgr_shmmrs_to_frag_recs_device (index side) in one call with one wait: the pair records are derived on the
 * device behind the list stage and written to d_recs (DEVICE memory, recs_capacity records; *n_pairs = records written).
 * PGR_ERR_INVALID_ARG when the buffer is too small (nothing is returned then). */
int pgr_shmmrs_compute_recs(pgr_ctx *ctx, const pgr_batch *b, const pgr_spec *spec, const uint32_t *sids, pgr_frag_rec *d_recs,
                            uint64_t recs_capacity, pgr_shmmrs **out, uint64_t *n_pairs);

/* ------------------------------------------------------------------ a software pipeline over resident batches
 * The reference's loaders call get_shmmrs_from_seqs batch after batch (load_index_from_reader, pgr-db/src/seq_db.rs:541-571:
 * read <= 129 contigs, compute, insert, repeat).  pgr_shmmrs_compute is that call, and it returns when its batch is done:
 * the list stage, the pair records and the host's round trip of batch i sit between the tile kernels of batches i and
 * i + 1.  A pgr_pipe takes the batches of such a loop WITHOUT waiting: submit enqueues the whole pass -- level-1 tiles on the
 * context's stream; behind an event, on a second (higher-priority) stream: the list stage and the index-side pair records
 * (seq_db.rs:381-400) -- and returns; collect hands back the oldest submitted job once it is done.  Two jobs may be in flight
 * (a third submit fails with PGR_ERR_STATE): the HBM-bound tail of batch i runs beside the VALU-bound tiles of batch i + 1.
 *   sids     sids[i] = sequence id of contig i in the records (NULL: i, or the index's running sid when ix != NULL); copied
 *   ix       != NULL: the pair records are appended to this index in submission order when the job is collected
 *            (load_index_from_seq_vec's insertion order, seq_db.rs:605-612); the index must not be finalized, queried or
 *            written to by anything else while one of its jobs is in flight
 *   d_recs   != NULL (ix == NULL): the pair records go to this DEVICE buffer of recs_capacity records
 * collect:  *out (may be NULL: the list is released) = the job's shimmer lists, as from pgr_shmmrs_compute;
 *           *n_pairs (may be NULL) = pair records written.  PGR_ERR_STATE when nothing is in flight.
 * The batch must stay alive, and unchanged, until its job has been collected.  Results are bit-identical to
 * pgr_shmmrs_compute + pgr_shmmrs_to_frag_recs_device: a job whose first, optimistic pass meets a flagged tile (non-ACGT byte,
 * palindromic k-mer) or an undersized buffer is finished synchronously when it is collected.
 * pgr_ctx_last_prof after a collect describes that job.  Destroying a pipe waits for its jobs and drops their results. */
typedef struct pgr_pipe pgr_pipe;
int pgr_pipe_create(pgr_ctx *ctx, const pgr_spec *spec, pgr_pipe **out);
int pgr_pipe_submit(pgr_pipe *p, const pgr_batch *b, const uint32_t *sids, pgr_index *ix, pgr_frag_rec *d_recs,
                    uint64_t recs_capacity);
int pgr_pipe_collect(pgr_pipe *p, pgr_shmmrs **out, uint64_t *n_pairs);
int pgr_pipe_in_flight(const pgr_pipe *p);
void pgr_pipe_destroy(pgr_pipe *p);

/* ------------------------------------------------------------------ profiling hooks
 * HIP-event timing of the kernels of the LAST pgr_shmmrs_compute on the context's stream. */
typedef struct {
    float level1_ms;     /* dominant kernel (level1_tile_kernel): k-mer hash + windowed minimizers */
    float level1_aux_ms; /* tail kernel + serial (exact state machine) kernel                       */
    float level2_ms;     /* gather + reduce x2 + min_span filter                                    */
    float total_ms;      /* first launch to last kernel of pgr_shmmrs_compute (incl. host syncs)    */
    uint64_t n_level1;   /* level-1 minimizers emitted                                              */
    uint64_t n_tiles;    /* workgroups of the dominant kernel                                       */
    uint64_t n_serial_contigs; /* contigs with at least one island of exact (state machine) tiles   */
    uint64_t bases_tiled;      /* bases covered by the dominant kernel                              */
    uint64_t exact_bases;      /* bases re-done by the exact state-machine kernel (islands)         */
} pgr_prof;
int pgr_ctx_last_prof(const pgr_ctx *ctx, pgr_prof *out);
int pgr_ctx_synchronize(pgr_ctx *ctx);

/* ------------------------------------------------------------------ index (ShmmrToFrags)
 * frag_map: FxHashMap<(u64,u64), Vec<FragmentSignature>> (pgr-db/src/seq_db.rs:75-76) as a
 * CSR sorted by key; within a key the records keep (sid, frg_id) insertion order
 * (seq_db.rs:605-612).                                                                      */
int pgr_index_create(pgr_ctx *ctx, const pgr_spec *spec, pgr_index **out);
void pgr_index_destroy(pgr_index *ix);
/* load_index_from_seq_vec (seq_db.rs:573-615): append contigs; sids[i] NULL -> running id */
int pgr_index_add_batch(pgr_ctx *ctx, pgr_index *ix, uint32_t n_seqs, const uint8_t *const *seqs,
                        const uint64_t *lens, const uint32_t *sids);
int pgr_index_add_packed(pgr_ctx *ctx, pgr_index *ix, uint32_t n_seqs, const uint64_t *lens, const uint64_t *planes,
                         const uint32_t *valid, const uint32_t *sids);
int pgr_index_add_resident(pgr_ctx *ctx, pgr_index *ix, const pgr_batch *b, const uint32_t *sids);
/* room for n_records appended records in all (a host that knows what it is going to index -- total bases x ~0.00304 pair
 * records per base at (80, 56, 4, 64) -- saves the index its grow-and-copy steps; records already appended are kept) */
int pgr_index_reserve(pgr_ctx *ctx, pgr_index *ix, uint64_t n_records);
/* merge pair records computed elsewhere (other GPUs, after the RCCL all-gather) */
int pgr_index_add_records(pgr_ctx *ctx, pgr_index *ix, const pgr_frag_rec *recs, uint64_t n,
                          int recs_on_device);
/* merge shimmer lists computed elsewhere: MM128 with y>>32 = sequence id; the shimmers of one sequence must be
 * contiguous and in position order (what pgr_shmmrs_compute produces with rids = global ids); the pair records
 * (seq_db.rs:381-400) are derived on the GPU.  mm is a host or a DEVICE pointer. */
int pgr_index_add_shmmrs(pgr_ctx *ctx, pgr_index *ix, const pgr_mm128 *mm, uint64_t n, int mm_on_device);
/* sort -> CSR (GPU).  Also builds the lookup side tables of the query path: a bucket table over the keys and the keys by
 * themselves (16 B per distinct key + 4 B per bucket, ~0.5 GB for the 3x10^7 keys of a 10 Gbp index). */
int pgr_index_finalize(pgr_ctx *ctx, pgr_index *ix);
uint64_t pgr_index_n_keys(const pgr_index *ix);
uint64_t pgr_index_n_records(const pgr_index *ix);
/* host copy of the sorted records (pgr_free) */
int pgr_index_download(pgr_ctx *ctx, const pgr_index *ix, pgr_frag_rec **out, uint64_t *n);
/* .mdb files: write_shmmr_map_file / read_mdb_file (pgr-db/src/seq_db.rs:1291-1326, 1328-1407).  Keys are
 * written sorted (the reference writes hash-map order; its readers are order agnostic).  load returns a
 * finalized index carrying the spec stored in the file. */
int pgr_index_write_mdb(pgr_ctx *ctx, const pgr_index *ix, const char *path);
int pgr_index_load_mdb(pgr_ctx *ctx, const char *path, pgr_index **out);
int pgr_index_spec(const pgr_index *ix, pgr_spec *out);

/* ------------------------------------------------------------------ B2: query_fragment_to_hps
 * Replaces SeqIndexDB::query_fragment_to_hps (pgr-db/src/ext.rs:252-282) =
 * raw_query_fragment (seq_db.rs:1200-1228) + aln::query_fragment_to_hps (aln.rs:147-242) +
 * aln::sparse_aln (aln.rs:12-142) for a whole batch of queries (the reference loops over
 * queries with rayon, pgr-bin/src/bin/pgr-query.rs:135-165).
 * Result layout (flat, all arrays pgr_free'd through pgr_hps_result_free):
 *   query q owns targets  [q_off[q], q_off[q+1])
 *   target t: sid = t_sid[t], owns chains [t_off[t], t_off[t+1])      (sorted by sid)
 *   chain c: score = c_score[c], owns hit pairs [c_off[c], c_off[c+1])                      */
typedef struct {
    uint32_t n_queries;
    uint64_t *q_off;
    uint64_t n_targets;
    uint32_t *t_sid;
    uint64_t *t_off;
    uint64_t n_chains;
    float *c_score;
    uint64_t *c_off;
    uint64_t n_hps;
    pgr_hitpair *hps;
    /* (query, target) groups on which the reference's chain extraction never terminates (aln.rs:105-131: every
     * unvisited hit pair has a non-positive score, e.g. bgn == end).  Such a group keeps the chains extracted up to
     * that point; the other groups of the batch are unaffected. */
    uint64_t n_nonterminating;
    void *_owner; /* the one host block all arrays above live in (released by pgr_hps_result_free) */
} pgr_hps_result;

int pgr_query_hps_batch(pgr_ctx *ctx, const pgr_index *ix, uint32_t n_queries,
                        const uint8_t *const *seqs, const uint64_t *lens, float penalty,
                        uint32_t max_count, uint32_t max_count_query, uint32_t max_count_target,
                        uint32_t max_aln_span, int has_max_gap, uint32_t max_gap, int oriented,
                        pgr_hps_result *out);
/* the same on queries that are already resident on the GPU (pgr_batch_from_ascii): the stages of B2 without the
 * host -> device copy of the ASCII queries */
int pgr_query_hps_resident(pgr_ctx *ctx, const pgr_index *ix, const pgr_batch *queries, float penalty,
                           uint32_t max_count, uint32_t max_count_query, uint32_t max_count_target,
                           uint32_t max_aln_span, int has_max_gap, uint32_t max_gap, int oriented,
                           pgr_hps_result *out);
void pgr_hps_result_free(pgr_hps_result *r);
/* Query batches through a pgr_pipe (two in flight): the reference loops over its queries with rayon
 * (pgr-bin/src/bin/pgr-query.rs:135-165); here the tiles of batch i + 1 run on the context's stream beside everything that is
 * behind the tiles of batch i -- list stage, pair records, the per-query kernel, packing, the chains' way back over PCIe -- on the
 * pipe's back stream.  submit enqueues and returns; collect hands back the oldest query job's result, which is the result of
 * pgr_query_hps_resident on that batch (a batch the chained path cannot take -- flagged tiles, long queries, a repeat key -- is
 * answered by that call at collect).  The pipe's spec must be the index's; `queries` and `ix` stay alive until the job is
 * collected; jobs of pgr_pipe_submit and query jobs may be mixed, each kind collected with its own call, oldest first. */
int pgr_pipe_submit_query(pgr_pipe *p, const pgr_batch *queries, const pgr_index *ix, float penalty, uint32_t max_count,
                          uint32_t max_count_query, uint32_t max_count_target, uint32_t max_aln_span, int has_max_gap,
                          uint32_t max_gap, int oriented);
int pgr_pipe_collect_query(pgr_pipe *p, pgr_hps_result *out);

/* counts and host-side stage times of the LAST pgr_query_hps_batch / _resident on the context (what bench.py prices the
 * query leg with: 24 B per hit pair emitted + 0.25 B per query base + 17 B per looked-up signature) */
typedef struct {
    uint64_t n_queries, query_bases;
    uint64_t n_query_pairs; /* shimmer pairs of the queries = index lookups                      */
    uint64_t n_signatures;  /* fragment signatures under the looked-up keys (before the filters) */
    uint64_t n_hits;        /* hit pairs after the count filters (input of the chaining)          */
    uint64_t n_groups;      /* (query, target) groups with >= 2 hits                              */
    uint64_t n_chains, n_hps;
    float stage_ms;  /* host time to stage the ASCII queries (pinned copy + enqueue of H2D and pack)   */
    float shmmr_ms;  /* query shimmers (pgr_shmmrs_compute)                                             */
    float lookup_ms; /* pair records, index lookup, multiplicities, count filters (1 round trip)       */
    float chain_ms;  /* hit expansion, sort by (group, qb), sparse_aln, packing, download (2 round trips) */
    float result_ms; /* host assembly of the flat result                                                */
    float total_ms;
    /* 0: one kernel per stage over the whole batch (any batch).  1: one wavefront per query does every stage behind the pair
     * records (batches of short queries, csrc/query_fused.hip): lookup_ms and result_ms are 0, chain_ms holds that stage.
     * 2: the same, enqueued behind the shimmer pipeline without a host wait in between (the usual case; 1 when the guess
     * of the queries' sizes was too small): shmmr_ms holds the device time of both, the call has ONE synchronization.
     * 3: the level-1 form of that kernel (round 6; the usual case for batches of short clean queries): the tile kernel of the
     * queries, then the per-query kernel on the tile segments -- it runs the list stage of its own query (reduce_shmmr twice,
     * shmmrutils.rs:359-415; min_span, :536-555; the pairs, seq_db.rs:1205-1217) in LDS --, no list stage of the batch: 8 launches,
     * ONE synchronization; n_query_pairs is what the device counted.  What the tile kernel flags (a palindromic k-mer, a non-ACGT
     * byte, an overflow) declines the batch on the device and it is answered as under 2 / 1 / 0.                              */
    uint32_t path;
    uint32_t _pad;
} pgr_query_prof;
int pgr_ctx_last_query_prof(const pgr_ctx *ctx, pgr_query_prof *out);

/* aln::sparse_aln on caller-provided hit pairs (pgr-tk/src/lib.rs:1539 `sparse_aln`):
 * n_groups groups, group g = hits[g_off[g], g_off[g+1]).  Output as above with one
 * "query" and one target per group (t_sid = group index).                                  */
int pgr_sparse_aln_batch(pgr_ctx *ctx, uint32_t n_groups, const pgr_hitpair *hits,
                         const uint64_t *g_off, uint32_t max_span, float penalty, int has_max_gap,
                         uint32_t max_gap, int oriented, pgr_hps_result *out);

/* ------------------------------------------------------------------ multi-GPU exchange
 * One process per GPU.  Contigs are sharded across ranks (the reference's unit of parallelism is the contig,
 * pgr-db/src/seq_db.rs:460-467); after every rank has computed the shimmers of its shard the per-rank MM128 lists
 * (rid = global sequence id) are all-gathered over RCCL / xGMI, and the rank that owns the frag_map -- or every rank,
 * for a replicated query index -- derives the pair records from them (pgr_index_add_shmmrs) in the order the serial
 * insert of seq_db.rs:605-612 produces.  The reference has no counterpart (single process); the only FFI precedent
 * is the opaque-handle style of the AGC binding (pgr-db/src/agc_io.rs:76-116), followed here.
 *   rank 0: pgr_exchange_unique_id() -> 128 bytes, handed to the other ranks by the HOST program (pipe, file, MPI ...)
 *   all   : pgr_exchange_create(ctx, id, rank, world)             (collective: ncclCommInitRank)
 *   step  : pgr_exchange_allgather_shmmrs_start(...)  enqueues the collective on the exchange's own stream behind the
 *           work already queued on the context's stream and returns at once;  pgr_exchange_wait() blocks until the
 *           lists have arrived and returns every rank's element count.  Rank r's list is d_out[r * cap_per_rank ...].
 *   cap_per_rank is a capacity agreed at start-up (all ranks pass the same value, >= every rank's count): the
 *   collective is one padded ncclAllGather, no rank needs another rank's size to post it.
 * RCCL is loaded at the first call (dlopen "librccl.so.1"); without it these functions fail with PGR_ERR_DEVICE. */
#define PGR_UNIQUE_ID_BYTES 128
typedef struct pgr_exchange pgr_exchange;
int pgr_exchange_unique_id(pgr_ctx *ctx, uint8_t *id /* PGR_UNIQUE_ID_BYTES */);
int pgr_exchange_create(pgr_ctx *ctx, const uint8_t *id, int rank, int world, pgr_exchange **out);
void pgr_exchange_destroy(pgr_exchange *x);
int pgr_exchange_rank(const pgr_exchange *x);
int pgr_exchange_world(const pgr_exchange *x);
int pgr_exchange_allgather_shmmrs_start(pgr_exchange *x, const pgr_mm128 *d_local, uint64_t n_local, pgr_mm128 *d_out,
                                        uint64_t cap_per_rank);
int pgr_exchange_wait(pgr_exchange *x, uint64_t *counts /* world entries, may be NULL */);
/* device pointer to the `world` element counts of the last all-gather */
const uint64_t *pgr_exchange_device_counts(const pgr_exchange *x);
/* Blocking convenience for host programs: all-gather the lists of `s` (rids[i] = global sequence id of its contig i;
 * s == NULL: nothing from this rank in this round) and add EVERY rank's lists to `ix` in rank order (ix == NULL: take
 * part in the collective only -- ranks that do not own the frag_map).  The counts travel first, their maximum pads the
 * payload, so no capacity has to be agreed.  All ranks must call it the same number of times. */
int pgr_exchange_gather_into_index(pgr_exchange *x, const pgr_shmmrs *s, const uint32_t *rids, pgr_index *ix,
                                   uint64_t *n_gathered /* may be NULL */);

/* ------------------------------------------------------------------ key-range sharded index
 * The frag_map of the reference is one hash map filled serially (pgr-db/src/seq_db.rs:605-612).  With one process per GPU
 * the key space is cut into `world` ranges of the first hash h0 and every pair record travels to the rank that owns its
 * range: each rank sorts total / world records however many ranks there are, and the shards' CSRs in rank order ARE the
 * single-process CSR (same keys, same per-key (sid, frg_id) order).
 *   pgr_exchange_shard_records   collective: pooled sample of h0 -> world - 1 splitters (returned in splitters_out when
 *                                not NULL), stable partition of this rank's records (device pointer), counts, one variable
 *                                all-to-all (grouped ncclSend / ncclRecv) straight into `ix`.  *n_received = records that
 *                                arrived in this call.  A host that feeds ONE index over several calls passes
 *                                reuse_splitters != 0 from the second call on (the ranges must not move between calls).
 *                                Finish the shard with pgr_index_finalize.
 *   pgr_exchange_allgather_index collective: the replicated query index from the finalized shards (the concatenation of
 *                                the sorted ranges needs no sort).
 * The collective-free pieces, for hosts that bring their own transport (the tests run two ranks on one GPU over gloo):
 *   pgr_shard_sample_keys  up to n_samples first hashes of the records, evenly spaced (host output)
 *   pgr_shard_splitters    host only: quantiles of the pooled samples; record -> rank = number of splitters <= h0
 *   pgr_shard_partition    d_out = the records grouped by destination rank (stable), counts[world] on the host
 *   pgr_records_checksum / pgr_index_records_checksum   order-independent 128-bit content checksum of a record set: the
 *                          sums over all ranks before and after the exchange must agree                              */
int pgr_exchange_shard_records(pgr_exchange *x, const pgr_frag_rec *d_recs, uint64_t n, pgr_index *ix,
                               int reuse_splitters, uint64_t *splitters_out /* world - 1, may be NULL */,
                               uint64_t *n_received /* may be NULL */);
int pgr_exchange_allgather_index(pgr_exchange *x, const pgr_index *shard, pgr_index **out);
int pgr_shard_sample_keys(pgr_ctx *ctx, const pgr_frag_rec *d_recs, uint64_t n, uint32_t n_samples, uint64_t *out,
                          uint32_t *n_out);
int pgr_shard_splitters(const uint64_t *samples, uint64_t n, int world, uint64_t *splitters /* world - 1 */);
int pgr_shard_partition(pgr_ctx *ctx, const pgr_frag_rec *d_recs, uint64_t n, const uint64_t *splitters, int world,
                        pgr_frag_rec *d_out, uint64_t *counts /* host, world */);
int pgr_records_checksum(pgr_ctx *ctx, const pgr_frag_rec *d_recs, uint64_t n, uint64_t out[2]);
int pgr_index_records_checksum(pgr_ctx *ctx, const pgr_index *ix, uint64_t out[2]);
/* device pointer to the index's records: sorted after pgr_index_finalize, in append order before */
const pgr_frag_rec *pgr_index_device_records(const pgr_index *ix);
/* first hash of the first and of the last record of a finalized index (a shard's key range) */
int pgr_index_key_range(pgr_ctx *ctx, const pgr_index *ix, uint64_t *h0_min, uint64_t *h0_max);

/* ------------------------------------------------------------------ next: MAP-graph + principal bundles
 * Consumers of the frag_map (BASELINE.json configs[3], pgr-pbundle-decomp).  The data-parallel parts run on
 * the GPU (adjacency list = one sort + a 2-point stencil over all records; bundle lookup of every shimmer pair);
 * the graph walks are small serial host code inside the library, like the reference's.                       */

/* ShmmrGraphNode (pgr-db/src/graph_utils.rs:47) + its weight = frag_map[(h0,h1)].len() (seq_db.rs:1038-1043) */
typedef struct {
    uint64_t h0, h1;
    uint32_t orient;
    uint32_t count;
} pgr_vertex;

/* AdjPair (graph_utils.rs:49): (sid, v, w) */
typedef struct {
    uint32_t sid, _pad;
    pgr_vertex v, w;
} pgr_adj_pair;

/* seq_db::frag_map_to_adj_list (pgr-db/src/seq_db.rs:876-945) on a finalized index.
 * keeps == NULL <=> None.  *out is pgr_free'd by the caller.                                                  */
int pgr_index_adj_list(pgr_ctx *ctx, const pgr_index *ix, uint32_t min_count, const uint32_t *keeps,
                       uint32_t n_keeps, pgr_adj_pair **out, uint64_t *n_out);

/* frag_map[(h0,h1)].len() for n keys (pgr-tk/src/lib.rs:636 get_shmmr_pair_count, batched); keys = n x {h0,h1} */
int pgr_index_key_counts(pgr_ctx *ctx, const pgr_index *ix, uint64_t n, const uint64_t *keys,
                         uint32_t *counts);

/* one element of seq_db::sort_adj_list_by_weighted_dfs (seq_db.rs:1006-1062):
 * (node, Option<previous node>, node weight = node.count, is_leaf, global_rank, branch, branch_rank)          */
typedef struct {
    pgr_vertex node, parent;
    uint32_t has_parent, is_leaf, rank, branch, branch_rank, _pad;
} pgr_dfs_node;
/* (host only: ctx may be NULL, then errors come back as codes without a message) */
int pgr_sort_adj_list_by_weighted_dfs(pgr_ctx *ctx, const pgr_adj_pair *adj, uint64_t n,
                                      const pgr_vertex *start, pgr_dfs_node **out, uint64_t *n_out);

/* principal bundles: bundle b = vertices[b_off[b], b_off[b+1]);  bundle_id / mean_ord are filled by the
 * "with id" entry points (ext.rs:552-650: bundles re-ordered by mean position along the sequences,
 * reversed by direction vote), otherwise bundle_id[b] = b and mean_ord[b] = 0.                               */
typedef struct {
    uint64_t n_bundles;
    uint64_t *b_off;
    uint64_t *bundle_id;
    uint64_t *mean_ord;
    uint64_t n_vertices;
    pgr_vertex *vertices;
} pgr_bundles;
void pgr_bundles_free(pgr_bundles *b);

/* SeqIndexDB::get_principal_bundles (ext.rs:491-510) = frag_map_to_adj_list +
 * get_principal_bundles_from_adj_list (seq_db.rs:1064-1186) */
int pgr_principal_bundles(pgr_ctx *ctx, const pgr_index *ix, uint32_t min_count, uint32_t path_len_cutoff,
                          const uint32_t *keeps, uint32_t n_keeps, pgr_bundles *out);
/* the same from a caller-provided adjacency list (seq_db.rs:1064); host only, ctx may be NULL */
int pgr_principal_bundles_from_adj_list(pgr_ctx *ctx, const pgr_adj_pair *adj, uint64_t n,
                                        uint32_t path_len_cutoff, pgr_bundles *out);

/* one shimmer pair of a sequence annotated with its principal bundle
 * ((h0,h1,p0,p1,orient), Option<(bundle_id, direction, position)>)  ext.rs:976-1014 */
typedef struct {
    uint64_t h0, h1;
    uint32_t bgn, end;
    uint32_t orient;     /* query-side orientation (strict <, ext.rs:534-548) */
    uint32_t sid;
    int32_t bundle_id;   /* -1: None */
    uint32_t bundle_dir;
    uint32_t bundle_pos;
    uint32_t _pad;
} pgr_smp_bundle;

/* get_principal_bundle_decomposition (pgr-tk/src/lib.rs:1066-1100, ext.rs:552-650 + 976-1014) over the
 * index's own sequences: bundles with id + every shimmer pair annotated, grouped by sequence in
 * ascending sid (smps of sequence j = smps[seq_off[j], seq_off[j+1]), its id seq_sid[j]).
 * Outputs are pgr_free'd / pgr_bundles_free'd by the caller.                                                  */
int pgr_principal_bundle_decomposition(pgr_ctx *ctx, const pgr_index *ix, uint32_t min_count,
                                       uint32_t path_len_cutoff, const uint32_t *keeps, uint32_t n_keeps,
                                       pgr_bundles *bundles, pgr_smp_bundle **smps, uint64_t *n_smps,
                                       uint32_t **seq_sid, uint64_t **seq_off, uint32_t *n_seqs);
/* get_principal_bundle_projection (pgr-tk/src/lib.rs:1128-1146): the same for caller-provided sequences
 * (ASCII, host), which also vote on bundle order and direction.                                              */
int pgr_principal_bundle_projection(pgr_ctx *ctx, const pgr_index *ix, uint32_t min_count,
                                    uint32_t path_len_cutoff, const uint32_t *keeps, uint32_t n_keeps,
                                    uint32_t n, const uint8_t *const *seqs, const uint64_t *lens,
                                    const uint32_t *sids, pgr_bundles *bundles, pgr_smp_bundle **smps,
                                    uint64_t *n_smps, uint64_t **seq_off);

#ifdef __cplusplus
}
#endif
#endif /* PGR_HIP_H */
