// pgr_internal.h -- shared between the HIP translation units of libpgrhip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/pgr_hip.h"

namespace pgr {

// ------------------------------------------------------------------ geometry of the level-1 kernel
#ifndef PGR_L1_BLOCK
#define PGR_L1_BLOCK 256
#endif
constexpr int L1_BLOCK = PGR_L1_BLOCK;   // threads per workgroup: 256 (4 wavefronts) measured 2 % faster than 512, 384 much slower
constexpr int L1_G = 16;                 // consecutive positions owned by one lane
constexpr int L1_EXT = L1_BLOCK * L1_G;  // positions per tile including both halos
constexpr int L1_WORDS = (L1_EXT + 96) / 32 + 5;  // plane words staged per tile (tile + k-mer look-back)
constexpr int L1_MIN_W = 17;             // window sizes below this use the serial kernel
// Batches of short contigs (reads) run the same kernel with ONE wavefront per tile of 1024 positions: a 1 kbp read in a
// 4096-position tile keeps two of four wavefronts busy and the other two hold its 35 KB of LDS while they wait at the barriers.
constexpr int L1_BLOCK_SHORT = 64;
constexpr int L1_EXT_SHORT = L1_BLOCK_SHORT * L1_G;
constexpr int L1_SHORT_MEAN_LEN = 2048;  // mean contig length of a batch at or below which the one-wavefront tiles are used
constexpr int L1_SHORT_MAX_W = 128;      // ... for windows up to this (tile core = 1024 - 2 (w - 1) positions)
// Tiles of a contig of L positions, tile core tc, extended tile ext: a contig of up to ext - L1_G positions is ONE tile (its
// last core position needs no halo behind it: nothing is selected beyond the contig's end.  Not the tile's last lane: the
// max pass reads the rows behind a lane clamped to the last row, which is that lane's own); otherwise tile t owns the core
// [t tc, min((t+1) tc, L)).  The first tile's extended range starts at position 0 (there is nothing in front of it), every
// other tile's at its core - (w - 1).
__host__ __device__ inline uint64_t l1_tiles_of(uint64_t L, uint32_t tc, uint32_t ext) {
    return L == 0 ? 0 : (L <= ext - L1_G ? 1 : (L + tc - 1) / tc);
}
__host__ __device__ inline long long l1_core_end(long long c0, long long L, uint32_t tc, uint32_t ext) {
    return (L <= (long long)(ext - L1_G) || c0 + (long long)tc > L) ? L : c0 + (long long)tc;
}
constexpr uint64_t U64MAX = 0xFFFFFFFFFFFFFFFFull;

// ------------------------------------------------------------------ device-resident batch
// Contig c occupies words [word_off[c], word_off[c] + ceil(len/32)) of the plane arrays.
// Word j of a contig holds bases 32j .. 32j+31, base i at bit (31 - i%32)  (MSB first), so a
// run of consecutive bases ending at base e is a contiguous bit field whose LSB is base e:
// exactly the k-mer bit-plane layout of shmmrutils.rs:459-476 (fmmer.0 = low bits, fmmer.1 =
// high bits of the 2-bit codes).  planes[j] = {low-bit plane, high-bit plane}.
// valid[j]: bit set = the byte was one of ACGTacgt\0\1\2\3 (shmmrutils.rs:426-436).
struct BatchDev {
    uint2 *planes = nullptr;
    uint32_t *valid = nullptr;
    uint64_t *word_off = nullptr;  // [n+1]
    uint32_t *len = nullptr;       // [n]
    uint32_t *n_invalid = nullptr; // [n] number of non-ACGT bytes
};

}  // namespace pgr

struct pgr_batch {
    pgr_ctx *ctx = nullptr;
    uint32_t n = 0;
    uint64_t total_bases = 0;
    uint64_t total_words = 0;
    pgr::BatchDev d;
    std::vector<uint64_t> h_word_off;  // [n+1]
    std::vector<uint32_t> h_len;       // [n]
    std::vector<uint32_t> h_n_invalid; // [n] non-ACGT bytes counted by the host packer (source of an async H2D copy)
    bool host_saw_invalid = false;     // the host packer counted at least one
    hipEvent_t ev_alloc = nullptr;     // behind the H2D copies of the batch's tables (batch_alloc), on the context's stream
};

struct pgr_shmmrs {
    pgr_ctx *ctx = nullptr;
    uint32_t n = 0;
    uint64_t count = 0;
    pgr_mm128 *d_mm = nullptr;   // [count]
    uint64_t *d_off = nullptr;   // [n+1]; = d_block + a few status words in front (one D2H copy brings both)
    uint64_t *d_block = nullptr; // the allocation d_off lives in
    std::vector<uint64_t> h_off; // [n+1]
    bool rid_is_index = false;   // MM128.y >> 32 is the contig index (no rids given, no padding sentinels)
    const pgr_mm128 *host_copy = nullptr;  // small results: already in the context's pinned buffer (valid until its next use)
};

namespace pgr {

// ------------------------------------------------------------------ kernel launch wrappers
// pack.hip
void launch_pack_ascii(hipStream_t st, const uint8_t *d_ascii, uint64_t w0, const BatchDev &b, uint32_t n,
                       uint64_t w1);
// packed host input: clean words [w0, w1) (bits past the contig end, plane bits of invalid positions) and count the
// non-ACGT positions per contig; has_valid == 0: every base is valid, the validity plane is written from the lengths
void launch_sanitize_packed(hipStream_t st, const BatchDev &b, uint32_t n, uint64_t w0, uint64_t w1, int has_valid);
// dst[0 .. n_words) = pinned_src[0 .. n_words): a table out of pinned host memory, read by a kernel (not by a copy engine)
void launch_copy_words(hipStream_t st, uint32_t *dst, const uint32_t *pinned_src, uint64_t n_words);
void launch_synth(hipStream_t st, const BatchDev &b, uint32_t n, uint64_t total_words, uint64_t seed,
                  uint64_t contig0, const uint64_t *d_ids /* NULL: contig0 + index */);

// level1.hip
// One level-1 minimizer as it crosses HBM between the level-1 kernels and the fused list kernel: 12 bytes instead of a
// 16-byte MM128.  x = key << 8 | k and the contig id are recomputable by the reader (k from the spec, the contig from
// the segment the record sits in: seg_cid), so only the 56-bit hash key and pos << 1 | strand travel.
struct L1Rec {
    uint32_t key_lo, key_hi;  // hash & (2^56 - 1)
    uint32_t ypos;            // pos << 1 | strand  (the low word of MM128.y)
};
static_assert(sizeof(L1Rec) == 12, "level-1 record");
struct TileDesc {  // one per tile, written by tile_desc_kernel (saves every workgroup a 10-step dependent search)
    uint64_t word_off;    // first plane word of the contig
    uint32_t len;         // contig length
    uint32_t contig;      // contig index
    uint32_t tile_local;  // tile ordinal inside the contig
    uint32_t skip;        // 1: a non-ACGT byte in the tile's reach (set by mark_invalid_tiles_kernel): the tile kernel leaves it to the islands
    uint32_t _pad[2];
};
// exact-machine chunks (level1_chunk_kernel)
struct ChunkState {
    uint64_t min_x, min_y, mdist;  // min_mer and the distance counter (shmmrutils.rs:450-453, 441)
    uint64_t F0, F1, R0, R1;       // rolling k-mer planes (fmmer / rmmer)
    uint64_t ring_sig;             // order-sensitive signature of the ring buffer + its fill
};
struct ChunkDesc {
    uint32_t contig;
    uint32_t seg;             // segment-table entry this chunk's list goes to (0xFFFFFFFF: probe, writes nothing)
    uint64_t cs, ce;          // emission steps [cs, ce), cs a multiple of 64
    uint64_t emit_lo_pos;     // elements below this position belong to the tile before an island: suppressed
    uint64_t drain_end;       // > ce: keep stepping to here, emitting only elements with position < ce
    uint64_t region_off, region_cap;
    uint32_t warm;            // machine warm-up positions before cs (multiple of 64)
    uint32_t override_state;  // 1: install in_state at cs instead of trusting the warm-up
    ChunkState in_state;
    // ring buffer hand-over (slots of CHUNK_RING_WORDS u64 in a device array): the chunk leaves its ring at ce in ring_out;
    // with override_state it takes the ring of the chunk in front from ring_in (0xFFFFFFFF: keep the warmed-up ring) --
    // inside a stretch of skipped pushes (palindromic k-mers, shmmrutils.rs:477-480) no warm-up can rebuild it
    uint32_t ring_in, ring_out;
    // Round 6, an island that ends inside a tile (Island::cutE).  ext_limit > ce, on the island's last chunk: while the machine
    // arrives at ce STUCK (mdist > w - 1: it emits nothing until a push reaches down to min_mer, shmmrutils.rs:505-520), ce and
    // drain_end move on by 64, up to ext_limit; the blocks added come back in bits 40.. of the chunk's 4th info word (the island's
    // probe, which ran beside the chunk at the end that was planned, then runs again at the new end in the next round).
    uint64_t ext_limit;
};
constexpr uint32_t CHUNK_RING_WORDS = 2 * 128 + 1;  // x[128], y[128] in push order, fill
constexpr uint32_t L1_TAIL_SLOT = 4;
struct L1Args {
    BatchDev b;
    uint32_t n_contigs;
    uint32_t n_tiles;
    const uint32_t *tile_first;  // [n+1] device
    TileDesc *desc;              // [n_tiles] scratch, filled by launch_level1_tiles
    uint32_t w, k, r, tc, sketch;
    uint32_t ext;                // positions per extended tile: L1_EXT, or L1_EXT_SHORT for batches of short contigs
    L1Rec *out;                  // level-1 segments: [0, n_tiles*slot) fixed tile slots, then the overflow region
    uint32_t slot;               // elements per tile slot
    uint64_t ovf_base;           // first element of the overflow region (= n_tiles * slot)
    uint64_t cap;                // capacity of the overflow region (elements)
    uint64_t tail_base;          // first element of the contigs' tail slots (L1_TAIL_SLOT elements per contig; the tail kernel
                                 // allocates from the overflow region only when a tail emits more)
    unsigned long long *cursor;  // [0] overflow elements allocated, [1] overflow-of-the-overflow flag, [2] bit0: a tile saw a
                                 // palindromic k-mer, bit1: a tile holds a non-ACGT byte (islands of exact tiles needed)
    uint64_t *seg_off;           // [n_tiles + n_contigs]
    uint32_t *seg_cnt;           // [n_tiles + n_contigs]
    uint32_t *seg_cid;           // [n_tiles + n_contigs] contig of the records of a segment
    uint32_t *contig_flags;      // [n] bit0: palindromic skip seen (needs the exact kernel)
    uint8_t *tile_flags;         // [n_tiles] 1: the tile's extended range holds a palindromic k-mer / non-ACGT byte
    uint16_t *tile_pal;          // [n_tiles] or NULL; of a tile with flag bit 0: first | last << 8 block of 64 core positions with a palindromic k-mer
    // [n_tiles] (contig + 1) << 32 | (last valid position <= the end of the tile's core) + 1, low word 0 when the contig has
    // no valid base up to there.  Written per tile by mark_invalid_tiles_kernel, made cumulative by an inclusive max-scan
    // before the exact-machine chunks run: their k-mer look-back crosses a multi-Mbp run of N in one step with it.
    uint64_t *tile_lv;
};
void launch_level1_pre(hipStream_t st, const L1Args &a, uint64_t *tile_lv, bool known_clean = false);  // tile descriptors, flags of tiles with a non-ACGT byte in reach (tile_lv: [n_tiles] last valid positions)
void launch_level1_tiles(hipStream_t st, const L1Args &a);                    // the tiles (and the contigs' tails), behind launch_level1_pre
void launch_level1_tails(hipStream_t st, const L1Args &a);
constexpr uint32_t LDS_GRANULE = 512;  // gfx950 hands out LDS in 512-byte units
uint32_t level1_tile_lds_bytes(const L1Args &a);  // LDS a workgroup of the tile kernel for `a` occupies (0: unknown)
// exact state machine, one wavefront per chunk; status bit0 = region overflow, bit1 = override impossible
void launch_level1_chunks(hipStream_t st, const L1Args &a, const ChunkDesc *d_descs, uint32_t n_chunks,
                          ChunkState *d_in, ChunkState *d_out, uint32_t *d_status, uint64_t *d_rings, uint64_t *d_info);
void launch_zero_contig_segs(hipStream_t st, const L1Args &a, const uint32_t *d_list, uint32_t n_list);
// seg_cnt[s] = 0 for s in [ranges[2i], ranges[2i+1])
void launch_zero_seg_ranges(hipStream_t st, const L1Args &a, const uint32_t *d_ranges, uint32_t n_ranges);
void launch_assemble_chunks(hipStream_t st, const L1Args &a, const uint64_t *d_copies, uint32_t n_copies, const uint64_t *d_segs, uint32_t n_segs);
void launch_splice_segs(hipStream_t st, const L1Args &a, const uint64_t *d_ents, uint32_t n_ents);  // tiles an island begins or ends inside
// tile_flags |= 1 for tiles (of the listed contigs) whose extended range contains a non-ACGT byte
void launch_mark_invalid_tiles(hipStream_t st, const L1Args &a);  // (called by launch_level1_tiles between the descriptors and the tiles)
// inclusive max-scan of a.tile_lv in place (rocPRIM); temp from scan_max_temp_bytes
size_t scan_max_temp_bytes(uint32_t n);
hipError_t scan_max_inplace(hipStream_t st, void *temp, size_t temp_bytes, uint64_t *v, uint32_t n);

// level2.hip
// dst holds dst_cap elements: segments that would end beyond it are skipped (the host sees the true total and retries)
void launch_gather_segments(hipStream_t st, const pgr_mm128 *src, const uint64_t *seg_off, const uint32_t *seg_cnt,
                            const uint64_t *seg_dst, uint32_t n_segs, pgr_mm128 *dst, uint64_t dst_cap);
void launch_frag_recs(hipStream_t st, const pgr_mm128 *mm, const uint64_t *off, const uint64_t *rec_off,
                      uint32_t n_contigs, uint64_t n, const uint32_t *sids, int query_side, int rid_is_index,
                      pgr_frag_rec *out);
// device-only variant (no host counts): rec_off[n+1] computed from `off`, the element count read from *total_ptr (capped by cap);
// garbage-tolerant (every index checked).  rid field = contig index.
void launch_frag_recs_dev(hipStream_t st, const pgr_mm128 *mm, const uint64_t *off, uint32_t n_contigs, uint64_t cap,
                          const uint64_t *total_ptr, int query_side, uint64_t *rec_off, pgr_frag_rec *out, uint64_t out_cap,
                          uint32_t *clear3 = nullptr,   // clear3: three 32-bit words zeroed on the way (the consumer's flags)
                          const uint32_t *sids = nullptr,   // sids[n_contigs] (device): the records' sequence ids; NULL: the contig index
                          const uint64_t *base_ptr = nullptr,  // device: first record goes to out[*base_ptr] (NULL: out[0])
                          uint32_t lds_match = 0);  // > 0: LDS per workgroup of the kernels that use any (see FusedArgsPub)
void launch_cursor_bump(hipStream_t st, uint64_t *cursor, const uint64_t *count, uint64_t *before);  // *before = *cursor; *cursor += *count

void launch_contig_offsets(hipStream_t st, const uint64_t *seg_dst, const uint32_t *tile_first, uint32_t n,
                           uint32_t n_segs, uint64_t *off);
void launch_copy_or_sentinel(hipStream_t st, const pgr_mm128 *in, const uint64_t *off_in, const uint64_t *off_out,
                             uint32_t n, pgr_mm128 *out);

// fused reduce x2 + min_span over the unordered level-1 segments (level2.hip)

constexpr uint32_t FUSED_BLOCK_ELEMS = 1024;
struct FusedArgsPub {
    const L1Rec *l1;
    const uint64_t *seg_off;
    const uint32_t *seg_cnt;
    const uint32_t *seg_cid;
    uint32_t k;                // spec.k: x = key << 8 | k
    const uint64_t *seg_dst;
    uint32_t n_segs;
    const uint64_t *total;     // device: number of level-1 elements (= seg_dst[n_segs]); workgroups beyond it exit
    uint32_t r, padding, min_span, do_reduce, halo;
    pgr_mm128 *out;            // [0, n_blocks*slot) fixed block slots, then the overflow region
    uint32_t slot;
    uint64_t ovf_base;
    uint64_t cap;              // overflow capacity
    unsigned long long *cursor;
    uint64_t *blk_off;
    uint32_t *blk_cnt;
    uint32_t *blk_first_seg;  // [n_blocks] scratch
    uint32_t block_elems = 1024;  // list elements per workgroup: FUSED_BLOCK_ELEMS, or 512 (halo <= 32 only): small workgroups for a pipe's back stream
    uint32_t lds_match = 0;   // > 0: every workgroup occupies exactly this much LDS (the tile kernel's, when the two run side by side)
    uint32_t persistent_grid = 0;  // > 0 (with block_elems 512): this many workgroups loop over the blocks (level2.hip: fused_select_persistent_kernel)
};
void launch_fused_select_pub(hipStream_t st, const FusedArgsPub &a, uint32_t n_blocks);
// n_ptr: device, number of elements (clamped to cap)
void launch_offsets_by_rid(hipStream_t st, const pgr_mm128 *mm, const uint64_t *n_ptr, uint64_t cap, uint32_t n_contigs,
                           uint64_t *off, const unsigned long long *cursor = nullptr, const uint64_t *total1 = nullptr,
                           uint64_t *status = nullptr);
void launch_patch_rid(hipStream_t st, pgr_mm128 *mm, const uint64_t *n_ptr, uint64_t cap, const uint32_t *rids,
                      uint32_t n_contigs);
// status[0..7] = cursor[0..7], status[8] = *total1, status[9] = *n_final: everything the host reads after the one sync
void launch_collect_status(hipStream_t st, const unsigned long long *cursor, const uint64_t *total1, const uint64_t *n_final,
                           uint64_t *status);
// 128-bit content checksum per contig (formula at shmmr_checksum_kernel): sums[2c], sums[2c+1]
void launch_shmmr_checksum(hipStream_t st, const pgr_mm128 *mm, const uint64_t *off, uint32_t n_contigs, uint64_t max_cnt,
                           uint64_t *sums);
void launch_copy_add_rid(hipStream_t st, const pgr_mm128 *in, uint64_t n, uint32_t rid_add, pgr_mm128 *out);
void launch_copy_map_rid(hipStream_t st, const pgr_mm128 *in, uint64_t n, const uint32_t *rids, pgr_mm128 *out);

// scan.hip (rocPRIM device scans / sorts: plain library primitives, not the hot path)
// exclusive scan of n+1 u32 counts (in[n] must be 0) into n+1 u64 offsets: out[n] = total
size_t scan_counts_temp_bytes(uint32_t n_plus_1);
hipError_t scan_counts(hipStream_t st, void *temp, size_t temp_bytes, const uint32_t *in, uint64_t *out,
                       uint32_t n_plus_1);
// the same without LDS (wave scans, partial sums through `temp`): for a stream that runs beside the tile kernel (level1.hip:
// level1_tile_lds_bytes says why).  temp: scan_counts_nolds_temp_bytes(n_plus_1)
size_t scan_counts_nolds_temp_bytes(uint32_t n_plus_1);
hipError_t scan_counts_nolds(hipStream_t st, void *temp, size_t temp_bytes, const uint32_t *in, uint64_t *out, uint32_t n_plus_1);
size_t sort_pairs_temp_bytes(uint64_t n);
hipError_t sort_pairs(hipStream_t st, void *temp, size_t temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                      const uint32_t *vals_in, uint32_t *vals_out, uint64_t n, unsigned end_bit);

}  // namespace pgr
