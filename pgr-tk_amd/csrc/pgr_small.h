// pgr_small.h -- one-launch path for batches of short contigs (csrc/small.hip)
#pragma once
#include "pgr_internal.h"

namespace pgr {

constexpr uint32_t SMALL_FALLBACK = 0xFFFFFFFFu;  // count of a contig the one-workgroup kernel hands back to the general path
constexpr uint32_t SMALL_MAX_LEN = 131072;        // longest contig the kernel takes
constexpr uint32_t SMALL_L1_CAP_MAX = 4096;       // level-1 minimizers per contig its LDS list holds at most
constexpr uint32_t SMALL_MAX_CONTIGS = 4096;
constexpr uint64_t SMALL_MAX_BASES = 16ull << 20;

struct SmallContig {
    uint64_t word_off;  // first plane word of the contig
    uint32_t len;
    uint32_t rid;       // value of MM128.y >> 32
    uint32_t out_off;   // first element of the contig's output slot
    uint32_t out_cap;   // elements the slot holds
};
struct SmallArgs {
    const uint2 *planes;      // 2-bit planes of all contigs (HBM or pinned host memory)
    const uint32_t *valid;    // validity plane, or NULL when the host has checked that every base is valid
    uint32_t l1_cap;          // entries of the level-1 list in dynamic LDS (small_l1_cap of the batch's longest contig)
    const SmallContig *desc;  // [n]
    uint32_t n, w, k, r, min_span, tc;
    pgr_mm128 *out;           // output slots
    uint32_t *counts;         // [n] final shimmers per contig, or SMALL_FALLBACK
    uint32_t *flags;          // [1] bit 0: some contig fell back
};
uint32_t small_l1_cap(uint32_t max_len, uint32_t w);
void launch_small_shmmr(hipStream_t st, const SmallArgs &a);
// resident results: clean[c] = counts[c] (fallback marker -> 0), clean[n] = 0;  then slots -> dst at off[c]
void launch_small_counts(hipStream_t st, const uint32_t *counts, uint32_t n, uint32_t *clean);
void launch_small_gather(hipStream_t st, const pgr_mm128 *slots, const SmallContig *desc, const uint32_t *counts, const uint64_t *off,
                         uint32_t n, pgr_mm128 *dst, uint64_t dst_cap);

}  // namespace pgr
