// pgr_small.h -- one-launch path for batches of short contigs (csrc/small.hip)
#pragma once
#include "pgr_internal.h"

namespace pgr {

constexpr uint32_t SMALL_FALLBACK = 0xFFFFFFFFu;  // count of a contig the one-workgroup kernel hands back to the general path
constexpr uint32_t SMALL_MAX_LEN = 131072;        // longest contig the kernel takes (its LDS list holds 4096 level-1 minimizers)
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
    const uint2 *planes;      // 2-bit planes of all contigs (HBM or pinned host memory); every base valid
    const SmallContig *desc;  // [n]
    uint32_t n, w, k, r, min_span, tc;
    pgr_mm128 *out;           // output slots
    uint32_t *counts;         // [n] final shimmers per contig, or SMALL_FALLBACK
    uint32_t *flags;          // [1] bit 0: some contig fell back
};
void launch_small_shmmr(hipStream_t st, const SmallArgs &a);

}  // namespace pgr
