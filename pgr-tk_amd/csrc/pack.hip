// pack.hip -- ASCII -> 2-bit bit-plane packing and the on-device synthetic contig generator.
//
// Byte semantics follow the reference's base2bits table (pgr-db/src/shmmrutils.rs:426-436):
// A/a/0 -> 0, C/c/1 -> 1, G/g/2 -> 2, T/t/3 -> 3, everything else is "not a base" (valid bit 0).
// HBM-bound streaming kernels (1 B/bp in, 0.375 B/bp out); never on the critical path of the
// VALU-bound level-1 kernel.
#include "pgr_device.h"
#include "pgr_internal.h"

namespace pgr {

namespace {

__device__ __forceinline__ uint32_t find_by_word(const uint64_t *__restrict__ word_off, uint32_t n, uint64_t wi) {
    uint32_t lo = 0, hi = n;  // largest c with word_off[c] <= wi
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (word_off[mid] <= wi) lo = mid;
        else hi = mid;
    }
    return lo;
}

// 2-bit code of one byte, 4 if not a base
__device__ __forceinline__ uint32_t base_code(uint32_t ch) {
    if (ch < 4) return ch;
    const uint32_t lc = ch | 0x20u;
    if (lc == 'a') return 0;
    if (lc == 'c') return 1;
    if (lc == 'g') return 2;
    if (lc == 't') return 3;
    return 4;
}

}  // namespace

// one lane per 32-base word.  The ASCII stream gives every batch word wi the 32 bytes [32*wi, 32*wi+32)
// (contigs start on word boundaries); `ascii` points at the bytes of word w0 (a staged window [w0, w1)).
__global__ __launch_bounds__(256) void pack_ascii_kernel(const uint8_t *__restrict__ ascii, uint64_t w0, BatchDev b,
                                                         uint32_t n, uint64_t w1) {
    const uint64_t wi = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= w1) return;
    const uint32_t c = find_by_word(b.word_off, n, wi);
    const uint64_t wl = wi - b.word_off[c];
    const uint64_t len = b.len[c];
    const uint64_t first = wl * 32;
    uint32_t p0 = 0, p1 = 0, v = 0;
    if (first < len) {
        const uint4 *src = reinterpret_cast<const uint4 *>(ascii + (wi - w0) * 32);
        const uint4 lo = src[0], hi = src[1];  // bytes past the contig end are masked below
        const uint32_t wds[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const uint32_t nb = (len - first) >= 32 ? 32u : (uint32_t)(len - first);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const uint32_t ch = (wds[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            const uint32_t code = base_code(ch);
            const bool ok = (code < 4) && ((uint32_t)i < nb);
            const uint32_t bit = 31u - i;
            if (ok) {
                p0 |= (code & 1u) << bit;
                p1 |= (code >> 1) << bit;
                v |= 1u << bit;
            }
        }
        const uint32_t n_bad = nb - __popc(v);
        if (n_bad) atomicAdd(b.n_invalid + c, n_bad);
    }
    b.planes[wi] = make_uint2(p0, p1);
    b.valid[wi] = v;
}

// packed host input (pgr_batch_from_packed): one lane per word
__global__ __launch_bounds__(256) void sanitize_packed_kernel(BatchDev b, uint32_t n, uint64_t w0, uint64_t w1, int has_valid) {
    const uint64_t wi = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= w1) return;
    const uint32_t c = find_by_word(b.word_off, n, wi);
    const uint64_t first = (wi - b.word_off[c]) * 32;
    const uint64_t len = b.len[c];
    const uint32_t nb = first >= len ? 0u : ((len - first) >= 32 ? 32u : (uint32_t)(len - first));
    const uint32_t tail = nb == 32 ? 0xFFFFFFFFu : (nb == 0 ? 0u : ~(0xFFFFFFFFu >> nb));  // base i at bit 31 - i
    const uint32_t v = has_valid ? (b.valid[wi] & tail) : tail;
    uint2 p = b.planes[wi];
    p.x &= v;
    p.y &= v;
    b.planes[wi] = p;
    b.valid[wi] = v;
    const uint32_t n_bad = nb - __popc(v);
    if (n_bad) atomicAdd(b.n_invalid + c, n_bad);
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// base(c,i) = (splitmix64(seed ^ c*GOLD ^ (i>>5)) >> (2*(i&31))) & 3
__global__ __launch_bounds__(256) void synth_kernel(BatchDev b, uint32_t n, uint64_t total_words, uint64_t seed,
                                                    uint64_t contig0, const uint64_t *__restrict__ ids) {
    const uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= total_words) return;
    const uint32_t c = find_by_word(b.word_off, n, wi);
    const uint64_t wl = wi - b.word_off[c];
    const uint64_t len = b.len[c];
    const uint64_t first = wl * 32;
    uint32_t p0 = 0, p1 = 0, v = 0;
    if (first < len) {
        const uint64_t cid = ids ? ids[c] : contig0 + c;  // global contig id (a shard of a partitioned set passes its list)
        const uint64_t z = splitmix64(seed ^ (cid * 0x9E3779B97F4A7C15ull) ^ wl);
        const uint32_t nb = (len - first) >= 32 ? 32u : (uint32_t)(len - first);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const uint32_t code = (uint32_t)(z >> (2 * i)) & 3u;
            const uint32_t bit = 31u - i;
            if ((uint32_t)i < nb) {
                p0 |= (code & 1u) << bit;
                p1 |= (code >> 1) << bit;
                v |= 1u << bit;
            }
        }
    }
    b.planes[wi] = make_uint2(p0, p1);
    b.valid[wi] = v;
}

void launch_pack_ascii(hipStream_t st, const uint8_t *d_ascii, uint64_t w0, const BatchDev &b, uint32_t n,
                       uint64_t w1) {
    if (w1 <= w0) return;
    const uint32_t blocks = (uint32_t)((w1 - w0 + 255) / 256);
    hipLaunchKernelGGL(pack_ascii_kernel, dim3(blocks), dim3(256), 0, st, d_ascii, w0, b, n, w1);
}

// a small table from PINNED host memory into device memory by the compute queue itself (coalesced reads over the link): a copy
// engine takes its work in the order it was queued, and a table that a pass needs at its start would wait behind every staging
// copy of the sub-batches queued in front of it (4 ms of a 7 ms call, api.hip: batch_stage)
__global__ void copy_words_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
void launch_copy_words(hipStream_t st, uint32_t *dst, const uint32_t *pinned_src, uint64_t n_words) {
    if (n_words == 0) return;
    hipLaunchKernelGGL(copy_words_kernel, dim3((uint32_t)((n_words + 255) / 256)), dim3(256), 0, st, dst, pinned_src, n_words);
}

void launch_sanitize_packed(hipStream_t st, const BatchDev &b, uint32_t n, uint64_t w0, uint64_t w1, int has_valid) {
    if (w1 <= w0) return;
    const uint32_t blocks = (uint32_t)((w1 - w0 + 255) / 256);
    hipLaunchKernelGGL(sanitize_packed_kernel, dim3(blocks), dim3(256), 0, st, b, n, w0, w1, has_valid);
}

void launch_synth(hipStream_t st, const BatchDev &b, uint32_t n, uint64_t total_words, uint64_t seed,
                  uint64_t contig0, const uint64_t *d_ids) {
    if (total_words == 0) return;
    const uint32_t blocks = (uint32_t)((total_words + 255) / 256);
    hipLaunchKernelGGL(synth_kernel, dim3(blocks), dim3(256), 0, st, b, n, total_words, seed, contig0, d_ids);
}

}  // namespace pgr
