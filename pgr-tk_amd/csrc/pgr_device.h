// pgr_device.h -- device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pgr {

// Thomas Wang / minimap2 hash64 with a full 64-bit mask -- reference: pgr-db/src/shmmrutils.rs:271-280
__device__ __forceinline__ uint64_t u64hash(uint64_t key) {
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

// The same hash with the two multiplications spelled as v_lshl_add_u64 (shift <= 4) + one 64-bit shift/add.
// hipcc otherwise canonicalises key*265 / key*21 into v_mad_u64_u32 pairs plus register-pair moves
// (15 cycles each on gfx950 against 13.4 / 9.0 for the shift-add forms; tools/ubench_valu.hip).
__device__ __forceinline__ uint64_t lshl2_add(uint64_t a, uint64_t b) {
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 2, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint64_t lshl3_add(uint64_t a, uint64_t b) {
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 3, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint64_t lshl4_add(uint64_t a, uint64_t b) {
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 4, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint64_t shl31(uint64_t a) {  // opaque: "key + (key << 31)" would become a 64-bit multiply
    uint64_t r;
    asm("v_lshlrev_b64 %0, 31, %1" : "=v"(r) : "v"(a));
    return r;
}
// (measured and rejected: "(key << 21) - key - 1" with the -1 on a preset borrow, 2 x v_subb_co_u32 -- the vcc
// dependency makes it slower than 2 x v_not_b32 + v_lshl_add_u64: 22.5 vs 22.2 ms per 10 Gbp)
__device__ __forceinline__ uint64_t u64hash_sa(uint64_t key) {
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = lshl3_add(key, key) + (key << 8);  // key * 265
    key = key ^ (key >> 14);
    key = lshl4_add(key, lshl2_add(key, key));  // key * 21
    key = key ^ (key >> 28);
    key = key + shl31(key);
    return key;
}

// ((hi:lo) >> s) & 0xffffffff, s in 0..31  (v_alignbit_b32)
__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t s) {
    return __builtin_amdgcn_alignbit(hi, lo, s);
}

__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a > b ? a : b; }

// The same hash with the two steps that are multiplications by a 32-bit constant issued as multiplications: measured on
// gfx950 (profiles/r02_ubench/valu_cycles.txt) v_mad_u64_u32 and v_mul_lo_u32 cost what one 64-bit shift or add costs
// (4.2 cycles per wave64 instruction), so
//   ~key + (key << 21) = key * (2^21 - 1) - 1 : high word (v_not_b32 + v_lshl_add_u32) + ONE v_mad_u64_u32 whose addend
//                                               carries that high word and the -1            (3 instead of 4 instructions)
//   key * 265                                 : v_mul_lo_u32 (high word) + ONE v_mad_u64_u32 (2 instead of 3)
// 17 VALU instructions per hash instead of 20.  The asm keeps hipcc from re-associating them into a full 64 x 64 multiply.
__device__ __forceinline__ uint64_t u64hash_mad(uint64_t key) {
    {
        const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
        uint32_t h;  // high word of ~key + (key << 21) without the carry from the low word: ~hi + (hi << 21)
        asm("v_not_b32 %0, %1\n\tv_lshl_add_u32 %0, %1, 21, %0" : "=&v"(h) : "v"(hi));
        const uint64_t add = ((uint64_t)h << 32) | 0xFFFFFFFFull;  // + (h << 32) - 1
        uint64_t cy;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(key), "=s"(cy) : "v"(lo), "s"(0x1FFFFFu), "v"(add));
    }
    key = key ^ (key >> 24);
    {
        const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
        uint32_t h;
        asm("v_mul_lo_u32 %0, %1, %2" : "=v"(h) : "v"(hi), "s"(265u));
        const uint64_t add = (uint64_t)h << 32;
        uint64_t cy;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(key), "=s"(cy) : "v"(lo), "s"(265u), "v"(add));
    }
    key = key ^ (key >> 14);
    key = lshl4_add(key, lshl2_add(key, key));  // key * 21
    key = key ^ (key >> 28);
    key = key + shl31(key);
    return key;
}

// canonical k-mer -> (x, strand).  f/r: forward / reverse-complement bit planes
// (shmmrutils.rs:485-500): strand decided on the LOW plane only; x = hash << 8 | k.
__device__ __forceinline__ uint64_t kmer_x(uint64_t f0, uint64_t f1, uint64_t r0, uint64_t r1, uint32_t k,
                                           uint32_t &strand, uint64_t &hash) {
    const bool rev = r0 < f0;
    const uint64_t m0 = rev ? r0 : f0;
    const uint64_t m1 = rev ? r1 : f1;
    const uint64_t h = u64hash(m0) ^ u64hash(m1 ^ 0xAD12CF59ull);
    strand = rev ? 1u : 0u;
    hash = h;
    return (h << 8) | (uint64_t)k;
}

// reverse-complement plane of a forward plane value (all k bases valid):
// r bit (k-1-m) = ~f bit m  (shmmrutils.rs:469-475 applied k times)
__device__ __forceinline__ uint64_t rc_plane(uint64_t f, uint32_t k) { return __brevll(~f) >> (64 - k); }

// wave64 helpers ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int m) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((uint64_t)hi << 32) | lo;
}
// value of lane `src` (the same for the whole wave: a constant or a ballot result) -- v_readlane_b32, no trip through the LDS
// crossbar like ds_bpermute_b32 (a lone wavefront waits ~100 cycles for each of those)
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src) {
    const int l = __builtin_amdgcn_readfirstlane(src);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
// one DPP step on both halves of a 64-bit value; lanes without a source keep all ones
__device__ __forceinline__ uint64_t dpp_u64_ones(uint64_t v, const int ctrl, const int row_mask) {
    uint32_t lo, hi;
    switch (ctrl) {  // (the control word is an immediate of the instruction)
#define PGR_DPP_CASE(C, RM)                                                                                        \
    case C:                                                                                                        \
        lo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)v, C, RM, 0xf, false);                       \
        hi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(v >> 32), C, RM, 0xf, false);               \
        break;
        PGR_DPP_CASE(0x111, 0xf)
        PGR_DPP_CASE(0x112, 0xf)
        PGR_DPP_CASE(0x114, 0xf)
        PGR_DPP_CASE(0x118, 0xf)
        PGR_DPP_CASE(0x142, 0xa)
        PGR_DPP_CASE(0x143, 0xc)
        PGR_DPP_CASE(0x138, 0xf)
#undef PGR_DPP_CASE
    default:
        lo = hi = 0xFFFFFFFFu;
    }
    (void)row_mask;
    return ((uint64_t)hi << 32) | lo;
}
// inclusive prefix minimum over the wave (DPP row shifts + the two row broadcasts, as wave_incl_sum)
__device__ __forceinline__ uint64_t wave_incl_min64(uint64_t v) {
    v = umin64(v, dpp_u64_ones(v, 0x111, 0xf));  // row_shr:1
    v = umin64(v, dpp_u64_ones(v, 0x112, 0xf));  // row_shr:2
    v = umin64(v, dpp_u64_ones(v, 0x114, 0xf));  // row_shr:4
    v = umin64(v, dpp_u64_ones(v, 0x118, 0xf));  // row_shr:8
    v = umin64(v, dpp_u64_ones(v, 0x142, 0xa));  // row_bcast:15 -> rows 1, 3
    v = umin64(v, dpp_u64_ones(v, 0x143, 0xc));  // row_bcast:31 -> rows 2, 3
    return v;
}
// lane l <- lane l - 1 (lane 0: all ones)
__device__ __forceinline__ uint64_t wave_shr1_ones(uint64_t v) { return dpp_u64_ones(v, 0x138, 0xf); }

// minimum over the wave, in every lane: the inclusive prefix minimum's last lane (6 DPP steps + one v_readlane pair; the butterfly
// over ds_bpermute_b32 was 12 trips through the LDS crossbar -- a lone wavefront of the exact machine waits for each)
__device__ __forceinline__ uint64_t wave_min64(uint64_t v) { return readlane64(wave_incl_min64(v), 63); }
// inclusive prefix sum over the wave: DPP row shifts inside the rows of 16 lanes, then the two row broadcasts
// (6 v_add_u32_dpp instead of 6 x (ds_bpermute + compare + select + add))
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

}  // namespace pgr
