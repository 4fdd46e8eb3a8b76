// ubench_valu.hip -- issue-rate micro-benchmarks for the integer VALU ops the SHIMMER kernels lean on.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
// Output: cycles per wave-instruction per SIMD (2.0 = full rate for wave64 on a SIMD-32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

__device__ unsigned long long g_cyc[256 * 8 * 4];
__device__ unsigned long long g_wall[256 * 8 * 4];  // the same interval in ticks of the constant 100 MHz clock  // shader cycles (s_memtime) every wave spent in its loop

#ifndef ITERS
#define ITERS 8192   // x 4 copies of the body per iteration: ~2-4 ms per launch (launch overhead, clock ramp and the loop's scalar instructions are negligible)
#endif
#define UNROLL 8

#define KERNEL(name, decl, body)                                            \
    __global__ __launch_bounds__(256) void name(uint32_t *out, uint32_t seed) { \
        decl;                                                               \
        const unsigned long long c0 = __builtin_readcyclecounter();         \
        const unsigned long long w0 = wall_clock64();                       \
        for (int it = 0; it < ITERS; ++it) {                                \
            body body body body                                             \
        }                                                                   \
        const unsigned long long c1 = __builtin_readcyclecounter();         \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(acc);              \
        const unsigned long long w1 = wall_clock64();                       \
        if ((threadIdx.x & 63) == 0) {                                      \
            g_cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;           \
            g_wall[blockIdx.x * 4 + (threadIdx.x >> 6)] = w1 - w0;          \
        }                                                                   \
    }

// 8 independent chains a0..a7 (32-bit) / q0..q7 (64-bit)
#define DECL32 uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1; uint32_t acc = 0
#define FIN32 acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
#define DECL64 uint64_t q0 = seed + threadIdx.x, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7, q4 = q0 * 11, q5 = q0 * 13, q6 = q0 * 17, q7 = q0 * 19; uint64_t b = ((uint64_t)seed << 20) | 1; uint64_t acc = 0
#define FIN64 acc = q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7;

#define REP8_32(INS) \
    asm volatile(INS : "+v"(a0) : "v"(b)); asm volatile(INS : "+v"(a1) : "v"(b)); asm volatile(INS : "+v"(a2) : "v"(b)); asm volatile(INS : "+v"(a3) : "v"(b)); \
    asm volatile(INS : "+v"(a4) : "v"(b)); asm volatile(INS : "+v"(a5) : "v"(b)); asm volatile(INS : "+v"(a6) : "v"(b)); asm volatile(INS : "+v"(a7) : "v"(b));
#define REP8_64(INS) \
    asm volatile(INS : "+v"(q0) : "v"(b)); asm volatile(INS : "+v"(q1) : "v"(b)); asm volatile(INS : "+v"(q2) : "v"(b)); asm volatile(INS : "+v"(q3) : "v"(b)); \
    asm volatile(INS : "+v"(q4) : "v"(b)); asm volatile(INS : "+v"(q5) : "v"(b)); asm volatile(INS : "+v"(q6) : "v"(b)); asm volatile(INS : "+v"(q7) : "v"(b));

KERNEL(k_add_u32, DECL32, REP8_32("v_add_u32 %0, %0, %1") FIN32)
KERNEL(k_xor_b32, DECL32, REP8_32("v_xor_b32 %0, %0, %1") FIN32)
KERNEL(k_or_b32, DECL32, REP8_32("v_or_b32 %0, %0, %1") FIN32)
KERNEL(k_and_b32, DECL32, REP8_32("v_and_b32 %0, %0, %1") FIN32)
KERNEL(k_not_b32, DECL32, REP8_32("v_not_b32 %0, %0") FIN32)
KERNEL(k_lshrrev_b32, DECL32, REP8_32("v_lshrrev_b32 %0, 3, %0") FIN32)
KERNEL(k_lshlrev_b32, DECL32, REP8_32("v_lshlrev_b32 %0, 3, %0") FIN32)
KERNEL(k_sub_u32, DECL32, REP8_32("v_sub_u32 %0, %0, %1") FIN32)
KERNEL(k_fma_f32, DECL32, REP8_32("v_fma_f32 %0, %0, %1, %1") FIN32)
// round 4: the guide's "2 cycles per wave64 instruction" against 3.7 for this v_fma_f32 -- is it the operand pattern (%1 twice)?
// Three distinct VGPR sources, the VOP2 forms, and the packed form the fp32 peak is quoted on.
#define REP8_32C(INS) \
    asm volatile(INS : "+v"(a0) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a1) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a2) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a3) : "v"(b), "v"(c)); \
    asm volatile(INS : "+v"(a4) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a5) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a6) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a7) : "v"(b), "v"(c));
#define REP8_64C(INS) \
    asm volatile(INS : "+v"(q0) : "v"(b), "v"(c)); asm volatile(INS : "+v"(q1) : "v"(b), "v"(c)); asm volatile(INS : "+v"(q2) : "v"(b), "v"(c)); asm volatile(INS : "+v"(q3) : "v"(b), "v"(c)); \
    asm volatile(INS : "+v"(q4) : "v"(b), "v"(c)); asm volatile(INS : "+v"(q5) : "v"(b), "v"(c)); asm volatile(INS : "+v"(q6) : "v"(b), "v"(c)); asm volatile(INS : "+v"(q7) : "v"(b), "v"(c));
KERNEL(k_fma_f32_3src, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_fma_f32 %0, %0, %1, %2") FIN32)
KERNEL(k_fmac_f32, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_fmac_f32 %0, %1, %2") FIN32)
KERNEL(k_mul_f32, DECL32, REP8_32("v_mul_f32 %0, %0, %1") FIN32)
KERNEL(k_add_f32, DECL32, REP8_32("v_add_f32 %0, %0, %1") FIN32)
KERNEL(k_pk_fma_f32, DECL64; uint64_t c = ((uint64_t)seed << 9) | 5, REP8_64C("v_pk_fma_f32 %0, %0, %1, %2") FIN64)
KERNEL(k_pk_mul_f32, DECL64, REP8_64("v_pk_mul_f32 %0, %0, %1") FIN64)
KERNEL(k_fma_f64, DECL64; uint64_t c = ((uint64_t)seed << 9) | 5, REP8_64C("v_fma_f64 %0, %0, %1, %2") FIN64)
// ... and the three-source integer opcodes, which round 3 had measured with one VGPR feeding two source operands as well
KERNEL(k_bfi_b32_3src, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_bfi_b32 %0, %1, %0, %2") FIN32)
KERNEL(k_and_or_b32_3src, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_and_or_b32 %0, %0, %1, %2") FIN32)
KERNEL(k_or3_b32_3src, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_or3_b32 %0, %0, %1, %2") FIN32)
KERNEL(k_add3_u32_3src, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_add3_u32 %0, %0, %1, %2") FIN32)
KERNEL(k_xad_u32_3src, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_xad_u32 %0, %0, %1, %2") FIN32)
KERNEL(k_bitop3_b32_3src, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96") FIN32)
KERNEL(k_min3_u32_3src, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_min3_u32 %0, %0, %1, %2") FIN32)
KERNEL(k_mad_u32_u24_3src, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_mad_u32_u24 %0, %0, %1, %2") FIN32)
KERNEL(k_alignbit_3src, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_alignbit_b32 %0, %0, %1, %2") FIN32)
KERNEL(k_lshl_add_u32_3src, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_lshl_add_u32 %0, %0, %1, %2") FIN32)
KERNEL(k_perm_b32_3src, DECL32; uint32_t c = seed * 7 + 3, REP8_32C("v_perm_b32 %0, %0, %1, %2") FIN32)
KERNEL(k_bfi_b32, DECL32, REP8_32("v_bfi_b32 %0, %1, %0, %1") FIN32)
KERNEL(k_bfe_i32, DECL32, REP8_32("v_bfe_i32 %0, %0, 3, 1") FIN32)
KERNEL(k_and_or_b32, DECL32, REP8_32("v_and_or_b32 %0, %0, %1, %1") FIN32)
KERNEL(k_cmp_eq_u32, DECL32, REP8_32("v_cmp_eq_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc") FIN32)
KERNEL(k_add_co_only, DECL32, REP8_32("v_add_co_u32 %0, vcc, %0, %1") FIN32)
KERNEL(k_cmp_lt_u64_only, DECL64, REP8_64("v_cmp_lt_u64 vcc, %0, %1\n v_lshl_add_u64 %0, %0, 1, %1") FIN64)
KERNEL(k_cmp_eq_u64_only, DECL64, REP8_64("v_cmp_eq_u64 vcc, %0, %1\n v_lshl_add_u64 %0, %0, 1, %1") FIN64)
KERNEL(k_cmp_lt_f64_only, DECL64, REP8_64("v_cmp_lt_f64 vcc, %0, %1\n v_lshl_add_u64 %0, %0, 1, %1") FIN64)
KERNEL(k_cmp_eq_f64_only, DECL64, REP8_64("v_cmp_eq_f64 vcc, %0, %1\n v_lshl_add_u64 %0, %0, 1, %1") FIN64)
KERNEL(k_cmp_cnd2, DECL32, REP8_32("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc") FIN32)
KERNEL(k_cmp_cnd4, DECL32, REP8_32("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc") FIN32)
KERNEL(k_cmp_sgpr_cnd2, DECL32, REP8_32("v_cmp_lt_u32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %0, %1, %0, s[20:21]") FIN32)
KERNEL(k_cmp_addc, DECL32, REP8_32("v_cmp_lt_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc") FIN32)
KERNEL(k_cmpf64_addc, DECL64; uint32_t acc2 = 0, REP8_64("v_cmp_lt_f64 vcc, %0, %1\n v_lshl_add_u64 %0, %0, 1, %1") asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc2) :: "vcc"); FIN64 acc ^= acc2;)
KERNEL(k_mov_b32, DECL32, REP8_32("v_mov_b32 %0, %1") FIN32)
KERNEL(k_ashrrev_i32, DECL32, REP8_32("v_ashrrev_i32 %0, 3, %0") FIN32)
KERNEL(k_or3_b32, DECL32, REP8_32("v_or3_b32 %0, %0, %1, %1") FIN32)
KERNEL(k_bcnt, DECL32, REP8_32("v_bcnt_u32_b32 %0, %0, %1") FIN32)
KERNEL(k_sub_co_subb, DECL32, REP8_32("v_sub_co_u32 %0, vcc, %0, %1\n v_subb_co_u32 %0, vcc, %0, %1, vcc") FIN32)
KERNEL(k_add_co_addc, DECL32, REP8_32("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc") FIN32)
KERNEL(k_mov_b64, DECL64, REP8_64("v_mov_b64 %0, %1") FIN64)
KERNEL(k_min_u32, DECL32, REP8_32("v_min_u32 %0, %0, %1") FIN32)
KERNEL(k_min3_u32, DECL32, REP8_32("v_min3_u32 %0, %0, %1, %1") FIN32)
KERNEL(k_alignbit, DECL32, REP8_32("v_alignbit_b32 %0, %0, %1, 7") FIN32)
KERNEL(k_bfrev, DECL32, REP8_32("v_bfrev_b32 %0, %0") FIN32)
KERNEL(k_mul_lo_u32, DECL32, REP8_32("v_mul_lo_u32 %0, %0, %1") FIN32)
KERNEL(k_mul_u32_u24, DECL32, REP8_32("v_mul_u32_u24 %0, %0, %1") FIN32)
KERNEL(k_mad_u32_u24, DECL32, REP8_32("v_mad_u32_u24 %0, %0, %1, %1") FIN32)
KERNEL(k_add3_u32, DECL32, REP8_32("v_add3_u32 %0, %0, %1, %1") FIN32)
KERNEL(k_lshl_add_u32, DECL32, REP8_32("v_lshl_add_u32 %0, %0, 3, %1") FIN32)
KERNEL(k_lshl_or_b32, DECL32, REP8_32("v_lshl_or_b32 %0, %0, 3, %1") FIN32)
KERNEL(k_xad_u32, DECL32, REP8_32("v_xad_u32 %0, %0, %1, %1") FIN32)
KERNEL(k_bfe_u32, DECL32, REP8_32("v_bfe_u32 %0, %0, 3, 17") FIN32)
KERNEL(k_cndmask, DECL32, REP8_32("v_cndmask_b32 %0, %0, %1, vcc") FIN32)
#define BAR64 asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
#define ALL64(EXPR) { uint64_t q; q = q0; q0 = EXPR; q = q1; q1 = EXPR; q = q2; q2 = EXPR; q = q3; q3 = EXPR; q = q4; q4 = EXPR; q = q5; q5 = EXPR; q = q6; q6 = EXPR; q = q7; q7 = EXPR; BAR64 }
__device__ __forceinline__ uint64_t h_ref(uint64_t key) {
    key = (~key) + (key << 21); key = key ^ (key >> 24); key = (key + (key << 3)) + (key << 8); key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4); key = key ^ (key >> 28); key = key + (key << 31); return key; }
__device__ __forceinline__ uint64_t mk64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
// 32-bit formulation: every step on (lo,hi) halves with funnel shifts and carry adds
__device__ __forceinline__ uint64_t h_32(uint64_t key) {
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32), tl, th;
    // key = ~key + (key << 21)
    tl = lo << 21; th = __builtin_amdgcn_alignbit(hi, lo, 11);
    { uint64_t s = mk64(~lo, ~hi) + mk64(tl, th); lo = (uint32_t)s; hi = (uint32_t)(s >> 32); }
    // key ^= key >> 24
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 24); hi ^= hi >> 24;
    // key *= 265  (= key + key<<3 + key<<8)
    { uint64_t s = mk64(lo, hi) + mk64(lo << 3, __builtin_amdgcn_alignbit(hi, lo, 29)) + mk64(lo << 8, __builtin_amdgcn_alignbit(hi, lo, 24)); lo = (uint32_t)s; hi = (uint32_t)(s >> 32); }
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 14); hi ^= hi >> 14;
    { uint64_t s = mk64(lo, hi) + mk64(lo << 2, __builtin_amdgcn_alignbit(hi, lo, 30)) + mk64(lo << 4, __builtin_amdgcn_alignbit(hi, lo, 28)); lo = (uint32_t)s; hi = (uint32_t)(s >> 32); }
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 28); hi ^= hi >> 28;
    // key += key << 31
    { uint64_t s = mk64(lo, hi) + mk64(lo << 31, __builtin_amdgcn_alignbit(hi, lo, 1)); lo = (uint32_t)s; hi = (uint32_t)(s >> 32); }
    return mk64(lo, hi);
}
// multiply formulation: 265 and 21 as 64-bit multiplies
__device__ __forceinline__ uint64_t h_mul(uint64_t key) {
    key = (~key) + (key << 21); key = key ^ (key >> 24); key = key * 265; key = key ^ (key >> 14);
    key = key * 21; key = key ^ (key >> 28); key = key + (key << 31); return key; }
__device__ __forceinline__ uint64_t umin64c(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t fmin64c(uint64_t a, uint64_t b) { return (uint64_t)__double_as_longlong(__builtin_fmin(__longlong_as_double((long long)a), __longlong_as_double((long long)b))); }
KERNEL(k_cpp_add64, DECL64, ALL64(q + b) FIN64)
KERNEL(k_cpp_shl64, DECL64, ALL64(q << 3) FIN64)
KERNEL(k_cpp_shr64, DECL64, ALL64(q >> 24) FIN64)
KERNEL(k_cpp_lshladd, DECL64, ALL64((q << 3) + b) FIN64)
KERNEL(k_cpp_mul265, DECL64, ALL64(q * 265) FIN64)
KERNEL(k_cpp_shift265, DECL64, ALL64((q + (q << 3)) + (q << 8)) FIN64)
KERNEL(k_cpp_umin64, DECL64, ALL64(umin64c(q, b)) FIN64)
KERNEL(k_cpp_fmin64, DECL64, ALL64(fmin64c(q, b)) FIN64)
KERNEL(k_cpp_xorshr, DECL64, ALL64(q ^ (q >> 24)) FIN64)
KERNEL(k_hash_ref, DECL64, ALL64(h_ref(q)) FIN64)
KERNEL(k_hash_32, DECL64, ALL64(h_32(q)) FIN64)
KERNEL(k_hash_mul, DECL64, ALL64(h_mul(q)) FIN64)
KERNEL(k_lshlrev_b64, DECL64, REP8_64("v_lshlrev_b64 %0, 3, %0") FIN64)
KERNEL(k_lshrrev_b64, DECL64, REP8_64("v_lshrrev_b64 %0, 3, %0") FIN64)
KERNEL(k_lshl_add_u64, DECL64, REP8_64("v_lshl_add_u64 %0, %0, 3, %1") FIN64)
KERNEL(k_mad_u64_u32, DECL64; uint32_t m = seed | 3, ALL64(q + (uint64_t)(uint32_t)q * m) FIN64)
KERNEL(k_min_f64, DECL64, REP8_64("v_min_f64 %0, %0, %1") FIN64)
KERNEL(k_max_f64, DECL64, REP8_64("v_max_f64 %0, %0, %1") FIN64)
KERNEL(k_cmp_lt_u32, DECL32, REP8_32("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc") FIN32)
KERNEL(k_mov_dpp, DECL32, REP8_32("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") FIN32)
KERNEL(k_min_dpp, DECL32, REP8_32("v_min_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") FIN32)

// ---- round 3: every opcode of level1_tile_kernel measured by itself (no "class" costs left in tools/isa_histogram.py).
// Compares write a scalar pair that nothing reads; selects / carries read a scalar pair that nothing rewrites (s[90:93] are
// far above what these tiny kernels allocate), so no dependency stall enters -- the issue rate is what is measured.
KERNEL(k_cndmask_sgpr, DECL32, REP8_32("v_cndmask_b32 %0, %0, %1, s[90:91]") FIN32)
KERNEL(k_addc_sgpr, DECL32, REP8_32("v_addc_co_u32 %0, s[92:93], %0, %1, s[90:91]") FIN32)
KERNEL(k_cmp_lt_u64_sgpr, DECL64, REP8_64("v_cmp_lt_u64 s[92:93], %0, %1") FIN64)
KERNEL(k_cmp_eq_u64_sgpr, DECL64, REP8_64("v_cmp_eq_u64 s[92:93], %0, %1") FIN64)
KERNEL(k_cmp_gt_u64_sgpr, DECL64, REP8_64("v_cmp_gt_u64 s[92:93], %0, %1") FIN64)
KERNEL(k_cmp_le_u64_sgpr, DECL64, REP8_64("v_cmp_le_u64 s[92:93], %0, %1") FIN64)
KERNEL(k_cmp_lt_i64_sgpr, DECL64, REP8_64("v_cmp_lt_i64 s[92:93], %0, %1") FIN64)
KERNEL(k_cmp_gt_i64_sgpr, DECL64, REP8_64("v_cmp_gt_i64 s[92:93], %0, %1") FIN64)
KERNEL(k_cmp_le_i64_sgpr, DECL64, REP8_64("v_cmp_le_i64 s[92:93], %0, %1") FIN64)
KERNEL(k_cmp_lt_f64_sgpr, DECL64, REP8_64("v_cmp_lt_f64 s[92:93], %0, %1") FIN64)
KERNEL(k_cmp_eq_u32_sgpr, DECL32, REP8_32("v_cmp_eq_u32 s[92:93], %0, %1") FIN32)
KERNEL(k_cmp_ne_u32_sgpr, DECL32, REP8_32("v_cmp_ne_u32 s[92:93], %0, %1") FIN32)
KERNEL(k_cmp_lt_u32_sgpr, DECL32, REP8_32("v_cmp_lt_u32 s[92:93], %0, %1") FIN32)
KERNEL(k_cmp_gt_u32_sgpr, DECL32, REP8_32("v_cmp_gt_u32 s[92:93], %0, %1") FIN32)
KERNEL(k_cmp_ge_u32_sgpr, DECL32, REP8_32("v_cmp_ge_u32 s[92:93], %0, %1") FIN32)
KERNEL(k_max_u32, DECL32, REP8_32("v_max_u32 %0, %0, %1") FIN32)
KERNEL(k_max_i32, DECL32, REP8_32("v_max_i32 %0, %0, %1") FIN32)
KERNEL(k_med3_i32, DECL32, REP8_32("v_med3_i32 %0, %0, %1, %1") FIN32)
KERNEL(k_mul_hi_u32, DECL32, REP8_32("v_mul_hi_u32 %0, %0, %1") FIN32)
KERNEL(k_ffbl_b32, DECL32, REP8_32("v_ffbl_b32 %0, %0") FIN32)
KERNEL(k_mbcnt_lo, DECL32, REP8_32("v_mbcnt_lo_u32_b32 %0, %1, %0") FIN32)
KERNEL(k_mbcnt_hi, DECL32, REP8_32("v_mbcnt_hi_u32_b32 %0, %1, %0") FIN32)
KERNEL(k_subrev_u32, DECL32, REP8_32("v_subrev_u32 %0, %0, %1") FIN32)
KERNEL(k_bitop3_b32, DECL32, REP8_32("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96") FIN32)
KERNEL(k_readfirstlane, DECL32, REP8_32("v_readfirstlane_b32 s92, %0") FIN32)
// v_mad_u64_u32 as the tile kernel issues it (one 32 x 32 -> 64 product added to a 64-bit register pair)
KERNEL(k_mad_u64_u32_asm, DECL64; uint32_t m0 = seed | 3,
       asm volatile("v_mad_u64_u32 %0, s[92:93], %1, %1, %0" : "+v"(q0) : "v"(m0)); asm volatile("v_mad_u64_u32 %0, s[92:93], %1, %1, %0" : "+v"(q1) : "v"(m0));
       asm volatile("v_mad_u64_u32 %0, s[92:93], %1, %1, %0" : "+v"(q2) : "v"(m0)); asm volatile("v_mad_u64_u32 %0, s[92:93], %1, %1, %0" : "+v"(q3) : "v"(m0));
       asm volatile("v_mad_u64_u32 %0, s[92:93], %1, %1, %0" : "+v"(q4) : "v"(m0)); asm volatile("v_mad_u64_u32 %0, s[92:93], %1, %1, %0" : "+v"(q5) : "v"(m0));
       asm volatile("v_mad_u64_u32 %0, s[92:93], %1, %1, %0" : "+v"(q6) : "v"(m0)); asm volatile("v_mad_u64_u32 %0, s[92:93], %1, %1, %0" : "+v"(q7) : "v"(m0)); FIN64)
// the left / right shift split of round 2 (4.15 vs 2.45 cycles): the same opcodes with the shift amount in a VGPR and in the
// 64-bit (VOP3) encoding
KERNEL(k_lshlrev_b32_vgpr, DECL32, REP8_32("v_lshlrev_b32 %0, %1, %0") FIN32)
KERNEL(k_lshrrev_b32_vgpr, DECL32, REP8_32("v_lshrrev_b32 %0, %1, %0") FIN32)
KERNEL(k_lshlrev_b32_e64, DECL32, REP8_32("v_lshlrev_b32_e64 %0, 3, %0") FIN32)
KERNEL(k_lshrrev_b32_e64, DECL32, REP8_32("v_lshrrev_b32_e64 %0, 3, %0") FIN32)
KERNEL(k_lshlrev_b32_by1, DECL32, REP8_32("v_lshlrev_b32 %0, 1, %0") FIN32)
KERNEL(k_ashrrev_i32_vgpr, DECL32, REP8_32("v_ashrrev_i32 %0, %1, %0") FIN32)

__global__ void k_clock(unsigned long long *out) {
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long w0 = wall_clock64();
    uint32_t a = threadIdx.x;
    for (int i = 0; i < 1000000; ++i) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a));
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = a; }
}
struct Case { const char *name; void (*fn)(uint32_t *, uint32_t); int per_iter; };

int main() {
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
    uint32_t *d; hipMalloc(&d, blocks * 256 * 4);
    std::vector<Case> cases = {
        {"v_add_u32", k_add_u32, 8}, {"v_xor_b32", k_xor_b32, 8}, {"v_or_b32", k_or_b32, 8}, {"v_and_b32", k_and_b32, 8}, {"v_not_b32", k_not_b32, 8},
        {"v_lshrrev_b32", k_lshrrev_b32, 8}, {"v_lshlrev_b32", k_lshlrev_b32, 8}, {"v_sub_u32", k_sub_u32, 8}, {"v_fma_f32", k_fma_f32, 8},
        {"v_fma_f32 (3 sources)", k_fma_f32_3src, 8}, {"v_fmac_f32", k_fmac_f32, 8}, {"v_mul_f32", k_mul_f32, 8}, {"v_add_f32", k_add_f32, 8},
        {"v_pk_fma_f32", k_pk_fma_f32, 8}, {"v_pk_mul_f32", k_pk_mul_f32, 8}, {"v_fma_f64", k_fma_f64, 8},
        {"bfi 3src", k_bfi_b32_3src, 8}, {"and_or 3src", k_and_or_b32_3src, 8}, {"or3 3src", k_or3_b32_3src, 8}, {"add3 3src", k_add3_u32_3src, 8},
        {"xad 3src", k_xad_u32_3src, 8}, {"bitop3 3src", k_bitop3_b32_3src, 8}, {"min3 3src", k_min3_u32_3src, 8}, {"mad_u32_u24 3src", k_mad_u32_u24_3src, 8},
        {"alignbit 3src", k_alignbit_3src, 8}, {"lshl_add_u32 3src", k_lshl_add_u32_3src, 8}, {"perm 3src", k_perm_b32_3src, 8},
        {"v_bfi_b32", k_bfi_b32, 8}, {"v_bfe_i32", k_bfe_i32, 8}, {"v_and_or_b32", k_and_or_b32, 8},
        {"v_cmp_eq_u32 + v_addc (pair)", k_cmp_eq_u32, 8}, {"v_add_co_u32", k_add_co_only, 8},
        {"v_cmp_lt_u64 + lshl_add_u64 (pair)", k_cmp_lt_u64_only, 8}, {"v_cmp_eq_u64 + lshl_add_u64 (pair)", k_cmp_eq_u64_only, 8},
        {"v_cmp_lt_f64 + lshl_add_u64 (pair)", k_cmp_lt_f64_only, 8}, {"v_cmp_eq_f64 + lshl_add_u64 (pair)", k_cmp_eq_f64_only, 8},
        {"cmp_u32 + 2 cndmask (vcc)", k_cmp_cnd2, 8}, {"cmp_u32 + 4 cndmask (vcc)", k_cmp_cnd4, 8},
        {"cmp_u32->sgpr + 2 cndmask_e64", k_cmp_sgpr_cnd2, 8}, {"cmp_u32 + v_addc (bit accumulate)", k_cmp_addc, 8},
        {"v_mov_b32", k_mov_b32, 8}, {"v_ashrrev_i32", k_ashrrev_i32, 8}, {"v_or3_b32", k_or3_b32, 8}, {"v_bcnt_u32_b32", k_bcnt, 8},
        {"v_sub_co_u32 + v_subb_co_u32 (pair)", k_sub_co_subb, 8}, {"v_add_co_u32 + v_addc_co_u32 (pair)", k_add_co_addc, 8}, {"v_mov_b64", k_mov_b64, 8},
        {"v_min_u32", k_min_u32, 8}, {"v_min3_u32", k_min3_u32, 8},
        {"v_alignbit_b32", k_alignbit, 8}, {"v_bfrev_b32", k_bfrev, 8}, {"v_mul_lo_u32", k_mul_lo_u32, 8},
        {"v_mul_u32_u24", k_mul_u32_u24, 8}, {"v_mad_u32_u24", k_mad_u32_u24, 8}, {"v_add3_u32", k_add3_u32, 8},
        {"v_lshl_add_u32", k_lshl_add_u32, 8}, {"v_lshl_or_b32", k_lshl_or_b32, 8}, {"v_xad_u32", k_xad_u32, 8},
        {"v_bfe_u32", k_bfe_u32, 8}, {"v_cndmask_b32", k_cndmask, 8},
        {"v_lshlrev_b64", k_lshlrev_b64, 8}, {"v_lshrrev_b64", k_lshrrev_b64, 8},
        {"v_lshl_add_u64", k_lshl_add_u64, 8}, {"v_mad_u64_u32", k_mad_u64_u32, 8}, {"v_min_f64", k_min_f64, 8},
        {"v_max_f64", k_max_f64, 8}, {"c++ q+b (add64)", k_cpp_add64, 8}, {"c++ q<<3", k_cpp_shl64, 8}, {"c++ q>>24", k_cpp_shr64, 8},
        {"c++ (q<<3)+b", k_cpp_lshladd, 8}, {"c++ q*265", k_cpp_mul265, 8}, {"c++ q+(q<<3)+(q<<8)", k_cpp_shift265, 8},
        {"c++ umin64", k_cpp_umin64, 8}, {"c++ fmin64 (f64 min)", k_cpp_fmin64, 8}, {"c++ q^(q>>24)", k_cpp_xorshr, 8},
        {"u64hash reference form", k_hash_ref, 8}, {"u64hash 32-bit halves", k_hash_32, 8}, {"u64hash with 64-bit muls", k_hash_mul, 8},
        {"v_cmp_lt_u32+cnd (pair)", k_cmp_lt_u32, 8}, {"v_mov_b32_dpp", k_mov_dpp, 8}, {"v_min_u32_dpp", k_min_dpp, 8},
        {"cndmask_sgpr", k_cndmask_sgpr, 8}, {"addc_sgpr", k_addc_sgpr, 8}, {"cmp_lt_u64_sgpr", k_cmp_lt_u64_sgpr, 8}, {"cmp_eq_u64_sgpr", k_cmp_eq_u64_sgpr, 8}, {"cmp_gt_u64_sgpr", k_cmp_gt_u64_sgpr, 8}, {"cmp_le_u64_sgpr", k_cmp_le_u64_sgpr, 8}, {"cmp_lt_i64_sgpr", k_cmp_lt_i64_sgpr, 8}, {"cmp_gt_i64_sgpr", k_cmp_gt_i64_sgpr, 8}, {"cmp_le_i64_sgpr", k_cmp_le_i64_sgpr, 8}, {"cmp_lt_f64_sgpr", k_cmp_lt_f64_sgpr, 8}, {"cmp_eq_u32_sgpr", k_cmp_eq_u32_sgpr, 8}, {"cmp_ne_u32_sgpr", k_cmp_ne_u32_sgpr, 8}, {"cmp_lt_u32_sgpr", k_cmp_lt_u32_sgpr, 8}, {"cmp_gt_u32_sgpr", k_cmp_gt_u32_sgpr, 8}, {"cmp_ge_u32_sgpr", k_cmp_ge_u32_sgpr, 8}, {"max_u32", k_max_u32, 8}, {"max_i32", k_max_i32, 8}, {"med3_i32", k_med3_i32, 8}, {"mul_hi_u32", k_mul_hi_u32, 8}, {"ffbl_b32", k_ffbl_b32, 8}, {"mbcnt_lo", k_mbcnt_lo, 8}, {"mbcnt_hi", k_mbcnt_hi, 8}, {"subrev_u32", k_subrev_u32, 8}, {"bitop3_b32", k_bitop3_b32, 8}, {"readfirstlane", k_readfirstlane, 8}, {"mad_u64_u32_asm", k_mad_u64_u32_asm, 8}, {"lshlrev_b32_vgpr", k_lshlrev_b32_vgpr, 8}, {"lshrrev_b32_vgpr", k_lshrrev_b32_vgpr, 8}, {"lshlrev_b32_e64", k_lshlrev_b32_e64, 8}, {"lshrrev_b32_e64", k_lshrrev_b32_e64, 8}, {"lshlrev_b32_by1", k_lshlrev_b32_by1, 8}, {"ashrrev_i32_vgpr", k_ashrrev_i32_vgpr, 8},
    };
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const double clk = prop.clockRate * 1e3;  // Hz
    printf("device %s  CUs %d  clock %.0f MHz\n", prop.gcnArchName, prop.multiProcessorCount, clk / 1e6);
    {
        unsigned long long *dc; hipMalloc(&dc, 64); unsigned long long hc[3];
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_clock<<<1, 64>>>(dc); hipDeviceSynchronize();
        hipEventRecord(e0); k_clock<<<1, 64>>>(dc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hc, dc, 24, hipMemcpyDeviceToHost);
        printf("clock calibration: %llu shader cycles, %llu wall ticks in %.3f ms -> shader %.0f MHz, wall %.0f MHz\n", hc[0], hc[1], ms, hc[0] / ms / 1e3, hc[1] / ms / 1e3);
    }
    printf("8 waves per SIMD, 8 independent chains per wave, %d iterations of 4 x 8 sequences.  cyc(nominal) = HIP-event time x nominal clock /\n"
           "sequences per SIMD.  cyc(memtime) = mean s_memtime ticks a wave spent in its loop / (sequences per wave x 8 waves).\n"
           "memtime MHz = s_memtime ticks per second of the constant 100 MHz wall clock, read by the same waves around the\n"
           "same loop: the rate s_memtime really ran at.  The authoritative cycle count is the PMC one: run this binary under\n"
           "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU and divide (GRBM_GUI_ACTIVE / 8 XCDs) by (SQ_INSTS_VALU / 1024 SIMDs)\n"
           "(tools/ubench_table.py does that).\n", ITERS);
    printf("%-36s %9s %12s %12s %12s\n", "sequence", "ms", "cyc(nominal)", "cyc(memtime)", "memtime MHz");
    for (auto &c : cases) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        c.fn<<<blocks, 256>>>(d, 12345);  // warm
        hipDeviceSynchronize();
        hipEventRecord(e0);
        c.fn<<<blocks, 256>>>(d, 12346);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // sequences issued per SIMD: waves per SIMD * ITERS * per_iter
        const double waves_per_simd = (double)blocks * 4 / (prop.multiProcessorCount * 4);
        const double seqs = waves_per_simd * ITERS * c.per_iter * 4;
        static unsigned long long hc[256 * 8 * 4];
        hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_cyc), sizeof(hc));
        double cs = 0;
        for (int i = 0; i < blocks * 4; ++i) cs += (double)hc[i];
        static unsigned long long hw[256 * 8 * 4];
        hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_wall), sizeof(hw));
        double ws = 0;
        for (int i = 0; i < blocks * 4; ++i) ws += (double)hw[i];
        const double cyc_wave = cs / (blocks * 4);
        const double cyc_mem = cyc_wave / ((double)ITERS * c.per_iter * 4 * waves_per_simd);
        printf("%-36s %9.3f %12.2f %12.2f %12.0f\n", c.name, ms, ms * 1e-3 * clk / seqs, cyc_mem, cs / ws * 100.0);
    }
    return 0;
}
