// scan.hip -- device-wide scan / sort primitives from rocPRIM (AMD's native primitives library).
// These are plain library calls around the hand-written kernels (counts -> offsets, records -> CSR);
// none of them is on the hot path's critical time.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "pgr_internal.h"

namespace pgr {

// rocPRIM asks the runtime for its last error after every launch (hipGetLastError) and hands whatever it finds back as ITS result:
// an error some earlier, unrelated call of this host thread left behind -- this library checks the return value of every call
// it makes and never looks at that state; PyTorch and the host program share the thread -- came back as "scan_counts(...): invalid
// argument" from a scan that had nothing wrong with it (round 6: once in a 149-test run, never in isolation).  Every wrapper
// takes stale state away first; with PGR_DEBUG_STALE=1 it says so.
void drop_stale_hip_error(const char *who) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        static const bool say = getenv("PGR_DEBUG_STALE") != nullptr;
        if (say) fprintf(stderr, "[pgr] %s: a stale HIP error of this thread was dropped: %s\n", who, hipGetErrorString(e));
    }
}

namespace {
struct U32toU64 {
    __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};
}  // namespace

// exclusive scan of n+1 u32 counts (caller guarantees in[n] == 0) into n+1 u64 offsets: out[n] = total
size_t scan_counts_temp_bytes(uint32_t n_plus_1) {
    size_t bytes = 0;
    auto it = rocprim::make_transform_iterator((const uint32_t *)nullptr, U32toU64());
    (void)rocprim::exclusive_scan(nullptr, bytes, it, (uint64_t *)nullptr, (uint64_t)0, (size_t)n_plus_1,
                                  rocprim::plus<uint64_t>());
    return bytes;
}

hipError_t scan_counts(hipStream_t st, void *temp, size_t temp_bytes, const uint32_t *in, uint64_t *out,
                       uint32_t n_plus_1) {
    drop_stale_hip_error("scan_counts");
    auto it = rocprim::make_transform_iterator(in, U32toU64());
    return rocprim::exclusive_scan(temp, temp_bytes, it, out, (uint64_t)0, (size_t)n_plus_1,
                                   rocprim::plus<uint64_t>(), st);
}

// ---- the same scan without a byte of LDS, for a stream whose kernels run BESIDE the tile kernel (pgr_internal.h).  A wavefront
// owns 1024 consecutive counts (16 rows of 64): (1) every wavefront adds up its counts, (2) ONE wavefront scans the partial sums
// (64 per step: 40 steps for the 2.56 M segments of a 10 Gbp batch), (3) every wavefront scans its rows again on top of its base.
// Wave scans by __shfl_up (the LDS crossbar, no allocation).  Two passes over 4 bytes per count: nothing next to the tiles' 2.5 GB.
namespace {
__device__ __forceinline__ unsigned long long wave_incl_sum64(unsigned long long v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(v, d, 64);
        if ((int)lane >= d) v += o;
    }
    return v;
}
constexpr uint32_t NL_ROWS = 16, NL_CHUNK = 64 * NL_ROWS;
__global__ __launch_bounds__(256) void nolds_partials_kernel(const uint32_t *__restrict__ in, uint32_t n, unsigned long long *__restrict__ part) {
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint64_t base = (uint64_t)w * NL_CHUNK;
    if (base >= n) return;
    unsigned long long s = 0;
#pragma unroll
    for (uint32_t j = 0; j < NL_ROWS; ++j) {
        const uint64_t e = base + j * 64 + lane;
        s += e < n ? in[e] : 0u;
    }
    s = wave_incl_sum64(s, lane);
    if (lane == 63) part[w] = s;
}
__global__ __launch_bounds__(64) void nolds_scan_partials_kernel(unsigned long long *__restrict__ part, uint32_t n_part) {
    const uint32_t lane = threadIdx.x;
    unsigned long long carry = 0;
    for (uint32_t p0 = 0; p0 < n_part; p0 += 64) {
        const unsigned long long v = p0 + lane < n_part ? part[p0 + lane] : 0ull;
        const unsigned long long inc = wave_incl_sum64(v, lane);
        if (p0 + lane < n_part) part[p0 + lane] = carry + inc - v;  // exclusive
        carry += __shfl(inc, 63, 64);
    }
}
__global__ __launch_bounds__(256) void nolds_apply_kernel(const uint32_t *__restrict__ in, uint32_t n, const unsigned long long *__restrict__ part,
                                                          uint64_t *__restrict__ out) {
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint64_t base = (uint64_t)w * NL_CHUNK;
    if (base >= n) return;
    unsigned long long carry = part[w];
#pragma unroll
    for (uint32_t j = 0; j < NL_ROWS; ++j) {
        const uint64_t e = base + j * 64 + lane;
        const unsigned long long v = e < n ? in[e] : 0u;
        const unsigned long long inc = wave_incl_sum64(v, lane);
        if (e < n) out[e] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
}
}  // namespace
size_t scan_counts_nolds_temp_bytes(uint32_t n_plus_1) { return ((size_t)(n_plus_1 + NL_CHUNK - 1) / NL_CHUNK + 1) * sizeof(unsigned long long); }
hipError_t scan_counts_nolds(hipStream_t st, void *temp, size_t temp_bytes, const uint32_t *in, uint64_t *out, uint32_t n_plus_1) {
    if (n_plus_1 == 0) return hipSuccess;
    drop_stale_hip_error("scan_counts_nolds");
    const uint32_t n_part = (n_plus_1 + NL_CHUNK - 1) / NL_CHUNK;
    if (temp_bytes < (size_t)n_part * sizeof(unsigned long long)) return hipErrorInvalidValue;
    unsigned long long *part = (unsigned long long *)temp;
    hipLaunchKernelGGL(nolds_partials_kernel, dim3((n_part + 3) / 4), dim3(256), 0, st, in, n_plus_1, part);
    hipLaunchKernelGGL(nolds_scan_partials_kernel, dim3(1), dim3(64), 0, st, part, n_part);
    hipLaunchKernelGGL(nolds_apply_kernel, dim3((n_part + 3) / 4), dim3(256), 0, st, in, n_plus_1, part, out);
    return hipGetLastError();
}

// inclusive max-scan of u64 values in place
size_t scan_max_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (size_t)n, rocprim::maximum<uint64_t>());
    return bytes;
}
hipError_t scan_max_inplace(hipStream_t st, void *temp, size_t temp_bytes, uint64_t *v, uint32_t n) {
    drop_stale_hip_error("scan_max_inplace");
    return rocprim::inclusive_scan(temp, temp_bytes, (const uint64_t *)v, v, (size_t)n, rocprim::maximum<uint64_t>(), st);
}

// stable LSD radix sort of 64-bit keys (bits [0, end_bit)) with a 32-bit payload
size_t sort_pairs_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                    (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0, 64);
    return bytes;
}

hipError_t sort_pairs(hipStream_t st, void *temp, size_t temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                      const uint32_t *vals_in, uint32_t *vals_out, uint64_t n, unsigned end_bit) {
    drop_stale_hip_error("sort_pairs");
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, end_bit, st);
}

}  // namespace pgr
