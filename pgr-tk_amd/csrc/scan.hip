// scan.hip -- device-wide scan / sort primitives from rocPRIM (AMD's native primitives library).
// These are plain library calls around the hand-written kernels (counts -> offsets, records -> CSR);
// none of them is on the hot path's critical time.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "pgr_internal.h"

namespace pgr {

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
    auto it = rocprim::make_transform_iterator(in, U32toU64());
    return rocprim::exclusive_scan(temp, temp_bytes, it, out, (uint64_t)0, (size_t)n_plus_1,
                                   rocprim::plus<uint64_t>(), st);
}

// inclusive max-scan of u64 values in place
size_t scan_max_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (size_t)n, rocprim::maximum<uint64_t>());
    return bytes;
}
hipError_t scan_max_inplace(hipStream_t st, void *temp, size_t temp_bytes, uint64_t *v, uint32_t n) {
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
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, end_bit, st);
}

}  // namespace pgr
