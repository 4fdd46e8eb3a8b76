// shard.hip -- key-range partition of shimmer-pair records ("key-range partitioned" index).
//
// The frag_map of the reference is ONE hash map filled by a serial insert (pgr-db/src/seq_db.rs:605-612).  With one
// process per GPU every rank computes the pair records of its own contigs (seq_db.rs:460-467); instead of gathering
// all records everywhere and sorting N times the same set, the key space is cut into N ranges [s_{r-1}, s_r) of the
// first hash h0, every record travels to the rank that owns its range (one variable all-to-all, csrc/exchange.hip) and
// each rank sorts only its range: the CSR of rank r followed by the CSR of rank r+1 ... IS the single-process CSR, and
// the sort work per rank stays what one GPU does for its own contigs however many ranks there are.
//
// This file holds the collective-free pieces (usable with any transport): sampling, splitters, the stable partition,
// and an order-independent content checksum to prove that what arrived is what was sent.
#include <algorithm>
#include <vector>

#include "pgr_device.h"
#include "pgr_index.h"

using namespace pgr;

namespace {

__global__ void sample_h0_kernel(const pgr_frag_rec *__restrict__ recs, uint64_t n, uint32_t n_samples,
                                 uint64_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples) return;
    // evenly spaced in the append order: the records of a contig come in position order, unrelated to their hash
    const uint64_t q = n / n_samples, r = n % n_samples;  // floor(i * n / n_samples) without a 128-bit product
    out[i] = recs[(uint64_t)i * q + ((uint64_t)i * r) / n_samples].h0;
}

// destination of a record = number of splitters <= h0 (records with one h0 -- hence every record of a key -- share it)
__global__ void dest_key_kernel(const pgr_frag_rec *__restrict__ recs, uint64_t n, const uint64_t *__restrict__ splitters,
                                uint32_t n_split, uint64_t *__restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t h = recs[i].h0;
    uint32_t lo = 0, hi = n_split;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (splitters[mid] <= h) lo = mid + 1;
        else hi = mid;
    }
    keys[i] = lo;
}

// off[d] = first position of destination d in the sorted destination keys, off[n_dest] = n
__global__ void dest_offsets_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint32_t n_dest, uint64_t *__restrict__ off) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_dest) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < d) lo = mid + 1;
        else hi = mid;
    }
    off[d] = lo;
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// order-independent 128-bit content checksum of a record set: two sums of independent mixes of every field
__global__ __launch_bounds__(256) void recs_checksum_kernel(const pgr_frag_rec *__restrict__ recs, uint64_t n,
                                                            unsigned long long *__restrict__ out) {
    uint64_t a = 0, b = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const pgr_frag_rec r = recs[i];
        const uint64_t p = ((uint64_t)r.frg_id << 32) | r.sid, q = ((uint64_t)r.bgn << 32) | r.end;
        const uint64_t m = mix64(r.h0) ^ mix64(r.h1 + 0xD1B54A32D192ED03ull) ^ mix64(p ^ 0xA0761D6478BD642Full) ^
                           mix64(q + ((uint64_t)r.orient << 62));
        a += mix64(m);
        b += mix64(m ^ 0xE7037ED1A0B428DBull) * 0x8EBC6AF09C88C6E3ull;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += shfl_xor64(a, d);
        b += shfl_xor64(b, d);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out, (unsigned long long)a);
        atomicAdd(out + 1, (unsigned long long)b);
    }
}

}  // namespace

extern "C" int pgr_shard_sample_keys(pgr_ctx *ctx, const pgr_frag_rec *d_recs, uint64_t n, uint32_t n_samples, uint64_t *out,
                                     uint32_t *n_out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!out || !n_out || (n && !d_recs)) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    PGR_ENTER(ctx);
    const uint32_t ns = (uint32_t)std::min<uint64_t>(n, n_samples);
    *n_out = ns;
    if (ns == 0) return PGR_OK;
    Tmp d(ctx);
    int rc = d.alloc((size_t)ns * 8);
    if (rc) return rc;
    hipLaunchKernelGGL(sample_h0_kernel, grid_for(ns), dim3(256), 0, ctx->stream, d_recs, n, ns, d.as<uint64_t>());
    PGR_HIP(ctx, hipMemcpyAsync(out, d.p, (size_t)ns * 8, hipMemcpyDeviceToHost, ctx->stream));
    PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PGR_HIP(ctx, hipGetLastError());
    return PGR_OK;
}

// host only: world - 1 splitters at the quantiles of the pooled samples (every rank computes the same from the same pool)
extern "C" int pgr_shard_splitters(const uint64_t *samples, uint64_t n, int world, uint64_t *splitters) {
    if (world < 1 || (world > 1 && !splitters) || (n && !samples)) return PGR_ERR_INVALID_ARG;
    std::vector<uint64_t> s(samples, samples + n);
    std::sort(s.begin(), s.end());
    for (int j = 1; j < world; ++j)
        splitters[j - 1] = n ? s[(size_t)(((unsigned __int128)j * n) / (uint64_t)world)] : (~0ull / (uint64_t)world) * (uint64_t)j;
    return PGR_OK;
}

extern "C" int pgr_shard_partition(pgr_ctx *ctx, const pgr_frag_rec *d_recs, uint64_t n, const uint64_t *splitters, int world,
                                   pgr_frag_rec *d_out, uint64_t *counts) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (world < 1 || !counts || (world > 1 && !splitters) || (n && (!d_recs || !d_out)))
        return ctx->fail(PGR_ERR_INVALID_ARG, "bad partition arguments");
    if (n >= (1ull << 32)) return ctx->fail(PGR_ERR_INVALID_ARG, "more than 2^32-1 records in one partition call");
    PGR_ENTER(ctx);
    hipStream_t st = ctx->stream;
    for (int d = 0; d < world; ++d) counts[d] = 0;
    if (n == 0) return PGR_OK;
    if (world == 1) {
        PGR_HIP(ctx, hipMemcpyAsync(d_out, d_recs, n * sizeof(pgr_frag_rec), hipMemcpyDeviceToDevice, st));
        PGR_HIP(ctx, hipStreamSynchronize(st));
        counts[0] = n;
        return PGR_OK;
    }
    int rc;
    Tmp d_split(ctx), keys_a(ctx), keys_b(ctx), idx_a(ctx), idx_b(ctx), d_off(ctx);
    if ((rc = d_split.alloc((size_t)(world - 1) * 8)) || (rc = keys_a.alloc(n * 8)) || (rc = keys_b.alloc(n * 8)) ||
        (rc = idx_a.alloc(n * 4)) || (rc = idx_b.alloc(n * 4)) || (rc = d_off.alloc((size_t)(world + 1) * 8)))
        return rc;
    // (pageable source of a small copy: staged by the runtime before the call returns; synchronized below anyway)
    PGR_HIP(ctx, hipMemcpyAsync(d_split.p, splitters, (size_t)(world - 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(dest_key_kernel, grid_for(n), dim3(256), 0, st, d_recs, n, d_split.as<uint64_t>(), (uint32_t)(world - 1),
                       keys_a.as<uint64_t>());
    launch_iota(st, idx_a.as<uint32_t>(), n);
    // stable: inside a destination the records keep their append order, i.e. (sid, frg_id) order of this rank's contigs
    const size_t tb = sort_pairs_temp_bytes(n);
    if ((rc = ctx->ws_scan_tmp.ensure(ctx, tb))) return rc;
    PGR_HIP(ctx, sort_pairs(st, ctx->ws_scan_tmp.p, tb, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), idx_a.as<uint32_t>(),
                            idx_b.as<uint32_t>(), n, bits_for((uint64_t)world)));
    launch_gather_recs(st, d_recs, idx_b.as<uint32_t>(), d_out, n);
    hipLaunchKernelGGL(dest_offsets_kernel, grid_for((uint64_t)world + 1), dim3(256), 0, st, keys_b.as<uint64_t>(), n, (uint32_t)world,
                       d_off.as<uint64_t>());
    std::vector<uint64_t> off((size_t)world + 1);
    PGR_HIP(ctx, hipMemcpyAsync(off.data(), d_off.p, off.size() * 8, hipMemcpyDeviceToHost, st));
    PGR_HIP(ctx, hipStreamSynchronize(st));
    PGR_HIP(ctx, hipGetLastError());
    for (int d = 0; d < world; ++d) counts[d] = off[(size_t)d + 1] - off[(size_t)d];
    return PGR_OK;
}

extern "C" int pgr_records_checksum(pgr_ctx *ctx, const pgr_frag_rec *d_recs, uint64_t n, uint64_t out[2]) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!out || (n && !d_recs)) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    out[0] = out[1] = 0;
    if (n == 0) return PGR_OK;
    PGR_ENTER(ctx);
    Tmp d(ctx);
    int rc = d.alloc(16);
    if (rc) return rc;
    PGR_HIP(ctx, hipMemsetAsync(d.p, 0, 16, ctx->stream));
    hipLaunchKernelGGL(recs_checksum_kernel, dim3((uint32_t)std::min<uint64_t>(2048, (n + 255) / 256)), dim3(256), 0, ctx->stream,
                       d_recs, n, d.as<unsigned long long>());
    PGR_HIP(ctx, hipMemcpyAsync(out, d.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PGR_HIP(ctx, hipGetLastError());
    return PGR_OK;
}

// the records of an index: the appended ones before pgr_index_finalize, the sorted ones after
extern "C" int pgr_index_records_checksum(pgr_ctx *ctx, const pgr_index *ix, uint64_t out[2]) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix) return ctx->fail(PGR_ERR_INVALID_ARG, "null index");
    return ix->finalized ? pgr_records_checksum(ctx, ix->recs, ix->n, out) : pgr_records_checksum(ctx, ix->raw, ix->n_raw, out);
}

extern "C" const pgr_frag_rec *pgr_index_device_records(const pgr_index *ix) {
    return ix ? (ix->finalized ? ix->recs : ix->raw) : nullptr;
}

// smallest and largest h0 of a finalized index (a shard's key range); 0 / 0 when empty
extern "C" int pgr_index_key_range(pgr_ctx *ctx, const pgr_index *ix, uint64_t *h0_min, uint64_t *h0_max) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || !h0_min || !h0_max) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (!ix->finalized) return ctx->fail(PGR_ERR_STATE, "index not finalized");
    *h0_min = *h0_max = 0;
    if (ix->n == 0) return PGR_OK;
    PGR_ENTER(ctx);
    PGR_HIP(ctx, hipMemcpyAsync(h0_min, &ix->recs[0].h0, 8, hipMemcpyDeviceToHost, ctx->stream));
    PGR_HIP(ctx, hipMemcpyAsync(h0_max, &ix->recs[ix->n - 1].h0, 8, hipMemcpyDeviceToHost, ctx->stream));
    PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PGR_OK;
}
