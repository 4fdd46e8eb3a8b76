// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths libpgrhip's dominant kernel uses
// (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other
// access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel moves a KNOWN number of bytes through buffers far larger than the 256 MiB Infinity Cache:
//   read_u4        16 B per lane, fully coalesced                                   (the guide's case)
//   read_u2         8 B per lane, fully coalesced
//   read_tile_like  the staging pattern of level1_tile_kernel: a workgroup of 256 lanes, lanes 0..135 load one uint2 each from
//                   136 consecutive words; consecutive workgroups advance by 122 words (the tile core): 14 words are read twice
//   write_u4       16 B per lane, coalesced
//   write_12B      12-byte records (L1Rec) written in runs of ~96 records at the start of 256-record slots (the tile kernel's output)
// usage (on the GPU box):  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o c -- ./fetch_calib     (and again with WRITE_SIZE)
// tools/summarize_calib.py turns the two CSVs into profiles/r05_calib/calibration.json.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/fetch_calib tools/probe/fetch_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(256) void read_u4(const uint4 *__restrict__ p, size_t n, uint32_t *__restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;  // (never true for the fill pattern: no write traffic)
}
__global__ __launch_bounds__(256) void read_u2(const uint2 *__restrict__ p, size_t n, uint32_t *__restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint2 v = p[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void read_tile_like(const uint2 *__restrict__ p, size_t n_words, uint32_t *__restrict__ out) {
    const size_t w = (size_t)blockIdx.x * 122 + threadIdx.x;
    uint32_t acc = 0;
    if (threadIdx.x < 136 && w < n_words) {
        const uint2 v = p[w];
        acc = v.x ^ v.y;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void write_u4(uint4 *__restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
struct Rec12 {
    uint32_t a, b, c;
};
__global__ __launch_bounds__(256) void write_12B(Rec12 *__restrict__ p, size_t n_slots) {  // one wavefront per slot of 256 records, 96 written
    const size_t slot = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= n_slots) return;
    const uint32_t lane = threadIdx.x & 63;
    Rec12 *o = p + slot * 256;
    for (uint32_t i = lane; i < 96; i += 64) o[i] = Rec12{(uint32_t)slot, i, 7u};
}

int main() {
    const size_t bytes = 4ull << 30;
    void *buf = nullptr;
    uint32_t *out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **)&out, 64u << 20) != hipSuccess) return 1;
    (void)hipMemset(buf, 0x5a, bytes);
    (void)hipDeviceSynchronize();
    const size_t n_tiles = (bytes / 8 - 136) / 122;
    const size_t n_slots = bytes / (256 * sizeof(Rec12));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_u4, dim3(16384), dim3(256), 0, nullptr, (const uint4 *)buf, bytes / 16, out);
        hipLaunchKernelGGL(read_u2, dim3(16384), dim3(256), 0, nullptr, (const uint2 *)buf, bytes / 8, out);
        hipLaunchKernelGGL(read_tile_like, dim3((uint32_t)n_tiles), dim3(256), 0, nullptr, (const uint2 *)buf, bytes / 8, out);
        hipLaunchKernelGGL(write_u4, dim3(16384), dim3(256), 0, nullptr, (uint4 *)buf, bytes / 16);
        hipLaunchKernelGGL(write_12B, dim3((uint32_t)((n_slots + 3) / 4)), dim3(256), 0, nullptr, (Rec12 *)buf, n_slots);
        (void)hipDeviceSynchronize();
    }
    // the byte counts the counters are compared with (requested = what the lanes asked for; distinct = without the words read twice)
    printf("{\"read_u4\": {\"requested\": %zu}, \"read_u2\": {\"requested\": %zu}, \"read_tile_like\": {\"requested\": %zu, \"distinct\": %zu}, "
           "\"write_u4\": {\"requested\": %zu}, \"write_12B\": {\"requested\": %zu}}\n",
           bytes, bytes, n_tiles * 136 * 8, n_tiles * 122 * 8 + 14 * 8, bytes, n_slots * 96 * sizeof(Rec12));
    return 0;
}
