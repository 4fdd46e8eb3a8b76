// Can a kernel write a query batch's result straight into the host's pinned block as fast as the copy engine downloads it?
// N single-wavefront workgroups each write `per` bytes (8 B per lane, consecutive) to pinned host memory (hipHostMalloc, mapped)
// after `spin` rounds of arithmetic; against hipMemcpyAsync D2H of the same bytes.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hw tools/probe/host_write_probe.hip && /tmp/hw
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

__global__ __launch_bounds__(64) void writer(uint64_t *dst, uint32_t words_per_wg, int spin, int nontemporal) {
    uint64_t v = blockIdx.x * 64 + threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 6364136223846793005ull + 1442695040888963407ull;
    uint64_t *o = dst + (size_t)blockIdx.x * words_per_wg;
    for (uint32_t i = threadIdx.x; i < words_per_wg; i += 64) {
        if (nontemporal) __builtin_nontemporal_store(v + i, o + i);
        else o[i] = v + i;
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const uint32_t n = 10000;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (uint32_t per : {768u, 3072u, 12288u}) {
        const uint32_t wpw = per / 8;
        const size_t bytes = (size_t)n * per;
        uint64_t *h, *d;
        CK(hipHostMalloc((void **)&h, bytes, hipHostMallocDefault));
        CK(hipMalloc((void **)&d, bytes));
        for (int spin : {0, 20000}) {
            for (int nt = 0; nt < 2; ++nt) {
                double best_h = 1e9, best_d = 1e9, best_c = 1e9;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipStreamSynchronize(st));
                    double t0 = now();
                    hipLaunchKernelGGL(writer, dim3(n), dim3(64), 0, st, h, wpw, spin, nt);
                    CK(hipStreamSynchronize(st));
                    double t1 = now();
                    hipLaunchKernelGGL(writer, dim3(n), dim3(64), 0, st, d, wpw, spin, nt);
                    CK(hipStreamSynchronize(st));
                    double t2 = now();
                    hipLaunchKernelGGL(writer, dim3(n), dim3(64), 0, st, d, wpw, spin, nt);
                    CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st));
                    CK(hipStreamSynchronize(st));
                    double t3 = now();
                    if (t1 - t0 < best_h) best_h = t1 - t0;
                    if (t2 - t1 < best_d) best_d = t2 - t1;
                    if (t3 - t2 < best_c) best_c = t3 - t2;
                }
                printf("%5u B per workgroup (%5.1f MB), spin %5d, %s stores: kernel -> host %7.1f us (%5.1f GB/s); kernel -> device %6.1f us; "
                       "kernel -> device + D2H copy %7.1f us\n",
                       per, bytes / 1e6, spin, nt ? "nontemporal" : "plain      ", best_h * 1e6, bytes / best_h / 1e9, best_d * 1e6, best_c * 1e6);
            }
        }
        CK(hipHostFree(h));
        CK(hipFree(d));
    }
    return 0;
}
