// What does fresh device memory cost?  hipMalloc / first touch / hipFree of blocks of 0.25 .. 16 GiB, and the same again
// (a process that has released a block gets it back faster?).   hipcc --offload-arch=gfx950 -O2 -o malloc_probe malloc_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    (void)hipFree(nullptr);
    for (int rep = 0; rep < 2; ++rep)
        for (double gib : {0.25, 1.0, 4.0, 16.0}) {
            const size_t bytes = (size_t)(gib * (1ull << 30));
            void *p = nullptr;
            double t0 = now();
            if (hipMalloc(&p, bytes) != hipSuccess) return 1;
            double t1 = now();
            (void)hipMemsetAsync(p, 1, bytes, nullptr);
            (void)hipDeviceSynchronize();
            double t2 = now();
            (void)hipMemsetAsync(p, 2, bytes, nullptr);
            (void)hipDeviceSynchronize();
            double t3 = now();
            (void)hipFree(p);
            double t4 = now();
            printf("%5.2f GiB (pass %d): hipMalloc %7.2f ms (%6.1f GB/s), first memset %6.2f ms, second %6.2f ms, hipFree %6.2f ms\n", gib, rep,
                   (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
        }
    return 0;
}
