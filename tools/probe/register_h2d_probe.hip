// H2D rate out of (a) hipHostMalloc'd memory, (b) malloc'd memory pinned with hipHostRegister (what pgr_host_register does),
// (c) plain pageable memory; 256 MiB in one copy and in windows of 20 MiB; plus what hipHostRegister / Unregister cost.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/register_h2d_probe tools/probe/register_h2d_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void run(const char *what, const void *src, void *dst, size_t bytes, hipStream_t st) {
    for (size_t win : {bytes, (size_t)20 << 20}) {
        double best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipStreamSynchronize(st);
            const double t0 = now();
            for (size_t o = 0; o < bytes; o += win)
                (void)hipMemcpyAsync((char *)dst + o, (const char *)src + o, bytes - o < win ? bytes - o : win, hipMemcpyHostToDevice, st);
            const double t1 = now();
            (void)hipStreamSynchronize(st);
            const double t2 = now();
            if (t2 - t0 < best) best = t2 - t0;
            if (rep == 3) printf("%-28s windows of %4zu MiB: %6.2f GB/s (calls returned after %.2f ms of %.2f ms)\n", what, win >> 20, bytes / best / 1e9, (t1 - t0) * 1e3, (t2 - t0) * 1e3);
        }
    }
}

int main() {
    const size_t bytes = 256u << 20;
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    void *dst = nullptr, *pin = nullptr;
    (void)hipMalloc(&dst, bytes);
    (void)hipHostMalloc(&pin, bytes, hipHostMallocDefault);
    memset(pin, 1, bytes);
    void *reg = aligned_alloc(4096, bytes), *page = aligned_alloc(4096, bytes);
    memset(reg, 2, bytes);
    memset(page, 3, bytes);
    double t0 = now();
    const hipError_t e = hipHostRegister(reg, bytes, hipHostRegisterDefault);
    double t1 = now();
    printf("hipHostRegister of 256 MiB: %.2f ms (%s)\n", (t1 - t0) * 1e3, hipGetErrorString(e));
    run("hipHostMalloc", pin, dst, bytes, st);
    run("malloc + hipHostRegister", reg, dst, bytes, st);
    run("pageable", page, dst, bytes, st);
    t0 = now();
    (void)hipHostUnregister(reg);
    printf("hipHostUnregister: %.2f ms\n", (now() - t0) * 1e3);
    return 0;
}
