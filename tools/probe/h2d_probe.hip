// What bounds the host-buffer entry points (pageable ASCII in -> pinned window -> H2D)?  Measures, on the GPU box:
//   (a) hipMemcpyAsync H2D from a pinned window,  (b) host threads copying pageable memory into the pinned window,
//   (c) both pipelined over two windows (what batch_stage does).     usage: h2d_probe [total MiB] [window MiB]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(uint8_t *dst, const uint8_t *src, size_t len, unsigned nthr) {
    const size_t piece = 1u << 20;
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t o; (o = next.fetch_add(piece)) < len;) memcpy(dst + o, src + o, std::min(piece, len - o));
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nthr; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

int main(int argc, char **argv) {
    const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 512) << 20, win = (size_t)(argc > 2 ? atoi(argv[2]) : 32) << 20;
    uint8_t *src = (uint8_t *)malloc(total);
    memset(src, 'A', total);
    uint8_t *pin, *dev;
    hipHostMalloc((void **)&pin, 2 * win, hipHostMallocDefault);
    hipMalloc((void **)&dev, 2 * win);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t ev[2];
    hipEventCreate(&ev[0]);
    hipEventCreate(&ev[1]);
    memset(pin, 1, 2 * win);
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (size_t o = 0; o < total; o += win) hipMemcpyAsync(dev, pin, win, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
        printf("(a) H2D from pinned, %zu MiB windows: %.1f GB/s\n", win >> 20, total / (now() - t0) / 1e9);
    }
    for (unsigned nthr : {1u, 2u, 4u, 8u, 12u, 16u}) {
        double t0 = now();
        for (size_t o = 0; o < total; o += win) par_copy(pin, src + o, win, nthr);
        printf("(b) pageable -> pinned with %2u threads: %.1f GB/s\n", nthr, total / (now() - t0) / 1e9);
    }
    for (unsigned nthr : {4u, 8u, 16u}) {
        double t0 = now();
        int slot = 0;
        bool used[2] = {false, false};
        for (size_t o = 0; o < total; o += win, slot ^= 1) {
            if (used[slot]) hipEventSynchronize(ev[slot]);
            par_copy(pin + slot * win, src + o, win, nthr);
            hipMemcpyAsync(dev + slot * win, pin + slot * win, win, hipMemcpyHostToDevice, st);
            hipEventRecord(ev[slot], st);
            used[slot] = true;
        }
        hipStreamSynchronize(st);
        printf("(c) pipelined, %2u threads: %.1f GB/s\n", nthr, total / (now() - t0) / 1e9);
    }
    {   // (d) hipHostRegister of the user's buffer + direct H2D
        double t0 = now();
        hipError_t e = hipHostRegister(src, total, hipHostRegisterDefault);
        double t1 = now();
        if (e == hipSuccess) {
            uint8_t *big;
            hipMalloc((void **)&big, total);
            hipMemcpyAsync(big, src, total, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
            double t2 = now();
            hipHostUnregister(src);
            double t3 = now();
            printf("(d) hipHostRegister %.2f ms, direct H2D %.1f GB/s, unregister %.2f ms -> %.1f GB/s overall\n", (t1 - t0) * 1e3,
                   total / (t2 - t1) / 1e9, (t3 - t2) * 1e3, total / (t3 - t0) / 1e9);
        } else {
            printf("(d) hipHostRegister failed: %s\n", hipGetErrorString(e));
        }
    }
    {   // (e) plain hipMemcpy from pageable memory (the runtime's own staging)
        uint8_t *big;
        hipMalloc((void **)&big, total);
        double t0 = now();
        hipMemcpy(big, src, total, hipMemcpyHostToDevice);
        printf("(e) hipMemcpy from pageable memory: %.1f GB/s\n", total / (now() - t0) / 1e9);
    }
    return 0;
}
