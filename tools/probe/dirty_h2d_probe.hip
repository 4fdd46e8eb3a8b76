// Is the H2D DMA slower when the pinned window has just been written by CPU threads (dirty lines in the 512 MB of L3) than when
// it sits clean in DRAM?  And do non-temporal stores fix it?   usage: dirty_h2d_probe
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par(unsigned nthr, size_t len, const std::function<void(size_t, size_t)> &f) {
    const size_t piece = 1u << 20;
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t o; (o = next.fetch_add(piece)) < len;) f(o, std::min(piece, len - o));
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthr; ++t) th.emplace_back(work);
    for (auto &t : th) t.join();
}
__attribute__((target("avx2"))) static void nt_copy(uint8_t *d, const uint8_t *s, size_t n) {
    for (size_t i = 0; i < n; i += 32) _mm256_stream_si256((__m256i *)(d + i), _mm256_loadu_si256((const __m256i *)(s + i)));
    _mm_sfence();
}

int main() {
    const size_t win = 24u << 20, total = 40 * win;
    uint8_t *src = (uint8_t *)malloc(total);
    memset(src, 'A', total);
    uint8_t *pin, *dev;
    hipHostMalloc((void **)&pin, 2 * win, hipHostMallocDefault);
    hipMalloc((void **)&dev, 2 * win);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    memset(pin, 1, 2 * win);
    for (int mode = 0; mode < 3; ++mode) {
        double t_copy = 0, t_h2d = 0;
        for (size_t o = 0; o < total; o += win) {
            double t0 = now();
            if (mode == 1) par(12, win, [&](size_t a, size_t n) { memcpy(pin + a, src + o + a, n); });
            if (mode == 2) par(12, win, [&](size_t a, size_t n) { nt_copy(pin + a, src + o + a, n); });
            double t1 = now();
            hipMemcpyAsync(dev, pin, win, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
            t_copy += t1 - t0;
            t_h2d += now() - t1;
        }
        printf("%-44s H2D %.1f GB/s   (CPU fill %.1f GB/s)\n",
               mode == 0 ? "window untouched (clean in DRAM):" : mode == 1 ? "window just written with memcpy, 12 threads:" : "window just written with NT stores, 12 threads:",
               total / t_h2d / 1e9, mode ? total / t_copy / 1e9 : 0.0);
    }
    // the same with the fill of window B overlapping the DMA of window A (what the pipeline does)
    for (int mode = 1; mode < 3; ++mode) {
        double t0 = now();
        int slot = 0;
        hipEvent_t ev[2];
        hipEventCreate(&ev[0]);
        hipEventCreate(&ev[1]);
        bool used[2] = {false, false};
        for (size_t o = 0; o < total; o += win, slot ^= 1) {
            if (used[slot]) hipEventSynchronize(ev[slot]);
            uint8_t *w = pin + slot * win;
            if (mode == 1) par(12, win, [&](size_t a, size_t n) { memcpy(w + a, src + o + a, n); });
            else par(12, win, [&](size_t a, size_t n) { nt_copy(w + a, src + o + a, n); });
            hipMemcpyAsync(dev + slot * win, w, win, hipMemcpyHostToDevice, st);
            hipEventRecord(ev[slot], st);
            used[slot] = true;
        }
        hipStreamSynchronize(st);
        printf("pipelined over two windows, %s: %.1f GB/s\n", mode == 1 ? "memcpy" : "NT stores", total / (now() - t0) / 1e9);
    }
    return 0;
}
