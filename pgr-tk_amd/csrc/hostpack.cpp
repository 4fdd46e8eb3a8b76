// hostpack.cpp -- host side of the input boundary: ASCII -> 2-bit planes + validity plane on the CPU, a small
// persistent thread pool for the staging copies.  Plain C++ (g++), no HIP: nothing here touches the GPU.
//
// Byte semantics = the reference's base2bits table (pgr-db/src/shmmrutils.rs:426-436): A/a/0 -> 0, C/c/1 -> 1,
// G/g/2 -> 2, T/t/3 -> 3, everything else "not a base" (valid bit 0, plane bits 0).  The packed layout is the one
// pgr_batch keeps in HBM (csrc/pgr_internal.h: BatchDev): word j of a contig holds bases 32j..32j+31, base i at bit
// 31 - (i % 32); planes[w] = low plane | high plane << 32.  Packing on the host puts 0.375 B per base on the PCIe
// link instead of 1 B (SURVEY.md section 7 step 3, K1 "or accept pre-packed").
#include <immintrin.h>
#include <sched.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/pgr_hip.h"
#include "pgr_host.h"

namespace pgr {

// CPUs this process may really use: scheduler affinity capped by the cgroup quota (a container that sees 256 CPUs with
// a 16-CPU quota gets 16 threads' worth of work done)
unsigned host_cpus() {
    static const unsigned n = [] {
        unsigned c = std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) c = (unsigned)CPU_COUNT(&set);
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64];
            double period = 0;
            if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                const double quota = atof(q);
                if (quota > 0) c = std::min<unsigned>(c, (unsigned)std::max(1.0, quota / period + 0.5));
            }
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            long quota = -1, period = 0;
            if (fscanf(g, "%ld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(h, "%ld", &period) != 1) period = 0;
                fclose(h);
            }
            if (quota > 0 && period > 0) c = std::min<unsigned>(c, (unsigned)std::max<long>(1, (quota + period / 2) / period));
        }
        if (const char *e = getenv("PGR_HOST_THREADS")) c = (unsigned)std::max(1, atoi(e));
        return std::max(1u, c);
    }();
    return n;
}

// Host memory for a big result the caller releases with pgr_free (= free): 2 MiB aligned and advised to use transparent
// huge pages, so that the first touch of a 50 MB shimmer list costs a few dozen page faults instead of twelve thousand.
void *host_result_alloc(size_t bytes) {
    if (bytes < (8u << 20)) return malloc(bytes ? bytes : 1);
    void *p = nullptr;
    const size_t rounded = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    if (posix_memalign(&p, 2u << 20, rounded) != 0) return malloc(bytes);
    (void)madvise(p, rounded, MADV_HUGEPAGE);
    return p;
}

// ------------------------------------------------------------------ thread pool
struct HostPool::Impl {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false;
    void worker() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
            }
            f();
        }
    }
};

HostPool::HostPool(unsigned n_workers) : impl(new Impl()) {
    for (unsigned i = 0; i < n_workers; ++i) impl->th.emplace_back([this] { impl->worker(); });
}

HostPool::~HostPool() {
    {
        std::lock_guard<std::mutex> lk(impl->mu);
        impl->stop = true;
    }
    impl->cv.notify_all();
    for (auto &t : impl->th) t.join();
    delete impl;
}

unsigned HostPool::workers() const { return (unsigned)impl->th.size(); }

// (Measured and rejected, tools/host_threads_sweep.py: binding the workers to the CPUs of the GPU's NUMA node.  The pinned
// windows live there, but the caller's ASCII does not have to -- processes whose input sat on the other socket packed at
// 78-92 Gbp/s bound against 121-137 unbound; packed input, which the workers only copy, did not care: 164-170 either way.)

HostPool &HostPool::instance() {
    // one pool per process, created at first use; the calling thread always takes part in its own loops, so the pool
    // holds host_cpus() - 1 workers (at most 31)
    static HostPool pool(std::min(31u, host_cpus() - 1));
    return pool;
}

void HostPool::parallel_for(size_t n, const std::function<void(size_t)> &fn, unsigned max_par) {
    if (n == 0) return;
    const unsigned helpers = (unsigned)std::min<size_t>({(size_t)workers(), n - 1, (size_t)(max_par ? max_par - 1 : ~0u)});
    if (helpers == 0) {
        for (size_t i = 0; i < n; ++i) fn(i);
        return;
    }
    struct Group {
        std::atomic<size_t> next{0}, done{0};
        size_t n = 0;
        std::function<void(size_t)> fn;
        std::mutex m;
        std::condition_variable c;
    };
    auto g = std::make_shared<Group>();  // shared: a helper that starts after the caller has finished finds nothing to do
    g->n = n;
    g->fn = fn;
    // items are taken `grain` at a time: the counter's cache line travels between the cores (two sockets on the GPU boxes,
    // ~150 ns per hop), and 10 000 small items -- a query batch: one item per 10 kbp query -- spent 1.2 ms on it
    const size_t grain = std::max<size_t>(1, n / ((size_t)(helpers + 1) * 16));
    auto body = [g, grain] {
        size_t mine = 0;
        for (size_t i0; (i0 = g->next.fetch_add(grain)) < g->n;) {
            const size_t i1 = std::min(g->n, i0 + grain);
            for (size_t i = i0; i < i1; ++i) g->fn(i);
            mine += i1 - i0;
        }
        if (mine && g->done.fetch_add(mine) + mine == g->n) {
            std::lock_guard<std::mutex> lk(g->m);
            g->c.notify_all();
        }
    };
    {
        std::lock_guard<std::mutex> lk(impl->mu);
        for (unsigned i = 0; i < helpers; ++i) impl->q.emplace_back(body);
    }
    impl->cv.notify_all();
    body();
    std::unique_lock<std::mutex> lk(g->m);
    g->c.wait(lk, [&] { return g->done.load() == g->n; });
}

// ------------------------------------------------------------------ the packer
namespace {

struct Lut {
    uint8_t code[256];  // 0..3, 4 = not a base
    Lut() {
        for (int i = 0; i < 256; ++i) code[i] = 4;
        code[0] = 0; code[1] = 1; code[2] = 2; code[3] = 3;
        code['A'] = code['a'] = 0;
        code['C'] = code['c'] = 1;
        code['G'] = code['g'] = 2;
        code['T'] = code['t'] = 3;
    }
};
const Lut g_lut;
struct Rev64 {  // vpermb index: byte j of each 32-byte half -> byte 31 - j of that half
    alignas(64) uint8_t idx[64];
    Rev64() {
        for (int j = 0; j < 64; ++j) idx[j] = (uint8_t)((j & 32) | (31 - (j & 31)));
    }
};
const Rev64 g_rev64;

inline void pack32_scalar(const uint8_t *s, uint32_t nb, uint32_t &lo, uint32_t &hi, uint32_t &v) {
    lo = hi = v = 0;
    for (uint32_t i = 0; i < nb; ++i) {
        const uint32_t c = g_lut.code[s[i]];
        if (c < 4) {
            const uint32_t bit = 31u - i;
            lo |= (c & 1u) << bit;
            hi |= (c >> 1) << bit;
            v |= 1u << bit;
        }
    }
}

uint64_t pack_words_scalar(const uint8_t *seq, uint64_t len, uint64_t w0, uint64_t w1, uint64_t *planes, uint32_t *valid) {
    uint64_t bad = 0;
    for (uint64_t w = w0; w < w1; ++w) {
        const uint64_t first = w * 32;
        const uint32_t nb = first >= len ? 0u : (uint32_t)std::min<uint64_t>(32, len - first);
        uint32_t lo, hi, v;
        pack32_scalar(seq + first, nb, lo, hi, v);
        planes[w - w0] = (uint64_t)lo | ((uint64_t)hi << 32);
        valid[w - w0] = v;
        bad += nb - (uint32_t)__builtin_popcount(v);
    }
    return bad;
}

// Stores of the packers.  NT = true: non-temporal (movnti) -- the staging path writes pinned windows that the CPU never reads
// again and the DMA engine fetches a moment later: regular stores allocate the lines in the caches (a read for ownership per
// line, then a write-back racing the DMA); measured with the fill of one window overlapping the DMA of the other
// (tools/probe/dirty_h2d_probe.hip): 34-51 GB/s with regular stores, 53-54 GB/s non-temporal.
template <bool NT>
inline __attribute__((always_inline)) void put_words(uint64_t *planes, uint32_t *valid, uint64_t i, uint64_t pl, uint32_t v) {
    if (NT) {
        _mm_stream_si64((long long *)(planes + i), (long long)pl);
        _mm_stream_si32((int *)(valid + i), (int)v);
    } else {
        planes[i] = pl;
        valid[i] = v;
    }
}

// 32 bases per step: the 32 bytes are reversed once (base i -> byte 31 - i), the per-byte plane bits moved to the
// byte's MSB, and vpmovmskb delivers one plane word
template <bool NT>
__attribute__((target("avx2,popcnt"))) uint64_t pack_words_avx2(const uint8_t *seq, uint64_t len, uint64_t w0, uint64_t w1,
                                                                 uint64_t *planes, uint32_t *valid) {
    const __m256i c20 = _mm256_set1_epi8(0x20), cA = _mm256_set1_epi8('a'), cC = _mm256_set1_epi8('c'),
                  cG = _mm256_set1_epi8('g'), cT = _mm256_set1_epi8('t'), cFC = _mm256_set1_epi8((char)0xFC),
                  zero = _mm256_setzero_si256();
    const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8,
                                         7, 6, 5, 4, 3, 2, 1, 0);
    uint64_t bad = 0;
    uint64_t w = w0;
    const uint64_t full_end = std::min<uint64_t>(w1, len / 32);  // words with 32 real bytes behind them
    for (; w < full_end; ++w) {
        __m256i v = _mm256_loadu_si256((const __m256i *)(seq + w * 32));
        v = _mm256_shuffle_epi8(v, rev);
        v = _mm256_permute2x128_si256(v, v, 1);
        const __m256i lc = _mm256_or_si256(v, c20);
        const __m256i letter = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(lc, cA), _mm256_cmpeq_epi8(lc, cC)),
                                               _mm256_or_si256(_mm256_cmpeq_epi8(lc, cG), _mm256_cmpeq_epi8(lc, cT)));
        const __m256i small = _mm256_cmpeq_epi8(_mm256_and_si256(v, cFC), zero);  // bytes 0..3 are their own code
        const __m256i ok = _mm256_or_si256(letter, small);
        // letters: code = c ^ (c >> 1) with c = (byte >> 1) & 3  ->  high bit = bit 2, low bit = bit 1 ^ bit 2
        // (16-bit shifts: what crosses from a byte into its neighbour never reaches the neighbour's MSB)
        const __m256i s5 = _mm256_slli_epi16(v, 5), s6 = _mm256_slli_epi16(v, 6), s7 = _mm256_slli_epi16(v, 7);
        const __m256i hi_b = _mm256_blendv_epi8(s5, s6, small);
        const __m256i lo_b = _mm256_blendv_epi8(_mm256_xor_si256(s6, s5), s7, small);
        const uint32_t vm = (uint32_t)_mm256_movemask_epi8(ok);
        const uint32_t lo = (uint32_t)_mm256_movemask_epi8(lo_b) & vm, hi = (uint32_t)_mm256_movemask_epi8(hi_b) & vm;
        put_words<NT>(planes, valid, w - w0, (uint64_t)lo | ((uint64_t)hi << 32), vm);
        bad += 32u - (uint32_t)__builtin_popcount(vm);
    }
    if (w < w1) bad += pack_words_scalar(seq, len, w, w1, planes + (w - w0), valid + (w - w0));
    return bad;
}

// one step = 64 (half-reversed) bytes -> two plane words.  `present`: the bytes of the step that exist (all of them except in a
// contig's last words: a masked load brings zeros for the rest, which must not count as the base with code 0), as bit
// positions AFTER the reversal of the two 32-byte halves
__attribute__((target("avx512f,avx512bw,avx512vbmi,popcnt"), always_inline)) inline void pack64_avx512(const __m512i v, uint64_t present,
                                                                                                   uint64_t &lo, uint64_t &hi,
                                                                                                   uint64_t &ok) {
    const __m512i c20 = _mm512_set1_epi8(0x20), cA = _mm512_set1_epi8('a'), cC = _mm512_set1_epi8('c'), cG = _mm512_set1_epi8('g'),
                  cT = _mm512_set1_epi8('t'), c4 = _mm512_set1_epi8(4), b0 = _mm512_set1_epi8(1), b1 = _mm512_set1_epi8(2),
                  b2 = _mm512_set1_epi8(4);
    const __m512i lc = _mm512_or_si512(v, c20);
    const uint64_t letter = _mm512_cmpeq_epi8_mask(lc, cA) | _mm512_cmpeq_epi8_mask(lc, cC) | _mm512_cmpeq_epi8_mask(lc, cG) |
                            _mm512_cmpeq_epi8_mask(lc, cT);
    const uint64_t small = _mm512_cmplt_epu8_mask(v, c4);  // bytes 0..3 are their own code
    ok = (letter | small) & present;
    const uint64_t t0 = _mm512_test_epi8_mask(v, b0), t1 = _mm512_test_epi8_mask(v, b1), t2 = _mm512_test_epi8_mask(v, b2);
    // letters: high bit = bit 2, low bit = bit 1 ^ bit 2 (see the AVX2 path)
    hi = ((small & t1) | (~small & t2)) & ok;
    lo = ((small & t0) | (~small & (t1 ^ t2))) & ok;
}

// 64 bases per step on CPUs with AVX-512 BW + VBMI (Zen 4/5, Ice Lake and later): one vpermb reverses the bytes of both
// 32-byte halves, every per-byte test lands in a 64-bit mask register, the planes are mask arithmetic
template <bool NT>
__attribute__((target("avx512f,avx512bw,avx512vbmi,popcnt"))) uint64_t pack_words_avx512(const uint8_t *seq, uint64_t len, uint64_t w0,
                                                                                          uint64_t w1, uint64_t *planes,
                                                                                          uint32_t *valid) {
    const __m512i rev = _mm512_load_si512((const void *)g_rev64.idx);
    uint64_t bad = 0;
    uint64_t w = w0;
    const uint64_t full_end = std::min<uint64_t>(w1, len / 32);
    for (; w + 2 <= full_end; w += 2) {
        const __m512i v = _mm512_permutexvar_epi8(rev, _mm512_loadu_si512((const void *)(seq + w * 32)));
        uint64_t lo, hi, ok;
        pack64_avx512(v, ~0ull, lo, hi, ok);
        put_words<NT>(planes, valid, w - w0, (lo & 0xFFFFFFFFull) | (hi << 32), (uint32_t)ok);
        put_words<NT>(planes, valid, w - w0 + 1, (lo >> 32) | (hi & 0xFFFFFFFF00000000ull), (uint32_t)(ok >> 32));
        bad += 64u - (uint32_t)__builtin_popcountll(ok);
    }
    // the last words of the range (an odd word, the contig's partial last word, words behind its end) with masked loads: a
    // batch of reads is all last words -- a 1 kbp read took the AVX2 path for its 31st word and the byte loop for its 32nd
    for (; w < w1; w += 2) {
        const uint64_t first = w * 32;
        const uint32_t nb = first >= len ? 0u : (uint32_t)std::min<uint64_t>(w + 1 < w1 ? 64 : 32, len - first);
        const uint32_t n_lo = std::min(nb, 32u), n_hi = nb - n_lo;
        const __mmask64 ld = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
        const __m512i v = _mm512_permutexvar_epi8(rev, _mm512_maskz_loadu_epi8(ld, (const void *)(seq + first)));
        // byte j of a half sits at bit 31 - j of that half
        const uint64_t p_lo = n_lo ? (0xFFFFFFFFull << (32 - n_lo)) & 0xFFFFFFFFull : 0ull;
        const uint64_t p_hi = n_hi ? (0xFFFFFFFFull << (32 - n_hi)) & 0xFFFFFFFFull : 0ull;
        uint64_t lo, hi, ok;
        pack64_avx512(v, p_lo | (p_hi << 32), lo, hi, ok);
        put_words<NT>(planes, valid, w - w0, (lo & 0xFFFFFFFFull) | (hi << 32), (uint32_t)ok);
        if (w + 1 < w1) put_words<NT>(planes, valid, w - w0 + 1, (lo >> 32) | (hi & 0xFFFFFFFF00000000ull), (uint32_t)(ok >> 32));
        bad += nb - (uint32_t)__builtin_popcountll(ok);
    }
    return bad;
}

bool have_avx512() {
    static const bool v = __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vbmi") && !getenv("PGR_NO_AVX512") &&
                          !getenv("PGR_NO_AVX2");
    return v;
}

bool have_avx2() {
    static const bool v = __builtin_cpu_supports("avx2") && !getenv("PGR_NO_AVX2");
    return v;
}

}  // namespace

uint64_t pack_words(const uint8_t *seq, uint64_t len, uint64_t w0, uint64_t w1, uint64_t *planes, uint32_t *valid) {
    if (have_avx512()) return pack_words_avx512<false>(seq, len, w0, w1, planes, valid);
    return have_avx2() ? pack_words_avx2<false>(seq, len, w0, w1, planes, valid) : pack_words_scalar(seq, len, w0, w1, planes, valid);
}

// the same into a pinned staging window: non-temporal stores, fenced before the caller hands the window to the DMA engine
uint64_t pack_words_stream(const uint8_t *seq, uint64_t len, uint64_t w0, uint64_t w1, uint64_t *planes, uint32_t *valid) {
    // (short pieces -- the contigs of a query batch -- end in partial write-combining lines: regular stores there)
    if (w1 - w0 < 2048) return pack_words(seq, len, w0, w1, planes, valid);
    uint64_t bad;
    if (have_avx512()) bad = pack_words_avx512<true>(seq, len, w0, w1, planes, valid);
    else if (have_avx2()) bad = pack_words_avx2<true>(seq, len, w0, w1, planes, valid);
    else return pack_words_scalar(seq, len, w0, w1, planes, valid);
    _mm_sfence();
    return bad;
}

// one contig of a GROUP of short contigs that a job packs back to back into the window: non-temporal stores whatever the
// contig's length (the group's destination is one contiguous range), no fence -- the job ends with stream_fence()
uint64_t pack_words_stream_nofence(const uint8_t *seq, uint64_t len, uint64_t w0, uint64_t w1, uint64_t *planes, uint32_t *valid) {
    if (have_avx512()) return pack_words_avx512<true>(seq, len, w0, w1, planes, valid);
    if (have_avx2()) return pack_words_avx2<true>(seq, len, w0, w1, planes, valid);
    return pack_words_scalar(seq, len, w0, w1, planes, valid);
}
void stream_fence() { _mm_sfence(); }

// memcpy into a pinned staging window with non-temporal stores (any alignment: the unaligned head and tail go through memcpy)
__attribute__((target("avx2"))) static void stream_copy_avx2(uint8_t *d, const uint8_t *s, size_t n) {
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(s + i)), b = _mm256_loadu_si256((const __m256i *)(s + i + 32)),
                      c = _mm256_loadu_si256((const __m256i *)(s + i + 64)), e = _mm256_loadu_si256((const __m256i *)(s + i + 96));
        _mm256_stream_si256((__m256i *)(d + i), a);
        _mm256_stream_si256((__m256i *)(d + i + 32), b);
        _mm256_stream_si256((__m256i *)(d + i + 64), c);
        _mm256_stream_si256((__m256i *)(d + i + 96), e);
    }
    for (; i + 32 <= n; i += 32) _mm256_stream_si256((__m256i *)(d + i), _mm256_loadu_si256((const __m256i *)(s + i)));
    if (i < n) memcpy(d + i, s + i, n - i);
    _mm_sfence();
}
void stream_copy(void *dst, const void *src, size_t n) {
    uint8_t *d = (uint8_t *)dst;
    const uint8_t *s = (const uint8_t *)src;
    if (!have_avx2() || n < 4096) {
        memcpy(d, s, n);
        return;
    }
    const size_t head = (32 - ((uintptr_t)d & 31)) & 31;
    if (head) memcpy(d, s, head);
    stream_copy_avx2(d + head, s + head, n - head);
}

}  // namespace pgr

extern "C" uint64_t pgr_packed_words(uint32_t n, const uint64_t *lens) {
    uint64_t w = 0;
    if (lens)
        for (uint32_t i = 0; i < n; ++i) w += (lens[i] + 31) / 32;
    return w;
}

extern "C" int pgr_pack_ascii(uint32_t n, const uint8_t *const *seqs, const uint64_t *lens, int n_threads, uint64_t *planes,
                              uint32_t *valid, uint64_t *n_invalid) {
    if (n && (!seqs || !lens)) return PGR_ERR_INVALID_ARG;
    if (pgr_packed_words(n, lens) && (!planes || !valid)) return PGR_ERR_INVALID_ARG;
    struct Job {
        uint32_t c;
        uint64_t w0, w1, out;
    };
    std::vector<Job> jobs;
    uint64_t woff = 0;
    constexpr uint64_t PIECE = 1u << 16;  // words per job (2 MiB of ASCII)
    for (uint32_t c = 0; c < n; ++c) {
        if (lens[c] && !seqs[c]) return PGR_ERR_INVALID_ARG;
        const uint64_t nw = (lens[c] + 31) / 32;
        for (uint64_t w = 0; w < nw; w += PIECE) jobs.push_back(Job{c, w, std::min(nw, w + PIECE), woff + w});
        woff += nw;
    }
    std::atomic<uint64_t> bad{0};
    pgr::HostPool::instance().parallel_for(
        jobs.size(),
        [&](size_t i) {
            const Job &j = jobs[i];
            const uint64_t b = pgr::pack_words(seqs[j.c], lens[j.c], j.w0, j.w1, planes + j.out, valid + j.out);
            if (b) bad.fetch_add(b);
        },
        n_threads > 0 ? (unsigned)n_threads : 0u);
    if (n_invalid) *n_invalid = bad.load();
    return PGR_OK;
}
