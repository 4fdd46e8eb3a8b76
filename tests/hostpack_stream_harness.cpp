// CPU-only harness for the non-temporal-store packers of csrc/hostpack.cpp (pack_words_stream, pack_words_stream_nofence,
// stream_copy): what the staging windows of the host entry points are filled with.  Compared against the byte table of
// shmmrutils.rs:426-436 applied position by position (a scalar reading written here, not the library's).
// Built and run by tests/test_hostpack_cpu.py: g++ -O2 -std=c++17 -pthread harness.cpp csrc/hostpack.cpp
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "pgr_host.h"

static int code_of(uint8_t b) {
    switch (b) {
    case 'A': case 'a': case 0: return 0;
    case 'C': case 'c': case 1: return 1;
    case 'G': case 'g': case 2: return 2;
    case 'T': case 't': case 3: return 3;
    default: return 4;
    }
}

// words [w0, w1) of one contig, base i of a word at bit 31 - (i % 32), planes = low | high << 32 (include/pgr_hip.h)
static uint64_t ref_pack(const uint8_t *seq, uint64_t len, uint64_t w0, uint64_t w1, uint64_t *planes, uint32_t *valid) {
    uint64_t bad = 0;
    for (uint64_t w = w0; w < w1; ++w) {
        uint32_t lo = 0, hi = 0, v = 0;
        for (uint32_t i = 0; i < 32; ++i) {
            const uint64_t p = w * 32 + i;
            if (p >= len) break;
            const int c = code_of(seq[p]);
            const uint32_t bit = 1u << (31 - i);
            if (c < 4) {
                v |= bit;
                if (c & 1) lo |= bit;
                if (c & 2) hi |= bit;
            } else {
                ++bad;
            }
        }
        planes[w - w0] = (uint64_t)lo | ((uint64_t)hi << 32);
        valid[w - w0] = v;
    }
    return bad;
}

int main(int argc, char **argv) {
    const int n_cases = argc > 1 ? atoi(argv[1]) : 3000;
    std::mt19937_64 rng(12345);
    const char alpha[] = "ACGTacgtNnX\x00\x01\x02\x03-";
    int fails = 0;
    for (int cs = 0; cs < n_cases && fails < 5; ++cs) {
        const uint64_t len = cs < 200 ? (uint64_t)cs : (rng() % 70000);
        const int mode = (int)(rng() % 3);  // 0: clean ACGT, 1: mixed bytes, 2: runs of N
        std::vector<uint8_t> seq(len + 64);
        for (uint64_t i = 0; i < len; ++i) {
            if (mode == 0) seq[i] = (uint8_t)"ACGT"[rng() & 3];
            else if (mode == 1) seq[i] = (uint8_t)alpha[rng() % (sizeof(alpha) - 1)];
            else seq[i] = ((i / 97) % 5 == 0) ? 'N' : (uint8_t)"acgt"[rng() & 3];
        }
        const uint64_t nw = (len + 31) / 32;
        uint64_t w0 = nw ? rng() % (nw + 1) : 0, w1 = nw ? rng() % (nw + 1) : 0;
        if (cs % 3 == 0) { w0 = 0; w1 = nw; }
        if (w0 > w1) std::swap(w0, w1);
        const uint64_t n = w1 - w0;
        // destinations at odd alignments (a window's words start wherever the previous contig ended)
        const uint64_t shift_p = rng() % 8, shift_v = rng() % 16;
        std::vector<uint64_t> rp(n + 1), gp(n + 24, 0xDEADBEEFDEADBEEFull);
        std::vector<uint32_t> rv(n + 1), gv(n + 40, 0xDEADBEEFu);
        const uint64_t rb = ref_pack(seq.data(), len, w0, w1, rp.data(), rv.data());
        for (int variant = 0; variant < 2; ++variant) {
            std::fill(gp.begin(), gp.end(), 0xDEADBEEFDEADBEEFull);
            std::fill(gv.begin(), gv.end(), 0xDEADBEEFu);
            uint64_t gb;
            if (variant == 0) gb = pgr::pack_words_stream(seq.data(), len, w0, w1, gp.data() + shift_p, gv.data() + shift_v);
            else {
                gb = pgr::pack_words_stream_nofence(seq.data(), len, w0, w1, gp.data() + shift_p, gv.data() + shift_v);
                pgr::stream_fence();
            }
            bool ok = gb == rb && (n == 0 || (!memcmp(gp.data() + shift_p, rp.data(), n * 8) && !memcmp(gv.data() + shift_v, rv.data(), n * 4)));
            // nothing written outside [dst, dst + n)
            for (uint64_t i = 0; i < shift_p; ++i) ok = ok && gp[i] == 0xDEADBEEFDEADBEEFull;
            for (uint64_t i = shift_p + n; i < gp.size(); ++i) ok = ok && gp[i] == 0xDEADBEEFDEADBEEFull;
            for (uint64_t i = 0; i < shift_v; ++i) ok = ok && gv[i] == 0xDEADBEEFu;
            for (uint64_t i = shift_v + n; i < gv.size(); ++i) ok = ok && gv[i] == 0xDEADBEEFu;
            if (!ok) {
                fprintf(stderr, "case %d variant %d: len %llu words [%llu, %llu) mode %d: mismatch (bad %llu vs %llu)\n", cs, variant,
                        (unsigned long long)len, (unsigned long long)w0, (unsigned long long)w1, mode, (unsigned long long)gb, (unsigned long long)rb);
                ++fails;
            }
        }
        // stream_copy: any length, any alignment of source and destination
        {
            const size_t nb = (size_t)(rng() % 5000), so = (size_t)(rng() % 64), dof = (size_t)(rng() % 64);
            std::vector<uint8_t> src(nb + 128), dst(nb + 256, 0xA5);
            for (auto &b : src) b = (uint8_t)rng();
            pgr::stream_copy(dst.data() + dof, src.data() + so, nb);
            pgr::stream_fence();
            bool ok = nb == 0 || !memcmp(dst.data() + dof, src.data() + so, nb);
            for (size_t i = 0; i < dof; ++i) ok = ok && dst[i] == 0xA5;
            for (size_t i = dof + nb; i < dst.size(); ++i) ok = ok && dst[i] == 0xA5;
            if (!ok) {
                fprintf(stderr, "case %d: stream_copy of %zu bytes (src +%zu, dst +%zu) differs\n", cs, nb, so, dof);
                ++fails;
            }
        }
    }
    printf("%d cases, %d failures\n", n_cases, fails);
    return fails ? 1 : 0;
}
