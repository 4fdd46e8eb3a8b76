// CPU check of csrc/arena_list.h (the free list under pgr_ctx_reserve): random take / give_back against a byte map.
//   usage: arena_list_harness <iterations>
// invariants after every step: handed-out ranges do not overlap and lie inside the arena; the free list is address ordered, its ranges
// are disjoint, NEVER adjacent (neighbours are merged) and cover exactly the bytes that are not handed out; take() fails only when no
// free range holds the request (checked against the byte map), and picks a smallest sufficient range.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "arena_list.h"

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937_64 rng(12345);
    for (int round = 0; round < 8; ++round) {
        const size_t SIZE = 4096 * (size_t)(64 + rng() % 960);
        pgr::ArenaList L;
        L.reset(SIZE);
        std::vector<uint8_t> used(SIZE / 4096, 0);  // one entry per 4 KiB unit
        std::vector<std::pair<size_t, size_t>> live;
        for (int it = 0; it < iters; ++it) {
            const bool do_take = live.empty() || (rng() % 100) < 55;
            if (do_take) {
                const size_t want = 4096 * (size_t)(1 + rng() % (rng() % 4 ? 8 : 200));
                // smallest sufficient free run according to the byte map
                size_t best_fit = ~(size_t)0, run = 0;
                for (size_t u = 0; u <= used.size(); ++u) {
                    if (u < used.size() && !used[u]) {
                        ++run;
                        continue;
                    }
                    if (run * 4096 >= want && run * 4096 < best_fit) best_fit = run * 4096;
                    run = 0;
                }
                const size_t off = L.take(want);
                if (off == pgr::ArenaList::NONE) {
                    if (best_fit != ~(size_t)0) return printf("FAIL: take(%zu) failed although a run of %zu is free\n", want, best_fit), 1;
                    continue;
                }
                if (best_fit == ~(size_t)0) return printf("FAIL: take(%zu) succeeded although nothing fits\n", want), 1;
                if (off % 4096 || off + want > SIZE) return printf("FAIL: range outside the arena\n"), 1;
                // the run the block was cut from must be a smallest sufficient one, and the block its front
                size_t lo = off / 4096, hi = (off + want) / 4096;
                while (lo > 0 && !used[lo - 1]) --lo;
                while (hi < used.size() && !used[hi]) ++hi;
                if ((hi - lo) * 4096 != best_fit) return printf("FAIL: not best fit (%zu vs %zu)\n", (hi - lo) * 4096, best_fit), 1;
                if (lo != off / 4096) return printf("FAIL: not the front of its range\n"), 1;
                for (size_t u = off / 4096; u < (off + want) / 4096; ++u) {
                    if (used[u]) return printf("FAIL: overlap\n"), 1;
                    used[u] = 1;
                }
                live.push_back({off, want});
            } else {
                const size_t i = rng() % live.size();
                L.give_back(live[i].first, live[i].second);
                for (size_t u = live[i].first / 4096; u < (live[i].first + live[i].second) / 4096; ++u) used[u] = 0;
                live[i] = live.back();
                live.pop_back();
            }
            if (it % 64 == 0 || it + 1 == iters) {  // the free list is exactly the complement, merged
                size_t prev_end = ~(size_t)0, covered = 0;
                for (const auto &kv : L.free_by_off) {
                    if (prev_end != ~(size_t)0 && kv.first <= prev_end) return printf("FAIL: free ranges touch or overlap\n"), 1;
                    for (size_t u = kv.first / 4096; u < (kv.first + kv.second) / 4096; ++u)
                        if (used[u]) return printf("FAIL: a free range covers a live block\n"), 1;
                    prev_end = kv.first + kv.second;
                    covered += kv.second;
                }
                size_t free_units = 0;
                for (uint8_t x : used) free_units += !x;
                if (covered != free_units * 4096 || covered != L.free_bytes()) return printf("FAIL: free bytes %zu vs %zu\n", covered, free_units * 4096), 1;
            }
        }
        for (auto &b : live) L.give_back(b.first, b.second);
        if (L.free_by_off.size() != 1 || L.free_by_off.begin()->first != 0 || L.free_by_off.begin()->second != SIZE)
            return printf("FAIL: the arena did not come back as one range\n"), 1;
    }
    printf("ok\n");
    return 0;
}
