// CPU harness (no GPU, no HIP): csrc/island_list.h -- the islands of the exact machine listed from the FLAGGED tiles only --
// against the tile-by-tile reading it replaced (four passes over every tile), on random flag patterns: isolated flags, clusters,
// runs of "nothing but non-ACGT bytes" tiles of every length around the deep-gap threshold, mostly-flagged contigs, short contigs,
// sketch specs.  Built and run by tests/test_island_list_cpu.py.   usage: island_list_harness <cases>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "island_list.h"

using pgr::Island;

static void dense(uint32_t n, const uint32_t *tile_first, const uint32_t *h_len, uint32_t tc, bool sketch, const uint32_t *flags,
                  const uint32_t *n_invalid, uint8_t *tf, std::vector<Island> &islands, std::vector<uint32_t> &gap_segs) {
    for (uint32_t c = 0; c < n; ++c) {
        if (n_invalid[c] == 0 && (sketch || !(flags[c] & 1u))) continue;
        const uint32_t t0 = tile_first[c], nt = tile_first[c + 1] - t0;
        const uint64_t L = h_len[c];
        for (uint32_t t = 0, run = 0; t < nt; ++t) {
            run = (tf[t0 + t] & 4) ? run + 1 : 0;
            if (run >= 5) tf[t0 + t - 2] |= 8;
        }
        for (uint32_t t = 0; t < nt; ++t)
            if (tf[t0 + t] & 8) {
                uint32_t e = t;
                while (e + 1 < nt && (tf[t0 + e + 1] & 8)) ++e;
                gap_segs.push_back(t0 + c + t);
                gap_segs.push_back(t0 + c + e + 1);
                for (uint32_t q = t; q <= e; ++q) tf[t0 + q] = 0;
                t = e;
            }
        uint32_t n_flag = 0;
        for (uint32_t t = 0; t < nt; ++t) n_flag += tf[t0 + t] != 0;
        if (n_flag == 0) continue;
        if (sketch && n_invalid[c] == 0) continue;
        if (3ull * n_flag > nt) {
            bool pal = false;
            for (uint32_t t = 0; t < nt; ++t) pal = pal || (tf[t0 + t] & 1);
            islands.push_back(Island{c, 0, L, false, pal});
            continue;
        }
        for (uint32_t t = 0; t < nt;) {
            if (!tf[t0 + t]) {
                ++t;
                continue;
            }
            uint32_t ta = t, tb = t;
            while (tb + 1 < nt && (tf[t0 + tb + 1] || (tb + 2 < nt && tf[t0 + tb + 2]))) ++tb;
            bool any_pal = false;
            for (uint32_t q = ta; q <= tb; ++q) any_pal = any_pal || (tf[t0 + q] & 1);
            if (tb + 1 < nt && any_pal) ++tb;
            Island is{c, (uint64_t)ta * tc, tb + 1 == nt ? L : std::min<uint64_t>(L, (uint64_t)(tb + 1) * tc), false, false};
            is.pal = any_pal;
            if (L - is.E < 2ull * tc) is.E = L;
            if (!islands.empty() && islands.back().contig == c && islands.back().E + tc >= is.B) {
                islands.back().E = std::max(islands.back().E, is.E);
                islands.back().pal = islands.back().pal || is.pal;
            } else {
                islands.push_back(is);
            }
            t = tb + 1;
        }
    }
}

int main(int argc, char **argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 1000;
    std::mt19937_64 rng(12345);
    int failures = 0;
    size_t n_islands = 0, n_gaps = 0, n_pal = 0, n_whole = 0, n_shorter = 0, n_cut = 0;
    const uint32_t tc = 3904;
    for (int it = 0; it < cases; ++it) {
        const uint32_t n = 1 + rng() % 6;
        std::vector<uint32_t> tile_first(n + 1, 0), h_len(n), flags(n, 0), n_inv(n, 0);
        for (uint32_t c = 0; c < n; ++c) {
            const uint32_t kind = rng() % 4;
            const uint32_t nt = kind == 0 ? 1 + rng() % 4 : kind == 1 ? 1 + rng() % 40 : 1 + rng() % 3000;
            tile_first[c + 1] = tile_first[c] + nt;
            h_len[c] = (nt - 1) * tc + 1 + rng() % (tc + 300);  // (a one-tile contig may exceed a core)
        }
        std::vector<uint8_t> tf(tile_first[n] + 8, 0);
        for (uint32_t c = 0; c < n; ++c) {
            const uint32_t t0 = tile_first[c], nt = tile_first[c + 1] - t0;
            const uint32_t mode = rng() % 6;  // 0: clean
            auto put = [&](uint32_t t, uint8_t f) {
                if (t < nt) {
                    tf[t0 + t] |= f;
                    if (f & 1) flags[c] |= 1;
                    if (f & 6) n_inv[c] += 1 + rng() % 100;
                }
            };
            if (mode >= 1) {
                const uint32_t k = mode == 5 ? nt / 2 + rng() % (nt / 2 + 1) : 1 + rng() % 12;
                for (uint32_t i = 0; i < k; ++i) {
                    const uint32_t t = rng() % nt;
                    const uint32_t what = rng() % 5;
                    if (what == 0) put(t, 1);
                    else if (what == 1) put(t, 2);
                    else if (what == 2) put(t, 3);
                    else if (what == 3) {  // a cluster
                        for (uint32_t q = 0, m = 1 + rng() % 6; q < m; ++q) put(t + rng() % 8, (uint8_t)(1 + rng() % 3));
                    } else {  // a run of tiles with nothing but non-ACGT bytes, around the threshold of five
                        const uint32_t len = rng() % 3 == 0 ? 1 + rng() % 400 : 1 + rng() % 9;
                        for (uint32_t q = 0; q < len; ++q) put(t + q, 6);
                        if (rng() % 2) put(t + len, 2);
                        if (rng() % 3 == 0 && t) put(t - 1, (uint8_t)(rng() % 2 ? 2 : 1));
                    }
                }
            }
            if (rng() % 16 == 0) n_inv[c] = 0, flags[c] |= 0;  // (a contig whose count says "clean": skipped unless a palindrome bit is set)
        }
        const bool sketch = rng() % 5 == 0;
        std::vector<uint8_t> tf_a = tf, tf_b = tf;
        std::vector<Island> ia, ib;
        std::vector<uint32_t> ga, gb;
        dense(n, tile_first.data(), h_len.data(), tc, sketch, flags.data(), n_inv.data(), tf_a.data(), ia, ga);
        pgr::list_islands_from_flags(n, tile_first.data(), h_len.data(), tc, sketch, flags.data(), n_inv.data(), tf_b.data(), nullptr, ib, gb);
        {
            // with the tiles' palindrome positions the islands may only END earlier (by whole tiles: the clean neighbour is left out),
            // never start elsewhere, and they cover every flagged tile all the same
            std::vector<uint16_t> pal(tile_first[n] + 8);
            for (auto &x : pal) {
                const uint32_t lo = rng() % 64, hi = lo + rng() % (64 - lo);
                x = (uint16_t)(lo | (hi << 8));
            }
            std::vector<uint8_t> tf_c = tf;
            std::vector<Island> ic;
            std::vector<uint32_t> gc;
            pgr::list_islands_from_flags(n, tile_first.data(), h_len.data(), tc, sketch, flags.data(), n_inv.data(), tf_c.data(), pal.data(), ic, gc);
            bool ok = gc == gb && tf_c == tf_b;
            uint64_t bases_b = 0, bases_c = 0;
            for (const Island &x : ib) bases_b += x.E - x.B;
            for (const Island &x : ic) bases_c += x.E - x.B;
            ok = ok && bases_c <= bases_b;
            // every island of ic lies inside an island of ib with the same start, or follows a split of one
            for (const Island &x : ic) {
                bool inside = false;
                for (const Island &y : ib) inside = inside || (y.contig == x.contig && y.B <= x.B && x.E <= y.E);
                ok = ok && inside;
            }
            // every flagged tile (after the deep-gap pass) is covered
            for (uint32_t c = 0; c < n && ok; ++c) {
                if (n_inv[c] == 0 && (sketch || !(flags[c] & 1u))) continue;
                if (sketch && n_inv[c] == 0) continue;
                for (uint32_t t = tile_first[c]; t < tile_first[c + 1] && ok; ++t) {
                    if (!tf_c[t]) continue;
                    const uint64_t p0 = (uint64_t)(t - tile_first[c]) * tc;
                    bool cov = false;
                    for (const Island &x : ic) cov = cov || (x.contig == c && x.B <= p0 && (p0 + tc <= x.E || x.E == h_len[c]));
                    ok = cov;
                }
            }
            n_shorter += bases_c < bases_b;
            if (!ok && ++failures <= 5) fprintf(stderr, "case %d: islands listed with palindrome positions are not a trimmed form of those without\n", it);
        }
        {
            // round 6, cut_margin > 0: islands may begin and end INSIDE tiles whose only flag is the palindromic one.  Every island lies
            // inside an island of the whole-tile list; tiles with a non-ACGT flag are covered whole; of a tile flagged for palindromic
            // k-mers only, the blocks [first, last] are covered with cut_margin in front and the settling distance behind; a cut
            // lies in a tile the tile kernel has computed (flag 1 or none); islands stay more than a tile apart.
            const uint32_t margin = 320, settle = (it & 1) ? 1408 : 320;
            std::vector<uint16_t> pal(tile_first[n] + 8);
            for (auto &x : pal) {
                const uint32_t lo = rng() % 61, hi = lo + rng() % (61 - lo);
                x = (uint16_t)(lo | (hi << 8));
            }
            std::vector<uint8_t> tf_c = tf;
            std::vector<Island> ic;
            std::vector<uint32_t> gc;
            pgr::list_islands_from_flags(n, tile_first.data(), h_len.data(), tc, sketch, flags.data(), n_inv.data(), tf_c.data(), pal.data(), ic, gc, 0, 0xFFFFFFFFu, margin, (it & 1) ? 0 : settle);
            bool ok = gc == gb && tf_c == tf_b;
            for (size_t i = 0; i < ic.size() && ok; ++i) {
                const Island &x = ic[i];
                const uint64_t L = h_len[x.contig];
                bool inside = false;
                for (const Island &y : ib) inside = inside || (y.contig == x.contig && y.B <= x.B + margin && x.E <= y.E);  // (B: up to `margin` into the clean tile in front)
                ok = inside && x.B < x.E && x.E <= L;
                // (islands do not share a tile, and the warm-up of one starts behind the end of the one in front -- wherever a stuck machine moves that)
                if (i && ic[i - 1].contig == x.contig) {
                    const uint64_t pe = std::max(ic[i - 1].E, ic[i - 1].ext_limit);
                    ok = ok && ic[i - 1].E + tc < x.B && pe + 256 <= x.B && ((pe - 1) / tc < x.B / tc || pe % tc == 0);
                }
                const uint32_t t0 = tile_first[x.contig];
                if (x.cutB) {  // in the first flagged tile, or within `margin` of it in the clean tile in front
                    const uint64_t tb_ = x.B / tc, tf_ = tf_c[t0 + tb_] ? tb_ : tb_ + 1;
                    ok = ok && x.B % 64 == 0 && x.B % tc != 0 && tf_c[t0 + tb_] <= 1 && tf_c[t0 + tf_] == 1 &&
                         x.B + margin == tf_ * (uint64_t)tc + (uint64_t)(pal[t0 + tf_] & 0xFF) * 64;
                }
                else ok = ok && (x.B % tc == 0);
                if (x.cutE) ok = ok && x.E + 2ull * tc <= L && tf_c[t0 + x.E / tc] <= 1 && x.E % 64 == 0;
                else ok = ok && (x.E == L || x.E % tc == 0);
                // how far a stuck machine may move the end on: inside the clean tile behind the last flagged one, a tile short of the next island
                if (x.ext_limit) ok = ok && x.cutE && x.ext_limit > x.E && x.ext_limit % 64 == 0 && x.ext_limit + 2ull * tc <= L + 63 &&
                                      tf_c[t0 + (x.ext_limit - 1) / tc] <= 1 &&
                                      (i + 1 == ic.size() || ic[i + 1].contig != x.contig || x.ext_limit + 256 <= ic[i + 1].B);
                n_cut += x.cutB + x.cutE;
            }
            for (uint32_t c = 0; c < n && ok; ++c) {
                if (n_inv[c] == 0 && (sketch || !(flags[c] & 1u))) continue;
                if (sketch && n_inv[c] == 0) continue;
                for (uint32_t t = tile_first[c]; t < tile_first[c + 1] && ok; ++t) {
                    if (!tf_c[t]) continue;
                    const uint64_t p0 = (uint64_t)(t - tile_first[c]) * tc;
                    uint64_t need_lo = p0, need_hi = std::min<uint64_t>(p0 + tc, h_len[c]);
                    if (tf_c[t] == 1) {
                        const uint64_t lo = (uint64_t)(pal[t] & 0xFF) * 64, hi = (uint64_t)(pal[t] >> 8) * 64 + 64;
                        need_lo = p0 + (lo > margin ? lo - margin : 0);
                        need_hi = std::min<uint64_t>(h_len[c], p0 + hi + settle);
                    }
                    bool cov = false;
                    for (const Island &x : ic) cov = cov || (x.contig == c && x.B <= need_lo && need_hi <= x.E);
                    ok = cov;
                }
            }
            if (!ok && ++failures <= 5) fprintf(stderr, "case %d: islands that begin / end inside tiles break an invariant\n", it);
        }
        {  // ranges of contigs listed one after the other give the same lists (what the product does on several threads)
            std::vector<uint8_t> tf_d = tf;
            std::vector<Island> id;
            std::vector<uint32_t> gd;
            const uint32_t cut = (uint32_t)(rng() % (n + 1));
            pgr::list_islands_from_flags(n, tile_first.data(), h_len.data(), tc, sketch, flags.data(), n_inv.data(), tf_d.data(), nullptr, id, gd, 0, cut);
            pgr::list_islands_from_flags(n, tile_first.data(), h_len.data(), tc, sketch, flags.data(), n_inv.data(), tf_d.data(), nullptr, id, gd, cut, n);
            bool ok = id.size() == ib.size() && gd == gb && tf_d == tf_b;
            for (size_t i = 0; ok && i < id.size(); ++i)
                ok = id[i].contig == ib[i].contig && id[i].B == ib[i].B && id[i].E == ib[i].E && id[i].whole == ib[i].whole && id[i].pal == ib[i].pal;
            if (!ok && ++failures <= 5) fprintf(stderr, "case %d: two ranges of contigs give other lists than all contigs at once\n", it);
        }
        bool same = ia.size() == ib.size() && ga == gb && tf_a == tf_b;
        n_islands += ia.size();
        n_gaps += ga.size() / 2;
        for (const Island &x : ia) n_pal += x.pal, n_whole += (x.B == 0 && x.E == h_len[x.contig]);
        for (size_t i = 0; same && i < ia.size(); ++i)
            same = ia[i].contig == ib[i].contig && ia[i].B == ib[i].B && ia[i].E == ib[i].E && ia[i].whole == ib[i].whole && ia[i].pal == ib[i].pal;
        if (!same) {
            if (++failures <= 5) fprintf(stderr, "case %d: %zu / %zu islands, %zu / %zu gap ranges\n", it, ia.size(), ib.size(), ga.size(), gb.size());
        }
    }
    printf("%d cases, %d failures\n", cases, failures);
    printf("%zu islands (%zu with a palindromic tile, %zu whole contigs), %zu deep-gap ranges; %zu cases end islands earlier with palindrome positions; %zu island ends inside tiles\n", n_islands, n_pal, n_whole, n_gaps, n_shorter, n_cut);
    return failures ? 1 : 0;
}
