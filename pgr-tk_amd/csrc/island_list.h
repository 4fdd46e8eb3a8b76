// island_list.h -- from tile flags to the islands of the exact state machine (host side, plain C++: tests/island_list_harness.cpp
// builds it with g++ against a tile-by-tile reading).
//
// flags[c] bit 0 = a tile of contig c saw a palindromic k-mer (skipped push, shmmrutils.rs:477-480), n_invalid[c] = its non-ACGT
// bytes, tf[tile] = tile flags (bit 0 palindromic k-mer, bit 1 non-ACGT byte in reach, bit 2 nothing but such bytes).
//
// A genome-sized batch has half a million tiles of which a few thousand are flagged: the tile-by-tile form of this (four passes
// over every tile of every flagged contig) took 0.7 ms of a 2 Gbp batch's 8 -- on the critical path, between the tile kernel's
// flags and the first round of the islands.  Here the flags are read eight tiles at a time and everything else works on the list
// of flagged tiles.
//
// Round 6: with the tile kernel's report of WHERE in a flagged tile the palindromic k-mers lie (pal: first | last << 8 block of 64
// core positions) and cut_margin > 0, an island around such an array begins and ends inside the tiles (Island::cutB / cutE /
// ext_limit); the harness checks the invariants of those cuts on random flag patterns.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace pgr {

// A stretch [B, E) of one contig that the exact machine computes instead of the tiles: every island's
// left edge trusts the warm-up (everything before it is regular by construction: islands keep a clean tile
// on both sides), its right edge is verified with a probe (warm-up only) at E and the island grows when the
// machine has not yet returned to its regular regime there.
struct Island {
    uint32_t contig;
    uint64_t B, E;
    bool whole;  // one chunk for the whole contig (last resort)
    // a tile of the island saw a palindromic k-mer: inside a stretch of skipped pushes a chunk needs the true state of the
    // chunk in front (one seam per round), so such islands keep long chunks; islands around non-ACGT bytes verify at the
    // first try and are cut short for parallelism
    bool pal = true;
    // Round 6: an island may begin and end INSIDE a tile the tile kernel has computed (a tile whose only flag is the palindromic
    // k-mer: its elements in front of B / from E on are exact, see list_islands_from_flags).  cutB: the elements of tile B / tc below
    // position B stay that tile's own; cutE: those of tile E / tc at or above E do (E < the contig's length).  Neither set: the island
    // is made of whole tiles, as before.
    bool cutB = false, cutE = false;
    // cutE: how far the island's last chunk may move E on while its machine is stuck (ChunkDesc::ext_limit; 0: not at all) -- the end
    // of the clean tile behind the last flagged one, where an island of whole tiles ends
    uint64_t ext_limit = 0;
};

// tf is modified: tiles deep inside a run of non-ACGT bytes end up 0 (their segment ranges go to gap_segs).
inline void list_islands_from_flags(uint32_t n, const uint32_t *tile_first, const uint32_t *h_len, uint32_t tc, bool sketch,
                                    const uint32_t *flags, const uint32_t *n_invalid, uint8_t *tf, const uint16_t *pal,
                                    std::vector<Island> &islands, std::vector<uint32_t> &gap_segs, uint32_t c_begin = 0,
                                    uint32_t c_end = 0xFFFFFFFFu, uint32_t cut_margin = 0, uint32_t cut_settle = 0) {
    // (contigs [c_begin, min(c_end, n)): islands never span contigs, so ranges of contigs can be listed side by side and the
    // lists put behind one another)
    // pal (or NULL): for a tile with flag bit 0, first | last << 8 block of 64 core positions that holds a palindromic k-mer
    // (0: in front of the core, >= tc / 64: behind it).  ISLAND_SETTLE: positions of regular sequence behind the last one
    // within which a machine that came out of the array stuck has practically always found back (a push at or below the stuck
    // minimum: one in ~w + 1 pushes does; the probe at the island's end checks it and the island grows if not)
    // cut_margin (0: islands of whole tiles): positions of regular sequence kept in front of the first palindromic block of an
    // island that begins inside its first tile -- a multiple of 64, >= w + k + 64: an element the tile keeps (position < B) has every
    // window it belongs to, and the k-mers of those windows, in front of the first skipped push; the machine that starts 256
    // positions in front of B warms up on regular sequence (the tile in front is clean, this tile is up to its first block).
    // cut_settle (with cut_margin; 0: ISLAND_SETTLE): positions behind the last palindromic block at which an island that ends
    // inside a tile is to end -- a multiple of 64, >= 2 w + k + 64: the ring holds nothing from in front of the array any more,
    // a rescan since has seen only pushes from behind it, and a machine that is NOT stuck there is the regular one; a machine that
    // IS stuck moves the end on itself, block by block, until a push frees it (ChunkDesc::ext_limit) -- so the settling distance
    // need not cover the stuck machines' tail (3 of 2 400 islands at 1408 positions: the tail falls off like 1 / distance).
    constexpr uint32_t ISLAND_SETTLE = 1408;
    struct FT {
        uint32_t t;
        uint8_t f;
    };
    std::vector<FT> F;
    for (uint32_t c = c_begin; c < n && c < c_end; ++c) {
        if (n_invalid[c] == 0 && (sketch || !(flags[c] & 1u))) continue;
        const uint32_t t0 = tile_first[c], nt = tile_first[c + 1] - t0;
        const uint64_t L = h_len[c];
        // the flagged tiles of the contig, in order
        F.clear();
        {
            const uint8_t *p = tf + t0;
            uint32_t t = 0;
            for (; t + 8 <= nt; t += 8) {
                uint64_t w8;
                memcpy(&w8, p + t, 8);
                if (!w8) continue;
                for (uint32_t q = 0; q < 8; ++q)
                    if (p[t + q]) F.push_back(FT{t + q, p[t + q]});
            }
            for (; t < nt; ++t)
                if (p[t]) F.push_back(FT{t, p[t]});
        }
        if (F.empty()) continue;
        // The inside of a long run of non-ACGT bytes (the gaps of a reference chromosome: up to 30 Mbp) needs no
        // machine at all.  Every position there pushes the same stale k-mer (shmmrutils.rs:461-476), so the level-1
        // list holds one element per position, all with one x -- ties keep them through both reductions
        // (:359-415) and the min_span stencil drops every one of them for having a neighbour with its x (:545-550).
        // What an element further than 2 r^2 list places from both ends of such a run contributes to the rest of
        // the list is nothing: tiles whose whole extended range is invalid AND whose two neighbours on either side
        // are too (>= 7 kbp of the run kept at each end) are left out -- their segments stay empty, the islands on
        // both sides end inside the run, where a warmed-up machine is exact.  (Round 3 pushed 40.9 Mbp of such
        // positions of a chromosome-like contig through the chunk kernel and the list stage: half of its 2.4 ms.)
        // A maximal run [s, e] of tiles with bit 2 of five or more: its tiles s + 2 .. e - 2 are deep.
        size_t n_flag = F.size();
        for (size_t i = 0; i < F.size();) {
            if (!(F[i].f & 4)) {
                ++i;
                continue;
            }
            size_t j = i;
            while (j + 1 < F.size() && (F[j + 1].f & 4) && F[j + 1].t == F[j].t + 1) ++j;
            if (j - i + 1 >= 5) {
                const uint32_t s = F[i].t + 2, e = F[j].t - 2;
                gap_segs.push_back(t0 + c + s);  // segment index of tile s of contig c
                gap_segs.push_back(t0 + c + e + 1);
                for (size_t q = i + 2; q + 2 <= j; ++q) F[q].f = 0;  // not flagged: no island over them
                memset(tf + t0 + s, 0, (size_t)e - s + 1);
                n_flag -= (size_t)e - s + 1;
            }
            i = j + 1;
        }
        if (n_flag == 0) continue;
        if (sketch && n_invalid[c] == 0) continue;  // sketch has no state machine: palindromes are exact
        if (3ull * n_flag > nt) {  // mostly irregular: one island
            bool pal = false;
            for (const FT &x : F) pal = pal || (x.f & 1);
            islands.push_back(Island{c, 0, L, false, pal});
            continue;
        }
        if (n_flag != F.size()) F.erase(std::remove_if(F.begin(), F.end(), [](const FT &x) { return x.f == 0; }), F.end());
        for (size_t i = 0; i < F.size();) {
            // (no tile in front of the first flagged one: tile t - 1 is clean, so nothing irregular lies within its reach -- which
            // ends w - 1 + 64 positions INTO tile t --, and the machine that starts 256 positions in front of tile t is regular
            // at its first step by construction; behind tiles deep inside a gap it starts inside the run, where a warmed-up
            // machine is exact)
            const uint32_t ta = F[i].t;
            uint32_t tb = ta;
            bool any_pal = (F[i].f & 1) != 0;
            size_t j = i, last_pal = i;
            while (j + 1 < F.size() && F[j + 1].t - tb <= 2) {  // bridge 1-tile gaps
                ++j;
                tb = F[j].t;
                if (F[j].f & 1) {
                    any_pal = true;
                    last_pal = j;
                }
            }
            // a clean neighbour on the right for the machine to find back into its regular regime -- behind skipped pushes
            // (palindromic k-mers) it may arrive stuck; behind a non-ACGT byte it cannot: the byte lies >= w + k + 64
            // positions in front of the first clean tile (or that tile would be flagged), every position pushes, and the
            // ring holds only pushes from behind the byte when the island ends.  The probe at the island's end checks it.
            // (the clean neighbour is 3904 more positions through the machine for an array of perhaps 100: where the tile kernel has
            // said where the last palindromic k-mer lies, and ISLAND_SETTLE positions of the flagged tiles follow it, they are the room)
            bool neighbour = tb + 1 < nt && any_pal;
            uint64_t cut_b = 0, cut_e = 0;  // (0: no cut)
            if (pal && cut_margin) {
                // Sub-tile ends.  Left: the group's first tile has no flag but the palindromic one -- the tile kernel computed it, and
                // what it selected below (first palindromic block - cut_margin) is exact.  Right: the group's last tile is the one with
                // the last palindromic k-mer and has no other flag -- ISLAND_SETTLE positions behind that block the machine has
                // practically always found back (the probe at E decides), and from E on the elements of the tile E lies in -- this one
                // or its clean neighbour, both computed -- are exact.
                // (an array within cut_margin of its tile's start: B lies in the tile in front -- clean, or the group would begin there;
                // a palindromic k-mer in FRONT of the core, block 0 of the report, would have flagged that tile)
                if (F[i].f == 1) {
                    const uint64_t first = (uint64_t)ta * tc + (uint64_t)(pal[t0 + ta] & 0xFF) * 64;
                    if (first > cut_margin && tc > 2 * cut_margin) cut_b = first - cut_margin;
                    if (cut_b + cut_margin >= L) cut_b = 0;  // (a block index beyond the contig's end: not from this tile kernel)
                    if (cut_b % tc == 0) cut_b = 0;          // (exactly the tile's start: an island of whole tiles on this side)
                }
                if (neighbour && last_pal == j && F[j].f == 1) {
                    const uint32_t hi_blk = (uint32_t)(pal[t0 + tb] >> 8);
                    if (hi_blk < tc / 64) cut_e = (uint64_t)tb * tc + (uint64_t)(hi_blk + 1) * 64 + (cut_settle ? cut_settle : ISLAND_SETTLE);
                    if (cut_e + 2ull * tc > L) cut_e = 0;  // (near the contig's end: the island takes the tail, as before)
                    if (cut_e / tc > (uint64_t)tb + 1) cut_e = 0;    // (tiles shorter than the settling distance: E must lie in this tile or its clean neighbour)
                }
                if (cut_e) neighbour = false;
            } else if (neighbour && pal) {
                const uint32_t hi_blk = (uint32_t)(pal[t0 + F[last_pal].t] >> 8);
                if (hi_blk < tc / 64 && (uint64_t)(tb - F[last_pal].t) * tc + (uint64_t)(tc / 64 - 1 - hi_blk) * 64 >= ISLAND_SETTLE) neighbour = false;
            }
            if (neighbour) ++tb;
            Island is{c, (uint64_t)ta * tc, tb + 1 == nt ? L : std::min<uint64_t>(L, (uint64_t)(tb + 1) * tc), false, false};
            is.pal = any_pal;
            if (cut_b) {
                is.B = cut_b;
                is.cutB = true;
            }
            if (cut_e) {
                is.E = cut_e;
                is.cutE = true;
                is.ext_limit = std::min<uint64_t>((uint64_t)(tb + 2) * tc, (L - 2ull * tc) / 64 * 64);
                if (is.ext_limit <= is.E) is.ext_limit = 0;
            }
            if (L - is.E < 2ull * tc) {  // the contig's tail region joins the island
                is.E = L;
                is.cutE = false;
                is.ext_limit = 0;
            }
            if (!islands.empty() && islands.back().contig == c && islands.back().E + tc >= is.B) {
                if (is.E > islands.back().E) {
                    islands.back().E = is.E;
                    islands.back().cutE = is.cutE;
                    islands.back().ext_limit = is.ext_limit;
                }
                islands.back().pal = islands.back().pal || is.pal;
            } else {
                islands.push_back(is);
            }
            i = j + 1;
        }
    }
}

}  // namespace pgr
