// pipeline.hip -- orchestration of ONE pass of the sequence_to_shmmrs pipeline over a resident batch,
// synchronous (pgr_shmmrs_compute) and as a software pipeline over batches (pgr_pipe_*: the list stage and the pair records of
// batch i on the back stream beside the tiles of batch i + 1).
//
//   level1_tile_kernel  (dominant)  -> unordered per-tile segments of level-1 minimizers
//   level1_tail_kernel              -> per-contig tail segment (rescan-only positions)
//   level1_chunk_kernel             -> exact state machine (chunked, verified seams) for what the closed form cannot do
//   scan(seg counts) + fused_select -> reduce x2 + min_span, survivors in block slots
//   scan(block counts) + gather     -> final MM128 lists, per-contig offsets (+ rid patch)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>

#include "pgr_ctx.h"
#include "pgr_index.h"
#include "pgr_aln.h"
#include "pgr_small.h"
#include "island_list.h"

using namespace pgr;

namespace {
using Tmp_list = pgr::Tmp;
}  // namespace

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Exact state machine (level1_chunk_kernel) over ISLANDS: position ranges [B, E) of a contig (tile aligned, or
// the whole contig) that replace the closed-form tiles near an irregularity (non-ACGT byte, palindromic
// k-mer) or everything when the spec has no tile path (w < 17).  Inside an island the chunks of 32 kbp are
// seamed by emission step and every seam is verified against the previous chunk's end state; the island's
// left edge trusts the warm-up (everything before it is regular by construction: islands keep a clean tile
// on both sides), its right edge is verified with a probe (warm-up only) at E and the island grows when the
// machine has not yet returned to its regular regime there.
// (struct Island and the listing of the islands from the tile flags: island_list.h)

// One run of the exact machine over a list of islands, in two halves so that the first round's chunk kernel can be enqueued
// BEFORE the host has anything to wait for (ShmmrJob::stage1: the islands around non-ACGT bytes are known while the tile
// kernel still runs): begin() builds the chunks and enqueues round 0, finish() waits, verifies the seams, runs the chunks
// that need the true state again (rounds 1 ..) and puts the tiles' lists together.  `a` (its .out may move) and `islands`
// (they may grow) are the run's own copies: the caller reads them back.
struct IslandRun {
    pgr_ctx *ctx;
    const pgr_batch *b;
    L1Args a;
    std::vector<Island> islands;
    const std::vector<uint32_t> &tile_first;
    uint32_t tc;
    hipStream_t st;
    uint64_t CS_SHORT = 1024, CS_PAL = 512;
    std::vector<uint32_t> zero_ranges;  // segment ranges of (re)built islands -- and of the tiles the caller leaves out --, cleared by ONE kernel before the next chunk launch
    std::vector<size_t> zero_owner;     // per pair of zero_ranges: the island it was built for (SIZE_MAX: the caller's)
    struct HChunk {
        ChunkDesc d;
        size_t island;
        bool full_cap = false, probe = false, retired = false;
        bool moved = false;     // a probe whose island's end has moved on since it ran (ChunkDesc::ext_limit): it runs again, at the new end
        bool final = false;     // the state this chunk started from is known to be the true one
        ChunkState t_out;       // final: the true state at ce
        uint32_t ring_src = 0;  // final: ring slot that holds the true ring at ce (this chunk's, or the one it passed through)
        uint64_t n_push = 0, bmin = 0;  // of the last run: pushes at the steps [cs, ce), smallest x of those with branch 2 enabled
        uint64_t n_out = 0;             // of the last run: elements in the chunk's region
        bool dropped = false;           // its output is not part of the list (a stuck machine passed through it)
    };
    std::vector<HChunk> ch;
    std::vector<ChunkState> s_in, s_out;
    std::vector<uint32_t> status;
    std::vector<size_t> todo;
    uint64_t next_region;
    std::chrono::steady_clock::time_point t_isl0;
    int round = 0;
    // of the round that is enqueued
    size_t nq = 0, desc_bytes = 0;
    size_t n_moved = 0;          // island ends a stuck machine has moved on (diagnostics)
    std::vector<ChunkDesc> descs;
    std::unique_ptr<Tmp_list> d_zr;  // (the source vector and this block live until the synchronization at the end of the round)
    bool enqueued = false;
    // round 0 may run on a stream of its own (beside the tile kernel: begin(side)): the chunk kernel touches nothing the tile
    // kernel touches -- its regions lie behind the tiles' part of the level-1 buffer, which must not have to grow for this
    hipStream_t st_chunks = nullptr;
    size_t n_spliced = 0;  // tiles whose own elements were kept beside an island's (diagnostics)
    size_t n_in_place = 0; // tiles whose segment is one chunk's region as it stands (diagnostics)
    bool critical = false; // the first round has been launched: what is built from here on is waited for

    IslandRun(pgr_ctx *ctx_, hipStream_t st_, const pgr_batch *b_, const L1Args &a_, const std::vector<Island> &islands_, const std::vector<uint32_t> &tile_first_,
              uint32_t tc_, uint64_t region_base, const std::vector<uint32_t> &empty_seg_ranges)
        : ctx(ctx_), b(b_), a(a_), islands(islands_), tile_first(tile_first_), tc(tc_), st(st_), zero_ranges(empty_seg_ranges),
          zero_owner(empty_seg_ranges.size() / 2, SIZE_MAX), next_region(region_base) {
        // chunk length: 32 kbp for big jobs, shorter when the islands are few so that there are still thousands of wavefronts
        // (one per chunk), down to 1024 positions: a round costs what its slowest chunk costs -- ~3 us per step of 64 positions,
        // 225 us for the 4096-position chunks that were the minimum while a chunk had to own the segment-table entry of the tile it
        // starts in.  The lists of the chunks that start in one tile are put together behind the last round (finish()).
        uint64_t island_bases = 0;
        for (const Island &is : islands) island_bases += is.E - is.B;
        const uint64_t CS_MIN = ctx->opt.island_chunk_min > 0 ? (uint64_t)((ctx->opt.island_chunk_min + 63) / 64 * 64) : 1024;
        // (~2.5 wavefronts per SIMD: below that a round waits for dependent instructions, above it the SIMDs are busy -- a step is
        // ~1.5 us of issue -- and shorter chunks only add warm-up steps)
        CS_SHORT = std::min<uint64_t>(32768, std::max<uint64_t>(CS_MIN, ((island_bases / 2560 + 1023) / 1024) * 1024));
        // Islands around palindromic k-mers come from the tile kernel's flags: their rounds are on the critical path, one after
        // the other, and a round costs what its slowest chunk costs -- 512 positions (+ 256 of warm-up: 12 steps instead of 20;
        // a chunk that runs again: 8 instead of 16) unless the islands are large.  The islands around non-ACGT bytes run beside
        // the tile kernel: more, shorter chunks there only take its slots (chromosome-like 0.85 -> 0.875 ms at 768).
        const uint64_t PAL_MIN = ctx->opt.island_chunk_min > 0 ? CS_MIN : 512;
        CS_PAL = pal_chunk(island_bases, PAL_MIN);
    }
    // (a run that is dropped with its first round still on the side stream -- the flags added islands, or the pass starts over:
    // whoever uses the workspaces, the pinned image and the regions next must find them idle)
    ~IslandRun() {
        if (enqueued && st_chunks && st_chunks != st) (void)hipStreamSynchronize(st_chunks);
    }
    IslandRun(const IslandRun &) = delete;
    IslandRun &operator=(const IslandRun &) = delete;
    // Chunk length of islands around palindromic k-mers: enough chunks to fill the chip (~2.5 wavefronts per SIMD), and -- once that
    // asks for chunks of a tile or more, i.e. there are thousands of islands (a genome-sized batch: one per (AT)n microsatellite
    // longer than k, each a flagged tile and its clean neighbour) -- ONE chunk per such island: a seam inside it costs a warm-up
    // and, where the machine passes it stuck, a second round (a 2 Gbp genome-like batch: 148 chunks run again at 3584 positions
    // per chunk, none at 8192; 7.06 -> 6.81 ms)
    uint64_t pal_chunk(uint64_t bases, uint64_t floor_) const {
        uint64_t cs = std::max<uint64_t>(floor_, ((bases / 2560 + 255) / 256) * 256);
        if (cs >= tc) cs = std::max<uint64_t>(cs, ((2ull * tc + 512 + 255) / 256) * 256);
        return std::min<uint64_t>(32768, cs);
    }
    uint64_t cap_of(uint64_t len, bool full) const { return full ? len + a.w + 320 : std::min<uint64_t>(len + a.w + 320, len / 4 + 1024); }
    void isl_lap(const char *what, int r) const {
        if (ctx->opt.debug_times)
            fprintf(stderr, "[pgr]     islands round %d %-34s at %7.1f us\n", r, what,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_isl0).count());
    }
    // (room for the chunks of these islands: a genome-sized batch has ten thousand, 300 bytes each, and the vector grew eight times)
    void reserve_chunks(const std::vector<Island> &isl, size_t have) {
        size_t nc = 0;
        for (const Island &is : isl) {
            uint64_t CS = (is.pal && ctx->opt.no_island_relay) ? 32768 : is.pal ? CS_PAL : CS_SHORT;
            if (critical && a.tile_pal) CS = std::min<uint64_t>(CS, 2816);
            nc += is.whole ? 2 : (size_t)((is.E - is.B + CS - 1) / CS) + 1;
        }
        ch.reserve(have + nc + 16);
        todo.reserve(todo.size() + nc + 16);
    }
    int build(size_t ii);
    int enqueue_round();
    int process_round();
    int begin(hipStream_t side = nullptr);
    int settle_first_round();
    int adopt(const std::vector<Island> &wanted);
    int finish();
};

// (re)build the chunks of one island; its tile (and tail) segments become empty first
int IslandRun::build(size_t ii) {
    const Island &is = islands[ii];
    const uint32_t c = is.contig;
    const uint64_t L = b->h_len[c];
    const uint32_t nt = tile_first[c + 1] - tile_first[c];
    const uint32_t seg0 = tile_first[c] + c;
    // (a contig that is one tile may be longer than a tile core: clamped)
    uint32_t rng[2] = {seg0 + (uint32_t)std::min<uint64_t>(nt ? nt - 1 : 0, is.B / tc), seg0 + (uint32_t)std::min<uint64_t>(nt, (is.E + tc - 1) / tc)};
    if (is.E >= L) rng[1] = seg0 + nt + 1;  // including the tail segment
    // (a tile the island begins or ends inside keeps its entry -- its own elements in front of B / from E on are spliced with the
    // chunks' lists behind the last round, finish())
    if (is.cutB) ++rng[0];
    if (is.cutE) rng[1] = seg0 + (uint32_t)(is.E / tc);
    if (rng[0] < rng[1]) {
        zero_ranges.push_back(rng[0]);
        zero_ranges.push_back(rng[1]);
        zero_owner.push_back(ii);
    }
    // (round 3 kept 32 kbp chunks for islands around palindromic k-mers: their seams were corrected one per host round.  A
    // state now passes through chunks without pushes and through chunks a stuck machine cannot emit in, on the host)
    uint64_t CS = (is.pal && ctx->opt.no_island_relay) ? 32768 : is.pal ? CS_PAL : CS_SHORT;
    // (an island that begins / ends inside its tiles is an array of palindromic k-mers and the stretch behind it in which the machine
    // finds back: a seam between the two is passed by a stuck machine more often than not -- one chunk, 30-60 steps)
    // (islands built behind the first round's launch -- grown, merged, added by the tile kernel's flags -- run on the critical path,
    // where a round costs what its slowest chunk costs: the long chunks that suit a round beside the tile kernel do not suit them)
    if (critical && a.tile_pal && !ctx->opt.no_island_relay && ctx->opt.island_chunk_min <= 0) CS = std::min<uint64_t>(CS, 2816);
    if (is.pal && a.tile_pal && is.E - is.B <= 4096 && !ctx->opt.no_island_relay) CS = std::max<uint64_t>(CS, is.E - is.B);
    const uint64_t nch = is.whole ? 1 : (is.E - is.B + CS - 1) / CS;
    for (uint64_t j = 0; j < nch; ++j) {
        HChunk h;
        memset(&h.d, 0, sizeof(h.d));
        memset(&h.t_out, 0, sizeof(h.t_out));
        h.island = ii;
        h.d.contig = c;
        h.d.cs = is.whole ? 0 : is.B + j * CS;
        h.d.ce = is.whole ? L : std::min<uint64_t>(is.E, is.B + (j + 1) * CS);
        h.d.emit_lo_pos = (j == 0) ? is.B : 0;
        h.d.drain_end = h.d.ce;
        if (j + 1 == nch && is.E < L) h.d.drain_end = std::min<uint64_t>(L, is.E + 320);
        if (j + 1 == nch && is.E < L && is.cutE && is.ext_limit > is.E) h.d.ext_limit = is.ext_limit;
        h.d.seg = seg0 + (uint32_t)std::min<uint64_t>(nt ? nt - 1 : 0, h.d.cs / tc);
        h.d.warm = 256;
        // a long island is a long irregular stretch (a run of N, low-complexity sequence): every position emits there
        // (ties, shmmrutils.rs:516-527), the sparse region estimate would overflow and the chunk run twice
        // (so is an island around non-ACGT bytes, however short: it may be one of the two ends of a long gap)
        // (and an island around palindromic k-mers is low-complexity sequence more often than not: a genome-like batch ran a
        // second round for a dozen chunks that had overflowed the sparse estimate -- 0.4 ms for their 225 steps.  Only a whole
        // contig, whose region would be 12 bytes per base, starts with the estimate)
        h.full_cap = !is.whole;
        todo.push_back(ch.size());
        ch.push_back(h);
    }
    if (is.E < L) {  // probe: what a warmed-up (regular) machine looks like at E
        HChunk h;
        memset(&h.d, 0, sizeof(h.d));
        memset(&h.t_out, 0, sizeof(h.t_out));
        h.island = ii;
        h.probe = true;
        h.d.contig = c;
        h.d.cs = h.d.ce = h.d.drain_end = is.E;
        h.d.seg = 0xFFFFFFFFu;
        h.d.warm = 256;
        todo.push_back(ch.size());
        ch.push_back(h);
    }
    return PGR_OK;
}

// chunk descriptors up, the chunk kernel, states and counts down (into the pinned image): enqueued, nothing waits
int IslandRun::enqueue_round() {
    int rc;
    nq = todo.size();
    for (size_t q = 0; q < nq; ++q) {
        HChunk &h = ch[todo[q]];
        h.d.ring_out = h.probe ? 0xFFFFFFFFu : (uint32_t)todo[q];
        if (!h.d.override_state) h.d.ring_in = 0xFFFFFFFFu;
        h.d.region_off = next_region;
        h.d.region_cap = h.probe ? 1 : cap_of(h.d.drain_end - h.d.cs, h.full_cap) + (h.d.ext_limit ? 128 : 0);
        next_region += h.d.region_cap;
    }
    isl_lap("round: regions placed", round);
    if (st_chunks != st && ctx->ws_l1.cap < (next_region + 1) * sizeof(L1Rec)) st_chunks = st;  // (the buffer grows: in stream order)
    if ((rc = ctx->ws_l1.ensure_keep(ctx, (next_region + 1) * sizeof(L1Rec), st))) return rc;
    a.out = (L1Rec *)ctx->ws_l1.p;
    const hipStream_t sc = st_chunks;
    // ONE pinned block on the host: [descriptors | states at cs | states at ce | push info | status].  The chunk kernel reads its
    // descriptor from it and writes its states into it across PCIe (a hundred bytes per wavefront each way): a round is the
    // chunk kernel alone -- the upload kernel, the clearing of the states and the download kernel in front of and behind it
    // were 12-15 us of a round's 45-60 (three rounds for the repeat-rich contigs)
    desc_bytes = nq * sizeof(ChunkDesc);
    const size_t down_bytes = nq * (2 * sizeof(ChunkState) + 4 * sizeof(uint64_t) + sizeof(uint32_t));
    if ((rc = ctx->ensure_imail(desc_bytes + down_bytes))) return rc;
    uint8_t *h_img = (uint8_t *)ctx->imail;
    ChunkDesc *d_desc = (ChunkDesc *)h_img;
    ChunkState *d_in = (ChunkState *)(d_desc + nq);
    ChunkState *d_out = d_in + nq;
    uint64_t *d_info = (uint64_t *)(d_out + nq);
    uint32_t *d_stat = (uint32_t *)(d_info + 4 * nq);
    // (straight into the pinned image: ten thousand descriptors of a genome-sized batch are 1.6 MB -- a vector of them first, then a
    // copy, was 0.1 ms between the tile kernel's flags and the chunk kernel)
    for (size_t q = 0; q < nq; ++q) d_desc[q] = ch[todo[q]].d;
    if (ctx->opt.debug) descs.assign(d_desc, d_desc + nq);
    isl_lap("round: descriptors in the pinned image", round);
    d_zr.reset(new Tmp_list(ctx));
    if (!zero_ranges.empty()) {
        if ((rc = d_zr->alloc(zero_ranges.size() * sizeof(uint32_t)))) return rc;
        PGR_HIP(ctx, hipMemcpyAsync(d_zr->p, zero_ranges.data(), zero_ranges.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        launch_zero_seg_ranges(st, a, (const uint32_t *)d_zr->p, (uint32_t)(zero_ranges.size() / 2));
    }
    // one ring slot per chunk ever built (ids = indices into `ch`), kept across the rounds
    if ((rc = ctx->ws_flags.ensure_keep(ctx, ch.size() * CHUNK_RING_WORDS * sizeof(uint64_t), sc))) return rc;
    isl_lap("chunks listed, buffers ready", round);
    launch_level1_chunks(sc, a, d_desc, (uint32_t)nq, d_in, d_out, d_stat, (uint64_t *)ctx->ws_flags.p, d_info);
    isl_lap(sc == st ? "chunk kernel enqueued" : "chunk kernel enqueued (side stream)", round);
    enqueued = true;
    return PGR_OK;
}

// behind the synchronization: the states of the round's chunks are in the pinned image.  Verifies the seams; todo = the next round
int IslandRun::process_round() {
    int rc;
    enqueued = false;
    const size_t down_bytes = nq * (2 * sizeof(ChunkState) + 4 * sizeof(uint64_t) + sizeof(uint32_t));
    (void)down_bytes;
    uint8_t *h_img = (uint8_t *)ctx->imail;
    const ChunkState *r_in = (const ChunkState *)(h_img + desc_bytes), *r_out = r_in + nq;
    const uint64_t *r_info = (const uint64_t *)(r_out + nq);
    const uint32_t *r_stat = (const uint32_t *)(r_info + 4 * nq);
    PGR_HIP(ctx, hipGetLastError());
    zero_ranges.clear();
    zero_owner.clear();
    d_zr.reset();
    s_in.resize(ch.size());
    s_out.resize(ch.size());
    status.resize(ch.size());
    for (size_t q = 0; q < nq; ++q) {
        s_in[todo[q]] = r_in[q];
        s_out[todo[q]] = r_out[q];
        status[todo[q]] = r_stat[q];
        ch[todo[q]].n_push = r_info[4 * q];
        ch[todo[q]].bmin = r_info[4 * q + 1];
        ch[todo[q]].n_out = r_info[4 * q + 3] & ((1ull << 40) - 1);
        ch[todo[q]].dropped = false;
        const uint64_t moved = (r_info[4 * q + 3] >> 40) * 64;
        if (moved && !ch[todo[q]].probe) {
            // the island's end has moved on behind a stuck machine: the chunk, the island and the probe behind it say so
            HChunk &h = ch[todo[q]];
            h.d.ce += moved;
            h.d.drain_end += moved;
            islands[h.island].E = h.d.ce;
            const size_t i = todo[q];
            // (the probe ran beside the chunk, at the end that was planned -- in the usual case, nothing moved, that is the answer; here
            // it runs again at the new end, in the next round.  A second launch behind the round's chunks for the probes of such
            // islands, reading the new end on the device, cost every round ~20 us: repeat-rich contigs 0.415 -> 0.466 ms)
            if (i + 1 < ch.size() && ch[i + 1].probe && ch[i + 1].island == h.island && !ch[i + 1].retired) {
                ch[i + 1].d.cs = ch[i + 1].d.ce = ch[i + 1].d.drain_end = h.d.ce;
                ch[i + 1].moved = true;
                ch[i + 1].final = false;
            }
            ++n_moved;
        }
    }
    if (ctx->opt.debug && descs.size() == nq && nq) {
        uint32_t worst = 0;
        size_t wq = 0;
        uint64_t steps = 0;
        uint64_t worst_t = 0;
        for (size_t q = 0; q < nq; ++q) {
            steps += r_stat[q] >> 8;
            if ((r_info[4 * q + 2] & 0xFFFFFFFFull) > (worst_t & 0xFFFFFFFFull)) {
                worst_t = r_info[4 * q + 2];
                worst = r_stat[q] >> 8;
                wq = q;
            }
        }
        fprintf(stderr, "[pgr]   slowest chunk: %.1f us (%.1f us before its first step), warm %u override %u drain_end %llu\n",
                (worst_t & 0xFFFFFFFFull) / 100.0, (worst_t >> 32) / 100.0, descs[wq].warm,
                descs[wq].override_state, (unsigned long long)descs[wq].drain_end);
        fprintf(stderr, "[pgr] exact islands round %d: %zu chunks run, %zu islands, region end %llu; %llu steps of 64 positions, "
                "the longest chunk %u (chunk [%llu, %llu) of contig %u%s)\n", round, nq, islands.size(),
                (unsigned long long)next_region, (unsigned long long)steps, worst, (unsigned long long)descs[wq].cs,
                (unsigned long long)descs[wq].ce, descs[wq].contig, descs[wq].seg == 0xFFFFFFFFu ? ", a probe" : "");
    }
    // ---- verify seams (chunks of an island are contiguous in `ch`, the probe comes last).  A chunk is FINAL once the state
    // it started from is known to be the true one: the island's first chunk (regular by construction), a chunk whose
    // recorded state at cs equals the true state its final predecessor left at ce (the warm-up was right, or the state was
    // installed), and a chunk without a push in [cs, ce) behind a final predecessor -- the machine does not move there
    // (shmmrutils.rs:477-480: a skipped position touches neither ring nor mdist), so its end state is its predecessor's
    // with the k-mer rolled on, and its (empty) output is right whatever state it ran with.  Only a chunk with a final
    // predecessor is corrected, with that predecessor's true state and ring: a correction never builds on a stale state.
    std::vector<size_t> next;
    std::vector<size_t> rebuild;  // islands to rebuild (grown or turned into one whole-contig chunk)
    const bool relay = !ctx->opt.no_island_relay;
    for (size_t i = 0; i < ch.size(); ++i) {
        HChunk &h = ch[i];
        if (h.retired) continue;
        // (settled in an earlier round and nothing to run again: a round of a dozen chunks behind a genome-sized batch's ten thousand
        // used to walk all of them through the rules below)
        if (h.final && !(status[i] & 3u) && !(h.probe && h.moved)) continue;
        Island &is = islands[h.island];
        if (status[i] & 2u) {  // the true state could not be installed: the contig as one chunk
            if (!is.whole) {
                is.whole = true;
                is.B = 0;
                is.E = b->h_len[is.contig];
                is.cutB = is.cutE = false;
                is.ext_limit = 0;
                rebuild.push_back(h.island);
            }
            continue;
        }
        bool again = false;
        if (status[i] & 1u) {  // region overflow
            h.full_cap = true;
            again = true;
        }
        const bool has_prev = i > 0 && !ch[i - 1].retired && ch[i - 1].island == h.island;
        auto grow = [&]() {
            // the machine is not back in its regular regime at E: grow the island
            const uint64_t L = b->h_len[is.contig];
            if (is.cutE) {  // it ended inside a tile: through that tile and the next one (what an island of whole tiles starts with)
                is.E = std::min<uint64_t>(L, (is.E / tc + 2) * (uint64_t)tc);
                is.cutE = false;
                is.ext_limit = 0;
            } else {
                is.E = std::min<uint64_t>(L, is.E + 4ull * tc);
            }
            if (L - is.E < 2ull * tc) is.E = L;
            rebuild.push_back(h.island);
        };
        // A probe proves TWO things or the island grows: the exact machine's state at E is what a warm-up gives (so nothing in front
        // of the warm-up window reaches E), AND that state is the regular regime the closed form of the tile behind E assumes --
        // mdist within the window.  A stuck machine (mdist beyond w - 1: no rescan can fire, shmmrutils.rs:505-514) whose cause lies
        // INSIDE the warm-up window is reproduced by the probe's warm-up exactly, the states compare equal, and the tile behind E
        // would be computed as if a minimizer were due every w positions: found by fuzz_parity seed 7123218 (round 6) -- a palindromic
        // (AT)n stretch in the last tile of an island around a run of N, a tile the tile kernel skips and therefore never flags for
        // its palindromes; the machine stayed stuck for 540 positions into the next tile.
        auto stuck = [&](const ChunkState &t) { return !a.sketch && t.mdist > (uint64_t)(a.w - 1); };
        if (!relay) {  // the round-3 scheme (A/B): every seam against whatever the chunk in front produced last
            if (h.probe) {
                if (has_prev && (memcmp(&s_in[i], &s_out[i - 1], sizeof(ChunkState)) != 0 || stuck(s_out[i - 1]))) grow();
            } else if (!is.whole && h.d.cs > is.B && has_prev && memcmp(&s_in[i], &s_out[i - 1], sizeof(ChunkState)) != 0) {
                h.d.override_state = 1;
                h.d.in_state = s_out[i - 1];
                h.d.ring_in = (uint32_t)(i - 1);
                h.d.warm = 1024;
                again = true;
            }
            if (again) next.push_back(i);
            continue;
        }
        if (h.probe && h.moved) {  // what it recorded is the state at the OLD end
            h.moved = false;
            next.push_back(i);
        } else if (h.probe) {
            if (!has_prev) h.final = true;
            else if (ch[i - 1].final && !h.final) {
                if (memcmp(&s_in[i], &ch[i - 1].t_out, sizeof(ChunkState)) != 0 || stuck(ch[i - 1].t_out)) {
                    if (ctx->opt.debug) {
                        const ChunkState &p = s_in[i], &q = ch[i - 1].t_out;
                        fprintf(stderr, "[pgr] probe mismatch contig %u E=%llu: min_x %llx/%llx min_y %llx/%llx mdist %llu/%llu "
                                "F0 %llx/%llx R0 %llx/%llx sig %llx/%llx\n", is.contig, (unsigned long long)is.E,
                                (unsigned long long)p.min_x, (unsigned long long)q.min_x, (unsigned long long)p.min_y,
                                (unsigned long long)q.min_y, (unsigned long long)p.mdist, (unsigned long long)q.mdist,
                                (unsigned long long)p.F0, (unsigned long long)q.F0, (unsigned long long)p.R0,
                                (unsigned long long)q.R0, (unsigned long long)p.ring_sig, (unsigned long long)q.ring_sig);
                    }
                    grow();
                } else {
                    h.final = true;
                }
            }
        } else if (is.whole || !has_prev || h.d.cs <= is.B) {  // the island's first chunk
            h.final = true;
            h.t_out = s_out[i];
            h.ring_src = (uint32_t)i;
        } else if (ch[i - 1].final) {
            const HChunk &pv = ch[i - 1];
            const ChunkState &t = pv.t_out;
            const bool kmer_ok = s_in[i].F0 == t.F0 && s_in[i].F1 == t.F1 && s_in[i].R0 == t.R0 && s_in[i].R1 == t.R1;
            if ((status[i] & 4u) && kmer_ok) {  // no push in [cs, ce): the state passes through
                h.final = true;
                h.t_out = t;
                h.t_out.F0 = s_out[i].F0;
                h.t_out.F1 = s_out[i].F1;
                h.t_out.R0 = s_out[i].R0;
                h.t_out.R1 = s_out[i].R1;
                h.ring_src = pv.ring_src;
            } else if (memcmp(&s_in[i], &t, sizeof(ChunkState)) == 0) {
                h.final = true;
                h.t_out = s_out[i];
                h.ring_src = (uint32_t)i;
            } else if (kmer_ok && !a.sketch && t.mdist > (uint64_t)(a.w - 1) && h.n_push >= a.w && h.bmin > t.min_x &&
                       h.d.drain_end <= h.d.ce && !(status[i] & 1u)) {
                // The machine arrives STUCK: mdist is beyond w - 1 (a rescan measured the distance to a minimum from in
                // front of a stretch of skipped pushes, shmmrutils.rs:505-514), so no rescan can fire, and no push of this
                // chunk reaches down to min_mer (branch 2, :516-520) -- nothing is emitted, min_mer stays, mdist counts the
                // pushes, and with >= w pushes the ring at ce holds this chunk's own last w pushes: exactly what its run
                // from a warmed-up state left there.  Its output of that run is dropped.
                h.final = true;
                h.t_out = s_out[i];
                h.t_out.min_x = t.min_x;
                h.t_out.min_y = t.min_y;
                h.t_out.mdist = t.mdist + h.n_push;
                h.ring_src = (uint32_t)i;
                h.dropped = true;
            } else {
                if (ctx->opt.debug)
                    fprintf(stderr, "[pgr]   chunk %zu [%llu, %llu) of contig %u runs again from the true state: mdist %llu (warm-up %llu), "
                            "min_x %llx (%llx), %llu pushes, smallest branch-2 x %llx\n", i, (unsigned long long)h.d.cs,
                            (unsigned long long)h.d.ce, is.contig, (unsigned long long)t.mdist, (unsigned long long)s_in[i].mdist,
                            (unsigned long long)t.min_x, (unsigned long long)s_in[i].min_x, (unsigned long long)h.n_push,
                            (unsigned long long)h.bmin);
                h.final = false;
                h.d.override_state = 1;
                h.d.in_state = t;
                h.d.ring_in = pv.ring_src;  // the ring the last chunk with a push left at its end
                h.d.warm = 0;               // (ring, minimizer state and rolling k-mer all come from the chunk in front)
                again = true;
            }
        }  // else: the chunk in front is not settled yet
        if (again) next.push_back(i);
        // (a last chunk that may move the island's end runs again: so does the probe behind it, at wherever that run ends)
        if (again && h.d.ext_limit && i + 1 < ch.size() && ch[i + 1].probe && ch[i + 1].island == h.island && !ch[i + 1].retired) {
            ch[i + 1].final = false;
            ch[i + 1].moved = true;  // (pushed when the loop gets to it)
        }
    }
    if (!rebuild.empty()) {
        std::sort(rebuild.begin(), rebuild.end());
        rebuild.erase(std::unique(rebuild.begin(), rebuild.end()), rebuild.end());
        for (auto &h : ch)
            if (std::binary_search(rebuild.begin(), rebuild.end(), h.island)) h.retired = true;
        next.erase(std::remove_if(next.begin(), next.end(), [&](size_t i) { return ch[i].retired; }), next.end());
        todo.swap(next);
        for (size_t ii : rebuild) {
            // merge with later islands of the same contig that the grown island now touches
            for (size_t jj = 0; jj < islands.size(); ++jj)
                if (jj != ii && islands[jj].contig == islands[ii].contig && islands[jj].B < islands[ii].E + tc &&
                    islands[jj].B >= islands[ii].B && islands[jj].E > islands[ii].B && !islands[jj].whole &&
                    islands[jj].E != 0) {
                    if (islands[jj].E > islands[ii].E) {
                        islands[ii].E = islands[jj].E;
                        islands[ii].cutE = islands[jj].cutE;
                        islands[ii].ext_limit = islands[jj].ext_limit;
                    }
                    islands[ii].pal = islands[ii].pal || islands[jj].pal;
                    for (auto &h : ch)
                        if (h.island == jj) h.retired = true;
                    islands[jj].E = islands[jj].B = 0;  // absorbed
                }
            todo.erase(std::remove_if(todo.begin(), todo.end(), [&](size_t i) { return ch[i].retired; }), todo.end());
            if ((rc = build(ii))) return rc;
        }
    } else {
        todo.swap(next);
    }
    return PGR_OK;
}

int IslandRun::begin(hipStream_t side) {
    int rc;
    st_chunks = side ? side : st;
    t_isl0 = std::chrono::steady_clock::now();
    reserve_chunks(islands, 0);
    for (size_t ii = 0; ii < islands.size(); ++ii)
        if ((rc = build(ii))) return rc;
    round = 0;
    rc = todo.empty() ? PGR_OK : enqueue_round();
    critical = st_chunks != st;  // (a first round beside the tile kernel; otherwise every round is waited for and the constructor's sizes hold)
    return rc;
}

// The round that begin() enqueued is waited for and its seams are verified: afterwards nothing of this run is pending on the device
// and nothing of it is left in the pinned image (the tile flags may come down into it).
int IslandRun::settle_first_round() {
    int rc;
    if (!enqueued) return PGR_OK;
    if (st_chunks != st) {
        PGR_HIP(ctx, hipStreamSynchronize(st_chunks));
        st_chunks = st;
    }
    PGR_HIP(ctx, hipStreamSynchronize(st));
    isl_lap("states back on the host (early round kept)", round);
    if ((rc = process_round())) return rc;
    ++round;
    return PGR_OK;
}

// The tile kernel's flags have added islands (tiles with a palindromic k-mer): `wanted` is the full list, a superset of what this
// run was begun with.  An island of this run that `wanted` holds unchanged keeps its chunks and what they have computed; one that
// `wanted` has merged into a longer island, or that has changed since (grown, the whole contig), is retired; every other island of
// `wanted` is built.  The real case: a reference assembly has gaps AND (AT)n microsatellites longer than k in every batch -- the
// islands around the gaps have run beside the tile kernel, only those around the microsatellites run behind it.
int IslandRun::adopt(const std::vector<Island> &wanted) {
    int rc;
    std::vector<char> keep(islands.size(), 0), covered(wanted.size(), 0);
    // islands the first round has rebuilt (grown: the probe at E found the machine not yet regular; the whole contig): their new
    // segment range is waiting to be emptied.  Such an island stays when every island of `wanted` it touches lies inside it
    std::vector<char> changed(islands.size(), 0);
    for (size_t o : zero_owner)
        if (o != SIZE_MAX) changed[o] = 1;
    for (size_t i = 0; i < islands.size(); ++i) {
        const Island &is = islands[i];
        if (!(changed[i] || is.whole) || is.E <= is.B) continue;
        bool inside = true;
        for (size_t j = 0; j < wanted.size() && inside; ++j) {
            const Island &wn = wanted[j];
            if (wn.contig != is.contig || !(wn.B < is.E + tc && wn.E + tc > is.B)) continue;
            inside = wn.B >= is.B && wn.E <= is.E && !covered[j];
        }
        if (!inside) continue;
        keep[i] = 1;
        for (size_t j = 0; j < wanted.size(); ++j) {
            const Island &wn = wanted[j];
            if (wn.contig == is.contig && wn.B < is.E + tc && wn.E + tc > is.B) {
                covered[j] = 1;
                islands[i].pal = islands[i].pal || wn.pal;
            }
        }
    }
    {
        // both lists are in (contig, B) order as list_islands makes them (this run's: the islands it was begun with, minus those the
        // first round has changed): one walk over the two.  (A std::map of this run's 2 461 islands and 4 857 look-ups was 0.26 ms
        // between the tile kernel's flags and the chunk kernel of a genome-sized batch.)
        std::vector<size_t> mine;
        for (size_t i = 0; i < islands.size(); ++i)
            if (!changed[i] && !islands[i].whole && islands[i].E > islands[i].B) mine.push_back(i);
        auto before = [&](size_t x, size_t y) {
            return islands[x].contig != islands[y].contig ? islands[x].contig < islands[y].contig : islands[x].B < islands[y].B;
        };
        if (!std::is_sorted(mine.begin(), mine.end(), before)) std::sort(mine.begin(), mine.end(), before);
        bool wanted_sorted = true;
        for (size_t j = 1; j < wanted.size() && wanted_sorted; ++j)
            wanted_sorted = wanted[j - 1].contig != wanted[j].contig ? wanted[j - 1].contig < wanted[j].contig : wanted[j - 1].B <= wanted[j].B;
        size_t p = 0;
        for (size_t j = 0; j < wanted.size(); ++j) {
            const Island &wn = wanted[j];
            if (covered[j]) continue;
            if (!wanted_sorted) p = 0;  // (never the case with list_islands' output: correct all the same)
            while (p < mine.size() && (islands[mine[p]].contig != wn.contig ? islands[mine[p]].contig < wn.contig : islands[mine[p]].B < wn.B)) ++p;
            if (p < mine.size() && islands[mine[p]].contig == wn.contig && islands[mine[p]].B == wn.B && !wn.whole &&
                islands[mine[p]].E == wn.E && islands[mine[p]].cutB == wn.cutB && islands[mine[p]].cutE == wn.cutE && !keep[mine[p]]) {
                keep[mine[p]] = 1;
                covered[j] = 1;
                islands[mine[p]].pal = islands[mine[p]].pal || wn.pal;
            }
        }
    }
    isl_lap("adopt: islands matched", round);
    for (auto &h : ch)
        if (!keep[h.island]) h.retired = true;
    todo.erase(std::remove_if(todo.begin(), todo.end(), [&](size_t i) { return ch[i].retired; }), todo.end());
    {  // (the segment range of a rebuilt island that goes: not emptied -- the longer list may not cover all of it)
        size_t kept = 0;
        for (size_t q = 0; q < zero_owner.size(); ++q)
            if (zero_owner[q] == SIZE_MAX || keep[zero_owner[q]]) {
                zero_ranges[2 * kept] = zero_ranges[2 * q];
                zero_ranges[2 * kept + 1] = zero_ranges[2 * q + 1];
                zero_owner[kept++] = zero_owner[q];
            }
        zero_ranges.resize(2 * kept);
        zero_owner.resize(kept);
    }
    for (size_t i = 0; i < keep.size(); ++i)
        if (!keep[i]) islands[i].E = islands[i].B = 0;  // absorbed
    // the islands the flags have added are on the critical path (nothing runs beside them): chunks short enough that the
    // slowest one takes about a third of what the round's steps take on the whole chip (~1.5 us of issue per step of 64
    // positions on 1024 SIMDs; a lone wavefront ~3 us per step), so that a second round -- one chunk deep -- stays cheap
    uint64_t fresh_bases = 0;
    for (size_t j = 0; j < wanted.size(); ++j)
        if (!covered[j]) fresh_bases += wanted[j].E - wanted[j].B;
    if (ctx->opt.island_chunk_min <= 0) CS_PAL = pal_chunk(fresh_bases, 512);
    // (islands that begin and end inside their tiles are one chunk each, build(); what is longer -- groups of flagged tiles, islands
    // merged with those around non-ACGT bytes -- is cut so that the round's slowest chunk is not three times its typical one)
    if (ctx->opt.island_chunk_min <= 0 && a.tile_pal) CS_PAL = std::min<uint64_t>(CS_PAL, 2816);
    {
        std::vector<Island> fresh;
        for (size_t j = 0; j < wanted.size(); ++j)
            if (!covered[j]) fresh.push_back(wanted[j]);
        reserve_chunks(fresh, ch.size());
    }
    for (size_t j = 0; j < wanted.size(); ++j) {
        if (covered[j]) continue;
        islands.push_back(wanted[j]);
        if ((rc = build(islands.size() - 1))) return rc;
    }
    isl_lap("adopt: new islands built", round);
    return PGR_OK;
}

int IslandRun::finish() {
    int rc;
    for (; !todo.empty(); ++round) {
        // Every round settles at least one seam or grows / merges an island, so the number of rounds is bounded by the number
        // of chunks plus the growth steps; in practice it is 1-3: a state is handed through chunks that cannot change it on the
        // host (see the verification in process_round), and only a chunk whose predecessor's state is final runs again.
        if (round > 1024 + 4 * (int)ch.size()) return ctx->fail(PGR_ERR_INTERNAL, "exact-machine islands did not converge");
        if (!enqueued && (rc = enqueue_round())) return rc;
        if (st_chunks != st) {
            PGR_HIP(ctx, hipStreamSynchronize(st_chunks));
            st_chunks = st;  // (the rounds behind the first: in stream order)
        }
        PGR_HIP(ctx, hipStreamSynchronize(st));
        isl_lap("states back on the host", round);
        if ((rc = process_round())) return rc;
    }
    if (!zero_ranges.empty()) {  // (segment ranges of islands built in the last round: none in practice)
        Tmp_list d_zr(ctx);
        if ((rc = d_zr.alloc(zero_ranges.size() * sizeof(uint32_t)))) return rc;
        PGR_HIP(ctx, hipMemcpyAsync(d_zr.p, zero_ranges.data(), zero_ranges.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        launch_zero_seg_ranges(st, a, (const uint32_t *)d_zr.p, (uint32_t)(zero_ranges.size() / 2));
        PGR_HIP(ctx, hipStreamSynchronize(st));
    }
    isl_lap("seams verified", -1);
    // ---- the lists of the chunks that start in one tile become that tile's segment: copied back to back into a fresh region
    // (the chunks of an island are contiguous in `ch`, in position order; their counts came back with the states).
    // A tile an island begins or ends INSIDE (Island::cutB / cutE) keeps its own elements below B / from E on: its entry is a SPLICE
    // -- room for that prefix in front of the chunks' lists (one element per position of [tile start, B) at most: the tile kernel
    // selects a position once) and for the suffix behind them; splice_segs_kernel finds the two split points in the tile's own
    // segment (in position order) and copies the prefix right-aligned against the chunks' lists, so the segment is contiguous.
    {
        constexpr uint64_t NONE = ~0ull;
        std::vector<uint64_t> img;  // copies (3 words each), then segment entries (3 words each), then splices (6 words each)
        std::vector<uint64_t> segs, splices;
        img.reserve(3 * ch.size() + 9 * islands.size() + 64);
        segs.reserve(3 * ch.size() + 64);
        splices.reserve(12 * islands.size() + 64);
        struct Open {
            uint32_t seg = 0xFFFFFFFFu, contig = 0;
            uint64_t off = 0, cnt = 0, lo = NONE, hi = NONE, sfx_room = 0;
            size_t island = SIZE_MAX, img_start = 0;  // (img_start: where this segment's copies begin in img)
        } cur;
        bool cur_open = false, end_seg_done = false;
        size_t cur_island = SIZE_MAX;
        auto seg0_of = [&](uint32_t c) { return tile_first[c] + c; };
        auto close_seg = [&]() {
            if (!cur_open) return;
            cur_open = false;
            if (cur.lo == NONE && cur.hi == NONE) {
                // ONE chunk's list is the whole segment (the usual case for the islands around runs of non-ACGT bytes, whose chunks are
                // longer than a tile): the entry points at the chunk's region, nothing is copied (the copies of a genome-like batch
                // were 160 MB through one wavefront per chunk, 65 us between the last round and the list stage)
                uint64_t off = cur.off, cnt = cur.cnt;
                if (img.size() - cur.img_start == 3) {
                    off = img[cur.img_start];
                    img.resize(cur.img_start);
                    ++n_in_place;
                }
                segs.push_back((uint64_t)cur.seg | ((uint64_t)cur.contig << 32));
                segs.push_back(off);
                segs.push_back(cnt);
            } else {
                splices.push_back((uint64_t)cur.seg | ((uint64_t)cur.contig << 32));
                splices.push_back(cur.off);
                splices.push_back(cur.cnt);
                splices.push_back(cur.lo);
                splices.push_back(cur.hi);
                splices.push_back(cur.sfx_room);
                next_region += cur.sfx_room;
            }
        };
        auto open_seg = [&](uint32_t seg, const Island &is) {
            close_seg();
            cur = Open();
            cur.seg = seg;
            cur.contig = is.contig;
            const uint32_t s0 = seg0_of(is.contig);
            if (is.cutB && seg == s0 + (uint32_t)(is.B / tc)) {
                cur.lo = is.B;
                next_region += is.B - (uint64_t)(seg - s0) * tc + 64;  // room for the tile's own elements below B
            }
            if (is.cutE && seg == s0 + (uint32_t)(is.E / tc)) {
                cur.hi = is.E;
                cur.sfx_room = (uint64_t)(seg - s0 + 1) * tc - is.E + 64;
                end_seg_done = true;
            }
            cur.off = next_region;
            cur.img_start = img.size();
            cur_open = true;
        };
        auto close_island = [&]() {
            if (cur_island == SIZE_MAX) return;
            const Island &is = islands[cur_island];
            // (no chunk starts in the tile the island ends in: that tile's entry is its own suffix alone)
            if (is.cutE && is.E > is.B) {
                // tiles between the last one a chunk starts in and the one the island ends in: wholly the island's, and empty -- an end
                // that has moved on into the next tile leaves the tile it was to end in with its own entry (build() kept it)
                const uint32_t seg_e = seg0_of(is.contig) + (uint32_t)(is.E / tc);
                const uint32_t last = cur_open ? cur.seg : seg_e;
                close_seg();
                for (uint32_t sg = last + 1; sg < seg_e; ++sg) {
                    segs.push_back((uint64_t)sg | ((uint64_t)is.contig << 32));
                    segs.push_back(next_region);
                    segs.push_back(0);
                }
                if (!end_seg_done) open_seg(seg_e, is);
            }
            close_seg();
            cur_island = SIZE_MAX;
        };
        for (const HChunk &h : ch) {
            if (h.retired || h.probe || h.d.seg == 0xFFFFFFFFu) continue;
            if (h.island != cur_island) {
                close_island();
                cur_island = h.island;
                end_seg_done = false;
            }
            if (!cur_open || h.d.seg != cur.seg) open_seg(h.d.seg, islands[h.island]);
            if (h.dropped || h.n_out == 0) continue;
            img.push_back(h.d.region_off);
            img.push_back(next_region);
            img.push_back(h.n_out);
            cur.cnt += h.n_out;
            next_region += h.n_out;
        }
        close_island();
        const size_t n_copies = img.size() / 3, n_set = segs.size() / 3, n_splice = splices.size() / 6;
        if (n_set || n_splice) {
            for (size_t i = 0; i < n_set; ++i)
                if (segs[3 * i + 2] > 0xFFFFFFFFull) return ctx->fail(PGR_ERR_INTERNAL, "a tile's exact list exceeds 2^32 elements");
            for (size_t i = 0; i < n_splice; ++i)
                if (splices[6 * i + 2] > 0x7FFFFFFFull) return ctx->fail(PGR_ERR_INTERNAL, "a tile's exact list exceeds 2^31 elements");
            img.insert(img.end(), segs.begin(), segs.end());
            img.insert(img.end(), splices.begin(), splices.end());
            const size_t bytes = img.size() * sizeof(uint64_t);
            if ((rc = ctx->ws_l1.ensure_keep(ctx, (next_region + 1) * sizeof(L1Rec), st)) || (rc = ctx->ws_serial.ensure(ctx, bytes)) ||
                (rc = ctx->ensure_imail(bytes)))
                return rc;
            a.out = (L1Rec *)ctx->ws_l1.p;
            memcpy(ctx->imail, img.data(), bytes);  // (pinned, and untouched until this context's next island call: no wait here)
            PGR_HIP(ctx, hipMemcpyAsync(ctx->ws_serial.p, ctx->imail, bytes, hipMemcpyHostToDevice, st));
            const uint64_t *d_img = (const uint64_t *)ctx->ws_serial.p;
            launch_assemble_chunks(st, a, d_img, (uint32_t)n_copies, d_img + 3 * n_copies, (uint32_t)n_set);
            if (n_splice) launch_splice_segs(st, a, d_img + 3 * n_copies + 3 * n_set, (uint32_t)n_splice);
            n_spliced = n_splice;
        }
    }
    if (ctx->opt.debug && (n_spliced || n_moved))
        fprintf(stderr, "[pgr] islands: %zu tiles keep elements of their own beside an island's, %zu island ends moved on behind a stuck machine\n", n_spliced, n_moved);
    isl_lap("tile lists assembled (enqueued)", -1);
    return PGR_OK;
}

static int run_exact_islands(pgr_ctx *ctx, hipStream_t st, const pgr_batch *b, L1Args &a, std::vector<Island> &islands,
                             const std::vector<uint32_t> &tile_first, uint32_t tc, uint64_t region_base,
                             const std::vector<uint32_t> &empty_seg_ranges) {
    IslandRun run(ctx, st, b, a, islands, tile_first, tc, region_base, empty_seg_ranges);
    int rc;
    if ((rc = run.begin()) || (rc = run.finish())) return rc;
    a.out = run.a.out;
    islands = run.islands;
    return PGR_OK;
}

// One pass of the hot path over a resident batch of SHORT contigs (query batches, reads, fragmented assemblies): the
// one-workgroup-per-contig kernel of csrc/small.hip replaces tiles + tails + segment scans + the fused list kernel -- 4 launches
// and one synchronization instead of ~15 dependent operations.  handled == false: not eligible, or a contig was handed back
// (non-ACGT byte, palindromic k-mer, low-complexity list): the caller runs the general pipeline.
static int shmmrs_compute_small(pgr_ctx *ctx, const pgr_batch *b, const pgr_spec *spec, const uint32_t *rids, pgr_shmmrs **out,
                                bool &handled) {
    handled = false;
    const uint32_t n = b->n;
    // Small batches only: the kernel trades throughput for latency (a workgroup walks its contig serially: tiles, then the tail
    // on one wavefront, then the list stage between barriers).  Measured on 10 kbp contigs: 0.045 ms + 58 ns per contig against
    // 0.13 ms + 34 ns per contig for the general pipeline -- the lines cross near 3500 contigs (35 Mbp).
    if (n == 0 || n > SMALL_MAX_CONTIGS || b->total_bases > SMALL_MAX_BASES || spec->sketch || spec->w < (uint32_t)L1_MIN_W ||
        b->host_saw_invalid || ctx->opt.no_small_path)
        return PGR_OK;
    if (ctx->skip_small_once) {  // shmmr_batch_small has just run this kernel on these contigs and was handed them back
        ctx->skip_small_once = false;
        return PGR_OK;
    }
    uint32_t max_len = 0;
    uint64_t total_slots = 0;
    for (uint32_t c = 0; c < n; ++c) {
        if (b->h_len[c] > SMALL_MAX_LEN) return PGR_OK;
        max_len = std::max(max_len, b->h_len[c]);
        total_slots += b->h_len[c] / 32 + 64;
    }
    if (total_slots >= (1ull << 32)) return PGR_OK;
    hipStream_t st = ctx->stream;
    std::vector<SmallContig> &desc = ctx->keep_small_desc;  // source of an async H2D copy: lives in the context
    desc.resize(n);
    uint64_t s_off = 0;
    for (uint32_t c = 0; c < n; ++c) {
        desc[c].word_off = b->h_word_off[c];
        desc[c].len = b->h_len[c];
        desc[c].rid = rids ? rids[c] : c;
        desc[c].out_off = (uint32_t)s_off;
        desc[c].out_cap = b->h_len[c] / 32 + 64;
        s_off += desc[c].out_cap;
    }
    constexpr size_t N_STATUS = 10;  // (same result layout as the general path: status words in front of the offsets)
    int rc;
    if ((rc = ctx->ws_small_desc.ensure(ctx, (size_t)n * sizeof(SmallContig))) ||
        (rc = ctx->ws_small_cnt.ensure(ctx, 2 * ((size_t)n + 1) * sizeof(uint32_t))) ||
        (rc = ctx->ws_list_a.ensure(ctx, (size_t)total_slots * sizeof(pgr_mm128))) ||
        (rc = ctx->ws_scan_tmp.ensure(ctx, scan_counts_temp_bytes(n + 1))) ||
        (rc = ctx->ensure_mailbox(((size_t)n + 2) * sizeof(uint64_t))))
        return rc;
    uint32_t *d_counts = (uint32_t *)ctx->ws_small_cnt.p, *d_clean = d_counts + (n + 1);
    pgr_shmmrs *res = new pgr_shmmrs();
    res->ctx = ctx;
    res->n = n;
    auto bail = [&](int code) {
        pgr_shmmrs_destroy(res);
        return code;
    };
    if ((rc = ctx->dmalloc((void **)&res->d_block, (N_STATUS + (size_t)n + 1) * sizeof(uint64_t)))) return bail(rc);
    res->d_off = res->d_block + N_STATUS;
    const double dens = 2.0 / (double)(spec->w + 1);
    const double spec_key = (double)spec->w * 1e9 + spec->k * 1e6 + spec->r * 1e4 + spec->min_span;
    const double ratio = (ctx->est_spec_key == spec_key && ctx->est_final_ratio > 0) ? ctx->est_final_ratio * 1.15 : dens / 3.0 + 1e-4;
    uint64_t cap_res = std::max<uint64_t>((uint64_t)((double)b->total_bases * ratio) + 64ull * n + 1024, 16);
    uint64_t *mbox = (uint64_t *)ctx->mailbox;
    hipError_t e = hipEventRecord(ctx->ev[0], st);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->ws_small_desc.p, desc.data(), (size_t)n * sizeof(SmallContig), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_counts + n, 0, sizeof(uint32_t), st);  // the fallback flag word
    if (e != hipSuccess) return bail(ctx->fail(PGR_ERR_DEVICE, std::string("small path set-up: ") + hipGetErrorString(e)));
    SmallArgs a;
    a.planes = b->d.planes;
    a.valid = b->d.valid;
    a.l1_cap = small_l1_cap(max_len, spec->w);
    a.desc = (const SmallContig *)ctx->ws_small_desc.p;
    a.n = n;
    a.w = spec->w;
    a.k = spec->k;
    a.r = spec->r;
    a.min_span = spec->min_span;
    a.tc = ((L1_EXT - 2 * (spec->w - 1)) / 64) * 64;
    a.out = (pgr_mm128 *)ctx->ws_list_a.p;
    a.counts = d_counts;
    a.flags = d_counts + n;
    launch_small_shmmr(st, a);
    launch_small_counts(st, d_counts, n, d_clean);
    if (scan_counts(st, ctx->ws_scan_tmp.p, scan_counts_temp_bytes(n + 1), d_clean, res->d_off, n + 1) != hipSuccess)
        return bail(ctx->fail(PGR_ERR_DEVICE, "scan failed"));
    for (int attempt = 0;; ++attempt) {
        ctx->dfree(res->d_mm);
        res->d_mm = nullptr;
        if ((rc = ctx->dmalloc((void **)&res->d_mm, cap_res * sizeof(pgr_mm128)))) return bail(rc);
        launch_small_gather(st, (const pgr_mm128 *)ctx->ws_list_a.p, a.desc, d_counts, res->d_off, n, res->d_mm, cap_res);
        e = hipEventRecord(ctx->ev_end, st);
        // (a consumer that does not wait for the host -- the query path -- enqueues its kernels here, see pgr_shmmrs_compute)
        if (e == hipSuccess && ctx->post_enqueue && (rc = ctx->post_enqueue(res->d_mm, res->d_off, cap_res, res->d_off + n)))
            return bail(rc);
        if (e == hipSuccess) e = hipMemcpyAsync(mbox, res->d_off, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(mbox + n + 1, d_counts + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) return bail(ctx->fail(PGR_ERR_DEVICE, std::string("small path: ") + hipGetErrorString(e)));
        if ((uint32_t)mbox[n + 1]) {  // a contig needs the general pipeline
            pgr_shmmrs_destroy(res);
            return PGR_OK;
        }
        if (mbox[n] <= cap_res || attempt > 0) break;
        cap_res = mbox[n] + 16;  // more survivors than estimated: gather again into a bigger buffer
    }
    res->h_off.assign(mbox, mbox + n + 1);
    res->count = mbox[n];
    res->rid_is_index = rids == nullptr;
    ctx->staged_unsynced = false;
    ctx->want_host_copy = false;
    if (b->total_bases) {
        ctx->est_spec_key = spec_key;
        ctx->est_final_ratio = (double)res->count / (double)b->total_bases;
    }
    pgr_prof prof;
    memset(&prof, 0, sizeof(prof));
    (void)hipEventElapsedTime(&prof.total_ms, ctx->ev[0], ctx->ev_end);
    prof.level1_ms = prof.total_ms;  // (one kernel does levels 1 and 2)
    prof.bases_tiled = b->total_bases;
    prof.n_tiles = n;
    ctx->prof = prof;
    *out = res;
    handled = true;
    return PGR_OK;
}

// ------------------------------------------------------------------------------------------------
// One pass of the hot path over a resident batch (ShmmrJob).  Everything is enqueued with sizes that are upper bounds or
// estimates; the host reads the true counts ONCE at the end of a pass (one wait per pass in the common case) and repeats a
// stage only when an estimate turned out too small:
//   stage 1  level-1 tiles + tails (+ exact islands when a tile flagged a palindromic k-mer / non-ACGT byte)      front stream
//   stage 2  scan of the segment counts (the level-1 total stays on the device)                                   back stream
//   stage 3  fused reduce x2 + min_span (grid = upper bound, surplus workgroups exit on the device-side total)
//   stage 4  scan of the block counts, ordered gather into the result, per-contig offsets, rid patch, the consumer's kernels
// The synchronous driver (pgr_shmmrs_compute) runs both halves on the context's stream and waits; batches of >= 1 Gbp look at
// the level-1 flags once before the list stage is enqueued (a 45 us round trip is nothing there and a flagged batch does not
// run stages 2-4 twice); smaller ones (the reference's real callers: <= 129 contigs per call, seq_db.rs:561, and single
// queries, ext.rs:252) run optimistically and redo stages 2-4 after the islands in the rare flagged case.
// The pipelined driver (pgr_pipe_*) enqueues the first pass of a job optimistically -- stage 1 on the context's stream, stages
// 2-4 behind an event on the back stream -- and returns; the counts are looked at when the job is collected, and a job that
// needs another pass gets it then, synchronously.
namespace {

constexpr size_t N_CURSOR = 8;  // [0..2] level 1 (L1Args::cursor), [4..5] fused list stage
constexpr size_t N_STATUS = 10;
constexpr int JOB_RESTART = 1;  // enqueue_pass: stage 1 overflowed its region, run the pass again

struct ShmmrJob {
    pgr_ctx *ctx = nullptr;
    const pgr_batch *b = nullptr;
    pgr_spec spec = {};
    const uint32_t *rids = nullptr;
    int padding = 0;
    hipStream_t sf = nullptr, sb = nullptr;  // front (level 1) and back (list stage) streams; the same for a synchronous call
    hipEvent_t ev_front = nullptr;           // sf != sb: stage 1 done (the back stream waits for it)
    bool optimistic = false;                 // the first pass of a pipelined job: no look at the flags between the stages
    // ... of a job whose predecessor (same spec, same context) needed islands -- a genome comes as many similar batches, and a real
    // assembly's are flagged every time --: stage 1 alone.  Its list stage would be thrown away at collect, and until then it runs
    // beside the other job's tile kernel and takes its slots one for one (0.5 ms of a 2 Gbp batch's 5.6).  collect looks at the
    // flags and enqueues the islands and THE list stage (or, for once not flagged, the list stage alone).
    bool stage1_only = false;
    bool list_pending = false;  // decide(): the pass that just ended had no list stage
    bool no_islands_pass = false;  // stage 1 for a consumer that declines what the tile kernel flags (the query path's level-1 form): a batch
                                   // the host packer found clean needs no look at its validity plane
    uint32_t lds_match = 0;                  // sf != sb: LDS per workgroup of the tile kernel; the back stream's kernels take as much or none
    // a consumer of the result that does not want to wait for the host: its kernels go behind stage 4 (stream, list, offsets,
    // capacity of the list, device address of the true count); a repeated pass calls it again
    std::function<int(hipStream_t, const pgr_mm128 *, const uint64_t *, uint64_t, const uint64_t *)> post;

    bool dbg_t = false;  // host-side timeline of the call on stderr
    std::chrono::steady_clock::time_point dbg_t0;
    void dbg_lap(const char *what) const {
        if (dbg_t)
            fprintf(stderr, "[pgr] shmmrs_compute %-28s at %7.1f us\n", what,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t0).count());
    }

    // ---- the plan
    uint32_t n = 0;
    bool sketch = false, tiled = false, short_tiles = false;
    uint32_t w_eff = 0, ext = 0, tc = 0;
    std::vector<uint32_t> serial;  // contigs that go through the exact machine as a whole (w < 17)
    uint64_t bases_tiled = 0;
    uint32_t n_tiles = 0, n_segs = 0;
    double dens = 0.0, l1_key = 0.0, spec_key = 0.0;
    uint32_t slot = 0;
    uint64_t slots_total = 0, cap_par = 0;
    unsigned long long *d_cursor = nullptr;
    uint32_t *d_cflags = nullptr;
    uint8_t *d_tflags = nullptr;
    uint16_t *d_tpal = nullptr;
    size_t pal_off = 0;     // of d_tpal behind d_tflags, bytes
    bool sub_tile = false;  // islands may begin / end inside tiles flagged for a palindromic k-mer only
    size_t zero_bytes = 0;
    uint64_t *mbox = nullptr;  // pinned: [0, N_STATUS) status, then the n + 1 result offsets
    uint32_t *d_rids = nullptr;
    L1Args a;
    pgr_prof prof;
    bool early_sync = false, pad_fix = false, do_reduce = false;
    uint32_t halo = 0, slot2 = 0;
    uint32_t fb = FUSED_BLOCK_ELEMS;  // list elements per workgroup of the fused kernel (512 for a pipelined job: see level2.hip)
    uint64_t serial_base = 0;  // first element of the serial regions inside the level-1 buffer
    bool islands_done = false;
    bool l2_cursor_clean = false;  // the list stage's cursor words were cleared by stage 1's memset
    // Islands around non-ACGT bytes, listed while the tile kernel runs: a batch the host packer has counted such bytes in gets the
    // tile flags (all but the palindrome bit, which the tile kernel sets) on the host as soon as mark_invalid_tiles has run -- the
    // copy, the listing (and the cumulative last-valid table) used to sit between the tile kernel and the chunk kernel, 0.15 ms
    // of a chromosome-like contig's 1.2.  Used when the tile kernel reports no palindromic k-mer; otherwise listed again.
    std::vector<Island> pre_islands;
    std::vector<uint32_t> pre_gap_segs;
    bool pre_listed = false;
    bool flags_prefetched = false;  // contig flags, tile flags and the contigs' non-ACGT counts came down with the status words
    std::unique_ptr<IslandRun> early_islands;  // round 0 of the pre-listed islands, enqueued behind the tile kernel (stage1)
    // ---- the list stage and the result
    pgr_shmmrs *res = nullptr;
    uint32_t n_blocks = 0;  // grid of the fused kernel
    uint64_t cap2 = 0;      // its overflow region
    uint64_t cap_res = 0;   // result capacity
    uint64_t res_alloc_elems = 0;  // elements res->d_mm was allocated for
    pgr_mm128 *d_list = nullptr;  // ordered final list (before the padding artefact)
    uint64_t *d_loff = nullptr;
    uint64_t *d_total1 = nullptr;
    std::vector<uint64_t> l1_off;  // only needed for the padding artefact
    bool scanned4 = false;
    uint64_t host_copy_elems = 0;
    // ---- the driver's state
    int from = 1;  // first stage of the next pass
    int attempt = 0;
    uint64_t n_final = 0;
    uint64_t l1_alloc_seen = 0;  // elements the level-1 kernels took from the overflow region
    uint64_t l2_alloc_seen = 0;  // ... the list kernel from its overflow region

    std::vector<uint32_t> &tile_first() const { return ctx->h_tile_first; }
    ~ShmmrJob() {
        if (res) pgr_shmmrs_destroy(res);
    }
    int plan();
    void list_islands(const uint32_t *flags, const uint32_t *n_invalid, uint8_t *tf, const uint16_t *pal, std::vector<Island> &islands,
                      std::vector<uint32_t> &gap_segs, uint32_t cut_margin = 0, uint32_t cut_settle = 0);
    int stage1();
    int run_islands(uint64_t need_word);
    int begin_result();
    int scan_back(const uint32_t *in, uint64_t *out, uint32_t n_plus_1);
    int stage2();
    int stage3();
    int stage4();
    int enqueue_pass();
    int decide(bool &done);
    int finish(pgr_shmmrs **out);
    int run_sync(pgr_shmmrs **out);
};

// host plan (tiles for the closed form, list for the serial kernel), workspaces, the tile table on its way to the device
int ShmmrJob::plan() {
    hipStream_t st = sf;
    n = b->n;
    sketch = spec.sketch != 0;
    tiled = sketch || spec.w >= (uint32_t)L1_MIN_W;
    w_eff = sketch ? 1u : spec.w;
    // tile core: extended tile minus both halos, rounded down to the 64-position step of the exact kernel so
    // that islands of exact tiles start and end on step boundaries
    // (batches of short contigs -- reads --: tiles of one wavefront's 1024 positions, pgr_internal.h)
    short_tiles = tiled && n && b->total_bases / n <= (uint64_t)L1_SHORT_MEAN_LEN && w_eff <= (uint32_t)L1_SHORT_MAX_W &&
                  !ctx->opt.no_short_tiles;
    ext = short_tiles ? (uint32_t)L1_EXT_SHORT : (uint32_t)L1_EXT;
    tc = ((ext - 2 * (w_eff - 1)) / 64) * 64;

    // (The table lives in the context: a million reads are 4 MB that a fresh vector would fault in page by page on every
    // call; big batches fill it on the pool's threads.)
    std::vector<uint32_t> &tf = tile_first();
    tf.resize((size_t)n + 1);
    serial.clear();
    uint64_t n_tiles64 = 0;
    bases_tiled = 0;
    if (tiled && n >= (1u << 17)) {
        constexpr uint32_t PIECE = 1u << 14;
        const uint32_t n_pieces = (n + PIECE - 1) / PIECE;
        std::vector<uint64_t> piece_tiles((size_t)n_pieces + 1, 0);
        HostPool::instance().parallel_for(n_pieces, [&](size_t p) {
            const uint32_t c0 = (uint32_t)p * PIECE, c1 = std::min<uint32_t>(n, c0 + PIECE);
            uint64_t t = 0;
            for (uint32_t c = c0; c < c1; ++c) t += l1_tiles_of(b->h_len[c], tc, ext);
            piece_tiles[p + 1] = t;
        });
        for (uint32_t p = 0; p < n_pieces; ++p) piece_tiles[p + 1] += piece_tiles[p];
        n_tiles64 = piece_tiles[n_pieces];
        bases_tiled = b->total_bases;
        if (n_tiles64 + n + 1 < (1ull << 31))
            HostPool::instance().parallel_for(n_pieces, [&](size_t p) {
                const uint32_t c0 = (uint32_t)p * PIECE, c1 = std::min<uint32_t>(n, c0 + PIECE);
                uint64_t t = piece_tiles[p];
                for (uint32_t c = c0; c < c1; ++c) {
                    tf[c] = (uint32_t)t;
                    t += l1_tiles_of(b->h_len[c], tc, ext);
                }
            });
    } else {
        for (uint32_t c = 0; c < n; ++c) {
            tf[c] = (uint32_t)n_tiles64;
            const uint64_t L = b->h_len[c];
            if (L == 0) continue;
            n_tiles64 += l1_tiles_of(L, tc, ext);  // every contig owns tile segments (the chunk kernel reuses them)
            if (tiled) bases_tiled += L;  // contigs with non-ACGT bytes too: only islands around them are replaced
            else serial.push_back(c);     // w < 17: the whole contig goes through the exact kernel
        }
    }
    if (n_tiles64 + n + 1 >= (1ull << 31)) return ctx->fail(PGR_ERR_INVALID_ARG, "batch too large (tile count)");
    tf[n] = (uint32_t)n_tiles64;
    n_tiles = (uint32_t)n_tiles64;
    n_segs = n_tiles + n;

    // level-1 buffer: one fixed slot per tile (2x the expected count: density 2/(w+1), sketch 2^-(4+r)) and
    // a cursor-allocated overflow region for dense tiles and the per-contig tails
    dens = sketch ? 1.0 / (double)(1ull << (4 + spec.r)) : 2.0 / (double)(spec.w + 1);
    slot = std::min<uint32_t>(tc, (((uint32_t)((double)tc * dens * 2.0) + 64 + 63) / 64) * 64);
    slots_total = (uint64_t)n_tiles * slot;
    cap_par = (uint64_t)((double)bases_tiled * dens * 0.02) + 65536 + 32ull * n;  // (a contig's tail has its own slot)
    // low-complexity / N-rich input overflows the fixed tile slots by far more than that: remember what the last call with
    // this spec needed per base (a genome comes as many similar batches) instead of running stage 1 twice every time
    l1_key = (double)spec.w * 1e3 + spec.k + (sketch ? 0.5 : 0.0);
    if (ctx->est_l1_key == l1_key && ctx->est_ovf_ratio > 0)
        cap_par = std::max<uint64_t>(cap_par, (uint64_t)((double)bases_tiled * ctx->est_ovf_ratio * 1.1) + 65536 + 32ull * n);

    int rc;
    if ((rc = ctx->ws_tile_first.ensure(ctx, ((size_t)n + 1) * sizeof(uint32_t))) ||
        (rc = ctx->ws_seg_off.ensure(ctx, ((size_t)n_segs + 1) * sizeof(uint64_t))) ||
        (rc = ctx->ws_tile_desc.ensure(ctx, ((size_t)n_tiles + 1) * sizeof(TileDesc))) ||
        (rc = ctx->ws_seg_cnt.ensure(ctx, ((size_t)n_segs + 1) * sizeof(uint32_t))) ||
        (rc = ctx->ws_seg_cid.ensure(ctx, ((size_t)n_segs + 1) * sizeof(uint32_t))) ||
        (rc = ctx->ws_seg_dst.ensure(ctx, ((size_t)n_segs + 1) * sizeof(uint64_t))) ||
        (rc = ctx->ws_tile_lv.ensure(ctx, ((size_t)n_tiles + 1) * sizeof(uint64_t))) ||
        // one block that a single memset clears per call: cursors | contig flags | tile flags  (+ the status words)
        (rc = ctx->ws_cursor.ensure(ctx, (N_CURSOR + N_STATUS) * sizeof(unsigned long long) +
                                             std::max<size_t>(n, 1) * sizeof(uint32_t) + (size_t)n_tiles + 64 + 4 + 2 * (size_t)n_tiles + 8)) ||
        (rc = ctx->ws_off_a.ensure(ctx, ((size_t)n + 1) * sizeof(uint64_t))) ||
        (rc = ctx->ws_off_b.ensure(ctx, (N_STATUS + (size_t)n + 1) * sizeof(uint64_t))) ||
        // pinned: the pass's status words + result offsets come back into it; behind them the tile table and the rids on their way up
        (rc = ctx->ensure_mailbox((N_STATUS + (size_t)n + 1) * sizeof(uint64_t) + (2 * (size_t)n + 2) * sizeof(uint32_t))))
        return rc;
    // debug_poison: whatever the workspaces hold now was left by an earlier call (or by the lane's previous job): nothing of this job
    // may depend on it.  On the job's front stream, in front of everything the job enqueues.
    if (ctx->opt.debug_poison && (rc = ctx->poison_workspaces(st))) return rc;
    d_cursor = (unsigned long long *)ctx->ws_cursor.p;
    d_cflags = (uint32_t *)(d_cursor + N_CURSOR);
    d_tflags = (uint8_t *)(d_cflags + std::max<size_t>(n, 1));
    // behind the tile flags and their slack: first | last << 8 block of 64 core positions with a palindromic k-mer, of the tiles the
    // tile kernel flags for one (not cleared: read only where this pass's tile kernel has set flag bit 0, i.e. has written it)
    pal_off = (((size_t)n_tiles + 64) + 3) & ~(size_t)3;
    d_tpal = (uint16_t *)(d_tflags + pal_off);
    sub_tile = !sketch && !ctx->opt.no_sub_tile_islands && tiled && bases_tiled && n_tiles;
    zero_bytes = N_CURSOR * sizeof(unsigned long long) + std::max<size_t>(n, 1) * sizeof(uint32_t) + (size_t)n_tiles + 16;
    mbox = (uint64_t *)ctx->mailbox;
    if (ctx->opt.debug_poison) memset(mbox, 0xFF, (N_STATUS + (size_t)n + 1) * sizeof(uint64_t));  // (status words + offsets of an earlier pass)
    // (through the pinned mailbox, read by a KERNEL of this stream: a copy engine would take these few kB in the order of its
    // queue -- behind every staging copy a pipelined host call has queued for the sub-batches to come)
    uint32_t *up = (uint32_t *)(mbox + N_STATUS + (size_t)n + 1);
    memcpy(up, tf.data(), ((size_t)n + 1) * sizeof(uint32_t));
    launch_copy_words(st, (uint32_t *)ctx->ws_tile_first.p, up, (uint64_t)n + 1);
    dbg_lap("plan + tile table uploaded");
    d_rids = nullptr;
    if (rids && n) {
        if ((rc = ctx->ws_rids.ensure(ctx, (size_t)n * sizeof(uint32_t)))) return rc;
        memcpy(up + n + 1, rids, (size_t)n * sizeof(uint32_t));
        launch_copy_words(st, (uint32_t *)ctx->ws_rids.p, up + n + 1, n);
        d_rids = (uint32_t *)ctx->ws_rids.p;
    }

    a.b = b->d;
    a.n_contigs = n;
    a.n_tiles = n_tiles;
    a.tile_first = (const uint32_t *)ctx->ws_tile_first.p;
    a.desc = (TileDesc *)ctx->ws_tile_desc.p;
    a.tile_flags = d_tflags;
    a.tile_pal = sub_tile ? d_tpal : nullptr;  // (round 6: islands begin and end inside tiles, island_list.h)
    a.tile_lv = nullptr;  // set once mark_invalid_tiles has filled it and run_islands has made it cumulative
    a.w = w_eff;
    a.k = spec.k;
    a.r = spec.r;
    a.tc = tc;
    a.ext = ext;
    a.sketch = sketch ? 1u : 0u;
    a.cursor = d_cursor;
    a.seg_off = (uint64_t *)ctx->ws_seg_off.p;
    a.seg_cnt = (uint32_t *)ctx->ws_seg_cnt.p;
    a.seg_cid = (uint32_t *)ctx->ws_seg_cid.p;
    a.contig_flags = d_cflags;

    memset(&prof, 0, sizeof(prof));
    prof.n_tiles = n_tiles;
    prof.bases_tiled = bases_tiled;
    // big batches look at the level-1 status words once before the list stage is enqueued (one more round trip, ~45 us):
    // if a tile asked for the exact path the islands are fixed first and the list stage runs once.  Smaller batches run
    // optimistically and repeat stages 2-4 in the (rare) flagged case: cheaper than the round trip below ~1 Gbp.
    const uint64_t early_bp = (uint64_t)std::max<int64_t>(0, ctx->opt.early_sync_bp);
    // (a batch the host packer has counted non-ACGT bytes in is known to need islands: look at the flags before the list stage)
    // (and so is a batch of a spec whose last batch on this context needed islands: a genome comes as many similar batches)
    early_sync = b->total_bases >= early_bp || !serial.empty() || b->host_saw_invalid ||
                 (ctx->est_flagged && ctx->est_l1_key == l1_key && b->total_bases >= (4u << 20));
    pad_fix = padding && !sketch && spec.r > 1;
    do_reduce = !sketch && spec.r > 1;
    halo = do_reduce ? 2 * spec.r * spec.r : 1;
    // (512-element workgroups for a pipelined job, measured and NOT the default: 14 KB of LDS fit beside a CU's four tile workgroups,
    // but the dispatcher gives them tile slots all the same -- the tile kernel beside them took 20.6 ms instead of 20.1 with the
    // 1024-element workgroups; with the context's stream at the highest priority they do not run beside the tiles at all)
    fb = (sf != sb && do_reduce && halo <= 32 && (ctx->opt.pipe_small_list || ctx->opt.pipe_persistent_list > 0)) ? 512u : FUSED_BLOCK_ELEMS;
    slot2 = do_reduce ? fb / 4 : fb;
    serial_base = 0;
    islands_done = false;
    l2_cursor_clean = false;
    pre_listed = false;
    // (a spec without a tile path -- w < 17: every contig goes through the exact machine -- ALWAYS needs the second pass: its first
    // pass has nothing for a list stage to work on)
    stage1_only = optimistic && !ctx->opt.no_stage1_only &&
                  (!serial.empty() || (tiled && bases_tiled && ctx->est_flagged && ctx->est_l1_key == l1_key && b->total_bases >= (4u << 20)));
    list_pending = false;
    return PGR_OK;
}

// islands of exact tiles from the flags: flags[c] bit 0 = a tile of contig c saw a palindromic k-mer, n_invalid[c] = its non-ACGT
// bytes, tf[tile] = tile flags (bit 0 palindromic k-mer, bit 1 non-ACGT byte in reach, bit 2 nothing but such bytes; bit 3 is set here)
void ShmmrJob::list_islands(const uint32_t *flags, const uint32_t *n_invalid, uint8_t *tf, const uint16_t *pal, std::vector<Island> &islands,
                            std::vector<uint32_t> &gap_segs, uint32_t cut_margin, uint32_t cut_settle) {
    const std::vector<uint32_t> &tfi = tile_first();
    // a genome-sized batch (half a million tiles in a dozen contigs): ranges of contigs on the pool's threads, lists in contig order
    const unsigned par = std::min<unsigned>(std::min<unsigned>(16, HostPool::instance().workers() + 1), n);
    if (n_tiles < (1u << 17) || par < 2) {
        list_islands_from_flags(n, tfi.data(), b->h_len.data(), tc, sketch, flags, n_invalid, tf, pal, islands, gap_segs, 0, 0xFFFFFFFFu, cut_margin, cut_settle);
        return;
    }
    std::vector<uint32_t> cut(par + 1, n);  // contig ranges of about n_tiles / par tiles each
    cut[0] = 0;
    for (unsigned p = 1, c = 0; p < par; ++p) {
        const uint64_t want = (uint64_t)n_tiles * p / par;
        while (c < n && tfi[c] < want) ++c;
        cut[p] = c;
    }
    std::vector<std::vector<Island>> isl(par);
    std::vector<std::vector<uint32_t>> gs(par);
    HostPool::instance().parallel_for(par, [&](size_t p) {
        if (cut[p] < cut[p + 1])
            list_islands_from_flags(n, tfi.data(), b->h_len.data(), tc, sketch, flags, n_invalid, tf, pal, isl[p], gs[p], cut[p], cut[p + 1], cut_margin, cut_settle);
    });
    for (unsigned p = 0; p < par; ++p) {
        islands.insert(islands.end(), isl[p].begin(), isl[p].end());
        gap_segs.insert(gap_segs.end(), gs[p].begin(), gs[p].end());
    }
}

int ShmmrJob::stage1() {
    hipStream_t st = sf;
    uint64_t serial_total = 0;
    for (uint32_t c : serial) serial_total += (uint64_t)b->h_len[c] / 4 + 4096;
    int r;
    if ((r = ctx->ws_l1.ensure(ctx, (slots_total + cap_par + serial_total + (uint64_t)n * L1_TAIL_SLOT + 1) * sizeof(L1Rec)))) return r;
    a.out = (L1Rec *)ctx->ws_l1.p;
    a.slot = slot;
    a.ovf_base = slots_total;
    a.cap = cap_par;
    a.tail_base = slots_total + cap_par;  // the contigs' tail slots sit between the overflow region and the exact regions
    serial_base = a.tail_base + (uint64_t)n * L1_TAIL_SLOT;  // (run_exact_islands grows the buffer behind this point)
    // cursors (both stages), contig flags, tile flags: cleared by the tile descriptor kernel when there are tiles
    if (!(tiled && bases_tiled)) PGR_HIP(ctx, hipMemsetAsync(d_cursor, 0, zero_bytes, st));
    // Without a tile kernel nobody writes the tile segments' entries (the tail kernel writes the contigs' tail segments, the islands
    // write theirs later): a list stage that runs before the islands -- the optimistic pass of a pipelined job -- scanned whatever the
    // workspace held and read level-1 records from wherever that pointed.  Found as a GPU memory fault that came and went with the
    // SIZE of an unrelated workspace (profiles/r05_fuzz/cursor_block_size_fault.txt): the counts start at zero.
    if (!(tiled && bases_tiled) && n_segs && !ctx->opt.debug_inject_stale_segments)
        PGR_HIP(ctx, hipMemsetAsync(ctx->ws_seg_cnt.p, 0, ((size_t)n_segs + 1) * sizeof(uint32_t), st));
    if (n == 0) PGR_HIP(ctx, hipMemsetAsync((uint32_t *)ctx->ws_seg_cnt.p + n_segs, 0, sizeof(uint32_t), st));
    l2_cursor_clean = true;  // (otherwise the tail kernel writes the scan sentinel)
    if (!(tiled && bases_tiled)) PGR_HIP(ctx, hipEventRecord(ctx->ev[1], st));
    pre_listed = false;
    flags_prefetched = false;
    early_islands.reset();
    if (tiled && bases_tiled) {
        launch_level1_pre(st, a, (uint64_t *)ctx->ws_tile_lv.p, no_islands_pass && !b->h_n_invalid.empty() && !b->host_saw_invalid);
        const bool pre = b->host_saw_invalid && !b->h_n_invalid.empty() && n_tiles && !ctx->opt.no_pre_islands;
        if (pre) {
            const size_t tb = scan_max_temp_bytes(n_tiles);
            if ((r = ctx->ws_scan_tmp.ensure(ctx, tb)) || (r = ctx->ensure_imail((size_t)n_tiles + 4))) return r;
            PGR_HIP(ctx, scan_max_inplace(st, ctx->ws_scan_tmp.p, tb, (uint64_t *)ctx->ws_tile_lv.p, n_tiles));
            PGR_HIP(ctx, hipEventRecord(ctx->pre_ev[0], st));
            PGR_HIP(ctx, hipStreamWaitEvent(ctx->pre_stream, ctx->pre_ev[0], 0));
            launch_copy_words(ctx->pre_stream, (uint32_t *)ctx->imail, (const uint32_t *)d_tflags, ((uint64_t)n_tiles + 3) / 4);  // (d_tflags: 4-byte aligned, 64 bytes of slack behind it)
            PGR_HIP(ctx, hipEventRecord(ctx->pre_ev[1], ctx->pre_stream));
        }
        PGR_HIP(ctx, hipEventRecord(ctx->ev[1], st));  // (prof.level1_ms is the tile kernel alone: descriptors and flags are in front of it)
        launch_level1_tiles(st, a);
        if (pre) {
            PGR_HIP(ctx, hipEventSynchronize(ctx->pre_ev[1]));
            std::vector<uint32_t> no_flags(n, 0u);
            pre_islands.clear();
            pre_gap_segs.clear();
            list_islands(no_flags.data(), b->h_n_invalid.data(), (uint8_t *)ctx->imail, nullptr, pre_islands, pre_gap_segs);
            pre_listed = true;
            dbg_lap("islands around non-ACGT bytes listed");
            // The synchronous driver is about to wait for the tile kernel's flags and would only then build and launch the
            // islands' first round (a chromosome-like contig: flags on the host at 0.64 ms, chunk kernel running from 0.71).
            // These islands do not depend on the flags: their round 0 goes behind the tile kernel NOW, and when the flags
            // arrive the chunks' states are on their way.  A flag that adds an island (a palindromic k-mer) makes
            // run_islands() list them again and start over: the early round's work is then wasted, not wrong -- it wrote
            // behind serial_base and emptied segments that the longer list empties again.
            if (early_sync && !optimistic && serial.empty() && !pre_islands.empty() && st == ctx->stream && !ctx->opt.no_early_islands) {
                L1Args as = a;
                as.w = sketch ? 1u : spec.w;
                as.tile_lv = (uint64_t *)ctx->ws_tile_lv.p;  // (made cumulative in front of the tile kernel, above)
                // (the pinned image must not move while the round's kernels are pending: room for the flags' download as well)
                if ((r = ctx->ensure_imail(2 * (std::max<size_t>(n, 1) * sizeof(uint32_t) + 16) + n_tiles + 64))) return r;
                early_islands.reset(new IslandRun(ctx, st, b, as, pre_islands, tile_first(), tc, serial_base, pre_gap_segs));
                // (beside the tile kernel when the device runs two streams side by side; the side stream has waited for everything
                // in front of the tile kernel: the copy of the tile flags above)
                if ((r = early_islands->begin(ctx->opt.early_islands_in_stream ? nullptr : ctx->pre_stream))) return r;
                a.out = early_islands->a.out;  // (the level-1 buffer may have grown -- behind the tile kernel, which has its pointer)
                dbg_lap("islands: round 0 enqueued behind the tile kernel");
            }
        }
    }
    PGR_HIP(ctx, hipEventRecord(ctx->ev[2], st));
    if (!(tiled && bases_tiled)) launch_level1_tails(st, a);  // (otherwise every contig's last tile has run its tail)
    islands_done = false;
    return PGR_OK;
}

// islands of exact tiles: around palindromic k-mers (skipped pushes, flagged by the tile kernel) and non-ACGT bytes (flagged by
// mark_invalid_tiles); whole contigs when the spec has no tile path.  Synchronizes (front stream).
int ShmmrJob::run_islands(uint64_t need_word) {
    hipStream_t st = sf;
    dbg_lap("islands: level-1 flags seen");
    std::vector<Island> islands;
    std::vector<uint32_t> gap_segs;  // [first, last + 1) segment ranges of tiles deep inside runs of non-ACGT bytes: emptied
    for (uint32_t c : serial) islands.push_back(Island{c, 0, b->h_len[c], false, true});
    const bool use_pre = pre_listed && !(need_word & 1ull) && serial.empty();  // (no tile saw a palindromic k-mer)
    // A tile saw a palindromic k-mer: the islands are listed again (a superset).  The early round is kept -- its islands around
    // non-ACGT bytes are still islands of the longer list, all but those a new neighbour is merged with -- but it is waited for and
    // its states are taken out of the pinned image BEFORE the flags come down into that image.  (no_early_merge: the first form,
    // the early round is dropped.)
    bool merge_early = !use_pre && early_islands && serial.empty() && !ctx->opt.no_early_merge;
    if (merge_early) {
        int r0 = early_islands->settle_first_round();
        if (r0) {
            early_islands.reset();
            return r0;
        }
    }
    if (!use_pre && !merge_early) early_islands.reset();
    if (use_pre) {
        islands = pre_islands;
        gap_segs = pre_gap_segs;
    } else if (tiled && bases_tiled && need_word) {
        // contig flags and tile flags are neighbours in the cursor block: two copies into the pinned image (three pageable ones
        // were 67 us of a 60 Mbp call's 660)
        const size_t nc = std::max<size_t>(n, 1) * sizeof(uint32_t);
        const size_t flag_bytes = nc + (sub_tile ? pal_off + 2 * (size_t)n_tiles : (size_t)n_tiles), inv_off = (flag_bytes + 15) & ~(size_t)15;
        int r0;
        if ((r0 = ctx->ensure_imail(inv_off + nc))) return r0;
        uint8_t *img = (uint8_t *)ctx->imail;
        if (!flags_prefetched) {
            PGR_HIP(ctx, hipMemcpyAsync(img, d_cflags, flag_bytes, hipMemcpyDeviceToHost, st));
            if (n) PGR_HIP(ctx, hipMemcpyAsync(img + inv_off, b->d.n_invalid, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            PGR_HIP(ctx, hipStreamSynchronize(st));
        }
        flags_prefetched = false;
        dbg_lap("islands: tile flags on the host");
        // (list_islands marks tiles in the flags it is given: in the pinned image itself, as stage1() does for the early list -- the image
        // is not read again; a copy of a genome-sized batch's half megabyte was 40 us on the way to the islands' round)
        uint8_t *tf = img + nc;
        std::vector<uint32_t> flags((const uint32_t *)img, (const uint32_t *)img + n), n_invalid((const uint32_t *)(img + inv_off), (const uint32_t *)(img + inv_off) + n);
        // (cut_margin: a multiple of 64 >= w + k + 64, island_list.h)
        const uint32_t cut_margin = sub_tile ? ((spec.w + spec.k + 64 + 63) / 64) * 64 + 64 : 0;
        // (cut_settle: a multiple of 64 >= 2 w + k + 64; option island_settle for A/B)
        const uint32_t cut_settle = !sub_tile ? 0 : ctx->opt.island_settle > 0 ? (uint32_t)((ctx->opt.island_settle + 63) / 64 * 64)
                                                                              : ((2 * spec.w + spec.k + 64 + 63) / 64) * 64;
        list_islands(flags.data(), n_invalid.data(), tf, sub_tile ? (const uint16_t *)(img + nc + pal_off) : nullptr, islands, gap_segs, cut_margin,
                     cut_settle);
    }
    {
        std::vector<uint32_t> cs;
        for (const auto &is : islands) cs.push_back(is.contig);
        std::sort(cs.begin(), cs.end());
        prof.n_serial_contigs = std::unique(cs.begin(), cs.end()) - cs.begin();
    }
    if (!islands.empty()) {
        L1Args as = a;
        as.w = sketch ? 1u : spec.w;  // the exact machine follows the spec literally (sketch ignores w)
        if (tiled && bases_tiled && n_tiles && !use_pre && !merge_early) {  // (use_pre, merge_early: done in front of the tile kernel)
            // per-tile "last valid position" (written by mark_invalid_tiles) -> cumulative: the chunks' k-mer look-back
            // and forward roll cross a run of N of any length in one step
            const size_t tb = scan_max_temp_bytes(n_tiles);
            int r2;
            if ((r2 = ctx->ws_scan_tmp.ensure(ctx, tb))) return r2;
            PGR_HIP(ctx, scan_max_inplace(st, ctx->ws_scan_tmp.p, tb, (uint64_t *)ctx->ws_tile_lv.p, n_tiles));
        }
        if (tiled && bases_tiled && n_tiles) as.tile_lv = (uint64_t *)ctx->ws_tile_lv.p;
        dbg_lap("islands: listed");
        int r;
        if (use_pre && early_islands) {  // round 0 is behind the tile kernel already
            r = early_islands->finish();
            as.out = early_islands->a.out;
            islands = early_islands->islands;
        } else if (merge_early) {  // ... and has been looked at: the islands the flags have added join the run
            if (!(r = early_islands->adopt(islands))) r = early_islands->finish();
            as.out = early_islands->a.out;
            islands = early_islands->islands;
        } else {
            r = run_exact_islands(ctx, st, b, as, islands, tile_first(), tc, serial_base, gap_segs);
        }
        early_islands.reset();
        if (r) return r;
        dbg_lap("islands: exact");
        a.out = as.out;
        prof.exact_bases = 0;
        for (const auto &is : islands) prof.exact_bases += is.E - is.B;  // final extents (islands may have grown)
    }
    islands_done = true;
    return PGR_OK;
}

// the result object and the estimates the list stage is sized by
int ShmmrJob::begin_result() {
    res = new pgr_shmmrs();
    res->ctx = ctx;
    res->n = n;
    // (the offsets of a million reads are 8 MB: the block of a destroyed result of this context is used again)
    if ((size_t)n + 1 >= (1u << 17) && ctx->spare_off.capacity() >= (size_t)n + 1) res->h_off.swap(ctx->spare_off);
    res->h_off.resize((size_t)n + 1);
    int rc;
    // the pipeline's status words sit right in front of the result offsets: ONE copy brings both to the mailbox
    if ((rc = ctx->dmalloc((void **)&res->d_block, (N_STATUS + (size_t)n + 1) * sizeof(uint64_t)))) return rc;
    res->d_off = res->d_block + N_STATUS;
    // estimates: level-1 count from the density (low-complexity sequence exceeds it: retried with the true count),
    // final count from this context's last result with the same spec (first call: a third of the level-1 estimate)
    const uint64_t l1_bound = slots_total + cap_par + b->total_bases / 4 + 4096ull * n + 4096;  // what stage 1 can emit at all
    // (per contig: the first window and the tail emit a few elements on top of the density -- NOT thousands: 2048 per contig made
    // the list stage of 10^6 reads a grid of 2 x 10^6 workgroups for 25 x 10^3 of work, 0.6 of its 0.8 ms, and 8 GB of slots)
    uint64_t l1_est = std::min<uint64_t>(l1_bound, (uint64_t)((double)b->total_bases * dens * 1.06) + 16ull * n + 8192);
    // (low-complexity sequence is denser than that: what the last call with this spec saw per base, like the result's size)
    if (ctx->est_l1_key == l1_key && ctx->est_l1_dens > dens)
        l1_est = std::min<uint64_t>(l1_bound, std::max<uint64_t>(l1_est, (uint64_t)((double)b->total_bases * ctx->est_l1_dens * 1.04) + 16ull * n + 8192));
    n_blocks = (uint32_t)((l1_est + fb - 1) / fb);
    cap2 = (uint64_t)((double)l1_est * 0.01) + 65536;
    if (ctx->est_l1_key == l1_key && ctx->est_l2_ovf > 0) cap2 = std::max<uint64_t>(cap2, (uint64_t)((double)b->total_bases * ctx->est_l2_ovf * 1.1) + 65536);
    spec_key = (double)spec.w * 1e9 + spec.k * 1e6 + spec.r * 1e4 + spec.min_span + (sketch ? 0.5 : 0.0) + (padding ? 0.25 : 0.0);
    const double ratio = (ctx->est_spec_key == spec_key && ctx->est_final_ratio > 0) ? ctx->est_final_ratio * 1.15 : dens / 3.0 + 1e-4;
    cap_res = std::max<uint64_t>((uint64_t)((double)b->total_bases * ratio) + 64ull * n + 1024, 16);
    d_list = nullptr;
    d_loff = nullptr;
    d_total1 = (uint64_t *)ctx->ws_seg_dst.p + n_segs;
    scanned4 = false;
    host_copy_elems = 0;
    return PGR_OK;
}

// counts -> offsets on the back stream: rocPRIM's scan, or (beside the tile kernel) the one that takes no LDS
int ShmmrJob::scan_back(const uint32_t *in, uint64_t *out, uint32_t n_plus_1) {
    int r;
    if (lds_match) {
        const size_t tb = scan_counts_nolds_temp_bytes(n_plus_1);
        if ((r = ctx->ws_scan_tmp.ensure(ctx, tb))) return r;
        PGR_HIP(ctx, scan_counts_nolds(sb, ctx->ws_scan_tmp.p, tb, in, out, n_plus_1));
        return PGR_OK;
    }
    const size_t tb = scan_counts_temp_bytes(n_plus_1);
    if ((r = ctx->ws_scan_tmp.ensure(ctx, tb))) return r;
    PGR_HIP(ctx, scan_counts(sb, ctx->ws_scan_tmp.p, tb, in, out, n_plus_1));
    return PGR_OK;
}

// debug_poison: the segment table a list stage is about to read has to describe records INSIDE the level-1 buffer (and its scan sentinel
// has to be zero).  An entry that does not is counted in cursor word 7 (it travels to the host with the status words: decide() fails the
// call) and cleared, so that the kernels behind this one do not follow it into unmapped memory.
__global__ void check_segments_kernel(uint32_t *__restrict__ seg_cnt, const uint64_t *__restrict__ seg_off, uint32_t n_segs,
                                      uint64_t l1_elems, unsigned long long *__restrict__ trip) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_segs) return;
    const uint32_t c = seg_cnt[i];
    if (c == 0) return;
    const uint64_t o = seg_off[i];
    if (i == n_segs || o > l1_elems || o + c > l1_elems) {
        atomicAdd(trip, 1ull);
        seg_cnt[i] = 0;
    }
}

int ShmmrJob::stage2() {
    hipStream_t st = sb;
    int r;
    if (ctx->opt.debug_poison)
        hipLaunchKernelGGL(check_segments_kernel, dim3((n_segs + 1 + 255) / 256), dim3(256), 0, st, (uint32_t *)ctx->ws_seg_cnt.p,
                           (const uint64_t *)ctx->ws_seg_off.p, n_segs, (uint64_t)(ctx->ws_l1.cap / sizeof(L1Rec)), d_cursor + 7);
    if ((r = scan_back((const uint32_t *)ctx->ws_seg_cnt.p, (uint64_t *)ctx->ws_seg_dst.p, n_segs + 1))) return r;
    if (pad_fix)
        launch_contig_offsets(st, (const uint64_t *)ctx->ws_seg_dst.p, (const uint32_t *)ctx->ws_tile_first.p, n, n_segs,
                              (uint64_t *)ctx->ws_off_a.p);
    return PGR_OK;
}

int ShmmrJob::stage3() {
    hipStream_t st = sb;
    int r;
    if ((r = ctx->ws_blk_cnt.ensure(ctx, ((size_t)n_blocks + 1) * sizeof(uint32_t))) ||
        (r = ctx->ws_blk_base.ensure(ctx, ((size_t)n_blocks + 1) * sizeof(uint64_t))) ||
        (r = ctx->ws_blk_off.ensure(ctx, ((size_t)n_blocks + 1) * sizeof(uint64_t))) ||
        (r = ctx->ws_start_rank.ensure(ctx, ((size_t)n_blocks + 1) * sizeof(uint32_t))))
        return r;
    // fixed output slot per fused workgroup (expected survivors: ~12 % after two reductions + min_span), plus
    // a cursor-allocated overflow region
    const uint64_t slots2 = (uint64_t)n_blocks * slot2;
    if ((r = ctx->ws_list_a.ensure(ctx, (slots2 + cap2 + 1) * sizeof(pgr_mm128)))) return r;
    if (!l2_cursor_clean) PGR_HIP(ctx, hipMemsetAsync(d_cursor + 4, 0, 2 * sizeof(unsigned long long), st));
    l2_cursor_clean = false;  // (a repeat of this stage alone clears it again)
    if (n_blocks == 0) PGR_HIP(ctx, hipMemsetAsync((uint32_t *)ctx->ws_blk_cnt.p, 0, sizeof(uint32_t), st));
    FusedArgsPub fa;
    fa.l1 = (const L1Rec *)ctx->ws_l1.p;
    fa.seg_cid = (const uint32_t *)ctx->ws_seg_cid.p;
    fa.k = spec.k;
    fa.seg_off = (const uint64_t *)ctx->ws_seg_off.p;
    fa.seg_cnt = (const uint32_t *)ctx->ws_seg_cnt.p;
    fa.seg_dst = (const uint64_t *)ctx->ws_seg_dst.p;
    fa.n_segs = n_segs;
    fa.total = d_total1;
    fa.r = spec.r;
    fa.padding = padding ? 1u : 0u;
    fa.min_span = spec.min_span;
    fa.do_reduce = do_reduce ? 1u : 0u;
    fa.halo = halo;
    fa.out = (pgr_mm128 *)ctx->ws_list_a.p;
    fa.slot = slot2;
    fa.ovf_base = slots2;
    fa.cap = cap2;
    fa.cursor = d_cursor + 4;
    fa.blk_off = (uint64_t *)ctx->ws_blk_off.p;
    fa.blk_cnt = (uint32_t *)ctx->ws_blk_cnt.p;
    fa.blk_first_seg = (uint32_t *)ctx->ws_start_rank.p;
    fa.lds_match = lds_match;
    fa.block_elems = fb;
    fa.persistent_grid = (sf != sb && fb == 512u && ctx->opt.pipe_persistent_list > 0) ? (uint32_t)ctx->opt.pipe_persistent_list : 0u;
    launch_fused_select_pub(st, fa, n_blocks);
    return PGR_OK;
}

int ShmmrJob::stage4() {
    hipStream_t st = sb;
    int r;
    if (!scanned4) {  // (a repeat of stage 4 alone only re-gathers into a bigger result buffer)
        if ((r = scan_back((const uint32_t *)ctx->ws_blk_cnt.p, (uint64_t *)ctx->ws_blk_base.p, n_blocks + 1))) return r;
        scanned4 = true;
    }
    const uint64_t *d_nfinal = (const uint64_t *)ctx->ws_blk_base.p + n_blocks;
    if (!pad_fix) {  // common case: gather straight into the result buffer
        // (a second pass whose result fits the block of the first keeps it: freeing it and asking again is a round through the
        // allocator's events for nothing)
        if (!res->d_mm || res_alloc_elems < cap_res) {
            ctx->dfree(res->d_mm);
            res->d_mm = nullptr;
            res_alloc_elems = 0;
            if ((r = ctx->dmalloc((void **)&res->d_mm, cap_res * sizeof(pgr_mm128)))) return r;
            res_alloc_elems = cap_res;
        }
        d_list = res->d_mm;
        d_loff = res->d_off;
    } else {
        if ((r = ctx->ws_list_b.ensure(ctx, cap_res * sizeof(pgr_mm128)))) return r;
        d_list = (pgr_mm128 *)ctx->ws_list_b.p;
        d_loff = (uint64_t *)ctx->ws_off_b.p + N_STATUS;
    }
    launch_gather_segments(st, (const pgr_mm128 *)ctx->ws_list_a.p, (const uint64_t *)ctx->ws_blk_off.p,
                           (const uint32_t *)ctx->ws_blk_cnt.p, (const uint64_t *)ctx->ws_blk_base.p, n_blocks, d_list, cap_res);
    launch_offsets_by_rid(st, d_list, d_nfinal, cap_res, n, d_loff, d_cursor, d_total1, d_loff - N_STATUS);  // + status words
    if (d_rids) launch_patch_rid(st, d_list, d_nfinal, cap_res, d_rids, n);
    PGR_HIP(ctx, hipEventRecord(ctx->ev_end, st));
    // a consumer of the result that does not want to wait for the host (the query path: pair records, lookup, chaining; a
    // pipelined index build: pair records) enqueues its kernels here, behind stage 4 and in front of the one wait; a repeated
    // pass calls it again.  (In front of the copies to the host as well: a DMA between two kernels costs ~20 us of bubbles.)
    if (ctx->post_enqueue && !pad_fix && (r = ctx->post_enqueue(d_list, d_loff, cap_res, d_nfinal))) return r;
    if (post && !pad_fix && (r = post(st, d_list, d_loff, cap_res, d_nfinal))) return r;
    // (status words + offsets into the pinned mailbox by a kernel of this stream, not by a copy engine: see plan())
    launch_copy_words(st, (uint32_t *)mbox, (const uint32_t *)(d_loff - N_STATUS), 2 * (N_STATUS + (uint64_t)n + 1));
    // a small result that the caller wants on the host anyway (pgr_shmmr_batch) rides along with this round trip
    host_copy_elems = 0;
    if (ctx->want_host_copy && !optimistic && !pad_fix && cap_res * sizeof(pgr_mm128) <= (256u << 10) &&
        ctx->ensure_pinned_out(256u << 10) == PGR_OK) {
        PGR_HIP(ctx, hipMemcpyAsync(ctx->pinned_out, d_list, cap_res * sizeof(pgr_mm128), hipMemcpyDeviceToHost, st));
        host_copy_elems = cap_res;
    }
    return PGR_OK;
}

// One pass from stage `from` on, enqueued; the caller waits for the back stream and calls decide().  JOB_RESTART: the early
// look found the overflow region of stage 1 too small (it has grown): enqueue again.
int ShmmrJob::enqueue_pass() {
    int rc;
    if (++attempt > 9) return ctx->fail(PGR_ERR_INTERNAL, "shimmer pipeline: buffers kept overflowing");
    if (sf == sb) lds_match = 0;  // (a synchronous pass: nothing runs beside it)
    if (from <= 1) {
        if ((rc = stage1())) return rc;
        if (early_sync && !optimistic) {
            // (a small batch that is looked at early is a batch that probably needs islands: its flags ride along with the status
            // words instead of costing a second round trip -- 20 us of the repeat-rich contigs' 500 -- unless the pinned image
            // belongs to the early round of the islands, which then needs no flags)
            flags_prefetched = false;
            const size_t nc = std::max<size_t>(n, 1) * sizeof(uint32_t);
            const size_t flag_bytes = nc + (sub_tile ? pal_off + 2 * (size_t)n_tiles : (size_t)n_tiles), inv_off = (flag_bytes + 15) & ~(size_t)15;
            if (tiled && bases_tiled && !early_islands && inv_off + nc <= (256u << 10)) {
                if ((rc = ctx->ensure_imail(inv_off + nc))) return rc;
                uint8_t *img = (uint8_t *)ctx->imail;
                launch_copy_words(sf, (uint32_t *)img, (const uint32_t *)d_cflags, (flag_bytes + 3) / 4);  // (64 bytes of slack behind the tile flags)
                if (n) launch_copy_words(sf, (uint32_t *)(img + inv_off), (const uint32_t *)b->d.n_invalid, n);
                flags_prefetched = true;
            }
            launch_copy_words(sf, (uint32_t *)mbox, (const uint32_t *)d_cursor, 8);  // (by a kernel: see plan())
            if (hipStreamSynchronize(sf) != hipSuccess || hipGetLastError() != hipSuccess)
                return ctx->fail(PGR_ERR_DEVICE, "level-1 kernels failed on the device");
            l1_alloc_seen = mbox[0];
            if (mbox[1] || mbox[0] > cap_par) {  // cursor region too small: grow and redo
                cap_par = (uint64_t)((double)mbox[0] * 1.1) + 65536;
                return JOB_RESTART;
            }
            if ((mbox[2] || !serial.empty()) && (rc = run_islands(mbox[2]))) return rc;
            islands_done = true;
        }
        if (optimistic && stage1_only) {  // the status words of stage 1, and that is all for now
            launch_copy_words(sf, (uint32_t *)mbox, (const uint32_t *)d_cursor, 8);  // (by a kernel: see plan())
            list_pending = true;
            dbg_lap("stage 1 enqueued (the list stage waits for the flags)");
            return PGR_OK;
        }
        if (sf != sb) {  // the list stage runs on the back stream, behind the level-1 kernels of THIS job only
            lds_match = (tiled && bases_tiled && ctx->opt.lds_match) ? level1_tile_lds_bytes(a) : 0;
            PGR_HIP(ctx, hipEventRecord(ev_front, sf));
            PGR_HIP(ctx, hipStreamWaitEvent(sb, ev_front, 0));
        }
        PGR_HIP(ctx, hipEventRecord(ctx->ev[3], sb));
    }
    if (from <= 2 && (rc = stage2())) return rc;
    if (from <= 3) {
        scanned4 = false;
        if ((rc = stage3())) return rc;
    }
    if ((rc = stage4())) return rc;
    if (pad_fix) {
        l1_off.resize((size_t)n + 1);
        PGR_HIP(ctx, hipMemcpyAsync(l1_off.data(), ctx->ws_off_a.p, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, sb));
    }
    dbg_lap("all stages enqueued");
    return PGR_OK;
}

// the one look at the device-side counts of a finished pass: done, or `from` = the stage the next pass starts at
int ShmmrJob::decide(bool &done) {
    done = false;
    int rc;
    const uint64_t l1_alloc = mbox[0], l1_ovf = mbox[1], need_islands = mbox[2], l2_alloc = mbox[4], l2_ovf = mbox[5];
    const uint64_t total1 = mbox[8];
    n_final = mbox[9];
    l1_alloc_seen = l1_alloc;
    l2_alloc_seen = l2_alloc;
    if (ctx->opt.debug_poison && !list_pending && mbox[7])
        return ctx->fail(PGR_ERR_INTERNAL, "debug_poison: " + std::to_string(mbox[7]) +
                                               " entries of the segment table a list stage read were not written by this job (they pointed "
                                               "outside the level-1 buffer): a workspace of an earlier call was read uninitialised");
    if (l1_ovf || l1_alloc > cap_par) {  // (only possible without the early look)
        cap_par = (uint64_t)((double)l1_alloc * 1.1) + 65536;
        list_pending = false;
        from = 1;
        return PGR_OK;
    }
    if (!islands_done && (need_islands || !serial.empty())) {
        if ((rc = run_islands(need_islands))) return rc;
        PGR_HIP(ctx, hipEventRecord(ctx->ev[3], sb));
        list_pending = false;
        from = 2;
        return PGR_OK;
    }
    if (list_pending) {  // (a stage-1-only pass that turned out clean: the list stage, now)
        list_pending = false;
        PGR_HIP(ctx, hipEventRecord(ctx->ev[3], sb));
        from = 2;
        return PGR_OK;
    }
    prof.n_level1 = total1;
    if (total1 > (uint64_t)n_blocks * fb) {  // denser than the estimate: the grid missed the tail
        n_blocks = (uint32_t)((total1 + fb - 1) / fb);
        cap2 = std::max<uint64_t>(cap2, (uint64_t)((double)total1 * 0.01) + 65536);
        from = 3;
        return PGR_OK;
    }
    if (l2_ovf || l2_alloc > cap2) {
        cap2 = (uint64_t)((double)l2_alloc * 1.05) + 65536;
        from = 3;
        return PGR_OK;
    }
    if (n_final > cap_res) {  // more survivors than estimated: bigger result buffer, gather again
        cap_res = n_final + 16;
        from = 4;
        return PGR_OK;
    }
    done = true;
    return PGR_OK;
}

// after the last pass: offsets, estimates for the next call, the padding artefact, the timings; hands the result over
int ShmmrJob::finish(pgr_shmmrs **out) {
    hipStream_t st = sb;
    int rc;
    memcpy(res->h_off.data(), mbox + N_STATUS, ((size_t)n + 1) * sizeof(uint64_t));
    res->count = n_final;
    if (host_copy_elems >= n_final && host_copy_elems) res->host_copy = (const pgr_mm128 *)ctx->pinned_out;
    res->rid_is_index = (d_rids == nullptr) && !pad_fix;
    if (b->total_bases) {
        ctx->est_spec_key = spec_key;
        ctx->est_final_ratio = (double)n_final / (double)b->total_bases;
    }
    if (bases_tiled) {
        ctx->est_l1_key = l1_key;
        ctx->est_ovf_ratio = (double)l1_alloc_seen / (double)bases_tiled;
    }
    if (b->total_bases) {
        ctx->est_l1_dens = (double)prof.n_level1 / (double)b->total_bases;
        ctx->est_l2_ovf = (double)l2_alloc_seen / (double)b->total_bases;
        ctx->est_flagged = prof.n_serial_contigs != 0 && serial.empty();
    }
    if (pad_fix) {
        // reference artefact: reduce_shmmr on an EMPTY list with padding emits its sentinels
        // (shmmrutils.rs:367-380), which survive as exactly two {MAX,MAX} after the second pass + filter
        std::vector<uint64_t> final_off((size_t)n + 1);
        uint64_t add = 0;
        for (uint32_t c = 0; c < n; ++c) {
            final_off[c] = res->h_off[c] + add;
            if (l1_off[c + 1] == l1_off[c]) add += 2;
        }
        final_off[n] = res->h_off[n] + add;
        res->count = final_off[n];
        if ((rc = ctx->dmalloc((void **)&res->d_mm, std::max<uint64_t>(res->count, 1) * sizeof(pgr_mm128)))) return rc;
        if (hipMemcpyAsync(res->d_off, final_off.data(), ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess)
            return ctx->fail(PGR_ERR_DEVICE, "H2D of the result offsets failed");
        launch_copy_or_sentinel(st, d_list, d_loff, res->d_off, n, res->d_mm);
        res->h_off = final_off;
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess)
            return ctx->fail(PGR_ERR_DEVICE, "pipeline failed on the device");
    }
    (void)hipEventElapsedTime(&prof.level1_ms, ctx->ev[1], ctx->ev[2]);
    (void)hipEventElapsedTime(&prof.level1_aux_ms, ctx->ev[2], ctx->ev[3]);
    (void)hipEventElapsedTime(&prof.level2_ms, ctx->ev[3], ctx->ev_end);  // (ev_end: end of stage 4, complete since the last wait)
    (void)hipEventElapsedTime(&prof.total_ms, ctx->ev[0], ctx->ev_end);
    ctx->prof = prof;
    *out = res;
    res = nullptr;
    dbg_lap("done");
    return PGR_OK;
}

int ShmmrJob::run_sync(pgr_shmmrs **out) {
    int rc;
    if ((rc = plan()) || (rc = begin_result())) return rc;
    PGR_HIP(ctx, hipEventRecord(ctx->ev[0], sf));
    for (;;) {
        rc = enqueue_pass();
        if (rc == JOB_RESTART) continue;
        if (rc) return rc;
        if (hipStreamSynchronize(sb) != hipSuccess || hipGetLastError() != hipSuccess)
            return ctx->fail(PGR_ERR_DEVICE, "pipeline failed on the device");
        dbg_lap("synchronized");
        bool done = false;
        if ((rc = decide(done))) return rc;
        if (done) break;
    }
    ctx->staged_unsynced = false;
    ctx->want_host_copy = false;
    return finish(out);
}

}  // namespace

extern "C" int pgr_shmmrs_compute(pgr_ctx *ctx, const pgr_batch *b, const pgr_spec *spec, const uint32_t *rids, int padding,
                                  pgr_shmmrs **out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!b || !out) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    int rc = check_spec(ctx, spec);
    if (rc) return rc;
    if (b->ctx != ctx) return ctx->fail(PGR_ERR_STATE, "batch belongs to another context");
    PGR_ENTER(ctx);
    if (!(padding && !spec->sketch && spec->r > 1)) {  // batches of short contigs: one workgroup per contig, 4 launches
        bool handled = false;
        if ((rc = shmmrs_compute_small(ctx, b, spec, rids, out, handled)) || handled) return rc;
    }
    ShmmrJob job;
    job.ctx = ctx;
    job.b = b;
    job.spec = *spec;
    job.rids = rids;
    job.padding = padding;
    job.sf = job.sb = ctx->stream;
    job.dbg_t = ctx->opt.debug_times != 0;
    job.dbg_t0 = std::chrono::steady_clock::now();
    return job.run_sync(out);  // (a failing pass releases the result and its device blocks with the job)
}

// Stage 1 ALONE of a resident batch -- tile descriptors, flags, the tile kernel with the contigs' tails -- and a consumer of the
// level-1 segments enqueued right behind it on the context's stream: the query path's level-1 form (pgr_aln.h: QfLevel1View;
// query_fused.hip), whose wavefronts run the list stage of their own query.  Nothing waits here.  *taken = false: the batch is
// not for this form (no tile path for the spec, the host packer has seen non-ACGT bytes, no bases) and nothing was enqueued.
namespace {
// plan + stage 1 of a job that is set up (context, batch, spec, streams) and the view of what the tile kernel leaves behind;
// *taken = false: the batch is not for the level-1 form (nothing was enqueued)
int level1_stage_and_view(ShmmrJob &job, QfLevel1View &v, bool *taken) {
    *taken = false;
    const pgr_batch *b = job.b;
    if (job.spec.sketch || job.spec.w < (uint32_t)L1_MIN_W || b->host_saw_invalid || b->n == 0 || b->total_bases == 0) return PGR_OK;
    pgr_ctx *ctx = job.ctx;
    int rc;
    if ((rc = job.plan())) return rc;
    if (!job.serial.empty() || !job.bases_tiled) return PGR_OK;
    PGR_HIP(ctx, hipEventRecord(ctx->ev[0], job.sf));
    job.no_islands_pass = true;
    rc = job.stage1();
    job.no_islands_pass = false;  // (a chained pass of the same job object -- the pipe's fall-through -- is an ordinary one)
    if (rc) return rc;
    v.l1 = job.a.out;
    v.seg_off = job.a.seg_off;
    v.seg_cnt = job.a.seg_cnt;
    v.tile_first = job.a.tile_first;
    v.status = job.d_cursor;
    v.ovf_cap = job.cap_par;
    v.flags = (uint32_t *)(job.d_cursor + 4);  // (the list stage's cursor words and the one behind them: cleared with the others, and no list stage runs)
    v.r = job.spec.r;
    v.min_span = job.spec.min_span;
    *taken = true;
    return PGR_OK;
}
}  // namespace

int pgr::shmmr_level1_then(pgr_ctx *ctx, const pgr_batch *b, const pgr_spec *spec, const std::function<int(const QfLevel1View &)> &consumer,
                           bool *taken) {
    ShmmrJob job;
    job.ctx = ctx;
    job.b = b;
    job.spec = *spec;
    job.rids = nullptr;
    job.padding = 0;
    job.sf = job.sb = ctx->stream;
    job.dbg_t = ctx->opt.debug_times != 0;
    job.dbg_t0 = std::chrono::steady_clock::now();
    QfLevel1View v;
    int rc = level1_stage_and_view(job, v, taken);
    if (rc || !*taken) return rc;
    rc = consumer(v);
    job.dbg_lap("stage 1 + the per-query kernel enqueued");
    return rc;
}

// B1 + seq_to_index in ONE call and ONE wait: the index-side pair records (seq_db.rs:381-400) are derived on the device behind the
// list stage (counts and offsets never visit the host) and land in the caller's device buffer; what pgr_shmmrs_compute followed
// by pgr_shmmrs_to_frag_recs_device does in two calls, two waits and two pageable table copies.
extern "C" int pgr_shmmrs_compute_recs(pgr_ctx *ctx, const pgr_batch *b, const pgr_spec *spec, const uint32_t *sids, pgr_frag_rec *d_recs,
                                       uint64_t recs_capacity, pgr_shmmrs **out, uint64_t *n_pairs) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!b || !out || !n_pairs) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    *n_pairs = 0;
    int rc = check_spec(ctx, spec);
    if (rc) return rc;
    if (b->ctx != ctx) return ctx->fail(PGR_ERR_STATE, "batch belongs to another context");
    PGR_ENTER(ctx);
    {   // batches of short contigs: the one-launch kernel, then the records from its (host-known) offsets
        bool handled = false;
        if ((rc = shmmrs_compute_small(ctx, b, spec, nullptr, out, handled))) return rc;
        if (handled) {
            rc = pgr_shmmrs_to_frag_recs_device(ctx, *out, sids, 0, d_recs, recs_capacity, n_pairs);
            if (rc) {
                pgr_shmmrs_destroy(*out);
                *out = nullptr;
            }
            return rc;
        }
    }
    if ((rc = ctx->ensure_qmail())) return rc;
    uint64_t *qm = (uint64_t *)ctx->qmail;
    qm[0] = 0;
    const uint32_t n = b->n;
    ShmmrJob job;
    job.ctx = ctx;
    job.b = b;
    job.spec = *spec;
    job.rids = nullptr;
    job.padding = 0;
    job.sf = job.sb = ctx->stream;
    job.dbg_t = ctx->opt.debug_times != 0;
    job.dbg_t0 = std::chrono::steady_clock::now();
    ShmmrJob *jp = &job;
    job.post = [ctx, jp, n, sids, d_recs, recs_capacity, qm](hipStream_t st, const pgr_mm128 *d_list, const uint64_t *d_off, uint64_t cap,
                                                             const uint64_t *d_count) -> int {
        int r;
        if ((r = ctx->ws_rec_off.ensure(ctx, ((size_t)n + 1) * sizeof(uint64_t)))) return r;
        uint32_t *d_sids = nullptr;
        if (sids && n) {  // (through the rids' place in the pinned mailbox, read by a kernel: ShmmrJob::plan)
            if ((r = ctx->ws_rids.ensure(ctx, (size_t)n * sizeof(uint32_t)))) return r;
            uint32_t *up = (uint32_t *)(jp->mbox + N_STATUS + (size_t)n + 1) + n + 1;
            memcpy(up, sids, (size_t)n * sizeof(uint32_t));
            launch_copy_words(st, (uint32_t *)ctx->ws_rids.p, up, n);
            d_sids = (uint32_t *)ctx->ws_rids.p;
        }
        launch_frag_recs_dev(st, d_list, d_off, n, cap, d_count, 0, (uint64_t *)ctx->ws_rec_off.p, d_recs, recs_capacity, nullptr, d_sids);
        launch_copy_words(st, (uint32_t *)qm, (const uint32_t *)((const uint64_t *)ctx->ws_rec_off.p + n), 2);
        return PGR_OK;
    };
    if ((rc = job.run_sync(out))) return rc;
    *n_pairs = qm[0];
    if (*n_pairs > recs_capacity || (*n_pairs && !d_recs)) {
        pgr_shmmrs_destroy(*out);
        *out = nullptr;
        return ctx->fail(PGR_ERR_INVALID_ARG, "output buffer too small for the pair records");
    }
    return PGR_OK;
}

// ------------------------------------------------------------------------------------------------
// pgr_pipe: two jobs in flight.  Job i's list stage and pair records run on the context's back stream while job i + 1's tiles
// run on the context's stream (include/pgr_hip.h).
//
// Records for an index.  The reference inserts batch after batch into ONE map (seq_db.rs:605-612): the records of job i lie
// behind those of job i - 1 in the index's append buffer.  How many those are is known on the device first, so the place is
// handed on THERE: a cursor word in device memory, set by the host when no index job is in flight and advanced by every job's
// pass behind its record kernel (stream order on the back stream); the pass reports the value it found.  A job is `direct`
// when the cursor chain is intact and the buffer has room for the largest list the job can produce -- its records are in
// place when it is collected, no copy.  Anything else is `staged`: the records go to the lane's own buffer and are copied into
// the index at the host-known offset when the job is collected.  A direct job that is finished synchronously (flagged tile,
// undersized estimate) first waits for the other job in flight (its optimistic pass placed its records behind a count that
// is about to change), writes its records at the host-known offset, and the other job, reporting a cursor value that is not
// the index's count when it is collected, writes its records again at the right place.
struct pgr_pipe {
    pgr_ctx *ctx = nullptr;
    pgr_spec spec = {};
    struct Slot {
        Lane *lane_p = nullptr;  // workspaces, mailbox, timing events of the job in this slot (swapped into the context while it is
                                 // worked on); from the context's spare lanes, and back there when the pipe goes
        std::unique_ptr<ShmmrJob> job;
        hipEvent_t ev_front = nullptr, ev_done = nullptr;
        uint32_t *sids = nullptr;        // PINNED: source of an asynchronous H2D copy on the back stream (a copy from pageable memory
        size_t sids_cap = 0;             // would block the host until that stream gets to it, i.e. until the job's tiles are done)
        pgr_index *ix = nullptr;
        pgr_frag_rec *d_recs = nullptr;  // the caller's record buffer (ix == NULL)
        uint64_t recs_cap = 0;
        uint64_t dst_cap = 0;            // capacity of the buffer the last pass wrote the records to
        uint64_t *pmail = nullptr;       // pinned: [0] = pair records the job's last pass counted, [1] = cursor value it found (direct),
                                         // [2] = the cursor's start value (source of an H2D copy)
        // a QUERY job (pgr_pipe_submit_query): behind the queries' shimmer pass, on the back stream, pair records + the per-query
        // kernel + packing + the download of the chains (csrc/query_fused.hip); whatever that path cannot take -- a flagged query
        // batch, long queries, a repeat key -- is answered at collect by the synchronous call (q_fallback)
        bool is_query = false, q_fallback = false;
        bool q_level1 = false;           // ... in its level-1 form: tiles on the context's stream, the per-query kernel (which runs the list stage of
                                         // its own query) + packing on the back stream, no list stage of the batch (pgr_aln.h: QfLevel1View)
        std::unique_ptr<QueryFusedRun> qrun;
        const pgr_index *qix = nullptr;
        const pgr_batch *qb = nullptr;
        QParams qqp = {};
        AlnParams qap = {};
        uint64_t *qmail = nullptr;       // pinned: the totals of the slot's query run
        hipEvent_t q_ev_packed = nullptr, q_ev_copied = nullptr;  // its download runs on the fix stream behind the packing
        bool has_sids = false;
        bool direct = false;             // records straight into the index's append buffer, placed by the device cursor
        bool placed_by_host = false;     // a synchronous pass wrote the records at the index's host-known count
        uint64_t reserved = 0;           // direct: list capacity of the optimistic pass = upper bound of its records
    } slot[2];
    std::deque<int> order;  // slots in flight, oldest first
    int next = 0;
    bool commit_pending = false;  // copies into an index are queued on the back stream
    hipEvent_t ev_commit = nullptr;  // ... recorded behind the last of them: a pass on another stream that rewrites the lane's record
                                     // buffer (the copy's source) waits for it
    uint64_t *d_cursor = nullptr; // device: [0] = where the next direct job's records go in chain_ix's append buffer
    pgr_index *chain_ix = nullptr;
    bool chain_ok = false;        // every index job in flight is direct on chain_ix (the cursor is what the host would compute)
    uint64_t chain_reserved = 0;  // sum of `reserved` of the direct jobs in flight

    int index_jobs_in_flight() const {
        int k = 0;
        for (int i : order) k += slot[i].ix != nullptr;
        return k;
    }
};

namespace {
struct LaneScope {
    pgr_ctx *ctx;
    Lane &lane;
    hipStream_t saved_alloc;
    LaneScope(pgr_ctx *c, Lane &l, hipStream_t alloc) : ctx(c), lane(l), saved_alloc(c->alloc_stream) {
        ctx->swap_lane(lane);
        ctx->alloc_stream = alloc;
    }
    ~LaneScope() {
        ctx->swap_lane(lane);
        ctx->alloc_stream = saved_alloc;
    }
};

}  // namespace
void pgr::lane_release(pgr_ctx *ctx, Lane &l) {
    DevBuf *bufs[] = {&l.ws_tile_first, &l.ws_seg_off, &l.ws_seg_cnt, &l.ws_seg_dst, &l.ws_cursor, &l.ws_flags, &l.ws_l1, &l.ws_serial,
                      &l.ws_scan_tmp, &l.ws_list_a, &l.ws_list_b, &l.ws_off_a, &l.ws_off_b, &l.ws_blk_cnt, &l.ws_blk_base,
                      &l.ws_start_rank, &l.ws_rids, &l.ws_rec_off, &l.ws_blk_off, &l.ws_tile_desc, &l.ws_tile_flags, &l.ws_seg_cid,
                      &l.ws_tile_lv, &l.ws_recs};
    for (DevBuf *b : bufs) b->release(ctx);
    if (l.mailbox) (void)hipHostFree(l.mailbox);
    l.mailbox = nullptr;
    l.mailbox_cap = 0;
    for (auto &e : l.ev)
        if (e) {
            (void)hipEventDestroy(e);
            e = nullptr;
        }
    if (l.ev_end) (void)hipEventDestroy(l.ev_end);
    l.ev_end = nullptr;
}
namespace {

// the consumer behind stage 4 of a pipelined job: index-side pair records (seq_db.rs:381-400) from the device list; counts and
// offsets stay on the device.  Called again by every repeated pass (then synchronously, on the context's stream).
int pipe_records(pgr_pipe *p, pgr_pipe::Slot *sp, uint32_t n, hipStream_t st, const pgr_mm128 *d_list, const uint64_t *d_off, uint64_t cap,
                 const uint64_t *d_count) {
    pgr_ctx *ctx = p->ctx;
    int r;
    if ((r = ctx->ws_rec_off.ensure(ctx, ((size_t)n + 1) * sizeof(uint64_t)))) return r;
    // (the sids are on the device already: pgr_pipe_submit sent them on the context's stream, in front of the job's tiles)
    uint32_t *d_sids = (sp->has_sids && n) ? (uint32_t *)ctx->ws_rids.p : nullptr;
    uint64_t *rec_off = (uint64_t *)ctx->ws_rec_off.p;
    const bool optimistic = sp->job && sp->job->optimistic;
    const uint32_t lds = optimistic ? sp->job->lds_match : 0;
    if (sp->ix && sp->direct && optimistic) {
        // behind the predecessor's records, wherever they end: the device knows
        pgr_index *ix = sp->ix;
        sp->dst_cap = ix->cap_raw;
        launch_frag_recs_dev(st, d_list, d_off, n, cap, d_count, 0, rec_off, ix->raw, ix->cap_raw, nullptr, d_sids, p->d_cursor, lds);
        launch_cursor_bump(st, p->d_cursor, rec_off + n, p->d_cursor + 1);
        PGR_HIP(ctx, hipMemcpyAsync(sp->pmail + 1, p->d_cursor + 1, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    } else if (sp->ix && sp->direct) {
        // a synchronous pass of a direct job: every job submitted before it has been collected, the index's count is exact, and the
        // other job in flight has been waited for (pgr_pipe_collect)
        pgr_index *ix = sp->ix;
        hipStream_t saved_alloc = ctx->alloc_stream;
        ctx->alloc_stream = nullptr;  // (the index's block belongs to the context's stream, whichever stream this pass runs on)
        int rc = index_grow_raw(ctx, ix, ix->n_raw + cap);
        ctx->alloc_stream = saved_alloc;
        if (rc) return rc;
        ctx->block_on_back(ix->raw);
        if (st == ctx->fix_stream) ctx->block_on_fix(ix->raw);  // (this pass writes it from the fix stream: a free remembers that stream too)
        sp->dst_cap = ix->cap_raw - ix->n_raw;
        launch_frag_recs_dev(st, d_list, d_off, n, cap, d_count, 0, rec_off, ix->raw + ix->n_raw, sp->dst_cap, nullptr, d_sids, nullptr, lds);
        sp->placed_by_host = true;
    } else {
        pgr_frag_rec *dst = sp->d_recs;
        sp->dst_cap = sp->recs_cap;
        if (sp->ix) {  // staged (the list holds at most `cap` shimmers, hence fewer pairs)
            if ((r = ctx->ws_recs.ensure(ctx, std::max<uint64_t>(cap, 1) * sizeof(pgr_frag_rec)))) return r;
            dst = (pgr_frag_rec *)ctx->ws_recs.p;
            sp->dst_cap = cap;
        }
        launch_frag_recs_dev(st, d_list, d_off, n, cap, d_count, 0, rec_off, dst, sp->dst_cap, nullptr, d_sids, nullptr, lds);
    }
    PGR_HIP(ctx, hipMemcpyAsync(sp->pmail, rec_off + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    return PGR_OK;
}
}  // namespace

extern "C" int pgr_pipe_create(pgr_ctx *ctx, const pgr_spec *spec, pgr_pipe **out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!out) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    int rc = check_spec(ctx, spec);
    if (rc) return rc;
    PGR_ENTER(ctx);
    if ((rc = ctx->enable_multi_stream())) return rc;
    pgr_pipe *p = new pgr_pipe();
    p->ctx = ctx;
    p->spec = *spec;
    hipError_t e = hipSuccess;
    for (auto &s : p->slot) {
        if (!ctx->spare_lanes.empty()) {
            s.lane_p = ctx->spare_lanes.back();
            ctx->spare_lanes.pop_back();
        } else {
            s.lane_p = new Lane();
            for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&s.lane_p->ev[i]);
            if (e == hipSuccess) e = hipEventCreate(&s.lane_p->ev_end);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_front, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s.pmail, 64, hipHostMallocDefault);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_cursor, 64);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_commit, hipEventDisableTiming);
    ++ctx->n_pipes;
    if (e != hipSuccess) {
        const std::string msg = std::string("pipe set-up: ") + hipGetErrorString(e);
        pgr_pipe_destroy(p);
        return ctx->fail(PGR_ERR_DEVICE, msg);
    }
    *out = p;
    return PGR_OK;
}

extern "C" int pgr_pipe_in_flight(const pgr_pipe *p) { return p ? (int)p->order.size() : 0; }

extern "C" int pgr_pipe_submit(pgr_pipe *p, const pgr_batch *b, const uint32_t *sids, pgr_index *ix, pgr_frag_rec *d_recs,
                               uint64_t recs_capacity) {
    if (!p) return PGR_ERR_INVALID_ARG;
    pgr_ctx *ctx = p->ctx;
    if (!b) return ctx->fail(PGR_ERR_INVALID_ARG, "null batch");
    if (b->ctx != ctx) return ctx->fail(PGR_ERR_STATE, "batch belongs to another context");
    if (ix && d_recs) return ctx->fail(PGR_ERR_INVALID_ARG, "pair records go to an index OR to a caller's buffer");
    if (ix && ix->ctx != ctx) return ctx->fail(PGR_ERR_STATE, "index belongs to another context");
    if (ix && memcmp(&ix->spec, &p->spec, sizeof(pgr_spec)) != 0) return ctx->fail(PGR_ERR_INVALID_ARG, "the index has another spec than the pipe");
    if (p->order.size() >= 2) return ctx->fail(PGR_ERR_STATE, "two jobs are in flight: collect one first");
    PGR_ENTER(ctx);
    pgr_pipe::Slot &s = p->slot[p->next];
    const uint32_t n = b->n;
    s.is_query = s.q_fallback = false;
    s.ix = ix;
    s.d_recs = d_recs;
    s.recs_cap = d_recs ? recs_capacity : 0;
    s.direct = false;
    s.placed_by_host = false;
    s.reserved = 0;
    s.has_sids = sids != nullptr || ix != nullptr;
    if (s.has_sids) {
        if (s.sids_cap < std::max<uint32_t>(n, 1)) {
            if (s.sids) (void)hipHostFree(s.sids);
            s.sids = nullptr;
            s.sids_cap = 0;
            const size_t want = std::max<size_t>((size_t)n + n / 4, 1024);
            if (hipHostMalloc((void **)&s.sids, want * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
                s.sids = nullptr;
                return ctx->fail(PGR_ERR_NOMEM, "hipHostMalloc of the sid block failed");
            }
            s.sids_cap = want;
        }
        if (sids) memcpy(s.sids, sids, (size_t)n * sizeof(uint32_t));
        else
            for (uint32_t i = 0; i < n; ++i) s.sids[i] = ix->next_sid + i;  // load_index_from_reader: running sid (seq_db.rs:543-553)
    }
    uint32_t next_sid_after = ix ? ix->next_sid : 0;  // (assigned once the job is in flight: a failed submit leaves no hole in the running sids)
    if (ix)
        for (uint32_t i = 0; i < n; ++i) next_sid_after = std::max(next_sid_after, s.sids[i] + 1);
    const bool want_recs = ix != nullptr || d_recs != nullptr;
    const int ix_jobs = p->index_jobs_in_flight();
    // (debug_poison fills the lane's workspaces when the job is planned: a staged copy out of its record buffer may still be queued)
    if (ctx->opt.debug_poison && p->commit_pending) (void)hipStreamSynchronize(ctx->back_stream);
    LaneScope scope(ctx, *s.lane_p, ctx->back_stream);
    s.job.reset(new ShmmrJob());
    ShmmrJob &job = *s.job;
    job.ctx = ctx;
    job.b = b;
    job.spec = p->spec;
    job.rids = nullptr;
    job.padding = 0;  // index path: padding = false (seq_db.rs:462)
    job.sf = ctx->stream;
    job.sb = ctx->back_stream;
    job.ev_front = s.ev_front;
    job.optimistic = true;
    job.dbg_t = ctx->opt.debug_times != 0;
    job.dbg_t0 = std::chrono::steady_clock::now();
    pgr_pipe::Slot *sp = &s;
    if (want_recs)
        job.post = [p, sp, n](hipStream_t st, const pgr_mm128 *d_list, const uint64_t *d_off, uint64_t cap, const uint64_t *d_count) -> int {
            return pipe_records(p, sp, n, st, d_list, d_off, cap, d_count);
        };
    s.pmail[0] = 0;
    s.pmail[1] = ~0ull;
    int rc;
    hipError_t e = hipSuccess;
    if (!(rc = job.plan()) && !(rc = job.begin_result())) {
        if (ix && job.stage1_only) p->chain_ok = false;  // (no records before collect: staged, and so is whoever comes behind)
        if (ix && !ctx->opt.pipe_staged_records && !job.stage1_only) {
            // direct: the chain starts here (nothing of an index in flight: the host knows the count) or continues on this index,
            // and the append buffer has room for everything the jobs in flight can produce (a job's list capacity bounds its pairs)
            const bool start = ix_jobs == 0;
            if (start) {
                p->chain_ix = ix;
                p->chain_ok = true;
                p->chain_reserved = 0;
            }
            if (p->chain_ok && p->chain_ix == ix && !ix->raw && start) rc = index_grow_raw(ctx, ix, job.cap_res);  // (first batch of an index without reserve)
            if (!rc && p->chain_ok && p->chain_ix == ix && ix->n_raw + p->chain_reserved + job.cap_res <= ix->cap_raw) {
                s.direct = true;
                s.reserved = job.cap_res;
                p->chain_reserved += s.reserved;
                ctx->block_on_back(ix->raw);
                if (start) {
                    s.pmail[2] = ix->n_raw;
                    e = hipMemcpyAsync(p->d_cursor, s.pmail + 2, sizeof(uint64_t), hipMemcpyHostToDevice, job.sb);
                }
            } else {
                p->chain_ok = false;  // a staged job in flight: the cursor no longer says where the next one's records go
            }
        }
        if (!rc && e == hipSuccess && s.has_sids && n && !(rc = ctx->ws_rids.ensure(ctx, (size_t)n * sizeof(uint32_t))))
            e = hipMemcpyAsync(ctx->ws_rids.p, s.sids, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, job.sf);
        if (!rc && e == hipSuccess) e = hipEventRecord(ctx->ev[0], job.sf);
        if (!rc && e == hipSuccess) rc = job.enqueue_pass();
        if (e == hipSuccess && !rc) e = hipEventRecord(s.ev_done, job.list_pending ? job.sf : job.sb);
    }
    if (rc || e != hipSuccess) {  // nothing of a failed submission may still be running on the lane's buffers
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->back_stream);
        s.job.reset();
        if (s.direct) p->chain_ok = false;
        return rc ? rc : ctx->fail(PGR_ERR_DEVICE, std::string("pipe submit: ") + hipGetErrorString(e));
    }
    if (ix) {
        ix->next_sid = next_sid_after;
        ++ix->pipe_jobs;
    }
    p->order.push_back(p->next);
    p->next ^= 1;
    return PGR_OK;
}

extern "C" int pgr_pipe_collect(pgr_pipe *p, pgr_shmmrs **out, uint64_t *n_pairs) {
    if (!p) return PGR_ERR_INVALID_ARG;
    pgr_ctx *ctx = p->ctx;
    if (out) *out = nullptr;
    if (n_pairs) *n_pairs = 0;
    if (p->order.empty()) return ctx->fail(PGR_ERR_STATE, "no job in flight");
    PGR_ENTER(ctx);
    if (p->slot[p->order.front()].is_query) return ctx->fail(PGR_ERR_STATE, "the oldest job is a query job: pgr_pipe_collect_query");
    pgr_pipe::Slot &s = p->slot[p->order.front()];
    p->order.pop_front();
    if (s.ix && s.ix->pipe_jobs > 0) --s.ix->pipe_jobs;
    std::unique_ptr<ShmmrJob> job = std::move(s.job);
    int rc = PGR_OK;
    pgr_shmmrs *res = nullptr;
    uint64_t np = 0;
    // the other job in flight has placed (or will place) its records behind this job's optimistic count: before this job's
    // records move, or its count changes, that pass has to be over -- and its cursor value will no longer be the index's count
    auto settle_others = [&]() {
        for (int i : p->order)
            if (p->slot[i].ix) (void)hipEventSynchronize(p->slot[i].ev_done);
        p->chain_ok = false;
    };
    {
        // what follows the optimistic pass is a synchronous call on the context's stream (the lane's buffers are free: the pass is over)
        LaneScope scope(ctx, *s.lane_p, nullptr);
        if (hipEventSynchronize(s.ev_done) != hipSuccess || hipGetLastError() != hipSuccess)
            rc = ctx->fail(PGR_ERR_DEVICE, "pipelined pass failed on the device");
        // A second pass (flagged tiles -> islands + the list stage again; an undersized estimate) runs on the fix stream when there is
        // one: beside the next job's tiles (the context's stream) and not behind that job's list stage (the back stream, which waits
        // for those tiles).  A batch of a real assembly is flagged every time -- gaps, (AT)n microsatellites longer than k --: on the
        // context's stream every job's islands and second list stage queued behind the next job's tile kernel, and the pipe ran at
        // the speed of the synchronous calls.
        hipStream_t fix = (ctx->fix_stream && !ctx->opt.no_fix_stream) ? ctx->fix_stream : ctx->stream;
        job->sf = job->sb = fix;
        ctx->alloc_stream = fix == ctx->stream ? nullptr : fix;
        // (a staged copy out of this lane's record buffer may still be queued on the back stream -- the lane's previous job was collected
        // without a second pass --: whatever this pass writes there comes behind it)
        if (!rc && p->commit_pending && hipStreamWaitEvent(fix, p->ev_commit, 0) != hipSuccess) rc = ctx->fail(PGR_ERR_DEVICE, "event wait failed");
        job->optimistic = false;
        bool done = false;
        // (a direct job whose records are about to move: the other job's pass has placed its own behind them -- waited for BEFORE
        // the islands are enqueued only when everything shares the context's stream; otherwise behind them, see below)
        if (!rc) rc = job->decide(done);
        if (!rc && !done && s.direct) settle_others();
        while (!rc && !done) {
            rc = job->enqueue_pass();
            if (rc == JOB_RESTART) {
                rc = PGR_OK;
                continue;
            }
            if (!rc && (hipStreamSynchronize(fix) != hipSuccess || hipGetLastError() != hipSuccess))
                rc = ctx->fail(PGR_ERR_DEVICE, "pipeline failed on the device");
            if (!rc) rc = job->decide(done);
        }
        ctx->alloc_stream = nullptr;
        const bool want_recs = s.ix != nullptr || s.d_recs != nullptr;
        if (!rc && want_recs) np = s.pmail[0];
        if (!rc) rc = job->finish(&res);
        if (!rc && s.ix && s.direct) {
            pgr_index *ix = s.ix;
            p->chain_reserved -= std::min(p->chain_reserved, s.reserved);
            const bool in_place = s.placed_by_host ? np <= s.dst_cap : (s.pmail[1] == ix->n_raw && ix->n_raw + np <= ix->cap_raw);
            if (!in_place && np) {
                // placed behind a count that was not the final one (a predecessor was finished synchronously), or the buffer was
                // too small: once more, at the index's count, from the finished list
                settle_others();
                if (!(rc = index_grow_raw(ctx, ix, ix->n_raw + np))) {
                    ctx->block_on_back(ix->raw);
                    std::vector<uint32_t> sid_copy(s.sids, s.sids + res->n);
                    rc = shmmrs_to_frag_recs_enqueue(ctx, res, sid_copy.data(), 0, ix->raw + ix->n_raw, ix->cap_raw - ix->n_raw);
                    if (!rc && (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess))
                        rc = ctx->fail(PGR_ERR_DEVICE, "pair records failed on the device");
                }
            }
            if (!rc) {
                ix->n_raw += np;
                ix->finalized = false;
            }
        } else if (!rc && want_recs) {
            if (np > s.dst_cap) rc = ctx->fail(PGR_ERR_INVALID_ARG, "output buffer too small for the pair records");
            if (!rc && s.ix && np) {
                // staged: one copy on the back stream, behind the pass that wrote the records and in front of the next job of this lane
                pgr_index *ix = s.ix;
                if (ix->n_raw + np > ix->cap_raw) {
                    if (p->commit_pending) (void)hipStreamSynchronize(ctx->back_stream);  // (copies into the block that is about to move)
                    p->commit_pending = false;
                    rc = index_grow_raw(ctx, ix, ix->n_raw + np);
                }
                ctx->block_on_back(ix->raw);
                if (!rc && hipMemcpyAsync(ix->raw + ix->n_raw, ctx->ws_recs.p, np * sizeof(pgr_frag_rec), hipMemcpyDeviceToDevice,
                                          ctx->back_stream) != hipSuccess)
                    rc = ctx->fail(PGR_ERR_DEVICE, "copy of the pair records into the index failed");
                if (!rc && hipEventRecord(p->ev_commit, ctx->back_stream) != hipSuccess) rc = ctx->fail(PGR_ERR_DEVICE, "event record failed");
                if (!rc) {
                    ix->n_raw += np;
                    ix->finalized = false;
                    p->commit_pending = true;
                }
            }
        }
    }
    job.reset();
    if (p->order.empty() && p->commit_pending) {  // the pipe has run dry: whoever uses the index next finds its records in place
        if (hipStreamSynchronize(ctx->back_stream) != hipSuccess && !rc) rc = ctx->fail(PGR_ERR_DEVICE, "back stream failed");
        p->commit_pending = false;
    }
    if (rc) {
        if (res) pgr_shmmrs_destroy(res);
        return rc;
    }
    if (n_pairs) *n_pairs = np;
    if (out) *out = res;
    else pgr_shmmrs_destroy(res);
    return PGR_OK;
}

// ------------------------------------------------------------------------------------------------
// Query batches through the pipe.  One batch of queries is a chain: a VALU-bound tile kernel, a dozen latency-bound kernels, the
// per-query kernel, and 7.5 MB of chains across PCIe (10 000 x 10 kbp: 0.69 ms, of which the tiles are 0.28 and the download
// 0.14).  With two batches in flight the tiles of batch i + 1 (the context's stream) run beside everything behind the tiles of
// batch i (the back stream) -- the reference loops over its queries with rayon (pgr-bin/src/bin/pgr-query.rs:135-165).  The
// answer of every batch is the synchronous call's: what the chained path cannot take (a flagged shimmer pass, long queries, a
// repeat key, a slot too small) is answered by that very call at collect.
extern "C" int pgr_pipe_submit_query(pgr_pipe *p, const pgr_batch *b, const pgr_index *ix, float penalty, uint32_t max_count,
                                     uint32_t max_count_query, uint32_t max_count_target, uint32_t max_aln_span, int has_max_gap,
                                     uint32_t max_gap, int oriented) {
    if (!p) return PGR_ERR_INVALID_ARG;
    pgr_ctx *ctx = p->ctx;
    if (!b || !ix) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (b->ctx != ctx) return ctx->fail(PGR_ERR_STATE, "batch belongs to another context");
    if (!ix->finalized) return ctx->fail(PGR_ERR_STATE, "index not finalized (call pgr_index_finalize)");
    if (memcmp(&ix->spec, &p->spec, sizeof(pgr_spec)) != 0) return ctx->fail(PGR_ERR_INVALID_ARG, "the index has another spec than the pipe");
    if (max_aln_span == 0) return ctx->fail(PGR_ERR_INVALID_ARG, "max_aln_span must be at least 1");
    // (two in flight, as for shimmer jobs.  Three were tried -- with two, the tiles of batch i + 2 are enqueued only when batch i's
    // chains are home, so the cycle is the chain behind a batch's tiles, ~0.42 ms, not the tiles' 0.3 -- and measured SLOWER: 0.54-0.62
    // ms per batch against 0.45-0.47: three lanes' allocations and cross-stream waits cost more than the idle front stream)
    if (p->order.size() >= 2) return ctx->fail(PGR_ERR_STATE, "two jobs are in flight: collect one first");
    PGR_ENTER(ctx);
    pgr_pipe::Slot &s = p->slot[p->next];
    const uint32_t n = b->n;
    s.is_query = true;
    s.q_fallback = false;
    s.ix = nullptr;
    s.d_recs = nullptr;
    s.direct = s.placed_by_host = s.has_sids = false;
    s.qix = ix;
    s.qb = b;
    s.qqp = QParams{max_count, max_count_query, max_count_target};
    s.qap = AlnParams{max_aln_span, penalty, has_max_gap, max_gap, oriented};
    s.qrun.reset();
    s.job.reset();
    // the same test as pgr_query_hps_resident's: the chained path takes batches of short queries on an index that has seen one
    uint32_t pairs_hint = ix->fused_pairs.load(std::memory_order_relaxed);
    if (!pairs_hint && n) {
        uint32_t max_len = 0;
        for (uint32_t c = 0; c < n; ++c) max_len = std::max(max_len, b->h_len[c]);
        const pgr_spec &sp = ix->spec;
        const double keep = sp.r > 1 ? 2.0 / (double)(sp.r + 1) : 1.0;
        const double dens = sp.sketch ? 1.0 / (double)(1ull << (4 + sp.r)) : 2.0 / (double)(sp.w + 1) * keep * keep;
        pairs_hint = (uint32_t)std::min<double>((double)max_len * dens * 1.6 + 4.0, 1e9);
    }
    // (a spec without a tile path needs the exact machine for every query: the synchronous call)
    const bool chained = n && ix->n && (ix->spec.sketch || ix->spec.w >= (uint32_t)L1_MIN_W) &&
                         ix->fused_skip.load(std::memory_order_relaxed) == 0 && pairs_hint &&
                         query_fused_eligible(ctx, n, pairs_hint, max_aln_span) && !ctx->opt.no_query_chaining;
    if (!chained) {
        s.q_fallback = true;
        p->order.push_back(p->next);
        p->next ^= 1;
        return PGR_OK;
    }
    if (!s.qmail) {
        if (hipHostMalloc((void **)&s.qmail, 256, hipHostMallocDefault) != hipSuccess) {
            s.qmail = nullptr;
            return ctx->fail(PGR_ERR_NOMEM, "hipHostMalloc of the query mailbox failed");
        }
    }
    memset(s.qmail, 0, 256);
    LaneScope scope(ctx, *s.lane_p, ctx->back_stream);
    s.qrun.reset(new QueryFusedRun(ctx, ix, n, pairs_hint, s.qqp, s.qap));
    s.qrun->stream = ctx->back_stream;
    s.qrun->mail = s.qmail;
    if (ctx->fix_stream && !ctx->opt.no_fix_stream) {  // (a stream that runs beside both others: the chains go home while the back stream works on)
        if (!s.q_ev_packed && hipEventCreateWithFlags(&s.q_ev_packed, hipEventDisableTiming) != hipSuccess) s.q_ev_packed = nullptr;
        if (!s.q_ev_copied && hipEventCreateWithFlags(&s.q_ev_copied, hipEventDisableTiming) != hipSuccess) s.q_ev_copied = nullptr;
        if (s.q_ev_packed && s.q_ev_copied) {
            s.qrun->copy_stream = ctx->fix_stream;
            s.qrun->ev_packed = s.q_ev_packed;
            s.qrun->ev_copied = s.q_ev_copied;
        }
    }
    s.job.reset(new ShmmrJob());
    ShmmrJob &job = *s.job;
    job.ctx = ctx;
    job.b = b;
    job.spec = p->spec;
    job.rids = nullptr;
    job.padding = 0;
    job.sf = ctx->stream;
    job.sb = ctx->back_stream;
    job.ev_front = s.ev_front;
    job.optimistic = true;
    job.dbg_t = ctx->opt.debug_times != 0;
    job.dbg_t0 = std::chrono::steady_clock::now();
    QueryFusedRun *run = s.qrun.get();
    int rc = PGR_OK;
    hipError_t e = hipSuccess;
    // the level-1 form (the same test as pgr_query_hps_resident's): the tiles on the context's stream, the per-query kernel on their
    // segments + offsets + packing on the back stream behind an event, the chains' download on the fix stream
    s.q_level1 = false;
    if (!ctx->opt.no_query_level1 && ix->fused_l1_skip.load(std::memory_order_relaxed) == 0) {
        uint32_t max_len = 0;
        for (uint32_t c = 0; c < n; ++c) max_len = std::max(max_len, b->h_len[c]);
        const uint32_t c1 = query_fused_level1_cap(max_len, ix->spec.w);
        if (c1) {
            QfLevel1View v;
            bool taken = false;
            rc = level1_stage_and_view(job, v, &taken);
            if (!rc && taken) {
                e = hipEventRecord(s.ev_front, job.sf);
                if (e == hipSuccess) e = hipStreamWaitEvent(job.sb, s.ev_front, 0);
                if (e == hipSuccess) rc = run->enqueue_from_level1(v, c1);
                if (e == hipSuccess && !rc) e = hipEventRecord(s.ev_done, job.sb);
                if (e == hipSuccess && !rc) {
                    s.q_level1 = true;
                    p->order.push_back(p->next);
                    p->next ^= 1;
                    return PGR_OK;
                }
            }
            if (rc || e != hipSuccess) {
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipStreamSynchronize(ctx->back_stream);
                s.qrun.reset();
                s.job.reset();
                s.is_query = false;
                return rc ? rc : ctx->fail(PGR_ERR_DEVICE, std::string("pipe submit (query, level-1 form): ") + hipGetErrorString(e));
            }
        }
    } else if (const uint32_t left = ix->fused_l1_skip.load(std::memory_order_relaxed)) {
        if (!ctx->opt.no_query_level1) ix->fused_l1_skip.store(left - 1, std::memory_order_relaxed);
    }
    job.post = [run](hipStream_t, const pgr_mm128 *d_list, const uint64_t *d_off, uint64_t cap, const uint64_t *d_count) -> int {
        return run->enqueue_from_shimmers(d_list, d_off, cap, d_count);
    };
    if (!(rc = job.plan()) && !(rc = job.begin_result())) {
        job.stage1_only = false;  // (the per-query kernel rides behind the list stage: always the whole pass)
        e = hipEventRecord(ctx->ev[0], job.sf);
        if (e == hipSuccess) rc = job.enqueue_pass();
        if (e == hipSuccess && !rc) e = hipEventRecord(s.ev_done, job.sb);
    }
    if (rc || e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->back_stream);
        s.qrun.reset();
        s.job.reset();
        s.is_query = false;
        return rc ? rc : ctx->fail(PGR_ERR_DEVICE, std::string("pipe submit (query): ") + hipGetErrorString(e));
    }
    p->order.push_back(p->next);
    p->next ^= 1;
    if (ctx->opt.debug_times)
        fprintf(stderr, "[pgr] pipe submit_query took %.1f us\n",
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - s.job->dbg_t0).count());
    return PGR_OK;
}

extern "C" int pgr_pipe_collect_query(pgr_pipe *p, pgr_hps_result *out) {
    if (!p) return PGR_ERR_INVALID_ARG;
    pgr_ctx *ctx = p->ctx;
    if (!out) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    memset(out, 0, sizeof(*out));
    if (p->order.empty()) return ctx->fail(PGR_ERR_STATE, "no job in flight");
    if (!p->slot[p->order.front()].is_query) return ctx->fail(PGR_ERR_STATE, "the oldest job is not a query job: pgr_pipe_collect");
    PGR_ENTER(ctx);
    pgr_pipe::Slot &s = p->slot[p->order.front()];
    p->order.pop_front();
    s.is_query = false;
    std::unique_ptr<ShmmrJob> job = std::move(s.job);
    std::unique_ptr<QueryFusedRun> run = std::move(s.qrun);
    bool fallback = s.q_fallback;
    int rc = PGR_OK;
    const auto tq0 = std::chrono::steady_clock::now();
    auto qlap = [&](const char *what) {
        if (ctx->opt.debug_times)
            fprintf(stderr, "[pgr] pipe collect_query %-28s at %7.1f us\n", what,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tq0).count());
    };
    if (!fallback) {
        // whatever this scope still enqueues (a repeated shimmer decision, the run's redo of the per-query kernel) runs on the back
        // stream: the blocks it takes and frees are the back stream's (round-5 advice: they were booked on the context's stream)
        LaneScope scope(ctx, *s.lane_p, ctx->back_stream);
        if (hipEventSynchronize(s.ev_done) != hipSuccess || hipGetLastError() != hipSuccess)
            rc = ctx->fail(PGR_ERR_DEVICE, "pipelined query pass failed on the device");
        qlap("back stream done");
        if (!rc && run && run->enqueued && run->copy_stream && hipEventSynchronize(run->ev_copied) != hipSuccess)
            rc = ctx->fail(PGR_ERR_DEVICE, "download of the chains failed");
        qlap("chains home");
        if (!rc && s.q_level1) {  // (no shimmer result to look at: the per-query kernel has seen the tile kernel's status words itself)
            QueryFusedCounts fc;
            bool declined = false;
            if (run->enqueued) rc = run->finish(out, &fc, &declined);
            else declined = true;
            if (!rc && declined) {
                memset(out, 0, sizeof(*out));
                if (run->l1_flagged) s.qix->fused_l1_skip.store(4, std::memory_order_relaxed);
                fallback = true;
            }
        }
        // a shimmer pass that needs anything more (flagged tiles, an undersized estimate): the synchronous call takes the batch
        bool done = false;
        if (!rc && !s.q_level1 && (job->mbox[1] || job->mbox[2])) fallback = true;
        if (!rc && !fallback && !s.q_level1) {
            job->sf = job->sb = ctx->back_stream;
            job->optimistic = false;
            rc = job->decide(done);
            if (!rc && !done) fallback = true;
        }
        if (!rc && !fallback && !s.q_level1) {
            pgr_shmmrs *res = nullptr;
            if (!(rc = job->finish(&res))) {
                uint64_t max_pairs = 0;
                for (uint32_t c = 0; c < res->n; ++c) max_pairs = std::max(max_pairs, res->h_off[c + 1] - res->h_off[c]);
                s.qix->fused_pairs.store((uint32_t)std::min<uint64_t>(std::max<uint64_t>(max_pairs, 1), 1u << 30), std::memory_order_relaxed);
                pgr_shmmrs_destroy(res);
                QueryFusedCounts fc;
                bool declined = false;
                if (run->enqueued) rc = run->finish(out, &fc, &declined);
                else declined = true;
                if (!rc && declined) {
                    memset(out, 0, sizeof(*out));
                    fallback = true;
                }
            }
        }
        qlap(fallback ? "for the synchronous call" : "result assembled");
        run.reset();  // (waits for its stream when its download may still be pending)
        job.reset();
        qlap("run and job released");
    }
    if (rc) return rc;
    if (fallback)
        return pgr_query_hps_resident(ctx, s.qix, s.qb, s.qap.penalty, s.qqp.max_count, s.qqp.max_count_query, s.qqp.max_count_target,
                                      s.qap.max_span, s.qap.has_max_gap, s.qap.max_gap, s.qap.oriented, out);
    return PGR_OK;
}

extern "C" void pgr_pipe_destroy(pgr_pipe *p) {
    if (!p) return;
    pgr_ctx *ctx = p->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->back_stream) (void)hipStreamSynchronize(ctx->back_stream);
    if (ctx->fix_stream) (void)hipStreamSynchronize(ctx->fix_stream);
    for (int i : p->order)  // jobs that were never collected
        if (!p->slot[i].is_query && p->slot[i].ix && p->slot[i].ix->pipe_jobs > 0) --p->slot[i].ix->pipe_jobs;
    for (auto &s : p->slot) {
        s.job.reset();
        if (s.lane_p) {
            bool complete = s.lane_p->ev_end != nullptr;
            for (auto &ev : s.lane_p->ev) complete = complete && ev != nullptr;
            if (complete) {
                ctx->spare_lanes.push_back(s.lane_p);  // (pgr_ctx_destroy / pgr_ctx_trim release them)
            } else {  // a lane of a pgr_pipe_create that failed half way: not for the next pipe
                pgr::lane_release(ctx, *s.lane_p);
                delete s.lane_p;
            }
        }
        s.lane_p = nullptr;
        if (s.ev_front) (void)hipEventDestroy(s.ev_front);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        s.qrun.reset();
        if (s.pmail) (void)hipHostFree(s.pmail);
        if (s.qmail) (void)hipHostFree(s.qmail);
        if (s.q_ev_packed) (void)hipEventDestroy(s.q_ev_packed);
        if (s.q_ev_copied) (void)hipEventDestroy(s.q_ev_copied);
        if (s.sids) (void)hipHostFree(s.sids);
    }
    if (p->d_cursor) (void)hipFree(p->d_cursor);
    if (p->ev_commit) (void)hipEventDestroy(p->ev_commit);
    if (ctx->n_pipes > 0 && --ctx->n_pipes == 0) ctx->multi_stream = false;  // (both streams are idle: one-stream rule again)
    delete p;
}
