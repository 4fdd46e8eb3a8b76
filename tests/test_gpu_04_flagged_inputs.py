"""GPU parity tests added in round 2: the single-synchronisation pipeline on flagged / overflowing inputs, the content
checksum, the resident query entry point, the C-ABI exchange (1 rank and 2 processes on one device), the strong-scaling
partitioner end to end, and BASELINE.json configs[4] as one GPU's slice."""
import os
import sys

import numpy as np
import pytest

import seqgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(ref, got):
    return len(ref) == len(got) and np.array_equal(ref["x"], got["x"]) and np.array_equal(ref["y"], got["y"])


def test_small_calls_take_the_optimistic_path_and_stay_exact(oracle, gpu_ctx):
    """batches below 64 Mbp run the whole pipeline before the host looks at any count; the rare cases that need a
    second pass (palindromic k-mers, non-ACGT bytes, denser lists than estimated) must come out exact all the same"""
    import pgrtk_amd as P
    rng = np.random.default_rng(5)
    sp_t = (80, 56, 4, 64)
    cases = {
        "clean": [seqgen.rnd(rng, 30_000), seqgen.rnd(rng, 777)],
        "palindrome": [seqgen.rnd(rng, 20_000) + b"AT" * 60 + seqgen.rnd(rng, 20_000)],
        "n_runs": [seqgen.rnd(rng, 9_000) + b"N" * 300 + seqgen.rnd(rng, 12_000), seqgen.rnd(rng, 5_000)],
        "low_complexity": [b"A" * 50_000, (b"ACGTTGCA" * 8000)],  # every position ties: far denser than the estimate
        "mixed": [seqgen.rnd(rng, 40_000), b"", b"ACGT" * 30, seqgen.rnd(rng, 100) + b"n" + seqgen.rnd(rng, 3000)],
    }
    for spec_t in (sp_t, (48, 56, 4, 12), (24, 24, 12, 24)):
        sp = oracle.spec(*spec_t)
        for name, seqs in cases.items():
            got = P.sequence_to_shmmrs_batch(seqs, P.make_spec(*spec_t), ctx=gpu_ctx)
            for i, s in enumerate(seqs):
                assert _same(oracle.sequence_to_shmmrs(i, s, sp), got[i]), (spec_t, name, i)
    # caller rids + a result far bigger than the estimate: the first gather leaves holes, the rid patch must not follow
    # their garbage (regression: memory access fault found by tools/fuzz_parity.py), the retry must be exact
    for seqs, rids in (([b"A" * 300_000, seqgen.rnd(rng, 5_000)], [2_000_000_000, 7]), ([b"ACAC" * 60_000], [4_000_000_000])):
        got = P.sequence_to_shmmrs_batch(seqs, P.make_spec(*sp_t), rids=rids, ctx=gpu_ctx)
        for i, s in enumerate(seqs):
            assert _same(oracle.sequence_to_shmmrs(rids[i], s, oracle.spec(*sp_t)), got[i]), ("rids", i)
    # the result-size estimate adapts to the previous call: a dense call after sparse ones, and back
    for seqs in ([seqgen.rnd(rng, 100_000)], [b"C" * 200_000], [seqgen.rnd(rng, 100_000)]):
        got = P.sequence_to_shmmrs_batch(seqs, P.make_spec(*sp_t), ctx=gpu_ctx)
        assert _same(oracle.sequence_to_shmmrs(0, seqs[0], oracle.spec(*sp_t)), got[0])


@pytest.mark.parametrize("early_bp", ["1000000", "100000000000"])
def test_large_batch_with_islands_is_exact(oracle, gpu_ctx, request, early_bp):
    """80 Mbp through both orchestrations: with the early look at the level-1 flags before the list stage (what batches of
    >= 1 Gbp do) and optimistically (list stage enqueued at once, repeated after the islands).  Non-ACGT bytes (single N, a
    300 kbp N run), palindromic (AT)n and a homopolymer stretch turn tiles into islands of the exact state machine,
    everything else stays on the closed-form tiles.  All contigs compared with the checker (128-bit checksums + counts)."""
    import pgrtk_amd as P
    gpu_ctx.set_option("early_sync_bp", int(early_bp))
    request.addfinalizer(lambda: gpu_ctx.set_option("early_sync_bp", 1 << 30))
    n, L = 8, 10_000_000
    seqs = []
    for i in range(n):
        s = oracle.synth_contig(9, i, L).copy()
        if i != 3:
            s[1234567 + i] = ord("N")
        if i % 4 == 0:
            s[5_000_000:5_300_000] = ord("N")
        if i == 2:
            s[7_000_000:7_000_200] = np.frombuffer(b"AT" * 100, dtype=np.uint8)
        if i == 5:
            s[2_000_000:2_050_000] = ord("C")
        seqs.append(s)
    # (a batch staged from ASCII knows from the host packer that it holds non-ACGT bytes and always looks at the flags early;
    # the optimistic orchestration is reached with packed input, where only the device knows)
    b = P.Batch.from_seqs(seqs, ctx=gpu_ctx) if early_bp == "1000000" else P.Batch.from_packed(P.pack_ascii(seqs)[0], ctx=gpu_ctx)
    sh = b.shmmrs(P.make_spec())
    prof = gpu_ctx.last_prof()
    assert prof.n_serial_contigs == 7 and 0 < prof.exact_bases < 0.2 * n * L  # islands, not whole contigs (contig 3 is clean)
    sums, off = sh.checksum(), sh.offsets()
    osp = oracle.spec()
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as pool:
        refs = list(pool.map(lambda i: oracle.sequence_to_shmmrs(i, seqs[i], osp), range(n)))
    for i in range(n):
        assert int(off[i + 1] - off[i]) == len(refs[i]), i
        assert np.array_equal(sums[i], oracle.shmmr_checksum(refs[i])), i


def test_multi_mbp_run_of_n_inside_a_contig(oracle, gpu_ctx):
    """a 7 Mbp run of N inside a 24 Mbp contig (reference chromosomes carry such gaps): fewer than a third of the tiles are
    irregular, so the run becomes an island of ~215 chunks; inside the run the machine carries the k-mer from before it, every
    seam has to be corrected with the true state of the chunk in front, one per round -- found by the fuzzer when the number
    of rounds was capped at 200"""
    import pgrtk_amd as P
    s = oracle.synth_contig(21, 0, 24_000_000).copy()
    s[9_000_000:16_000_000] = ord("N")
    t = oracle.synth_contig(21, 1, 3_000_000).copy()
    sh = P.Batch.from_seqs([s, t], ctx=gpu_ctx).shmmrs(P.make_spec())
    sums, off = sh.checksum(), sh.offsets()
    for i, q in enumerate((s, t)):
        ref = oracle.sequence_to_shmmrs(i, q, oracle.spec())
        assert int(off[i + 1] - off[i]) == len(ref), i
        assert np.array_equal(sums[i], oracle.shmmr_checksum(ref)), i


def test_long_palindromic_stretch_stays_an_island(oracle, gpu_ctx):
    """120 kbp of (AT)n and 70 kbp of (ACGT)n inside 6 Mbp contigs: every k-mer there is its own reverse complement and is not
    pushed (shmmrutils.rs:477-480), so the ring buffer keeps what was pushed in FRONT of the stretch and no warm-up inside it can
    rebuild that.  The chunks of the island hand their ring over (one per round); before, the whole contig fell back to ONE
    serial chunk (1.2 s for 30 Mbp, found with tools/repeat_like_bench.py)"""
    import pgrtk_amd as P
    seqs = []
    for i, (unit, n) in enumerate(((b"AT", 120_000), (b"ACGT", 70_000))):
        s = oracle.synth_contig(23, i, 6_000_000).copy()
        rep = np.frombuffer(unit * (n // len(unit)), dtype=np.uint8)
        s[2_500_000:2_500_000 + len(rep)] = rep
        seqs.append(s)
    sh = P.Batch.from_seqs(seqs, ctx=gpu_ctx).shmmrs(P.make_spec())
    prof = gpu_ctx.last_prof()
    assert prof.n_serial_contigs == 2 and prof.exact_bases < 600_000  # islands around the stretches, not whole contigs
    sums, off = sh.checksum(), sh.offsets()
    for i, q in enumerate(seqs):
        ref = oracle.sequence_to_shmmrs(i, q, oracle.spec())
        assert int(off[i + 1] - off[i]) == len(ref), i
        assert np.array_equal(sums[i], oracle.shmmr_checksum(ref)), i


def test_shmmrs_checksum_matches_the_checker(oracle, gpu_ctx):
    import pgrtk_amd as P
    lens = [300_000, 0, 5_000, 1_000_000, 80]
    b = P.Batch.synthetic(lens, seed=9, contig0=40, ctx=gpu_ctx)
    sh = b.shmmrs(P.make_spec())
    sums = sh.checksum()
    off = sh.offsets()
    sp = oracle.spec()
    for i, L in enumerate(lens):
        ref = oracle.sequence_to_shmmrs(i, oracle.synth_contig(9, 40 + i, L), sp)
        assert int(off[i + 1] - off[i]) == len(ref)
        assert np.array_equal(sums[i], oracle.shmmr_checksum(ref)), i
    # the threaded checker (what bench.py runs over all 1000 contigs) agrees with the one-contig form
    counts, csums, busy = oracle.synth_checksums_threads(sp, len(lens) - 1, 9, 40, 300_000, 3)
    assert np.array_equal(csums[0], sums[0]) and int(counts[0]) == int(off[1] - off[0])
    # explicit contig ids (a shard of a partitioned set) generate the same contigs
    b2 = P.Batch.synthetic([1_000_000, 300_000], seed=9, ctx=gpu_ctx, contig_ids=[43, 40])
    s2 = b2.shmmrs(P.make_spec()).checksum()
    assert np.array_equal(s2[0], sums[3]) and np.array_equal(s2[1], sums[0])


def test_query_resident_equals_host_entry_and_reports_counts(oracle, gpu_ctx):
    import pgrtk_amd as P
    rng = np.random.default_rng(12)
    seqs = [seqgen.rnd(rng, 400_000) for _ in range(4)]
    seqs.append(seqs[1][1000:90_000] + seqgen.rnd(rng, 50_000))  # shared segment: two targets per query
    sp = P.make_spec()
    ix = P.Index(sp, ctx=gpu_ctx)
    ix.add_seqs(seqs)
    ix.finalize()
    qs = [seqs[1][5_000:25_000], seqs[3][100_000:101_000], seqgen.rc(seqs[0][200_000:260_000]), b"ACGT" * 10, b""]
    r_host = ix.query_hps_raw(qs, 0.025)
    qb = P.Batch.from_seqs(qs, ctx=gpu_ctx)
    r_res = ix.query_hps_resident_raw(qb, 0.025)
    for k in ("q_off", "t_sid", "t_off", "c_score", "c_off", "hps"):
        assert np.array_equal(r_host[k], r_res[k]), k
    prof = gpu_ctx.last_query_prof()
    assert prof["n_queries"] == len(qs) and prof["query_bases"] == sum(len(q) for q in qs)
    assert prof["n_hps"] == len(r_res["hps"]) and prof["n_chains"] == len(r_res["c_score"])
    assert prof["n_signatures"] >= prof["n_hits"] >= prof["n_hps"] > 0
    # content against the checker
    oix = oracle.Index(oracle.spec())
    for i, s in enumerate(seqs):
        oix.add_seq(i, s)
    for qi, q in enumerate(qs):
        if len(q) == 0:
            continue
        ref = oix.query_fragment_to_hps(q, 0.025)
        got = []
        for t in range(int(r_res["q_off"][qi]), int(r_res["q_off"][qi + 1])):
            ch = []
            for c in range(int(r_res["t_off"][t]), int(r_res["t_off"][t + 1])):
                hp = r_res["hps"][int(r_res["c_off"][c]):int(r_res["c_off"][c + 1])]
                ch.append((float(r_res["c_score"][c]), [tuple(int(v) for v in h) for h in hp]))
            got.append((int(r_res["t_sid"][t]), ch))
        assert got == ref, qi


def test_degenerate_group_does_not_fail_the_batch(gpu_ctx):
    """a group whose hits all have end <= bgn never terminates in the reference (aln.rs:105-131); it is reported and
    cut short, the other groups of the same call are chained normally"""
    import pgrtk_amd as P
    good = [((10, 200, 0), (1000, 1190, 0)), ((300, 500, 0), (1300, 1500, 0)), ((600, 900, 0), (1600, 1900, 0))]
    bad = [((50, 50, 0), (7, 7, 0)), ((80, 80, 0), (9, 9, 0))]
    res = P.sparse_aln_groups([good, bad, good], 8, 0.025, ctx=gpu_ctx)
    assert res["n_nonterminating"] == 1
    assert res["chains"][0] == res["chains"][2] and len(res["chains"][0]) == 1 and len(res["chains"][0][0][1]) == 3
    assert res["chains"][1] == []


def test_max_aln_span_above_64(oracle, gpu_ctx):
    """aln.rs:91 accepts any max_span; above 64 every group runs on the one-thread kernel with its span set in global memory"""
    import pgrtk_amd as P
    rng = np.random.default_rng(99)
    hits, q, t = [], 0, 5000
    for _ in range(700):
        q += int(rng.integers(1, 50))
        t += int(rng.integers(-40, 120))
        ln = int(rng.integers(30, 200))
        hits.append(((q, q + ln, int(rng.integers(0, 2))), (max(1, t), max(1, t) + ln, int(rng.integers(0, 2)))))
        if rng.random() < 0.2:
            hits.append(((q, q + ln + 3, 0), (max(1, t) + 9000, max(1, t) + 9000 + ln, 0)))
    hits = list(dict.fromkeys(hits))
    flat = [a + b for a, b in hits]
    for span, pen, gap, ori in [(65, 0.01, None, False), (200, 0.002, None, True), (5000, 0.0005, 100000, False)]:
        got = P.sparse_aln(hits, span, pen, gap, ori, ctx=gpu_ctx)
        ref = oracle.sparse_aln(flat, span, pen, gap, ori)
        ref = [(sc, [((x[0], x[1], x[2]), (x[3], x[4], x[5])) for x in hp]) for sc, hp in ref]
        assert got == ref, span


def _random_group(rng, n, dup=0.15):
    hits, q, t = [], int(rng.integers(0, 100)), int(rng.integers(1000, 9000))
    while len(hits) < n:
        q += int(rng.integers(0, 40))  # 0: equal query bgn runs (value slots, span-set tail scan)
        t += int(rng.integers(-60, 150))
        ln = int(rng.integers(20, 160))
        h = ((q, q + ln, int(rng.integers(0, 2))), (max(1, t), max(1, t) + ln, int(rng.integers(0, 2))))
        hits.append(h)
        if rng.random() < dup and len(hits) < n:
            hits.append(h if rng.random() < 0.5 else ((q, q + ln, h[0][2]), (h[1][0] + 5000, h[1][1] + 5000, h[1][2])))
    return hits


@pytest.mark.parametrize("many", [False, True])
def test_sparse_aln_group_size_classes(oracle, gpu_ctx, many):
    """short groups are chained by one thread each with their hits staged in LDS (2048 hits per wavefront: the 64 groups of
    the first wavefronts below overflow it and finish from global memory), longer ones by a wavefront each (up to 256 hits
    in a small LDS image, above in a big one); "short" is < 64 hits in a call of >= 4096 groups and < 16 otherwise.  All of
    them against the checker, with duplicates and equal-bgn runs"""
    import pgrtk_amd as P
    rng = np.random.default_rng(4242)
    sizes = [int(rng.integers(40, 64)) for _ in range(128)] + [2, 3, 15, 16, 17, 63, 64, 65, 255, 256, 257, 300] + \
            [int(rng.integers(2, 64)) for _ in range(300)] + [int(rng.integers(64, 200)) for _ in range(20)]
    if many:  # >= 4096 groups in the call: the one-thread path takes everything below 64 hits (else: below 16)
        sizes += [int(rng.integers(2, 12)) for _ in range(4000)]
    groups = [_random_group(rng, n) for n in sizes]
    cases = [(8, 0.025, None, False), (3, 0.1, 200, True)] if many else \
        [(8, 0.025, None, False), (3, 0.1, 200, True), (64, 0.001, None, False), (100, 0.01, None, False)]
    for span, pen, gap, ori in cases:
        res = P.sparse_aln_groups(groups, span, pen, gap, ori, ctx=gpu_ctx)
        assert res["n_nonterminating"] == 0
        for gi, g in enumerate(groups):
            ref = oracle.sparse_aln([a + b for a, b in g], span, pen, gap, ori)
            ref = [(sc, [((x[0], x[1], x[2]), (x[3], x[4], x[5])) for x in hp]) for sc, hp in ref]
            assert res["chains"][gi] == ref, (span, gi, len(g))


def _raw_to_lists(r, qi):
    got = []
    for t in range(int(r["q_off"][qi]), int(r["q_off"][qi + 1])):
        ch = []
        for c in range(int(r["t_off"][t]), int(r["t_off"][t + 1])):
            hp = r["hps"][int(r["c_off"][c]):int(r["c_off"][c + 1])]
            ch.append((float(r["c_score"][c]), [tuple(int(v) for v in h) for h in hp]))
        got.append((int(r["t_sid"][t]), ch))
    return got


def test_query_many_targets_per_query_grouping_paths(oracle, gpu_ctx, monkeypatch):
    """queries that hit MANY targets (a pangenome-like index: 40 near-identical haplotypes): the per-query LDS grouping of
    the hits must give what the global sort gives and what the checker gives"""
    import pgrtk_amd as P
    rng = np.random.default_rng(77)
    anc = seqgen.rnd(rng, 60_000)
    haps = []
    for h in range(40):
        s = bytearray(anc)
        for p in rng.integers(0, len(s), 60):
            s[int(p)] = b"ACGT"[int(rng.integers(0, 4))]
        haps.append(bytes(s))
    ix = P.Index(P.make_spec(), ctx=gpu_ctx)
    ix.add_seqs(haps)
    ix.finalize()
    queries = [bytes(anc[o:o + 12_000]) for o in (0, 7_000, 20_000, 33_333, 47_999)]
    queries.append(seqgen.rc(queries[1]))
    got = ix.query_hps_raw(queries, 0.025)
    with gpu_ctx.options(query_global_sort=1):
        alt = ix.query_hps_raw(queries, 0.025)
    for k in ("q_off", "t_sid", "t_off", "c_score", "c_off", "hps"):
        assert np.array_equal(got[k], alt[k]), k
    oix = oracle.Index(oracle.spec())
    for sid, s in enumerate(haps):
        oix.add_seq(sid, s)
    n_targets = 0
    for qi, q in enumerate(queries):
        ref = oix.query_fragment_to_hps(q, 0.025)
        mine = _raw_to_lists(got, qi)
        assert sorted(mine) == sorted(ref), qi
        n_targets += len(ref)
    assert n_targets >= 5 * 30


def test_index_sort_shortcut_equals_full_sort(gpu_ctx, monkeypatch):
    """records appended in (sid, frg_id) order are sorted by the key alone (stable); any other append order, and the forced
    full sort, must give the same CSR"""
    import pgrtk_amd as P
    rng = np.random.default_rng(5)
    base = seqgen.rnd(rng, 120_000)
    seqs = [base[i * 7_000:i * 7_000 + 60_000] + seqgen.rnd(rng, 20_000) for i in range(6)]  # shared keys across sids
    sp = P.make_spec()

    def build(order, full):
        ix = P.Index(sp, ctx=gpu_ctx)
        for i in order:  # one call per sequence: the append order is `order`
            ix.add_seqs([seqs[i]], sids=[i])
        with gpu_ctx.options(index_full_sort=int(full)):
            ix.finalize()
        return ix.download()
    ref = build(range(6), True)
    assert len(ref) > 500 and len(np.unique(ref["sid"])) == 6
    for order, full in [(range(6), False), ([3, 1, 5, 0, 2, 4], False), ([5, 4, 3, 2, 1, 0], True)]:
        got = build(order, full)
        assert np.array_equal(got, ref), (list(order), full)


@pytest.mark.parametrize("max_target", [128, 20, 4000])
def test_query_through_a_repeat_heavy_keys(oracle, gpu_ctx, max_target):
    """the shimmer pairs of a repeat unit that 300 sequences share are keys with hundreds of records (more than HITS_HEAVY = 64):
    the wavefront takes such keys 64 records per step (runs of one sid pass or fail the target filter as a whole); against
    the checker, with three settings of the target filter"""
    import pgrtk_amd as P
    rng = np.random.default_rng(3)
    unit = seqgen.rnd(rng, 600)
    seqs = [seqgen.rnd(rng, 2500) + unit * 25 + seqgen.rnd(rng, 2500) for _ in range(300)]
    queries = [seqs[17][500:2500] + unit * 5 + seqs[17][-2500:-300], unit * 3 + seqs[5][-2500:], seqs[9][:3000]]
    ix = P.Index(P.make_spec(), ctx=gpu_ctx)
    ix.add_seqs(seqs)
    ix.finalize()
    got = ix.query_hps_raw(queries, 0.025, 128, 128, max_target)
    oix = oracle.Index(oracle.spec())
    for sid, s in enumerate(seqs):
        oix.add_seq(sid, s)
    n_targets = 0
    for qi, q in enumerate(queries):
        ref = oix.query_fragment_to_hps(q, 0.025, 128, 128, max_target)
        assert sorted(_raw_to_lists(got, qi)) == sorted(ref), qi
        n_targets += len(ref)
    assert gpu_ctx.last_query_prof()["n_signatures"] > 300
    assert n_targets >= 100


def test_a_probe_does_not_accept_a_stuck_machine(oracle, gpu_ctx):
    """Round 6, found by tools/fuzz_parity.py seed 7123218 (1 of 37 300 cases; the defect dates from the islands of round 4): an island
    around a run of N ends at the end of its last flagged tile, where a probe -- the machine warmed up over the 256 positions in front
    -- must show the exact machine back in the regular regime the next tile's closed form assumes.  A palindromic (AT)n stretch near
    the END of that tile (a tile the tile kernel skips, so nobody flags its palindromes) leaves the machine STUCK (mdist beyond w - 1,
    shmmrutils.rs:505-514) beyond the tile's end; the probe's warm-up window holds the stretch as well, reproduces the stuck state
    exactly, the states compare equal -- and the library emitted closed-form minimizers where the reference emits none for hundreds of
    positions.  A probe now also asks for mdist <= w - 1, else the island grows.  The fuzz case itself (regenerated from its seed) and
    a family built on purpose: a run of N that ends inside a tile, 40-80 x (AT) ending 60-400 positions in front of that tile's end."""
    import os
    import sys
    import pgrtk_amd as P
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_case
    spec_t, padding, seqs, rids, _ = fuzz_case.gen(7123218, 2_000_000)
    assert spec_t == (48, 56, 4, 12, False) and len(seqs[16]) == 79076
    cases = [(spec_t[:4], seqs[16]), ((48, 56, 1, 0), seqs[16])]
    rng = np.random.default_rng(81)
    for w, k in ((48, 56), (80, 56), (31, 24)):
        tc = ((4096 - 2 * (w - 1)) // 64) * 64
        for _ in range(12):
            t_end = int(rng.integers(4, 9)) * tc  # the end of the tile in which the run of N ends
            n_lo = int(rng.integers(2000, t_end - tc - 500))
            n_hi = t_end - tc + int(rng.integers(50, tc - 1200))  # the run ends inside the tile [t_end - tc, t_end)
            at = b"AT" * int(rng.integers(40, 81))
            at_end = t_end - int(rng.integers(60, 400))
            body = bytearray(seqgen.rnd(rng, t_end + 3 * tc))
            body[n_lo:n_hi] = b"N" * (n_hi - n_lo)
            body[at_end - len(at):at_end] = at
            cases.append(((w, k, 4, 12), bytes(body)))
            cases.append(((w, k, 1, 0), bytes(body)))
    for spec4, s in cases:
        ref = oracle.sequence_to_shmmrs(0, s, oracle.spec(*spec4), False)
        for opts in ({}, {"no_small_path": 1, "early_sync_bp": 0}):
            with gpu_ctx.options(**opts):
                got = P.sequence_to_shmmrs_batch([s], P.make_spec(*spec4), ctx=gpu_ctx)[0]
            assert len(ref) == len(got) and np.array_equal(ref["x"], got["x"]) and np.array_equal(ref["y"], got["y"]), (spec4, len(s), opts)


def test_islands_that_begin_and_end_inside_tiles(oracle, gpu_ctx):
    """Round 6 (island_list.h, IslandRun::finish, splice_segs_kernel): an island around an array of palindromic k-mers begins
    cut_margin in front of the first 64-position block the tile kernel reports one in and ends 2 w + k + 64 behind the last one --
    inside the tiles, whose own elements in front of B / from E on are spliced with the exact machine's list; a machine that
    arrives at E stuck moves the end on itself.  Arrays at every kind of place: at a tile's start (B lies in the clean tile in
    front), in its middle, at its end (E lies in the next tile), across a boundary, two in one tile, one to three tiles apart,
    near the contig's end; (AT)n and (ACGT)n; three specs; against the oracle, and against the islands of whole tiles
    (no_sub_tile_islands), which must be the same list with more positions through the machine."""
    import pgrtk_amd as P
    rng = np.random.default_rng(606)
    cases = []
    for w, k in ((80, 56), (48, 56), (31, 24)):
        tc = ((4096 - 2 * (w - 1)) // 64) * 64
        for rep in range(10):
            nt = int(rng.integers(40, 60))  # (a contig with more than a third of its tiles flagged is one island: arrays every 4-8 tiles)
            body = bytearray(seqgen.rnd(rng, nt * tc + int(rng.integers(0, tc))))
            t = 2
            while t < nt - 1:
                kind = int(rng.integers(0, 7))
                unit = b"AT" if rng.random() < 0.7 else b"ACGT"
                arr = unit * int(rng.integers(40, 220) * 2 // len(unit))
                if kind == 0:
                    off = int(rng.integers(0, 200))                        # at the tile's start
                elif kind == 1:
                    off = int(rng.integers(400, tc - 600))                 # in the middle
                elif kind == 2:
                    off = tc - len(arr) - int(rng.integers(0, 300))        # at its end: E in the next tile
                elif kind == 3:
                    off = tc - len(arr) // 2                               # across the boundary
                elif kind == 4:
                    off = int(rng.integers(300, 1200))                     # two in one tile
                    o2 = off + len(arr) + int(rng.integers(200, 1800))
                    if o2 + len(arr) < tc:
                        body[t * tc + o2:t * tc + o2 + len(arr)] = arr
                elif kind == 5:
                    off = int(rng.integers(0, tc - len(arr)))
                else:
                    off = int(rng.integers(0, 64))
                p0 = t * tc + off
                if p0 + len(arr) < len(body):
                    body[p0:p0 + len(arr)] = arr
                t += int(rng.integers(1, 4)) if rng.random() < 0.2 else int(rng.integers(4, 9))
            if rep % 3 == 0:  # ... and one within two tiles of the contig's end
                p0 = len(body) - int(rng.integers(300, 2 * tc))
                body[p0:p0 + 120] = b"AT" * 60
            cases.append(((w, k, 4, 12) if rep % 2 else (w, k, 1, 0), bytes(body)))
    less = 0
    for spec4, s in cases:
        ref = oracle.sequence_to_shmmrs(0, s, oracle.spec(*spec4), False)
        through = {}
        for name, opts in (("default", {"no_small_path": 1}), ("early look", {"no_small_path": 1, "early_sync_bp": 0}),
                           ("whole tiles", {"no_small_path": 1, "no_sub_tile_islands": 1})):
            with gpu_ctx.options(**opts):
                got = P.sequence_to_shmmrs_batch([s], P.make_spec(*spec4), ctx=gpu_ctx)[0]
                through[name] = gpu_ctx.last_prof().exact_bases
            assert len(ref) == len(got) and np.array_equal(ref["x"], got["x"]) and np.array_equal(ref["y"], got["y"]), (spec4, len(s), name)
        assert through["default"] <= through["whole tiles"], (spec4, through)
        less += through["default"] < through["whole tiles"]
    assert less >= len(cases) // 2  # (the sub-tile form is what runs: fewer positions through the machine in most cases)
