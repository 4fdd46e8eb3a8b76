"""GPU parity of the one-wavefront-per-query path (csrc/query_fused.hip): batches of short queries run every stage behind the
pair records -- lookup (seq_db.rs:1200-1228), count filters (aln.rs:147-242), grouping, aln::sparse_aln (aln.rs:12-142) -- in one
kernel.  Checked against the CPU oracle AND against the stage-by-stage kernels (context option no_fused_query), with the path that ran
read back from pgr_query_prof.path; batches the kernel cannot hold must decline and still give the oracle's answer."""
import os

import numpy as np
import pytest

import seqgen
from test_gpu_01_index_query import _build_pair, _make_db_seqs, _oracle_hps_to_tuples, revcomp

pytestmark = pytest.mark.gpu

KW = dict(penalty=0.025, max_count=128, max_count_query=128, max_count_target=128, max_aln_span=8, max_gap=None, orientated=False)

# pgr_query_prof.path: 0 stage by stage, 1 the per-query kernel behind a host look at the shimmers, 2 ... enqueued behind the shimmer
# pipeline, 3 its level-1 form (tile kernel + per-query kernel, no list stage of the batch: round 6).  Every test of this file runs
# twice: with the level-1 form switched off (context option no_query_level1: the paths exactly as before) and with the default,
# where a clean batch may (and, where a test says so, must) take path 3.
LEVEL1 = False


@pytest.fixture(autouse=True, params=["chained", "level1"])
def _form(request, gpu_ctx):
    global LEVEL1
    LEVEL1 = request.param == "level1"
    with gpu_ctx.options(no_query_level1=0 if LEVEL1 else 1):
        yield
    LEVEL1 = False


def _paths(*p):
    """the paths a call may have taken: those of the chained forms, and the level-1 form when it is on"""
    return set(p) | ({3} if LEVEL1 and any(x in (1, 2) for x in p) else set())


def _args(kw):
    return (kw["penalty"], kw["max_count"], kw["max_count_query"], kw["max_count_target"], kw["max_aln_span"], kw["max_gap"],
            kw["orientated"])


def _check_vs_oracle(oix, queries, got, kw):
    n_chains = 0
    for q, g in zip(queries, got):
        ref = _oracle_hps_to_tuples(oix.query_fragment_to_hps(q, *_args(kw)))
        assert [t[0] for t in ref] == [t[0] for t in g]
        for (sid, rc), (_, gc) in zip(ref, g):
            assert len(rc) == len(gc), sid
            for (rs, rh), (gs, gh) in zip(rc, gc):
                assert rs == gs, (sid, rs, gs)  # f32, bit exact
                assert rh == gh
            n_chains += len(rc)
    return n_chains


def _general(sdb, queries, kw):
    with sdb.ctx.options(no_fused_query=1):
        return sdb.query_fragments_to_hps(queries, *_args(kw))


def _short_queries(rng, seqs, n, lo=1500, hi=11000):
    out = []
    for i in range(n):
        src = seqs[int(rng.integers(0, 12))]
        a = int(rng.integers(0, max(1, len(src) - hi)))
        q = src[a:a + int(rng.integers(lo, hi))]
        if i % 2:
            q = revcomp(q)
        if i % 5 == 0:  # a few SNPs
            qa = bytearray(q)
            for p in rng.integers(0, len(qa), 5):
                qa[p] = ord("ACGT"[int(rng.integers(0, 4))])
            q = bytes(qa)
        out.append(q)
    return out


@pytest.mark.parametrize("variant", ["default", "max_gap", "oriented", "tight_counts", "span1", "span2", "span64"])
def test_short_query_batches_take_the_fused_path_and_match(oracle, gpu_ctx, variant):
    rng = np.random.default_rng(41)
    seqs, core = _make_db_seqs(rng)
    sdb, oix = _build_pair(oracle, gpu_ctx, seqs)
    kw = dict(KW)
    if variant == "max_gap":
        kw["max_gap"] = 2000
    elif variant == "oriented":
        kw["orientated"] = True
    elif variant == "tight_counts":
        kw.update(max_count=1, max_count_query=1, max_count_target=1)
    elif variant == "span1":
        kw.update(max_aln_span=1)
    elif variant == "span2":
        kw.update(max_aln_span=2, penalty=0.5)
    elif variant == "span64":
        kw.update(max_aln_span=64, penalty=0.001)
    queries = _short_queries(rng, seqs, 150)
    queries += [b"", seqgen.rnd(rng, 50), seqgen.rnd(rng, 5000), core[0][:9000], revcomp(core[1][100:7000]), b"N" * 3000,
                core[2][:4000] + b"NNNN" + core[2][4000:8000]]
    got = sdb.query_fragments_to_hps(queries, *_args(kw))
    assert gpu_ctx.last_query_prof()["path"] in _paths(1, 2)
    assert _check_vs_oracle(oix, queries, got, kw) > 100
    assert _general(sdb, queries, kw) == got
    assert gpu_ctx.last_query_prof()["path"] == 0
    one = sdb.query_fragment_to_hps(queries[3], *_args(kw))
    assert gpu_ctx.last_query_prof()["path"] in _paths(1, 2) and one == got[3]


def test_repeated_hits_share_value_slots_and_intervals(oracle, gpu_ctx):
    """tandem copies inside the targets: one query pair hits several places of one target, so groups hold hit pairs with equal
    query intervals (the span set's distinct-interval rule, aln.rs:70/:91) and identical hit pairs (shared value slots)"""
    rng = np.random.default_rng(42)
    unit = seqgen.rnd(rng, 6000)
    seqs = []
    for i in range(6):
        parts = [seqgen.rnd(rng, 3000)]
        for _ in range(int(rng.integers(2, 5))):
            parts += [unit, seqgen.rnd(rng, int(rng.integers(200, 1500)))]
        seqs.append(b"".join(parts))
    seqs += [seqs[0], seqs[1]] + [seqgen.rnd(rng, 20000) for _ in range(4)]
    sdb, oix = _build_pair(oracle, gpu_ctx, seqs)
    queries = [unit, unit[500:5500], revcomp(unit), unit[:3000] + unit[:3000], seqs[8][2000:9000]]
    for kw in (dict(KW), dict(KW, max_aln_span=2), dict(KW, max_aln_span=3, orientated=True), dict(KW, max_gap=500),
               dict(KW, max_count_target=2), dict(KW, max_count_query=1)):
        got = sdb.query_fragments_to_hps(queries, *_args(kw))
        path = gpu_ctx.last_query_prof()["path"]
        _check_vs_oracle(oix, queries, got, kw)
        assert _general(sdb, queries, kw) == got
        assert path in _paths(0, 1)


def test_batches_that_do_not_fit_decline_and_fall_back(oracle, gpu_ctx):
    rng = np.random.default_rng(43)
    seqs, core = _make_db_seqs(rng)
    sdb, oix = _build_pair(oracle, gpu_ctx, seqs)
    short = _short_queries(rng, seqs, 20)
    # (a) a query with more pairs than the kernel holds: not eligible, no attempt
    got = sdb.query_fragments_to_hps(short + [core[0] + core[1] + core[2]], *_args(KW))
    assert gpu_ctx.last_query_prof()["path"] == 0
    _check_vs_oracle(oix, short + [core[0] + core[1] + core[2]], got, KW)
    got = sdb.query_fragments_to_hps(short, *_args(KW))
    assert gpu_ctx.last_query_prof()["path"] in _paths(1, 2)
    # (b) more hits than the first guess of the slot size: the kernel asks for larger slots and runs again
    copies = [core[0][:20000]] * 12 + [seqgen.rnd(rng, 5000)]
    sdb2, oix2 = _build_pair(oracle, gpu_ctx, copies)
    q2 = [core[0][1000:9000], core[0][5000:9000], seqgen.rnd(rng, 3000)]
    got2 = sdb2.query_fragments_to_hps(q2, *_args(KW))
    assert gpu_ctx.last_query_prof()["path"] in _paths(1, 2)
    assert _check_vs_oracle(oix2, q2, got2, KW) > 5
    assert _general(sdb2, q2, KW) == got2
    # (c) more hits than the largest slot: declined on the device, answered by the stage-by-stage kernels
    copies = [core[0][:20000]] * 40 + [seqgen.rnd(rng, 5000)]
    sdb4, oix4 = _build_pair(oracle, gpu_ctx, copies)
    got4 = sdb4.query_fragments_to_hps(q2, *_args(KW))
    assert gpu_ctx.last_query_prof()["path"] == 0
    assert _check_vs_oracle(oix4, q2, got4, KW) > 5
    # the index remembers: the next calls do not try again, later ones do (and decline again)
    for _ in range(20):
        g = sdb4.query_fragments_to_hps(q2[2:], *_args(KW))
    assert g == got4[2:]
    # (d) a group of more than 64 hits: the look-back with its work arrays in LDS
    copies3 = [core[1][:30000], seqgen.rnd(rng, 5000), core[1][2000:28000]]
    sdb3, oix3 = _build_pair(oracle, gpu_ctx, copies3)
    q3 = [core[1][:29000], revcomp(core[1][1000:27000]), core[1][:9000]]
    for kw in (KW, dict(KW, max_aln_span=2), dict(KW, max_gap=3000, orientated=True), dict(KW, max_aln_span=64, penalty=0.0)):
        got3 = sdb3.query_fragments_to_hps(q3, *_args(kw))
        assert gpu_ctx.last_query_prof()["path"] in _paths(1, 2)
        assert _check_vs_oracle(oix3, q3, got3, kw) >= 3
        assert _general(sdb3, q3, kw) == got3


def test_many_short_queries_equal_the_stage_by_stage_path(oracle, gpu_ctx):
    """5000 queries of 2-10 kbp against 40 Mbp: the shape of BASELINE.json configs[2] (scaled); every array of the flat result
    equal between the two paths, a sample against the oracle"""
    import pgrtk_amd as P
    rng = np.random.default_rng(44)
    seqs = [seqgen.rnd(rng, 1_000_000) for _ in range(40)]
    sdb, oix = _build_pair(oracle, gpu_ctx, seqs)
    queries = []
    for i in range(5000):
        src = seqs[int(rng.integers(0, 40))]
        a = int(rng.integers(0, len(src) - 10000))
        q = src[a:a + int(rng.integers(2000, 10000))]
        queries.append(revcomp(q) if i % 2 else q)
    got = sdb.query_fragments_to_hps(queries, *_args(KW))
    assert gpu_ctx.last_query_prof()["path"] in _paths(1, 2)
    assert _general(sdb, queries, KW) == got
    idx = [int(i) for i in rng.integers(0, 5000, 60)]
    assert _check_vs_oracle(oix, [queries[i] for i in idx], [got[i] for i in idx], KW) >= 60


def test_second_batch_on_an_index_is_enqueued_behind_the_shimmer_pipeline(oracle, gpu_ctx):
    """The per-query kernel runs behind the shimmer pipeline without a host wait in between (pgr_query_prof.path == 2): pair
    records and their offsets are derived on the device from a guess of the queries' sizes (the spec's shimmer density for the
    first batch on an index, the previous batch afterwards).  Same results; a batch that outgrows the guess is answered all the
    same (path 1: done again with the host's numbers)."""
    rng = np.random.default_rng(45)
    seqs = [seqgen.rnd(rng, 1_000_000) for _ in range(24)]
    sdb, oix = _build_pair(oracle, gpu_ctx, seqs)

    def batch(n, lo, hi):
        out = []
        for i in range(n):
            src = seqs[int(rng.integers(0, 24))]
            a = int(rng.integers(0, len(src) - hi))
            q = src[a:a + int(rng.integers(lo, hi))]
            out.append(revcomp(q) if i % 2 else q)
        return out
    gpu_ctx.set_option("no_small_path", 1)  # (small batches of short contigs have a shimmer kernel of their own: not this test)
    try:
        q1 = batch(300, 2000, 9000)
        g1 = sdb.query_fragments_to_hps(q1, *_args(KW))
        assert gpu_ctx.last_query_prof()["path"] in _paths(2)  # (the first batch too: the guess comes from the spec's shimmer density)
        g1b = sdb.query_fragments_to_hps(q1, *_args(KW))
        assert gpu_ctx.last_query_prof()["path"] in _paths(2) and g1b == g1
        assert _check_vs_oracle(oix, q1[:40], g1b[:40], KW) >= 40
        # other parameters, other queries of the same kind
        kw = dict(KW, max_aln_span=2, max_gap=4000, orientated=True)
        q2 = batch(500, 1500, 9500) + [b"", seqgen.rnd(rng, 300), b"N" * 2000]
        g2 = sdb.query_fragments_to_hps(q2, *_args(kw))
        assert gpu_ctx.last_query_prof()["path"] in _paths(2)
        assert _general(sdb, q2, kw) == g2
        assert _check_vs_oracle(oix, q2[:30], g2[:30], kw) >= 30
        # longer queries than the guess (pairs per query): declined on the device, done again with the host's numbers
        q3 = batch(200, 24000, 33000)
        g3 = sdb.query_fragments_to_hps(q3, *_args(KW))
        assert gpu_ctx.last_query_prof()["path"] in _paths(1)
        assert _check_vs_oracle(oix, q3[:10], g3[:10], KW) >= 10
        g3b = sdb.query_fragments_to_hps(q3, *_args(KW))
        assert gpu_ctx.last_query_prof()["path"] in _paths(2) and g3b == g3
        # a batch the kernel cannot hold at all (a query of 400 kbp): stage by stage, and still right
        q4 = q1[:50] + [seqs[3][:400000]]
        g4 = sdb.query_fragments_to_hps(q4, *_args(KW))
        assert gpu_ctx.last_query_prof()["path"] == 0
        assert g4[:50] == g1[:50] and _check_vs_oracle(oix, q4[50:], g4[50:], KW) >= 1
        # empty batch of pairs
        g5 = sdb.query_fragments_to_hps([b"ACGT" * 10, b""], *_args(KW))
        assert g5 == [[], []]
    finally:
        gpu_ctx.set_option("no_small_path", 0)


def test_single_pass_result_option_gives_the_same_answer(oracle, gpu_ctx):
    """Context option direct_query_result (an experiment, off by default: query_fused_kernel<true> finds every query's place
    by a look-back over tiles of 64 queries and writes the host's block itself, sections placed from the previous batch's
    counts): the same answer as the default two-pass form -- second batch on an index (single pass), a batch with more hits
    than the previous one promised (sections too small: done again in two passes), a ragged batch (empty and N-only queries,
    a tile of 64 that is not full), and 1 and 65 queries (one tile, two tiles)."""
    if LEVEL1:
        pytest.skip("the single-pass experiment belongs to the chained form")
    rng = np.random.default_rng(46)
    seqs = [seqgen.rnd(rng, 1_000_000) for _ in range(24)]
    sdb, oix = _build_pair(oracle, gpu_ctx, seqs)

    def batch(n, lo, hi):
        out = []
        for i in range(n):
            src = seqs[int(rng.integers(0, 24))]
            a = int(rng.integers(0, len(src) - hi))
            q = src[a:a + int(rng.integers(lo, hi))]
            out.append(revcomp(q) if i % 2 else q)
        return out
    misses = [seqgen.rnd(rng, 5000) for _ in range(700)]          # no hit at all: the next batch's sections are sized for nothing
    q_hits = batch(700, 2000, 9000)
    ragged = batch(333, 1500, 9500) + [b"", seqgen.rnd(rng, 300), b"N" * 2000]
    gpu_ctx.set_option("no_small_path", 1)
    try:
        ref_hits = sdb.query_fragments_to_hps(q_hits, *_args(KW))
        ref_ragged = sdb.query_fragments_to_hps(ragged, *_args(KW))
        ref_small = [sdb.query_fragments_to_hps(q_hits[:n], *_args(KW)) for n in (1, 65)]
        assert _check_vs_oracle(oix, q_hits[:30], ref_hits[:30], KW) >= 30
        delivered = lambda: gpu_ctx.get_option("direct_query_results_delivered")  # noqa: E731
        with gpu_ctx.options(direct_query_result=1):
            d0 = delivered()
            assert sdb.query_fragments_to_hps(q_hits, *_args(KW)) == ref_hits      # single pass (the counts of the batches above)
            assert sdb.query_fragments_to_hps(q_hits, *_args(KW)) == ref_hits
            assert sdb.query_fragments_to_hps(ragged, *_args(KW)) == ref_ragged
            assert all(len(r) == 0 for r in sdb.query_fragments_to_hps(misses, *_args(KW)))
            assert delivered() == d0 + 4
            assert sdb.query_fragments_to_hps(q_hits, *_args(KW)) == ref_hits      # sections placed for no hits: two passes after all
            assert delivered() == d0 + 4
            assert sdb.query_fragments_to_hps(q_hits, *_args(KW)) == ref_hits
            for n, ref in zip((1, 65), ref_small):
                assert sdb.query_fragments_to_hps(q_hits[:n], *_args(KW)) == ref
            assert delivered() == d0 + 7
    finally:
        gpu_ctx.set_option("no_small_path", 0)


def test_level1_form_clean_batches_take_it_and_equal_every_other_path(oracle, gpu_ctx):
    """Round 6: the per-query kernel reads ITS query's level-1 minimizers from the tile kernel's segments and runs the list stage
    (reduce_shmmr twice, shmmrutils.rs:359-415, 533-535; min_span, :536-555; the pairs, seq_db.rs:1205-1217) inside its wavefront --
    no list stage of the batch.  A clean batch (ACGT only, no palindromic k-mer) must take it (path 3) and give, array for array,
    what the chained form, the stage-by-stage kernels and the oracle give: four specs (r = 1: no reduction; r = 2; a window of 48;
    the default), ragged lengths from shorter than k to three tiles, reverse complements, SNPs, empty queries, misses."""
    if not LEVEL1:
        pytest.skip("level-1 form only")
    rng = np.random.default_rng(61)
    seqs = [seqgen.rnd(rng, 300_000) for _ in range(12)]
    # (query lengths per spec: at most 128 pairs per query.  The first batch on an index is sized from the spec's shimmer density
    # x 1.6: where that guess exceeds 128 pairs the first call goes through the shimmer pipeline -- path 1 -- and tells the index)
    for spec_t, hi in (((80, 56, 4, 64), 12500), ((48, 56, 4, 32), 12500), ((80, 56, 2, 0), 8000), ((31, 24, 1, 8), 1400), ((64, 40, 6, 64), 12500)):
        sdb, oix = _build_pair(oracle, gpu_ctx, seqs, spec_t)
        queries = _short_queries(rng, seqs, 200, lo=40, hi=hi) + [b"", seqgen.rnd(rng, 30), seqgen.rnd(rng, hi // 2), b"ACGT" * 5]
        sdb.query_fragments_to_hps(queries, *_args(KW))
        assert gpu_ctx.last_query_prof()["path"] in (1, 3), spec_t
        for kw in (KW, dict(KW, max_aln_span=2, max_gap=3000, orientated=True)):
            got = sdb.query_fragments_to_hps(queries, *_args(kw))
            prof = gpu_ctx.last_query_prof()
            assert prof["path"] == 3, (spec_t, prof["path"])
            with gpu_ctx.options(no_query_level1=1):
                chained = sdb.query_fragments_to_hps(queries, *_args(kw))
                p2 = gpu_ctx.last_query_prof()
            assert p2["path"] in (1, 2) and chained == got
            # the device's count of the queries' pairs (the host never sees them in this form) against the chained form's
            assert prof["n_query_pairs"] == p2["n_query_pairs"] and prof["n_hits"] == p2["n_hits"] and prof["n_signatures"] == p2["n_signatures"]
            assert _general(sdb, queries, kw) == got
        assert _check_vs_oracle(oix, queries[:60] + queries[-4:], got[:60] + got[-4:], kw) >= 40
        one = sdb.query_fragment_to_hps(queries[7], *_args(kw))
        assert gpu_ctx.last_query_prof()["path"] == 3 and one == got[7]


def test_level1_form_declines_what_the_tile_kernel_flags_and_grows_what_outgrows_its_guess(oracle, gpu_ctx):
    """(a) a non-ACGT byte or a palindromic k-mer (skipped push, shmmrutils.rs:477-480) in ONE query: the tile kernel's status words
    decline the batch on the device, the shimmer pipeline (islands) answers -- same chains -- and the index skips the form for its
    next four batches; (b) low-complexity queries hold more level-1 minimizers than the density promises: the kernel asks for a
    larger list and runs again; (c) longer queries than the previous batch promised (pairs): the same with a larger P."""
    if not LEVEL1:
        pytest.skip("level-1 form only")
    rng = np.random.default_rng(62)
    seqs = [seqgen.rnd(rng, 400_000) for _ in range(12)]
    pal = seqgen.rnd(rng, 28)
    pal = pal + revcomp(pal)  # a palindromic 56-mer
    seqs.append(seqs[0][:5000] + pal + seqs[0][5000:9000])
    sdb, oix = _build_pair(oracle, gpu_ctx, seqs)
    clean = _short_queries(rng, seqs[:12], 80)
    got = sdb.query_fragments_to_hps(clean, *_args(KW))
    assert gpu_ctx.last_query_prof()["path"] == 3
    # a non-ACGT byte the HOST packer has counted: the form is not even tried (nothing to remember)
    bad = clean[5][:2000] + b"N" + clean[5][2000:]
    g = sdb.query_fragments_to_hps(clean + [bad], *_args(KW))
    assert gpu_ctx.last_query_prof()["path"] in (1, 2)
    assert g[:80] == got and _check_vs_oracle(oix, [bad], g[80:], KW) >= 1
    assert sdb.query_fragments_to_hps(clean, *_args(KW)) == got and gpu_ctx.last_query_prof()["path"] == 3
    # a palindromic k-mer (the host packer does not look for those): flagged by the tile kernel, declined on the device, remembered
    # for four batches
    bad = seqs[12][3000:8000]
    g = sdb.query_fragments_to_hps(clean + [bad], *_args(KW))
    assert gpu_ctx.last_query_prof()["path"] in (1, 2)
    assert g[:80] == got and _check_vs_oracle(oix, [bad], g[80:], KW) >= 1
    for i in range(4):
        assert sdb.query_fragments_to_hps(clean, *_args(KW)) == got and gpu_ctx.last_query_prof()["path"] in (1, 2), i
    assert sdb.query_fragments_to_hps(clean, *_args(KW)) == got and gpu_ctx.last_query_prof()["path"] == 3
    # (b) tandem repeats of a short unit: every window holds ties, the level-1 list is several times as dense
    unit = seqgen.rnd(rng, 9)
    dense = [seqs[1][:3000] + unit * 400 + seqs[1][3000:6000], seqs[2][1000:9000]]
    g = sdb.query_fragments_to_hps(dense + clean[:10], *_args(KW))
    assert g[2:] == got[:10] and _general(sdb, dense + clean[:10], KW) == g
    _check_vs_oracle(oix, dense, g[:2], KW)
    # (c) pairs: a batch of 2 kbp queries first (the hint), then 30 kbp ones
    sdb.query_fragments_to_hps([seqs[3][i * 3000:i * 3000 + 2000] for i in range(50)], *_args(KW))
    assert gpu_ctx.last_query_prof()["path"] == 3
    long_q = [seqs[4][i * 31000:i * 31000 + 30000] for i in range(8)]
    g = sdb.query_fragments_to_hps(long_q, *_args(KW))
    assert gpu_ctx.last_query_prof()["path"] == 3
    assert _check_vs_oracle(oix, long_q, g, KW) >= 8 and _general(sdb, long_q, KW) == g


def test_per_key_table_of_the_query_kernel_changes_nothing(oracle, gpu_ctx):
    """pgr_index.h: qkeys -- 32 B per key, the key with its record when it has exactly one: a pair's lookup (seq_db.rs:1200-1228) is two
    dependent trips to memory instead of six.  An index built without it (context option no_query_keys at finalize) answers the
    same: contigs that share segments (keys with several records, several of one sid), an exact duplicate, unique sequence; count
    filters at 1, 2 and 128 (the single-record shortcut applies max_count_target itself); both forms of the kernel."""
    rng = np.random.default_rng(63)
    seqs, core = _make_db_seqs(rng)
    seqs = seqs + [seqgen.rnd(rng, 200_000) for _ in range(3)]
    sdb, oix = _build_pair(oracle, gpu_ctx, seqs)
    with gpu_ctx.options(no_query_keys=1):
        plain, _ = _build_pair(oracle, gpu_ctx, seqs)
    queries = _short_queries(rng, seqs, 120) + [seqs[14][1000:9000], revcomp(seqs[15][50_000:61_000]), core[0][:9000], b""]
    for kw in (KW, dict(KW, max_count_target=1), dict(KW, max_count=2, max_count_query=1, max_count_target=2),
               dict(KW, max_aln_span=3, max_gap=5000, orientated=True)):
        got = sdb.query_fragments_to_hps(queries, *_args(kw))
        assert gpu_ctx.last_query_prof()["path"] in _paths(1, 2)
        ref = plain.query_fragments_to_hps(queries, *_args(kw))
        assert gpu_ctx.last_query_prof()["path"] in _paths(1, 2)
        assert got == ref
        assert _general(sdb, queries, kw) == got
        assert _check_vs_oracle(oix, queries[-30:], got[-30:], kw) >= 10
