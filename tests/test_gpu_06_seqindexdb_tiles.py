"""Round-4 parity tests (run with -m gpu on a MI355X): BASELINE.json configs[0] through the pgr-mdb counterparts
(`--synthetic NxL --seed S`), the remaining SeqIndexDB methods of row H3."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "pgr-tk_amd", "bin")


@pytest.fixture(scope="module")
def gpu_ctx():
    import pgrtk_amd as P
    return P.default_context(0)


def test_seqindexdb_remaining_methods(oracle, gpu_ctx, golden_dir, tmp_path):
    """pgr-tk/src/lib.rs:24 pgr_lib_version, :1337 write_midx_to_text_file, :1374 write_frag_and_index_files -- the
    last one is the call tests/golden's `test_seqs_frag.mdb` was made with (gen_frag_db.py): its .mdb/.midx half
    reproduces the golden files' content."""
    import pgrtk_amd as P
    assert P.pgr_lib_version().startswith("pgr-hip ")
    fa = os.path.join(golden_dir, "test_seqs.fa")
    sdb = P.SeqIndexDB(ctx=gpu_ctx)
    sdb.load_from_fastx(fa)
    prefix = str(tmp_path / "test_seqs_frag")
    sdb.write_frag_and_index_files(prefix)
    gspec, g = oracle.read_mdb(os.path.join(golden_dir, "test_seqs_frag.mdb"))
    spec2, g2 = oracle.read_mdb(prefix + ".mdb")
    assert (spec2, g2) == (gspec, g) and os.path.getsize(prefix + ".mdb") == 15291
    ref_midx = [l.split("\t")[:3] for l in open(os.path.join(golden_dir, "test_seqs_frag.midx")).read().splitlines()]
    assert [l.split("\t")[:3] for l in open(prefix + ".midx").read().splitlines()] == ref_midx
    a, b = str(tmp_path / "a.idx"), str(tmp_path / "b.idx")
    sdb.write_mapg_idx(a)
    sdb.write_midx_to_text_file(b)
    assert open(a).read() == open(b).read() and open(a).read().startswith("K\t80\t56\t4\t64\tfalse\n")
    # an index-file backend holds no sequences: the reference's `seq_db.is_some()` is false and nothing is written
    mdb = P.SeqIndexDB(ctx=gpu_ctx)
    mdb.load_from_mdb_index(prefix)
    mdb.write_frag_and_index_files(str(tmp_path / "none"))
    assert not os.path.exists(str(tmp_path / "none.mdb"))


def test_index_sorts_non_canonical_external_records(gpu_ctx):
    """pgr_index_add_records takes records from outside (other GPUs, other programs): nothing says h0 <= h1 there.  The
    radix passes are bounded by max(h0, h1) over ALL records (round 3 used max(h1): a record whose h0 exceeded every h1
    lost its high bits and the CSR came out mis-sorted).  Appended in (sid, frg_id) order (one-key sort + run fix-ups) and
    in random order (four-field sort): both must equal numpy's lexicographic order."""
    import pgrtk_amd as P
    rng = np.random.default_rng(99)
    n = 50_000
    recs = np.zeros(n, dtype=P.FRAG_REC)
    recs["h0"] = rng.integers(0, 1 << 40, n, dtype=np.uint64)
    recs["h1"] = rng.integers(0, 1 << 30, n, dtype=np.uint64)   # mostly h0 > h1: not canonical
    recs["h0"][::97] = rng.integers(1 << 54, 1 << 56, len(recs["h0"][::97]), dtype=np.uint64)  # above every h1, up to 56 bits
    recs["h0"][5] = (1 << 63) + 12345                            # beyond the library's own 56-bit hashes
    recs["h0"][1000:1040] = 777                                  # a run of equal h0 with mixed h1
    recs["sid"] = np.sort(rng.integers(0, 50, n)).astype(np.uint32)
    recs["frg_id"] = np.arange(n, dtype=np.uint32)
    recs["bgn"] = rng.integers(0, 1 << 20, n)
    recs["end"] = recs["bgn"] + 100
    want = np.sort(recs, order=["h0", "h1", "sid", "frg_id"])
    for order in (np.arange(n), rng.permutation(n)):
        ix = P.Index(P.make_spec(), ctx=gpu_ctx)
        ix.add_records(recs[order])
        ix.finalize()
        got = ix.download()
        for f in ("h0", "h1", "sid", "frg_id", "bgn", "end"):
            assert np.array_equal(got[f], want[f]), f


def _same(ref, got, what=""):
    assert len(ref) == len(got), "%s: %d vs %d shimmers" % (what, len(ref), len(got))
    assert np.array_equal(ref["x"], got["x"]) and np.array_equal(ref["y"], got["y"]), what


def test_one_wavefront_tiles_for_batches_of_short_contigs(oracle, gpu_ctx):
    """csrc/level1.hip: a batch whose mean contig length is <= 2048 (reads) runs the level-1 kernel with ONE wavefront per
    tile of 1024 positions; a contig of up to ext - 16 positions is one tile whose extended range starts at position 0.
    Lengths around every boundary of that geometry (one tile / two tiles, tile core, k, w), several specs incl. the sketch
    variant, non-ACGT bytes and palindromic k-mers (islands of the exact machine on the short geometry), a long contig in
    the batch; compared with the oracle and with the same call on the 4096-position tiles."""
    import pgrtk_amd as P
    import seqgen
    rng = np.random.default_rng(404)
    for spec_t in ((80, 56, 4, 64, False), (48, 56, 4, 12, False), (33, 31, 3, 8, False), (128, 56, 2, 64, False), (17, 9, 2, 0, False),
                   (80, 56, 1, 64, False), (80, 21, 2, 16, True)):
        w, k, r, ms, sk = spec_t
        tc = (1024 - 2 * ((1 if sk else w) - 1)) // 64 * 64
        lens = [0, 1, k - 1, k, k + 1, k + w - 2, k + w - 1, k + w, 2 * w + k, 3 * w, 500, 999, 1000, 1001, 1007, 1008, 1009, 1023, 1024,
                1025, tc - 1, tc, tc + 1, 2 * tc - w, 2 * tc - 1, 2 * tc, 2 * tc + 1, 2 * tc + w + k, 3 * tc, 3000, 4017, 5000]
        lens = [n for n in lens if n >= 0]
        seqs = [seqgen.rnd(rng, n) for n in lens] * 3
        # adversarial short contigs: N runs, lower case, bytes 0..3, low complexity, palindromic stretches, tandem repeats
        for mode in range(seqgen.N_MODES):
            for L in (300, 900, 1008, 1100, 1700, 2600):
                seqs.append(seqgen.adversarial(rng, mode, L))
        seqs.append(seqgen.rnd(rng, 400) + b"N" * 700 + seqgen.rnd(rng, 500))
        seqs.append(b"N" * 1008)
        seqs.append(b"AT" * 504)
        seqs.append(b"A" * 1000)
        seqs.append(seqgen.rnd(rng, 30_000))  # one long contig among the reads (mean stays below 2048)
        seqs.append(seqgen.rnd(rng, 9_000) + b"N" * 3000 + b"AT" * 90 + seqgen.rnd(rng, 9_000))
        assert sum(len(s) for s in seqs) / len(seqs) <= 2048
        rids = [int(v) for v in rng.integers(0, 2 ** 31, len(seqs))]
        spec = P.make_spec(w, k, r, ms, sketch=sk) if sk else P.make_spec(w, k, r, ms)
        osp = oracle.spec(w, k, r, ms, sketch=sk) if sk else oracle.spec(w, k, r, ms)
        with gpu_ctx.options(no_small_path=1):
            short = P.sequence_to_shmmrs_batch(seqs, spec, rids=rids, ctx=gpu_ctx)
            with gpu_ctx.options(no_short_tiles=1):
                long_ = P.sequence_to_shmmrs_batch(seqs, spec, rids=rids, ctx=gpu_ctx)
        for i, s in enumerate(seqs):
            ref = oracle.sequence_to_shmmrs(rids[i], s, osp)
            _same(ref, short[i], "one-wavefront tiles, spec %s contig %d len %d" % (spec_t, i, len(s)))
            _same(ref, long_[i], "4096-position tiles, spec %s contig %d len %d" % (spec_t, i, len(s)))
    # resident batch of 20 000 reads of ragged length (past the small path's 4096 contigs): whole-contig checksums
    n = 20_000
    lens = [int(v) for v in rng.integers(700, 1400, n)]
    b = P.Batch.synthetic(lens, seed=43, ctx=gpu_ctx)
    spec_t = (80, 56, 4, 64)
    sh = b.shmmrs(P.make_spec(*spec_t))
    with gpu_ctx.options(no_short_tiles=1):
        sh4 = b.shmmrs(P.make_spec(*spec_t))
    assert sh.count == sh4.count and np.array_equal(sh.checksum(), sh4.checksum()) and np.array_equal(sh.offsets(), sh4.offsets())
    sums, off = sh.checksum(), sh.offsets()
    import bench
    for c in range(0, n, 97):
        ref = oracle.sequence_to_shmmrs(c, bench.synth_contig_ascii(43, c, lens[c]), oracle.spec(*spec_t))
        assert int(off[c + 1] - off[c]) == len(ref) and np.array_equal(sums[c], oracle.shmmr_checksum(ref)), c


def test_many_dense_short_contigs_take_the_retry_paths(oracle, gpu_ctx):
    """20 000 low-complexity reads (homopolymers, (AC)n, two-letter noise: ties emit every position, shmmrutils.rs:516-527): the
    level-1 overflow region and the list stage's grid are sized from the density of random sequence plus a small allowance per
    contig, so this batch overflows both and runs again with the true counts -- still bit exact, and the next call (which
    remembers what this one needed) as well."""
    import pgrtk_amd as P
    rng = np.random.default_rng(91)
    seqs = []
    for i in range(20_000):
        L = int(rng.integers(150, 1200))
        m = i % 4
        if m == 0:
            seqs.append(b"A" * L)
        elif m == 1:
            seqs.append((b"AC" * (L // 2 + 1))[:L])
        elif m == 2:
            seqs.append(bytes(rng.choice(np.frombuffer(b"AC", dtype=np.uint8), L)))
        else:
            seqs.append(bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), L)))
    spec_t = (80, 56, 4, 64)
    for attempt in range(2):
        got = P.sequence_to_shmmrs_batch(seqs, P.make_spec(*spec_t), ctx=gpu_ctx)
        for i in range(0, len(seqs), 7):
            ref = oracle.sequence_to_shmmrs(i, seqs[i], oracle.spec(*spec_t))
            _same(ref, got[i], "attempt %d contig %d len %d" % (attempt, i, len(seqs[i])))


def test_island_options_agree_with_the_oracle(oracle, gpu_ctx):
    """the exact islands' variants (context options): chunks shorter than a tile / the round-3 minimum of 4096
    positions, islands listed beside the tile kernel / behind it, the state relay on / off -- on contigs with gaps, isolated N,
    palindromic arrays and low-complexity stretches, through the host entry point (the packer counts the non-ACGT bytes: the
    islands are listed while the tiles run) and as a resident batch; every variant bit-identical to the oracle."""
    import pgrtk_amd as P
    import seqgen
    rng = np.random.default_rng(606)

    def contig(L):
        parts, n = [], 0
        while n < L:
            r = rng.random()
            if r < 0.55:
                p = seqgen.rnd(rng, int(rng.integers(2_000, 60_000)))
            elif r < 0.7:
                p = b"N" * int(rng.integers(1, 30_000))
            elif r < 0.8:
                p = (b"AT", b"ACGT", b"GAATTC")[int(rng.integers(0, 3))] * int(rng.integers(50, 3000))
            elif r < 0.9:
                p = seqgen.rnd(rng, int(rng.integers(100, 9000)), b"AC")
            else:
                p = seqgen.rnd(rng, 500) + b"N" + seqgen.rnd(rng, 700)
            parts.append(p)
            n += len(p)
        return b"".join(parts)[:L]
    seqs = [contig(700_000), contig(150_000), seqgen.rnd(rng, 50_000), b"N" * 40_000 + seqgen.rnd(rng, 30_000) + b"N" * 9_000, contig(20_000)]
    spec_t = (80, 56, 4, 64)
    spec, osp = P.make_spec(*spec_t), oracle.spec(*spec_t)
    refs = [oracle.sequence_to_shmmrs(i, s, osp) for i, s in enumerate(seqs)]
    variants = [{}, {"no_pre_islands": 1}, {"island_chunk_min": 4096}, {"island_chunk_min": 1024, "no_island_relay": 1},
                {"island_chunk_min": 32768}, {"no_short_tiles": 1, "no_small_path": 1},
                # round 5: the first round of the islands around non-ACGT bytes beside the tile kernel (default) / behind it on
                # its stream / only when the flags are in; a batch with a palindromic array drops the early round and starts over
                {"early_islands_in_stream": 1}, {"no_early_islands": 1}]
    for opt in variants:
        with gpu_ctx.options(**dict(opt, no_small_path=1)):
            got = P.sequence_to_shmmrs_batch(seqs, spec, ctx=gpu_ctx)  # host entry point
            b = P.Batch.from_seqs(seqs, ctx=gpu_ctx)
            sh = b.shmmrs(spec)
        for i in range(len(seqs)):
            _same(refs[i], got[i], "host entry, options %s, contig %d" % (opt, i))
        sums, off = sh.checksum(), sh.offsets()
        for i in range(len(seqs)):
            assert int(off[i + 1] - off[i]) == len(refs[i]) and np.array_equal(sums[i], oracle.shmmr_checksum(refs[i])), (opt, i)


def test_reserved_arena_serves_every_device_allocation(oracle, gpu_ctx):
    """pgr_ctx_reserve (include/pgr_hip.h): one device block, allocated and touched up front, that workspaces, batches, results and the
    index of a context are carved from.  A build (pipe, two batches in flight, records into one index) + queries in a context with a
    2 GiB arena: the runtime's allocator is never asked (fallback_calls == 0), the index and the chains equal those of the shared
    context (which has no arena) and the shimmers of sampled contigs equal the oracle's; a request the arena cannot hold is served
    by the runtime and counted; blocks come back: after the build is destroyed the arena is as good as empty."""
    import pgrtk_amd as P
    sp_t = (80, 56, 4, 64, False)
    spec, osp = P.make_spec(*sp_t), oracle.spec(*sp_t)
    lens = [[3_000_000, 1_500_000, 70_000, 0, 999], [2_000_000, 2_500_000], [4_000_000]]

    def build(ctx):
        ix = P.Index(spec, ctx=ctx)
        pipe = P.Pipe(spec, ctx=ctx)
        kept, c0 = [], 0
        for ls in lens:
            ids = list(range(c0, c0 + len(ls)))
            c0 += len(ls)
            b = P.Batch.synthetic(ls, seed=21, ctx=ctx, contig_ids=ids)
            if pipe.in_flight == 2:
                kept.append(pipe.collect()[0])
            pipe.submit(b, sids=ids, index=ix)
        while pipe.in_flight:
            kept.append(pipe.collect()[0])
        pipe.close()
        ix.finalize()
        return ix, kept

    ctx = P.Context(0)
    ctx.reserve(2 << 30)
    st0 = ctx.arena_stats()
    assert st0["reserved"] == 2 << 30 and st0["used"] == 0
    ix, kept = build(ctx)
    ref_ix, _ = build(gpu_ctx)
    assert ix.n_records == ref_ix.n_records > 30_000 and ix.records_checksum() == ref_ix.records_checksum()
    mm, off = kept[0].download()
    for c in (0, 2, 4):
        ref = oracle.sequence_to_shmmrs(c, oracle.synth_contig(21, c, lens[0][c]).tobytes(), osp)
        g = mm[int(off[c]):int(off[c + 1])]
        assert len(ref) == len(g) and np.array_equal(ref["x"], g["x"]) and np.array_equal(ref["y"], g["y"]), c
    qs = [oracle.synth_contig(21, 5, 2_000_000).tobytes()[100_000:112_000], oracle.synth_contig(21, 7, 4_000_000).tobytes()[3_000_000:3_009_000]]
    ra, rb = ix.query_hps_raw(qs, 0.025), ref_ix.query_hps_raw(qs, 0.025)
    assert all(np.array_equal(ra[k], rb[k]) for k in ("q_off", "t_sid", "t_off", "c_score", "c_off", "hps")) and len(ra["hps"]) > 20
    st = ctx.arena_stats()
    assert st["fallback_calls"] == 0 and st["fallback_bytes"] == 0, st
    assert 0 < st["used"] <= st["peak_used"] <= st["reserved"]
    big = P.Batch.synthetic([1_000_000_000] * 8, seed=3, ctx=ctx)  # 8 Gbp: 2-bit planes + validity = 3 GB > the arena
    st2 = ctx.arena_stats()
    assert st2["fallback_calls"] >= 1 and st2["fallback_bytes"] >= 900_000_000, st2  # (the planes are separate blocks: what still fits is carved, the rest comes from the runtime)
    assert big.total_bases == 8_000_000_000
    del big, ix, kept
    ctx.trim()
    st3 = ctx.arena_stats()
    assert st3["used"] < st["used"], (st, st3)  # (workspaces of the context stay; results, batches, the index and the cache went back)
