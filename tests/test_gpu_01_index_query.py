"""GPU parity: ShmmrToFrags index, raw query, count filters and sparse hit chaining vs the CPU oracle.

Reference path: load_index_from_seq_vec (pgr-db/src/seq_db.rs:573-615), raw_query_fragment
(seq_db.rs:1200-1228), aln::query_fragment_to_hps (aln.rs:147-242), aln::sparse_aln (aln.rs:12-142),
through the SeqIndexDB surface of pgr-tk/src/lib.rs.
Chaining has no expected output in the reference (aln.rs:484): parity here is oracle <-> GPU.
"""
import os

import numpy as np
import pytest

import procutil
import seqgen

pytestmark = pytest.mark.gpu

COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgtN", b"TGCAtgcaN"):
    COMP[_a] = _b


def revcomp(s):
    return COMP[np.frombuffer(s, dtype=np.uint8)][::-1].tobytes()


def _oracle_hps_to_tuples(res):
    return [(sid, [(sc, [((h[0], h[1], h[2]), (h[3], h[4], h[5])) for h in hps]) for sc, hps in chains])
            for sid, chains in res]


def test_seqindexdb_golden_mdb(oracle, gpu_ctx, golden_dir, tmp_path):
    """G1 through the public API, exactly like gen_frag_db.py: load_from_fastx('test_seqs.fa') ->
    frag_map == test_seqs_frag.mdb (820 signatures, 55 keys, global fragment ids, per-key order);
    written .mdb/.midx round-trip to the same content."""
    import pgrtk_amd as P
    fa = os.path.join(golden_dir, "test_seqs.fa")
    sdb = P.SeqIndexDB(ctx=gpu_ctx)
    sdb.load_from_fastx(fa)
    gspec, g = oracle.read_mdb(os.path.join(golden_dir, "test_seqs_frag.mdb"))
    assert sdb.get_shmmr_spec() == (80, 56, 4, 64, False)
    assert sdb.get_shmmr_map() == g
    assert sorted(sdb.get_shmmr_pair_list()) == sorted((k[0], k[1], s[1], s[2], s[3], s[4]) for k, v in g.items() for s in v)
    k0 = next(iter(g))
    assert sdb.get_shmmr_pair_count(k0) == len(g[k0]) and sdb.get_shmmr_pair_count((1, 2)) == 0
    sdb.write_shmmr_map_index(str(tmp_path / "out"))
    spec2, g2 = oracle.read_mdb(str(tmp_path / "out.mdb"))
    assert spec2 == gspec and g2 == g
    assert os.path.getsize(str(tmp_path / "out.mdb")) == 15291
    ref_midx = [l.split("\t")[:3] for l in open(os.path.join(golden_dir, "test_seqs_frag.midx")).read().splitlines()]
    got_midx = [l.split("\t") for l in open(str(tmp_path / "out.midx")).read().splitlines()]
    assert [g[:3] for g in got_midx] == ref_midx and all(g[3] == fa for g in got_midx)
    assert len(sdb.seq_info) == 66 and sdb.seq_info[0][2] == 3385


def _build_pair(oracle, gpu_ctx, seqs, spec_t=(80, 56, 4, 64)):
    import pgrtk_amd as P
    sdb = P.SeqIndexDB(ctx=gpu_ctx)
    sdb.load_from_seq_list([("s%d" % i, s) for i, s in enumerate(seqs)], w=spec_t[0], k=spec_t[1], r=spec_t[2],
                           min_span=spec_t[3])
    oix = oracle.Index(oracle.spec(*spec_t))
    for i, s in enumerate(seqs):
        oix.add_seq(i, s)
    oix.finalize()
    return sdb, oix


def _make_db_seqs(rng, n=12, L=120000):
    """contigs sharing segments (so that queries hit several targets) + an exact duplicate"""
    core = [seqgen.rnd(rng, 30000) for _ in range(4)]
    seqs = []
    for i in range(n):
        parts = [seqgen.rnd(rng, int(rng.integers(1000, 20000)))]
        for j in rng.permutation(4)[: int(rng.integers(1, 4))]:
            c = core[j]
            parts.append(c if rng.random() < 0.6 else revcomp(c))
            parts.append(seqgen.rnd(rng, int(rng.integers(500, 8000))))
        seqs.append(b"".join(parts)[:L])
    seqs.append(seqs[0])
    seqs.append(b"")
    seqs.append(seqgen.rnd(rng, 100))
    return seqs, core


def test_index_records_vs_oracle(oracle, gpu_ctx):
    rng = np.random.default_rng(21)
    seqs, _ = _make_db_seqs(rng)
    sdb, oix = _build_pair(oracle, gpu_ctx, seqs)
    ref = oix.records()
    import ctypes as C
    from pgrtk_amd import _ffi
    p, n = C.c_void_p(), C.c_uint64()
    gpu_ctx.check(_ffi.lib().pgr_index_download(gpu_ctx.handle, sdb._ix, C.byref(p), C.byref(n)))
    got = _ffi.take(p, int(n.value), _ffi.FRAG_REC)
    assert len(got) == len(ref) and _ffi.lib().pgr_index_n_keys(sdb._ix) == oix.n_keys()
    for f in ("h0", "h1", "frg_id", "sid", "bgn", "end", "orient"):
        assert np.array_equal(ref[f], got[f]), f


@pytest.mark.parametrize("variant", ["default", "max_gap", "oriented", "tight_counts", "span2"])
def test_query_fragment_to_hps_vs_oracle(oracle, gpu_ctx, variant):
    rng = np.random.default_rng(22)
    seqs, core = _make_db_seqs(rng)
    sdb, oix = _build_pair(oracle, gpu_ctx, seqs)
    kw = dict(penalty=0.025, max_count=128, max_count_query=128, max_count_target=128, max_aln_span=8, max_gap=None,
              orientated=False)
    if variant == "max_gap":
        kw["max_gap"] = 2000
    elif variant == "oriented":
        kw["orientated"] = True
    elif variant == "tight_counts":
        kw.update(max_count=1, max_count_query=1, max_count_target=1)
    elif variant == "span2":
        kw.update(max_aln_span=2, penalty=0.5)
    queries = []
    for i in range(40):
        src = seqs[int(rng.integers(0, 12))]
        a = int(rng.integers(0, max(1, len(src) - 12000)))
        q = src[a:a + int(rng.integers(2000, 12000))]
        if i % 2:
            q = revcomp(q)
        if i % 5 == 0:  # a few SNPs
            qa = bytearray(q)
            for p in rng.integers(0, len(qa), 5):
                qa[p] = ord("ACGT"[int(rng.integers(0, 4))])
            q = bytes(qa)
        queries.append(q)
    queries += [core[0] + core[1], core[2] * 3, b"", seqgen.rnd(rng, 50), seqgen.rnd(rng, 5000)]
    got = sdb.query_fragments_to_hps(queries, kw["penalty"], kw["max_count"], kw["max_count_query"],
                                     kw["max_count_target"], kw["max_aln_span"], kw["max_gap"], kw["orientated"])
    n_chains = 0
    for q, g in zip(queries, got):
        ref = _oracle_hps_to_tuples(oix.query_fragment_to_hps(
            q, kw["penalty"], kw["max_count"], kw["max_count_query"], kw["max_count_target"], kw["max_aln_span"],
            kw["max_gap"], kw["orientated"]))
        assert [t[0] for t in ref] == [t[0] for t in g]
        for (sid, rc), (_, gc) in zip(ref, g):
            assert len(rc) == len(gc), (variant, sid)
            for (rs, rh), (gs, gh) in zip(rc, gc):
                assert rs == gs, (variant, sid, rs, gs)  # f32, bit exact
                assert rh == gh
            n_chains += len(rc)
    assert n_chains > 20
    # single-query entry point == batched
    one = sdb.query_fragment_to_hps(queries[1], kw["penalty"], kw["max_count"], kw["max_count_query"],
                                    kw["max_count_target"], kw["max_aln_span"], kw["max_gap"], kw["orientated"])
    assert one == got[1]


def test_raw_query_fragment_vs_oracle(oracle, gpu_ctx):
    rng = np.random.default_rng(23)
    seqs, core = _make_db_seqs(rng)
    sdb, oix = _build_pair(oracle, gpu_ctx, seqs)
    q = core[1][2000:20000]
    raw = sdb.query_fragment(q)
    sh = oracle.sequence_to_shmmrs(0, q, oracle.spec())
    qr = oracle.frag_recs(sh, 0, query_side=True)
    assert len(raw) == len(qr) > 10
    recs = oix.records()
    n_sig = 0
    for r, (key, pos, sigs) in zip(qr, raw):
        assert key == (int(r["h0"]), int(r["h1"])) and pos == (int(r["bgn"]), int(r["end"]), int(r["orient"]))
        m = recs[(recs["h0"] == r["h0"]) & (recs["h1"] == r["h1"])]
        assert [(s[1], s[2], s[3], s[4]) for s in sigs] == [(int(x["sid"]), int(x["bgn"]), int(x["end"]), int(x["orient"])) for x in m]
        n_sig += len(sigs)
    assert n_sig > 10
    mp = sdb.get_match_positions_with_fragment(q)
    assert all(v == sorted(v) for v in mp.values()) and len(mp) >= 1


def test_sparse_aln_test_hits(oracle, gpu_ctx, golden_dir):
    """the reference's own input for sparse_aln (aln.rs:458-485: test_hits, max_span 8, penalty 0.5)"""
    import pgrtk_amd as P
    h = np.loadtxt(os.path.join(golden_dir, "test_hits"), dtype=np.uint32)
    hits = [((int(r[0]), int(r[1]), int(r[2])), (int(r[3]), int(r[4]), int(r[5]))) for r in h]
    for (span, pen, gap, ori) in [(8, 0.5, None, False), (8, 0.025, None, False), (3, 0.5, 5000, False), (8, 0.5, None, True)]:
        got = P.sparse_aln(hits, span, pen, gap, ori, ctx=gpu_ctx)
        ref = oracle.sparse_aln([(a[0], a[1], a[2], b[0], b[1], b[2]) for a, b in hits], span, pen, gap, ori)
        ref = [(sc, [((x[0], x[1], x[2]), (x[3], x[4], x[5])) for x in hp]) for sc, hp in ref]
        assert len(got) == len(ref)
        assert got == ref


def test_sparse_aln_duplicates_and_small(oracle, gpu_ctx):
    """identical hit pairs share one map slot in the reference (FxHashMap keyed by value)"""
    import pgrtk_amd as P
    rng = np.random.default_rng(5)
    for trial in range(20):
        n = int(rng.integers(2, 40))
        hits = []
        q = 0
        for i in range(n):
            q += int(rng.integers(0, 300))
            ln = int(rng.integers(50, 400))
            t = int(rng.integers(0, 5000))
            hits.append(((q, q + ln, int(rng.integers(0, 2))), (t, t + ln, int(rng.integers(0, 2)))))
        hits += [hits[int(i)] for i in rng.integers(0, n, 3)]  # duplicates
        order = rng.permutation(len(hits))
        hits = [hits[int(i)] for i in order]
        got = P.sparse_aln(hits, 4, 0.1, None, False, ctx=gpu_ctx)
        ref = oracle.sparse_aln([(a[0], a[1], a[2], b[0], b[1], b[2]) for a, b in hits], 4, 0.1, None, False)
        ref = [(sc, [((x[0], x[1], x[2]), (x[3], x[4], x[5])) for x in hp]) for sc, hp in ref]
        assert got == ref, trial


@pytest.mark.parametrize("n", [15, 16, 17, 64, 129, 256, 257, 700, 3584, 3585, 6000, 20000])
def test_sparse_aln_long_groups(oracle, gpu_ctx, n):
    """groups of >= 16 hits take the wavefront-per-group kernel (9 KB LDS image up to 256 hits, 129 KB up to 3584, global
    memory above; shorter groups one thread each):
    collinear runs with noise, ties in qb, equal query intervals on different targets, duplicates, both orientations"""
    import pgrtk_amd as P
    rng = np.random.default_rng(1000 + n)
    hits = []
    q, t = 0, int(rng.integers(0, 10 ** 6))
    while len(hits) < n:
        q += int(rng.integers(1, 400))  # strictly increasing: a duplicate never shares its qb with a third hit
        t += int(rng.integers(-200, 600))
        t = max(t, 0)
        ln = int(rng.integers(60, 500))
        qo, to = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        hits.append(((q, q + ln, qo), (t, t + ln, to)))
        r = rng.random()
        if r < 0.15:    # same query interval, another target position (repeat)
            t2 = int(rng.integers(0, 10 ** 6))
            hits.append(((q, q + ln, qo), (t2, t2 + ln, to)))
        elif r < 0.25:  # same qb, different qe
            hits.append(((q, q + ln + 7, qo), (t + 3, t + ln + 10, to)))
        elif r < 0.30:  # exact duplicate
            hits.append(hits[-1])
        elif r < 0.33:  # a jump: starts a new chain
            t = int(rng.integers(0, 10 ** 6))
    hits = hits[:n]
    hits = [hits[int(i)] for i in rng.permutation(n)]
    flat = [(a[0], a[1], a[2], b[0], b[1], b[2]) for a, b in hits]
    compared = 0
    for (span, pen, gap, ori) in [(8, 0.025, None, False), (2, 0.5, 3000, False), (8, 0.1, None, True), (64, 0.05, None, False),
                                  (1, 0.01, None, False)]:
        got = P.sparse_aln(hits, span, pen, gap, ori, ctx=gpu_ctx)
        try:
            ref = oracle.sparse_aln(flat, span, pen, gap, ori)
        except RuntimeError:
            # exact duplicates can make the predecessor map cyclic (a value slot re-scored after a later hit chose
            # it); the reference then never terminates (aln.rs:121-128 marks visited only after the walk) -- the
            # GPU path returned something finite, there is nothing to compare it with
            continue
        ref = [(sc, [((x[0], x[1], x[2]), (x[3], x[4], x[5])) for x in hp]) for sc, hp in ref]
        assert got == ref, (n, span, pen, gap, ori)
        compared += 1
    assert compared >= 4


def test_sparse_aln_look_back_beyond_the_lds_ring(oracle, gpu_ctx):
    """groups above 3584 hits keep the last 3584 hits in an LDS ring; an `oriented` look-back that has to skip
    thousands of hits of the other orientation class reaches below the ring and reads global memory"""
    import pgrtk_amd as P
    rng = np.random.default_rng(77)
    hits, q, t = [], 0, 1000

    def run(n, cls):
        nonlocal q, t
        for _ in range(n):
            q += int(rng.integers(1, 300))
            t += int(rng.integers(1, 300))
            ln = int(rng.integers(60, 300))
            qo = int(rng.integers(0, 2))
            hits.append(((q, q + ln, qo), (t, t + ln, qo ^ cls)))
    run(150, 0)
    run(5200, 1)
    run(150, 0)
    run(4100, 1)
    run(60, 0)
    flat = [(a[0], a[1], a[2], b[0], b[1], b[2]) for a, b in hits]
    for (span, pen, gap, ori) in [(8, 0.001, None, True), (3, 0.01, None, True), (8, 0.001, None, False)]:
        got = P.sparse_aln(hits, span, pen, gap, ori, ctx=gpu_ctx)
        ref = oracle.sparse_aln(flat, span, pen, gap, ori)
        ref = [(sc, [((x[0], x[1], x[2]), (x[3], x[4], x[5])) for x in hp]) for sc, hp in ref]
        assert got == ref, (span, pen, gap, ori)
    # with orientation the three class-0 runs chain across the class-1 blocks
    got = P.sparse_aln(hits, 8, 0.001, None, True, ctx=gpu_ctx)
    assert max(len(hp) for _, hp in got) > 5200


def test_get_shmmr_pairs_from_seq(oracle, gpu_ctx):
    import pgrtk_amd as P
    rng = np.random.default_rng(6)
    s = seqgen.rnd(rng, 30000)
    for pad in (False, True):
        got = P.get_shmmr_pairs_from_seq(s, 80, 56, 4, 16, pad, ctx=gpu_ctx)
        sh = oracle.sequence_to_shmmrs(0, s, oracle.spec(80, 56, 4, 16), pad)
        ref = oracle.frag_recs(sh, 0, query_side=True)
        assert got == [(int(r["h0"]), int(r["h1"]), int(r["bgn"]), int(r["end"]), int(r["orient"])) for r in ref]


def test_shmmr_dots_and_source_count(oracle, gpu_ctx, golden_dir, tmp_path):
    """get_shmmr_dots (lib.rs:1649-1697) and get_shmmr_pair_source_count (lib.rs:669-727) from oracle shimmers"""
    import pgrtk_amd as P
    rng = np.random.default_rng(8)
    a = seqgen.rnd(rng, 30000)
    b = a[5000:20000] + revcomp(a[1000:9000]) + seqgen.rnd(rng, 4000) + a[5000:9000]
    for (w, k, r, ms) in [(80, 56, 4, 16), (24, 24, 2, 8)]:
        x, y = P.get_shmmr_dots(a, b, w, k, r, ms, ctx=gpu_ctx)
        sp = oracle.spec(w, k, r, ms)
        s0, s1 = oracle.sequence_to_shmmrs(0, a, sp), oracle.sequence_to_shmmrs(1, b, sp)
        base = {}
        for m in s0:
            base.setdefault(int(m["x"]) >> 8, []).append((int(m["y"]) & 0xFFFFFFFF) >> 1)
        rx, ry = [], []
        for m in s1:
            for px in base.get(int(m["x"]) >> 8, ()):
                rx.append(px)
                ry.append((int(m["y"]) & 0xFFFFFFFF) >> 1)
        assert (x, y) == (rx, ry) and len(x) > 20
    # two FASTA sources with shared content
    f1, f2 = tmp_path / "one.fa", tmp_path / "two.fa"
    f1.write_text(">a\n%s\n>b\n%s\n" % (a.decode(), a[2000:25000].decode()))
    f2.write_text(">c\n%s\n" % a[:28000].decode())
    sdb = P.SeqIndexDB(ctx=gpu_ctx)
    sdb.load_from_fastx(str(f1))
    sdb.append_from_fastx(str(f2))
    key = next(k for k, v in sdb.get_shmmr_map().items() if len(v) == 3)
    assert sdb.get_shmmr_pair_source_count(key) == sorted([(str(f1), 2), (str(f2), 1)])
    assert sdb.get_shmmr_pair_source_count(key, 2) == [(str(f2), 1)]
    assert sdb.get_shmmr_pair_source_count((1, 2)) == []


def test_cli_mdb_and_query(oracle, gpu_ctx, golden_dir, tmp_path):
    """pgr-mdb / pgr-query counterparts end to end: index-only .mdb (per-contig fragment ids) == the golden
    frag_map after undoing the FASTX-backend renumbering; the index file round-trips through the GPU; a query
    cut out of a contig is reported on that contig at the right place."""
    import pgrtk_amd as P
    from pgrtk_amd import cli
    fa = os.path.join(golden_dir, "test_seqs.fa")
    lst = tmp_path / "list.txt"
    lst.write_text(fa + "\n")
    prefix = str(tmp_path / "idx")
    cli.main(["mdb", str(lst), prefix])
    spec_t, m = oracle.read_mdb(prefix + ".mdb")
    _, g = oracle.read_mdb(os.path.join(golden_dir, "test_seqs_frag.mdb"))
    assert spec_t == (80, 56, 4, 64, 0) and set(m) == set(g)
    # golden ids are global (Prefix +1, pairs, Suffix +1 per sequence); the index-only path numbers per contig
    recs = oracle.read_fasta(fa)
    pairs = [0] * len(recs)
    for v in g.values():
        for s in v:
            pairs[s[1]] += 1
    base, acc = [], 0
    for c in pairs:
        base.append(acc)
        acc += 2 if c == 0 else c + 2
    for key, sigs in g.items():
        assert [(s[0] - base[s[1]] - 1, s[1], s[2], s[3], s[4]) for s in sigs] == m[key]
    assert open(prefix + ".midx").read().splitlines()[0].split("\t")[:3] == ["0", "3385", recs[0][0].decode()]
    # query through the index file
    qfa = tmp_path / "q.fa"
    src = recs[5][1]
    qfa.write_text(">q0 some comment\n%s\n>q1\n%s\n" % (src[200:3000].decode(), P.cli.reverse_complement(src[100:3300]).decode()))
    out = str(tmp_path / "res")
    cli.main(["query", prefix, str(qfa), out, "--only-summary"])
    for i in (0, 1):
        lines = [l.split("\t") for l in open(out + ".%03d.hit" % i).read().splitlines() if not l.startswith("#")]
        assert any(l[7] == recs[5][0].decode() for l in lines)  # the source contig is among the targets
    # same query against the FASTA directly (FASTX backend) + sequences of the hit regions
    cli.main(["query", fa, str(qfa), out + "_fx", "--fastx_file"])
    a = [l.split("\t")[1:11] for l in open(out + ".000.hit").read().splitlines() if not l.startswith("#")]
    b = [l.split("\t")[1:11] for l in open(out + "_fx.000.hit").read().splitlines() if not l.startswith("#")]
    assert [x[:5] + x[6:] for x in a] == [x[:5] + x[6:] for x in b]  # identical but for the src column
    assert os.path.getsize(out + "_fx.000.fa") > 0
    sdb = P.SeqIndexDB(ctx=gpu_ctx)
    sdb.load_from_mdb_index(prefix)
    assert sdb.get_shmmr_map() == m and sdb.get_shmmr_spec() == (80, 56, 4, 64, False)
    # ext.rs:285: the file-backed entry point pgr-query / pgr-web call; an error on in-memory backends as in the reference
    q = src[200:3000]
    got = sdb.query_fragment_to_hps_from_mmap_file(q, 0.025)
    assert got == sdb.query_fragment_to_hps(q, 0.025) and any(sid == 5 for sid, _ in got)
    mem = P.SeqIndexDB(ctx=gpu_ctx)
    mem.load_from_seq_list([(n.decode(), s) for n, s in recs[:3]])
    with pytest.raises(RuntimeError):
        mem.query_fragment_to_hps_from_mmap_file(q, 0.025)


def test_cpp_host_programs_match_python_cli(oracle, gpu_ctx, golden_dir, tmp_path):
    """the C++ host programs above the C ABI (pgr-tk_amd/bin/pgr-mdb, pgr-query: the reference's callers are compiled
    code too) write the same files as the Python counterparts: .mdb / .midx byte for byte, every .hit / .hit.bed / .fa"""
    from pgrtk_amd import cli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bindir = os.path.join(root, "pgr-tk_amd", "bin")
    assert os.path.exists(os.path.join(bindir, "pgr-mdb")), "host programs not built: python __graft_entry__.py build"
    fa = os.path.join(golden_dir, "test_seqs.fa")
    fq = tmp_path / "extra.fq"   # a FASTQ input next to the FASTA one
    recs = oracle.read_fasta(fa)
    fq.write_text("".join("@r%d x\n%s\n+\n%s\n" % (i, recs[i][1][:2500].decode(), "I" * len(recs[i][1][:2500])) for i in (3, 9)))
    import gzip
    fgz = tmp_path / "some.fa.gz"   # gzip input, CRLF line ends, a record without sequence
    with gzip.open(fgz, "wb") as f:
        f.write(b">g0 first\r\n" + recs[12][1][:1500] + b"\r\n" + recs[12][1][1500:4000] + b"\r\n>empty\r\n>g1\r\n" + recs[13][1][:3000] + b"\n")
    lst = tmp_path / "list.txt"
    lst.write_text(fa + "\n" + str(fq) + "\n" + str(fgz) + "\n")
    py, cc = str(tmp_path / "py"), str(tmp_path / "cc")
    for args in ([], ["-w", "48", "-k", "56", "-r", "4", "-m", "12"], ["--reference-sid-quirk"]):
        cli.main(["mdb", str(lst), py] + args)
        procutil.run_bounded([os.path.join(bindir, "pgr-mdb"), str(lst), cc] + args, check=True, timeout=120)
        assert open(py + ".mdb", "rb").read() == open(cc + ".mdb", "rb").read()
        assert open(py + ".midx").read() == open(cc + ".midx").read()
    cli.main(["mdb", str(lst), py])
    procutil.run_bounded([os.path.join(bindir, "pgr-mdb"), str(lst), cc], check=True, timeout=120)
    # queries: forward, reverse complement, one spanning two contigs, one without hits
    rng = np.random.default_rng(4)
    src = recs[5][1]
    qfa = tmp_path / "q.fa"
    qfa.write_text(">q0 c\n%s\n>q1\n%s\n>q2\n%s\n>q3\n%s\n" % (
        src[200:3000].decode(), cli.reverse_complement(src[100:3300]).decode(),
        (recs[7][1][:1500] + recs[11][1][:1800]).decode(), seqgen.rnd(rng, 3000).decode()))
    for db_py, db_cc, extra in ((py, cc, []), (py, cc, ["--bed-summary"]), (fa, fa, ["--fastx-file"]),
                                (py, cc, ["--merge-range-tol", "10", "--max-aln-chain-span", "2", "-g", "0.5"])):
        for f in os.listdir(tmp_path):
            if f.startswith(("opy.", "occ.")):
                os.remove(tmp_path / f)
        cli.main(["query", db_py, str(qfa), str(tmp_path / "opy")] + extra)
        procutil.run_bounded([os.path.join(bindir, "pgr-query"), db_cc, str(qfa), str(tmp_path / "occ")] + extra, check=True, timeout=120)
        outs_py = sorted(f[4:] for f in os.listdir(tmp_path) if f.startswith("opy."))
        outs_cc = sorted(f[4:] for f in os.listdir(tmp_path) if f.startswith("occ."))
        assert outs_py == outs_cc and len(outs_py) >= 4
        for f in outs_py:
            a = open(tmp_path / ("opy." + f)).read()
            b = open(tmp_path / ("occ." + f)).read()
            if db_py != db_cc:
                pass
            assert a.replace(py, "DB") == b.replace(cc, "DB"), f
    assert any(len(open(tmp_path / ("occ." + f)).read().splitlines()) > 1 for f in outs_cc)


def test_index_from_exchanged_shimmer_lists(oracle, gpu_ctx):
    """multi-GPU merge path on one GPU: two "ranks" index disjoint contig shards, copy their MM128 lists with
    global sequence ids (what the RCCL all-gather moves), and an index built from the concatenated lists
    (pgr_index_add_shmmrs) equals the index built from all sequences directly"""
    import ctypes as C
    import torch
    import pgrtk_amd as P
    rng = np.random.default_rng(31)
    seqs = [seqgen.rnd(rng, int(L)) for L in (50_000, 120_000, 100, 0, 80_000, 30_000, 64_000)]
    sp = P.make_spec()
    shards = [(0, seqs[:3]), (3, seqs[3:])]
    parts = []
    for contig0, sub in shards:
        b = P.Batch.from_seqs(sub, ctx=gpu_ctx)
        sh = b.shmmrs(sp)
        t = torch.empty((sh.count + 4, 2), dtype=torch.int64, device="cuda:0")
        n = sh.copy_into(t.data_ptr(), t.shape[0], rid_add=contig0)
        parts.append(t[:n])
    gathered = torch.cat(parts[::-1], dim=0)  # rank order does not matter
    ix = P.Index(sp, ctx=gpu_ctx)
    ix.add_shmmrs(device_ptr=gathered.data_ptr(), n=gathered.shape[0])
    ix.finalize()
    ref = P.Index(sp, ctx=gpu_ctx)
    ref.add_seqs(seqs)
    ref.finalize()
    a, b2 = ix.download(), ref.download()
    assert len(a) == len(b2) > 100 and ix.n_keys == ref.n_keys
    for f in ("h0", "h1", "frg_id", "sid", "bgn", "end", "orient"):
        assert np.array_equal(a[f], b2[f]), f
    # host-array form
    ix2 = P.Index(sp, ctx=gpu_ctx)
    ix2.add_shmmrs(mm=np.frombuffer(gathered.cpu().numpy().tobytes(), dtype=P.MM128))
    ix2.finalize()
    assert np.array_equal(ix2.download()["bgn"], b2["bgn"])


def test_index_from_large_host_batch_pipelined(oracle, gpu_ctx):
    """pgr_index_add_batch stages large host inputs (>= 512 Mbp) sub-batch by sub-batch on a second stream: same
    records as the single-batch path, for explicit and for running sequence ids"""
    import ctypes as C
    import pgrtk_amd as P
    from pgrtk_amd import _ffi
    rng = np.random.default_rng(43)
    lens = [int(x) for x in rng.integers(5_000_000, 40_000_000, 28)]
    seqs = [oracle.synth_contig(10, i, L).tobytes() for i, L in enumerate(lens)]
    assert sum(lens) > 560_000_000

    def build(sids):
        ix = P.Index(P.make_spec(80, 56, 4, 64), ctx=gpu_ctx)
        ix.add_seqs(seqs, sids=sids)
        ix.finalize()
        r = ix.download()
        ix.close()
        return r
    for sids in (None, [int(x) for x in rng.permutation(len(seqs)) + 5]):
        a = build(sids)
        with gpu_ctx.options(no_pipeline=1):
            b = build(sids)
        assert len(a) == len(b) > 1_000_000
        for f in ("h0", "h1", "frg_id", "sid", "bgn", "end", "orient"):
            assert np.array_equal(a[f], b[f]), f


def _records(gpu_ctx, sdb):
    import ctypes as C
    from pgrtk_amd import _ffi
    p, n = C.c_void_p(), C.c_uint64()
    gpu_ctx.check(_ffi.lib().pgr_index_download(gpu_ctx.handle, sdb._ix, C.byref(p), C.byref(n)))
    return _ffi.take(p, int(n.value), _ffi.FRAG_REC)


@pytest.mark.parametrize("shape", ["unique", "copies40", "copies3000", "mixed"])
def test_index_sort_by_one_key_and_run_fixups(oracle, gpu_ctx, shape):
    """pgr_index_finalize sorts append-ordered records by h0 alone and orders the runs of equal h0 by h1 afterwards: runs of up
    to 16 records by themselves, up to 4096 one workgroup each, longer ones send the index to the two-key sort.  Every class
    against the oracle's record order and against the two-key sort (context option index_two_key_sort)."""
    import pgrtk_amd as P
    rng = np.random.default_rng(61)
    if shape == "unique":
        seqs = [seqgen.rnd(rng, 300_000) for _ in range(6)]
    elif shape == "copies40":  # runs of ~80 records per h0
        base = seqgen.rnd(rng, 60_000)
        seqs = [base[int(rng.integers(0, 2000)):] for _ in range(40)] + [seqgen.rnd(rng, 50_000)]
    elif shape == "copies3000":  # runs of ~6000: the fallback
        base = seqgen.rnd(rng, 3_000)
        seqs = [base] * 3000 + [seqgen.rnd(rng, 20_000)]
    else:
        base = seqgen.rnd(rng, 40_000)
        seqs = [seqgen.rnd(rng, 100_000)] + [base] * 9 + [revcomp(base)] * 5 + [base[:20_000] * 3] + [seqgen.rnd(rng, 1000), b""]
    sdb, oix = _build_pair(oracle, gpu_ctx, seqs)
    ref = oix.records()
    got = _records(gpu_ctx, sdb)
    assert len(got) == len(ref)
    for f in ("h0", "h1", "frg_id", "sid", "bgn", "end", "orient"):
        assert np.array_equal(ref[f], got[f]), (shape, f)
    with gpu_ctx.options(index_two_key_sort=1):
        sdb2 = P.SeqIndexDB(ctx=gpu_ctx)
        sdb2.load_from_seq_list([("s%d" % i, s) for i, s in enumerate(seqs)], w=80, k=56, r=4, min_span=64)
        got2 = _records(gpu_ctx, sdb2)
    for f in ("h0", "h1", "frg_id", "sid", "bgn", "end", "orient"):
        assert np.array_equal(got2[f], got[f]), (shape, f)
