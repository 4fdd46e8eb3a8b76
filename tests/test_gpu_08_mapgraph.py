"""GPU parity: MAP-graph adjacency list, weighted DFS, principal bundles and the bundle decomposition
(BASELINE.json configs[3]) vs the CPU oracle (oracle/mapgraph.py).

Reference path: frag_map_to_adj_list (pgr-db/src/seq_db.rs:876-945), sort_adj_list_by_weighted_dfs (:1006-1062),
get_principal_bundles_from_adj_list (:1064-1186), get_principal_bundles_with_id (ext.rs:552-650),
get_principal_bundle_decomposition (ext.rs:976-1014), pgr-pbundle-decomp's .bed (rs:61-137, 340-395).
The reference has no expected output for any of these: parity is oracle <-> product (unpinned).
"""
import numpy as np
import pytest

import procutil
import seqgen

pytestmark = pytest.mark.gpu


def _oracle_side(oracle, seqs, spec_t):
    import mapgraph as og
    sp = oracle.spec(*spec_t)
    oix = oracle.Index(sp)
    for i, s in enumerate(seqs):
        oix.add_seq(i, s)
    oix.finalize()
    fm = {}
    for r in oix.records():
        fm.setdefault((int(r["h0"]), int(r["h1"])), []).append(
            (int(r["frg_id"]), int(r["sid"]), int(r["bgn"]), int(r["end"]), int(r["orient"])))
    smps = []
    for i, s in enumerate(seqs):
        q = oracle.frag_recs(oracle.sequence_to_shmmrs(0, s, sp), i, query_side=True)
        smps.append((i, [(int(r["h0"]), int(r["h1"]), int(r["bgn"]), int(r["end"]), int(r["orient"])) for r in q]))
    return og, fm, smps


def _gpu_side(gpu_ctx, seqs, spec_t):
    import pgrtk_amd as P
    sdb = P.SeqIndexDB(ctx=gpu_ctx)
    sdb.load_from_seq_list([("h%03d" % i, s) for i, s in enumerate(seqs)], w=spec_t[0], k=spec_t[1], r=spec_t[2],
                           min_span=spec_t[3])
    return sdb


def _check_all(oracle, gpu_ctx, seqs, spec_t, min_count, cutoff, keeps=None, bed_args=(2500, 10000)):
    og, fm, smps = _oracle_side(oracle, seqs, spec_t)
    sdb = _gpu_side(gpu_ctx, seqs, spec_t)
    # adjacency list: same edges in the same order
    ref_adj = og.frag_map_to_adj_list(fm, min_count, keeps)
    got_adj = sdb.get_smp_adj_list(min_count, keeps)
    assert got_adj == ref_adj
    if not ref_adj:
        assert sdb.get_principal_bundles(min_count, cutoff, keeps) == []
        return 0
    # weighted DFS from the first vertex
    start = ref_adj[0][1]
    ref_dfs = og.sort_adj_list_by_weighted_dfs(fm, ref_adj, start)
    got_dfs = sdb.sort_adj_list_by_weighted_dfs(got_adj, start)
    assert got_dfs == ref_dfs
    # principal bundles
    ref_pb = og.get_principal_bundles(fm, min_count, cutoff, keeps)
    got_pb = sdb.get_principal_bundles(min_count, cutoff, keeps)
    assert got_pb == ref_pb
    # bundles with id + decomposition of every sequence
    ref_with_id, vmap = og.get_principal_bundles_with_id(fm, smps, min_count, cutoff, keeps)
    ref_dec = og.get_principal_bundle_decomposition(vmap, smps)
    got_with_id, got_dec = sdb.get_principal_bundle_decomposition(min_count, cutoff, keeps)
    assert got_with_id == ref_with_id
    assert got_dec == ref_dec
    # the .bed body pgr-pbundle-decomp writes
    from pgrtk_amd import cli
    names = {sid: v[0] for sid, v in sdb.seq_info.items()}
    ref_bed = og.bed_lines(names, ref_dec, ref_with_id, spec_t[1], *bed_args)
    got_bed = cli.pbundle_bed_lines(names, got_dec, got_with_id, spec_t[1], *bed_args)
    assert got_bed == ref_bed
    return len(ref_pb)


def test_config4_amy1a_like(oracle, gpu_ctx):
    """BASELINE.json configs[3]: 96 haplotypes x ~250 kbp, pgr-pbundle-decomp defaults (48,56,4,12), min_cov 0,
    min_branch_size 8"""
    haps = seqgen.amy1a_like(seed=4, n_hap=96, L=200_000)
    n = _check_all(oracle, gpu_ctx, haps, (48, 56, 4, 12), 0, 8)
    assert n > 10


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_small_pangenomes(oracle, gpu_ctx, seed):
    """structural variation: inversions, deletions, duplications, private insertions; min_count filter and keeps"""
    rng = np.random.default_rng(seed)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    anc = seqgen.rnd(rng, 60_000)
    haps = []
    for h in range(int(rng.integers(4, 10))):
        s = bytearray(anc)
        for _ in range(int(rng.integers(1, 5))):
            a = int(rng.integers(0, len(s) - 8000))
            ln = int(rng.integers(500, 7000))
            op = int(rng.integers(0, 4))
            seg = bytes(s[a:a + ln])
            if op == 0:
                s[a:a + ln] = seg.translate(comp)[::-1]       # inversion
            elif op == 1:
                del s[a:a + ln]                                # deletion
            elif op == 2:
                s[a:a] = seg * int(rng.integers(1, 4))         # tandem duplication
            else:
                s[a:a] = seqgen.rnd(rng, ln)                   # private insertion
        haps.append(bytes(s))
    haps.append(haps[0].translate(comp)[::-1])                 # one haplotype given as its reverse complement
    for (mc, cutoff, keeps) in [(0, 2, None), (2, 3, None), (3, 1, [0, len(haps) - 1]), (1, 0, [])]:
        _check_all(oracle, gpu_ctx, haps, (24, 24, 2, 8), mc, cutoff, keeps, bed_args=(200, 2000))


def test_projection_and_edge_cases(oracle, gpu_ctx):
    import mapgraph as og
    rng = np.random.default_rng(5)
    haps = seqgen.amy1a_like(seed=6, n_hap=8, L=40_000, unit=3000)
    spec_t = (24, 24, 2, 8)
    _, fm, _ = _oracle_side(oracle, haps, spec_t)
    sdb = _gpu_side(gpu_ctx, haps, spec_t)
    ext = [(7, haps[1][5000:30000]), (3, seqgen.rnd(rng, 5000)), (9, b""), (1, haps[2])]
    sp = oracle.spec(*spec_t)
    smps = []
    for sid, s in ext:
        q = oracle.frag_recs(oracle.sequence_to_shmmrs(0, s, sp), sid, query_side=True)
        smps.append((sid, [(int(r["h0"]), int(r["h1"]), int(r["bgn"]), int(r["end"]), int(r["orient"])) for r in q]))
    ref_with_id, vmap = og.get_principal_bundles_with_id(fm, smps, 0, 2)
    ref_dec = og.get_principal_bundle_decomposition(vmap, smps)
    got_with_id, got_dec = sdb.get_principal_bundle_projection(0, 2, ext)
    assert got_with_id == ref_with_id and got_dec == ref_dec
    # an index without any adjacent pair: no edges, no bundles
    import pgrtk_amd as P
    e = P.SeqIndexDB(ctx=gpu_ctx)
    e.load_from_seq_list([("a", seqgen.rnd(rng, 40)), ("b", b"")], w=24, k=24, r=2, min_span=8)
    assert e.get_smp_adj_list(0) == [] and e.get_principal_bundles(0, 0) == []
    b, d = e.get_principal_bundle_decomposition(0, 0)
    assert b == [] and all(bi is None for _, smp in d for _, bi in smp)
    # min_count above every multiplicity: empty
    assert sdb.get_smp_adj_list(10_000) == []
    # weighted DFS from a vertex that is not in the list is an error, not a crash
    adj = sdb.get_smp_adj_list(0)
    with pytest.raises(P.PgrError):
        sdb.sort_adj_list_by_weighted_dfs(adj, (1, 2, 0))


def test_gfa_and_idx_writers(oracle, gpu_ctx, tmp_path):
    """generate_mapg_gfa (both methods), generate_principal_mapg_gfa, write_mapg_idx: same lines as the oracle's
    restatement (the reference writes the S / L / F blocks in hash-map order, so lines are compared as sorted sets)"""
    import mapgraph as og
    haps = seqgen.amy1a_like(seed=8, n_hap=10, L=50_000, unit=4000)
    spec_t = (24, 24, 2, 8)
    _, fm, smps = _oracle_side(oracle, haps, spec_t)
    sdb = _gpu_side(gpu_ctx, haps, spec_t)
    for mc, keeps in [(0, None), (3, None), (11, [2, 5])]:
        adj = og.frag_map_to_adj_list(fm, mc, keeps)
        sdb.generate_mapg_gfa(mc, str(tmp_path / "a.gfa"), "from_fragmap", keeps)
        got = open(tmp_path / "a.gfa").read().splitlines()
        assert got == og.gfa_lines(fm, adj, spec_t[1])  # same insertion order too
        adj2 = []
        for sid, sm in smps:
            adj2 += og.smp_adj_list_for_seq(sm, sid, fm, 0 if (keeps is not None and sid in keeps) else mc)
        sdb.generate_mapg_gfa(mc, str(tmp_path / "b.gfa"), "from_seqs", keeps)
        assert open(tmp_path / "b.gfa").read().splitlines() == og.gfa_lines(fm, adj2, spec_t[1])
        if adj:
            pb, filtered = og.get_principal_bundles_from_adj_list(fm, adj, 2)
            vmap = og.vertex_map_from_bundles(pb)
            sdb.generate_principal_mapg_gfa(mc, 2, str(tmp_path / "p.gfa"), keeps)
            assert open(tmp_path / "p.gfa").read().splitlines() == og.gfa_lines(fm, filtered, spec_t[1], vmap)
    sdb.write_mapg_idx(str(tmp_path / "m.idx"))
    lines = open(tmp_path / "m.idx").read().splitlines()
    assert lines[0] == "K\t24\t24\t2\t8\tfalse"
    assert [l for l in lines if l[0] == "C"] == ["C\t%d\th%03d\tMemory\t%d" % (i, i, len(s)) for i, s in enumerate(haps)]
    # MEMORY backend: global fragment ids (seq_db.rs:189-357) -- the oracle index built with fastx ids gives them
    oix = oracle.Index(oracle.spec(*spec_t))
    for i, s in enumerate(haps):
        oix.add_seq(i, s, fastx_ids=True)
    oix.finalize()
    ref_f = sorted("F\t%016x_%016x\t%d\t%d\t%d\t%d\t%d" % (r["h0"], r["h1"], r["frg_id"], r["sid"], r["bgn"], r["end"],
                                                         r["orient"]) for r in oix.records())
    assert sorted(l for l in lines if l[0] == "F") == ref_f


def test_cli_pbundle_decomp(oracle, gpu_ctx, tmp_path):
    """pgr-pbundle-decomp counterpart end to end: FASTA -> .bed + .ctg.summary.tsv, and the -d variant that
    decomposes other sequences with the bundles of the first file"""
    import mapgraph as og
    from pgrtk_amd import cli
    haps = seqgen.amy1a_like(seed=9, n_hap=12, L=60_000, unit=5000)
    spec_t = (24, 24, 2, 8)
    fa = tmp_path / "haps.fa"
    with open(fa, "w") as f:
        for i, s in enumerate(haps):
            f.write(">hap%02d\n%s\n" % (11 - i, s.decode()))  # names sort differently from the ids
    argv = ["pbundle-decomp", str(fa), str(tmp_path / "out"), "-w", "24", "-k", "24", "-r", "2", "--min-span", "8",
            "--min-branch-size", "8", "--bundle-length-cutoff", "300", "--bundle-merge-distance", "3000"]
    cli.main(argv)
    og_, fm, smps = _oracle_side(oracle, haps, spec_t)
    with_id, vmap = og.get_principal_bundles_with_id(fm, smps, 0, 8)
    dec = og.get_principal_bundle_decomposition(vmap, smps)
    names = {i: "hap%02d" % (11 - i) for i in range(len(haps))}
    bed = open(tmp_path / "out.bed").read().splitlines()
    assert bed[0].startswith("# cmd: ")
    ref_bed = og.bed_lines(names, dec, with_id, 24, 300, 3000)
    assert bed[1:] == ref_bed and len(ref_bed) > len(haps)
    assert any(l.endswith(":R") for l in ref_bed) and any(l.endswith(":U") for l in ref_bed)
    info = {i: (names[i], str(fa), len(s)) for i, s in enumerate(haps)}
    assert open(tmp_path / "out.ctg.summary.tsv").read().splitlines() == og.ctg_summary_lines(info, dec, 24, 300, 3000)
    # -d: other sequences (one mutated haplotype, one unrelated) against the same bundles
    rng = np.random.default_rng(3)
    other = [haps[3][2000:50000], seqgen.rnd(rng, 8000)]
    fb = tmp_path / "other.fa"
    with open(fb, "w") as f:
        for i, s in enumerate(other):
            f.write(">o%d\n%s\n" % (i, s.decode()))
    cli.main(argv[:2] + [str(tmp_path / "out2")] + argv[3:] + ["-d", str(fb)])
    sp = oracle.spec(*spec_t)
    osm = []
    for i, s in enumerate(other):
        q = oracle.frag_recs(oracle.sequence_to_shmmrs(0, s, sp), i, query_side=True)
        osm.append((i, [(int(r["h0"]), int(r["h1"]), int(r["bgn"]), int(r["end"]), int(r["orient"])) for r in q]))
    odec = og.get_principal_bundle_decomposition(vmap, osm)
    assert open(tmp_path / "out2.bed").read().splitlines()[1:] == og.bed_lines({0: "o0", 1: "o1"}, odec, with_id, 24, 300, 3000)
    # .pdb cache (rs:357-383 / :155-218): the first run wrote it; a run that reads it back (different, ignored, spec on
    # the command line) reproduces both decompositions without recomputing the bundles
    from pgrtk_amd import pdb
    w_, k_, r_, ms_, mbs_, mc_, pb, vm = pdb.read_pdb(str(tmp_path / "out.pdb"))
    assert (w_, k_, r_, ms_, mbs_, mc_) == (24, 24, 2, 8, 8, 0)
    assert [(b[0], b[1], [tuple(v) for v in b[2]]) for b in pb] == [(b[0], b[1], [tuple(v) for v in b[2]]) for b in with_id]
    assert vm == {k2: tuple(v) for k2, v in vmap.items()}
    cli.main(["pbundle-decomp", str(fa), str(tmp_path / "out3"), "-w", "80", "-r", "4", "--precomputed-bundles",
              str(tmp_path / "out.pdb"), "--bundle-length-cutoff", "300", "--bundle-merge-distance", "3000"])
    assert open(tmp_path / "out3.bed").read().splitlines()[1:] == bed[1:]
    assert open(tmp_path / "out3.ctg.summary.tsv").read() == open(tmp_path / "out.ctg.summary.tsv").read()
    cli.main(["pbundle-decomp", str(fa), str(tmp_path / "out4"), "--precomputed-bundles", str(tmp_path / "out.pdb"), "-d", str(fb),
              "--bundle-length-cutoff", "300", "--bundle-merge-distance", "3000"])
    assert open(tmp_path / "out4.bed").read().splitlines()[1:] == open(tmp_path / "out2.bed").read().splitlines()[1:]
    # the C++ host program above the C ABI writes the same files
    import os
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pgr-tk_amd", "bin", "pgr-pbundle-decomp")
    assert os.path.exists(exe), "host programs not built: python __graft_entry__.py build"
    procutil.run_bounded([exe, str(fa), str(tmp_path / "cc")] + argv[3:], check=True, timeout=120)
    assert open(tmp_path / "cc.bed").read().splitlines()[1:] == bed[1:]
    assert open(tmp_path / "cc.ctg.summary.tsv").read() == open(tmp_path / "out.ctg.summary.tsv").read()
    procutil.run_bounded([exe, str(fa), str(tmp_path / "cc2")] + argv[3:] + ["-d", str(fb)], check=True, timeout=120)
    assert open(tmp_path / "cc2.bed").read().splitlines()[1:] == open(tmp_path / "out2.bed").read().splitlines()[1:]
    assert open(tmp_path / "cc2.ctg.summary.tsv").read() == open(tmp_path / "out2.ctg.summary.tsv").read()
    inc = tmp_path / "inc.txt"
    inc.write_text("hap03\nhap07\n")
    cli.main(argv[:2] + [str(tmp_path / "out3")] + argv[3:] + ["-i", str(inc)])
    procutil.run_bounded([exe, str(fa), str(tmp_path / "cc3")] + argv[3:] + ["-i", str(inc)], check=True, timeout=120)
    assert open(tmp_path / "cc3.bed").read().splitlines()[1:] == open(tmp_path / "out3.bed").read().splitlines()[1:]
    assert len(open(tmp_path / "cc3.ctg.summary.tsv").read().splitlines()) == 3
    assert open(tmp_path / "cc3.ctg.summary.tsv").read() == open(tmp_path / "out3.ctg.summary.tsv").read()


def test_golden_fixture_graph(oracle, gpu_ctx, golden_dir):
    """the reference's own fixture (test_seqs.fa -> test_seqs_frag.mdb): the MAP-graph of the golden frag_map,
    loaded from the .mdb/.midx pair, vs the oracle fed with the same file"""
    import os
    import mapgraph as og
    import pgrtk_amd as P
    spec, fm = oracle.read_mdb(os.path.join(golden_dir, "test_seqs_frag.mdb"))
    sdb = P.SeqIndexDB(ctx=gpu_ctx)
    sdb.load_from_mdb_index(os.path.join(golden_dir, "test_seqs_frag"))
    seqs = oracle.read_fasta(os.path.join(golden_dir, "test_seqs.fa"))
    sp = oracle.spec(*spec[:4])
    smps = []
    for i, (_name, s) in enumerate(seqs):
        q = oracle.frag_recs(oracle.sequence_to_shmmrs(0, s, sp), i, query_side=True)
        smps.append((i, [(int(r["h0"]), int(r["h1"]), int(r["bgn"]), int(r["end"]), int(r["orient"])) for r in q]))
    for mc, cutoff in [(0, 0), (2, 1), (4, 2)]:
        adj = og.frag_map_to_adj_list(fm, mc)
        assert sdb.get_smp_adj_list(mc) == adj and len(adj) > 100
        assert sdb.sort_adj_list_by_weighted_dfs(adj, adj[0][1]) == og.sort_adj_list_by_weighted_dfs(fm, adj, adj[0][1])
        assert sdb.get_principal_bundles(mc, cutoff) == og.get_principal_bundles(fm, mc, cutoff)
        with_id, vmap = og.get_principal_bundles_with_id(fm, smps, mc, cutoff)
        got_with_id, got_dec = sdb.get_principal_bundle_decomposition(mc, cutoff)
        assert got_with_id == with_id
        assert got_dec == og.get_principal_bundle_decomposition(vmap, smps)


def test_bundle_bed_for_query_helper(oracle, gpu_ctx):
    """pgrtk.get_principle_bundle_bed_file_for_query (pgrtk/__init__.py:470-508) composed from the oracle's pieces"""
    import mapgraph as og
    import pgrtk_amd as P
    haps = seqgen.amy1a_like(seed=12, n_hap=7, L=40_000, unit=3000)
    spec_t = (24, 24, 2, 8)
    names = ["src::ctg%d_%d_%d_%d" % (6 - i, 1000 * i, 1000 * i + len(h), i % 2) for i, h in enumerate(haps)]
    got = P.get_principle_bundle_bed_file_for_query(list(zip(names, haps)), 24, 24, 2, 8, 2, 3, ctx=gpu_ctx)
    _, fm, smps = _oracle_side(oracle, haps, spec_t)
    with_id, vmap = og.get_principal_bundles_with_id(fm, smps, 2, 3)
    dec = dict(og.get_principal_bundle_decomposition(vmap, smps))
    ref = []
    for sid in sorted(range(len(haps)), key=lambda i: names[i]):
        for p in reversed(og.group_smps_by_principle_bundle_id(dec[sid], 50, 100000)):
            ref.append((names[sid], 1000 * sid + p[0][0][2], 1000 * sid + p[-1][0][3] + 24, "%d:%d:%d:%d" % (p[0][1], p[0][2], p[0][3], p[-1][3])))
    assert got == ref and len(ref) >= len(haps)


def test_adj_list_of_the_golden_frag_map_through_the_gpu(oracle, gpu_ctx, golden_dir):
    """the pinned half of SURVEY.md section 8f rank 3: `pgr_index_adj_list` (GPU: multiplicities, 6-pass radix sort, 2-point
    stencil) against the adjacency lists derived from the reference's own golden .mdb (tests/golden/test_seqs_adj_list.json,
    made by make_adj_list_fixture.py from the file's bytes alone) -- edge for edge, in the reference's order; both on the
    index loaded from the golden .mdb and on the index the product builds itself from test_seqs.fa."""
    import os
    import pgrtk_amd as P
    from test_mapgraph_cpu import _adj_fixture
    cases = _adj_fixture(golden_dir)
    from_file = P.SeqIndexDB(ctx=gpu_ctx)
    from_file.load_from_mdb_index(os.path.join(golden_dir, "test_seqs_frag"))
    built = P.SeqIndexDB(ctx=gpu_ctx)
    built.load_from_fastx(os.path.join(golden_dir, "test_seqs.fa"))
    for mc, keeps, adj in cases:
        assert from_file.get_smp_adj_list(mc, keeps) == adj, (mc, keeps)
        assert built.get_smp_adj_list(mc, keeps) == adj, (mc, keeps)
