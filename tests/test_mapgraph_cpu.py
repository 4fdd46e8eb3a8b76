"""CPU tests of the MAP-graph host logic: the oracle's container models (petgraph GraphMap / IndexMap, std BinaryHeap),
structural properties of its outputs, and the product's host-side post-processing (pgrtk_amd.cli pbundle functions)
against the oracle's restatement of pgr-pbundle-decomp (rs:61-137, 340-530).  No GPU needed."""
import numpy as np

import seqgen


def _small_pangenome(oracle, seed=1, n_hap=6, spec_t=(24, 24, 2, 8)):
    import mapgraph as og
    haps = seqgen.amy1a_like(seed=seed, n_hap=n_hap, L=30_000, unit=2500)
    sp = oracle.spec(*spec_t)
    oix = oracle.Index(sp)
    for i, s in enumerate(haps):
        oix.add_seq(i, s)
    oix.finalize()
    fm = {}
    for r in oix.records():
        fm.setdefault((int(r["h0"]), int(r["h1"])), []).append(
            (int(r["frg_id"]), int(r["sid"]), int(r["bgn"]), int(r["end"]), int(r["orient"])))
    smps = []
    for i, s in enumerate(haps):
        q = oracle.frag_recs(oracle.sequence_to_shmmrs(0, s, sp), i, query_side=True)
        smps.append((i, [(int(r["h0"]), int(r["h1"]), int(r["bgn"]), int(r["end"]), int(r["orient"])) for r in q]))
    return og, haps, fm, smps


def test_container_models():
    import mapgraph as og
    # IndexMap: insertion order, swap_remove moves the last entry into the hole
    m = og.IndexMap()
    for k in "abcde":
        assert m.insert(k, k.upper())
    assert not m.insert("b", "B2") and m.get("b") == "B2"
    assert m.swap_remove("b") == "B2" and m.keys == ["a", "e", "c", "d"] and m.pos["e"] == 1
    assert m.swap_remove("d") == "D" and m.keys == ["a", "e", "c"] and m.swap_remove("zz") is None
    # GraphMap: neighbour order = edge insertion order; self loops have no Incoming entry but count as both
    g = og.DiGraphMap()
    for a, b in [(1, 2), (1, 3), (2, 3), (3, 3), (4, 1), (1, 2)]:
        g.add_edge(a, b)
    assert g.node_list() == [1, 2, 3, 4] and g.all_edges() == [(1, 2), (1, 3), (2, 3), (3, 3), (4, 1)]
    assert g.neighbors_directed(1, og.OUT) == [2, 3] and g.neighbors_directed(1, og.IN) == [4]
    assert g.neighbors_directed(3, og.OUT) == [3] and g.neighbors_directed(3, og.IN) == [1, 2, 3]
    g.remove_node(1)
    assert g.node_list() == [4, 2, 3] and g.neighbors_directed(2, og.IN) == [] and g.neighbors_directed(4, og.OUT) == []
    assert g.all_edges() == [(2, 3), (3, 3)]  # three swap_removes: (1,2)<-(4,1), (1,3)<-(3,3), (4,1)<-(2,3)
    # BinaryHeap: pops a maximum every time; equal weights come out in the order fixed by the sift mechanics
    rng = np.random.default_rng(0)
    for _ in range(50):
        h = og.BinaryHeap()
        items = [(int(w), i) for i, w in enumerate(rng.integers(0, 5, int(rng.integers(1, 40))))]
        for it in items:
            h.push(it)
            d = h.d
            assert all(d[(i - 1) // 2][0] >= d[i][0] for i in range(1, len(d)))
        out = [h.pop() for _ in range(len(items))]
        assert [w for w, _ in out] == sorted((w for w, _ in items), reverse=True) and sorted(out) == sorted(items)
    h = og.BinaryHeap()
    for it in [(1, "a"), (1, "b"), (1, "c"), (1, "d")]:
        h.push(it)
    # hand simulation of std's sift_down_to_bottom + sift_up with all weights equal: [a,b,c,d] -> a; [c,b,d] -> c;
    # [b,d] -> b; [d] -> d   (the right child wins "<=" ties on the way down)
    assert [h.pop()[1] for _ in range(4)] == ["a", "c", "b", "d"]


def test_adj_list_and_bundle_properties(oracle):
    og, haps, fm, smps = _small_pangenome(oracle)
    adj = og.frag_map_to_adj_list(fm, 0)
    assert adj and len(adj) % 2 == 0
    # every edge comes with its reverse-complement twin; edges connect pairs adjacent on the sequence
    for i in range(0, len(adj), 2):
        sid, v, w = adj[i]
        assert adj[i + 1] == (sid, og.rev(w), og.rev(v))
        ev = [s for s in fm[(v[0], v[1])] if s[1] == sid and s[4] == v[2]]
        ew = [s for s in fm[(w[0], w[1])] if s[1] == sid and s[4] == w[2]]
        assert any(a[3] == b[2] for a in ev for b in ew)
    # min_count keeps a subset; keeps=all sequences restores everything
    sub = og.frag_map_to_adj_list(fm, 4)
    assert set(sub) <= set(adj) and len(sub) < len(adj)
    assert og.frag_map_to_adj_list(fm, 10 ** 6, keeps=list(range(len(haps)))) == adj
    # weighted DFS visits every vertex key reachable once (a vertex and its reverse share the visit)
    dfs = og.sort_adj_list_by_weighted_dfs(fm, adj, adj[0][1])
    keys = [(n[0][0], n[0][1]) for n in dfs]
    assert len(keys) == len(set(keys)) and dfs[0][0] == adj[0][1] and dfs[0][1] is None
    assert all(n[2] == len(fm[(n[0][0], n[0][1])]) for n in dfs)
    # principal bundles: longest first, every vertex key in at most one bundle position, all from the graph
    pb = og.get_principal_bundles(fm, 0, 2)
    assert pb and [len(p) for p in pb] == sorted((len(p) for p in pb), reverse=True)
    allk = [(v[0], v[1]) for p in pb for v in p]
    assert len(allk) == len(set(allk)) and set(allk) <= {(v[0], v[1]) for _, v, _ in adj} | {(w[0], w[1]) for _, _, w in adj}
    with_id, vmap = og.get_principal_bundles_with_id(fm, smps, 0, 2)
    assert sorted(b[0] for b in with_id) == list(range(len(pb)))
    assert [b[1] for b in with_id] == sorted(b[1] for b in with_id)
    for bid, _ord, bundle in with_id:  # a bundle is kept or reverse-complemented as a whole
        assert bundle == pb[bid] or bundle == [(v[0], v[1], 1 - v[2]) for v in reversed(pb[bid])]


def test_pbundle_postprocessing_vs_oracle(oracle):
    """product host code (cli.py) == oracle restatement on the same decomposition"""
    import sys, os
    from pgrtk_amd import cli
    og, haps, fm, smps = _small_pangenome(oracle, seed=3, n_hap=8)
    with_id, vmap = og.get_principal_bundles_with_id(fm, smps, 0, 4)
    dec = og.get_principal_bundle_decomposition(vmap, smps)
    names = {i: "ctg%02d" % (7 - i) for i in range(len(haps))}
    info = {i: (names[i], "x.fa", len(s)) for i, s in enumerate(haps)}
    for cutoff, dist in [(100, 1000), (300, 3000), (2500, 10000), (0, 0)]:
        for sid, sm in dec:
            assert cli.group_smps_by_principle_bundle_id(sm, cutoff, dist) == og.group_smps_by_principle_bundle_id(sm, cutoff, dist)
        assert cli.pbundle_bed_lines(names, dec, with_id, 24, cutoff, dist) == og.bed_lines(names, dec, with_id, 24, cutoff, dist)
        parts = cli.pbundle_partitions(names, dec, cutoff, dist)
        assert cli.pbundle_summary_lines(info, parts, 24) == og.ctg_summary_lines(info, dec, 24, cutoff, dist)
    assert any(l.split("\t")[2] != "0" for l in og.ctg_summary_lines(info, dec, 24, 100, 1000)[1:])  # repeats found
    assert cli._f32(100.0) == "100" and cli._f32(0.1) == "0.1" and cli._f32(1.0 / 3.0) == "0.33333334"


def test_product_graph_walks_on_cpu(oracle):
    """the library's host-only graph entry points (weighted DFS, principal bundles from an adjacency list) need no
    GPU: run them here against the oracle on the oracle's adjacency list"""
    import ctypes as C
    from pgrtk_amd import _ffi
    og, haps, fm, smps = _small_pangenome(oracle, seed=5, n_hap=7)
    L = _ffi.lib()
    for mc, cutoff in [(0, 0), (0, 3), (3, 1), (7, 2)]:
        adj = og.frag_map_to_adj_list(fm, mc)
        a = np.zeros(len(adj), dtype=_ffi.ADJ_PAIR)
        for i, (sid, v, w) in enumerate(adj):
            a[i]["sid"] = sid
            for side, n in (("v", v), ("w", w)):
                a[i][side]["h0"], a[i][side]["h1"], a[i][side]["orient"] = n
                a[i][side]["count"] = len(fm[(n[0], n[1])])
        if not adj:
            continue
        # weighted DFS
        sv = a[:1]["v"].copy()
        p, n = C.c_void_p(), C.c_uint64()
        assert L.pgr_sort_adj_list_by_weighted_dfs(None, a.ctypes.data, len(a), sv.ctypes.data, C.byref(p), C.byref(n)) == 0
        d = _ffi.take(p, int(n.value), _ffi.DFS_NODE)
        got = [((int(r["node"]["h0"]), int(r["node"]["h1"]), int(r["node"]["orient"])),
                (int(r["parent"]["h0"]), int(r["parent"]["h1"]), int(r["parent"]["orient"])) if r["has_parent"] else None,
                int(r["node"]["count"]), bool(r["is_leaf"]), int(r["rank"]), int(r["branch"]), int(r["branch_rank"])) for r in d]
        assert got == og.sort_adj_list_by_weighted_dfs(fm, adj, adj[0][1])
        # principal bundles
        b = _ffi.Bundles()
        assert L.pgr_principal_bundles_from_adj_list(None, a.ctypes.data, len(a), cutoff, C.byref(b)) == 0
        nb = int(b.n_bundles)
        verts = np.zeros(int(b.n_vertices), dtype=_ffi.VERTEX)
        if b.n_vertices:
            C.memmove(verts.ctypes.data, b.vertices, verts.nbytes)
        got_pb = [[(int(v["h0"]), int(v["h1"]), int(v["orient"])) for v in verts[int(b.b_off[i]):int(b.b_off[i + 1])]]
                  for i in range(nb)]
        L.pgr_bundles_free(C.byref(b))
        assert got_pb == og.get_principal_bundles_from_adj_list(fm, adj, cutoff)[0]
    # a start vertex outside the graph is an error code, not a crash (reference: expect("Node not found"))
    bad = np.zeros(1, dtype=_ffi.VERTEX)
    bad["h0"] = 1
    assert L.pgr_sort_adj_list_by_weighted_dfs(None, a.ctypes.data, len(a), bad.ctypes.data, C.byref(p), C.byref(n)) < 0


def _adj_fixture(golden_dir):
    """tests/golden/test_seqs_adj_list.json -> [(min_count, keeps, [(sid, (h0,h1,o), (h0,h1,o))])]"""
    import json
    import os
    fx = json.load(open(os.path.join(golden_dir, "test_seqs_adj_list.json")))
    keys = [tuple(k) for k in fx["keys"]]

    def node(i):
        return (keys[i >> 1][0], keys[i >> 1][1], i & 1)
    return [(c["min_count"], c["keeps"], [(sid, node(v), node(w)) for sid, v, w in c["adj_list"]])
            for c in fx["cases"]]


def test_adj_list_of_the_golden_frag_map(oracle, golden_dir):
    """SURVEY.md section 8f rank 3, the half that can be pinned: frag_map_to_adj_list (seq_db.rs:876-945) is a sort + a
    2-point stencil, independent of petgraph and hash-map order.  Expected lists derived from the reference's own golden
    .mdb by tests/golden/make_adj_list_fixture.py (a plain third reading, no product / oracle code); the oracle's
    restatement must reproduce them exactly, edge order included."""
    import os
    import mapgraph as og
    spec, fm = oracle.read_mdb(os.path.join(golden_dir, "test_seqs_frag.mdb"))
    cases = _adj_fixture(golden_dir)
    assert [(mc, len(adj)) for mc, _, adj in cases] == [(0, 1508), (2, 1462), (16, 1134), (16, 1148), (10 ** 6, 16)]
    for mc, keeps, adj in cases:
        assert og.frag_map_to_adj_list(fm, mc, keeps=keeps) == adj, (mc, keeps)


def test_order_independent_bundle_facts_on_the_golden_frag_map(oracle, golden_dir):
    """what seq_db.rs:1064-1186 / ext.rs:552-650, 976-1014 guarantee whatever petgraph / BinaryHeap / FxHash iterate like
    (tests/bundle_invariants.py: B1-B5, D1-D3), checked on the reference's own golden frag_map and its pinned adjacency lists
    (tests/golden/test_seqs_adj_list.json) -- the order-DEPENDENT part (which vertices share a bundle at a tie, bundle order
    among equal lengths, ids) stays unpinned: product == oracle only"""
    import os
    import bundle_invariants as bi
    og = __import__("mapgraph")
    spec, fm = oracle.read_mdb(os.path.join(golden_dir, "test_seqs_frag.mdb"))
    seqs = oracle.read_fasta(os.path.join(golden_dir, "test_seqs.fa"))
    sp = oracle.spec(*spec[:4])
    smps = []
    for i, (_name, s) in enumerate(seqs):
        q = oracle.frag_recs(oracle.sequence_to_shmmrs(0, s, sp), i, query_side=True)
        smps.append((i, [(int(r["h0"]), int(r["h1"]), int(r["bgn"]), int(r["end"]), int(r["orient"])) for r in q]))
    n_checked = 0
    for mc, keeps, adj in _adj_fixture(golden_dir):  # (the pinned half: derived from the reference's golden .mdb)
        assert og.frag_map_to_adj_list(fm, mc, keeps=keeps) == adj
        if not adj:
            continue
        for cutoff in (0, 1, 3):
            pb, filtered = og.get_principal_bundles_from_adj_list(fm, adj, cutoff)
            keys = bi.check_bundles(adj, pb, cutoff)
            # the filtered list is the list restricted to a key set that contains every bundle key (:1101-1112)
            fk = {(v[0], v[1]) for _s, v, _w in filtered} | {(w[0], w[1]) for _s, _v, w in filtered}
            assert filtered == [e for e in adj if (e[1][0], e[1][1]) in fk and (e[2][0], e[2][1]) in fk]
            assert {k for k in keys if k in fk} == fk  # every key with an edge inside g0 ends up in a bundle (B5)
            if keeps is None:
                with_id, vmap = og.get_principal_bundles_with_id(fm, smps, mc, cutoff)
                dec = og.get_principal_bundle_decomposition(vmap, smps)
                n_checked += bi.check_with_id_and_decomposition(pb, with_id, dec, smps)
    assert n_checked > 500
