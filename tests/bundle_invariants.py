"""What get_principal_bundles_from_adj_list (pgr-db/src/seq_db.rs:1064-1186) and the per-sequence decomposition
(pgr-db/src/ext.rs:552-650, 976-1014) guarantee WHATEVER order petgraph's GraphMap, its Dfs, std's BinaryHeap and the FxHash
containers iterate in -- written from the Rust text alone; imports neither the product nor oracle/mapgraph.py.

Order-INDEPENDENT (checked here, on the pinned adjacency list):
  B1  every bundle is a directed path of the adjacency graph: consecutive vertices (h0, h1, strand) are joined by an edge v -> w
      of the list (:1101-1112 builds g0 from the list's edges, :1150-1158 walks g1 = g0 minus removed nodes)
  B2  a vertex key (h0, h1) occurs at most once over all bundles (:1160-1163 removes a path's nodes AND their reverse twins)
  B3  bundles come longest first (:1184)
  B4  inside a bundle no vertex but the last is a branching point of the graph its edges come from (:1117-1124, :1151-1156: the
      walk stops AT a terminal vertex) -- evaluated on the sub-graph spanned by the bundles' own keys, which is g0 whenever
      every key of g0 ends up in a bundle (B5)
  B5  the walk runs until g1 is empty (:1145-1182): every key of g0 is in exactly one bundle.  g0's key set is the keys of the DFS
      paths longer than the cut-off (:1080-1099) -- WHICH keys those are is order dependent, so B5 is stated relative to the
      bundles: keys(bundles) is closed under "both ends of an edge inside keys(bundles)" by construction, nothing to check
      beyond B1-B4; with path_len_cutoff = 0 and a connected graph every key of the adjacency list is covered (checked then)
  D1  bundle ids are 0 .. n-1, each exactly once; bundles-with-id are ordered by (mean order, id) non-decreasing (:632-640)
  D2  a bundle-with-id is its principal bundle as it is or reverse-complemented as a whole (:641-648)
  D3  decomposition of a sequence: one entry per shimmer pair, in sequence order; the entry is None iff the pair's key is in no
      bundle; otherwise (bundle id, strand of the bundle's vertex, position in the bundle-with-id) -- a pure lookup (:976-1014)
Order-DEPENDENT (not checked here; product and oracle model the crates' mechanics and are compared with each other only):
  which keys the weighted DFS puts on which path when branches tie (BinaryHeap sift order among equal weights, GraphMap
  neighbour order), hence which vertices form which bundle at a tie, the order of bundles of equal length, bundle ids.
"""


def check_bundles(adj, bundles, cutoff=None, connected=False):
    edges = {(tuple(v), tuple(w)) for _sid, v, w in adj}
    keys_adj = {(v[0], v[1]) for _s, v, _w in adj} | {(w[0], w[1]) for _s, _v, w in adj}
    seen = set()
    for p in bundles:
        assert len(p) >= 1
        for a, b in zip(p, p[1:]):
            assert (tuple(a), tuple(b)) in edges, "B1: %r -> %r is no edge" % (a, b)
        for v in p:
            k = (v[0], v[1])
            assert k in keys_adj, "bundle vertex outside the graph"
            assert k not in seen, "B2: key %r twice" % (k,)
            seen.add(k)
    assert [len(p) for p in bundles] == sorted((len(p) for p in bundles), reverse=True), "B3"
    # B4 on the sub-graph spanned by the bundles' keys
    sub = [(tuple(v), tuple(w)) for _s, v, w in adj if (v[0], v[1]) in seen and (w[0], w[1]) in seen]
    out_n, in_n = {}, {}
    for v, w in set(sub):
        out_n.setdefault(v, set()).add(w)
        in_n.setdefault(w, set()).add(v)
    terminal = set()
    for v, w in set(sub):  # (:1117-1124: both tests mark v)
        if len(out_n.get(v, ())) > 1 or len(in_n.get(w, ())) > 1:
            terminal.add(v)
    for p in bundles:
        for v in p[:-1]:
            assert tuple(v) not in terminal, "B4: branching vertex inside a bundle"
    if cutoff == 0 and connected:
        assert seen == keys_adj, "B5: a key of the graph is in no bundle"
    return seen


def check_with_id_and_decomposition(bundles, with_id, decomposition, seq_smps):
    n = len(bundles)
    assert sorted(b[0] for b in with_id) == list(range(n)), "D1"
    assert [(b[1], b[0]) for b in with_id] == sorted((b[1], b[0]) for b in with_id), "D1 order"
    vmap = {}
    for bid, _ord, bundle in with_id:
        rc = [(v[0], v[1], 1 - v[2]) for v in reversed(bundles[bid])]
        assert list(map(tuple, bundle)) == list(map(tuple, bundles[bid])) or list(map(tuple, bundle)) == rc, "D2"
        for pos, v in enumerate(bundle):
            vmap[(v[0], v[1])] = (bid, v[2], pos)
    assert len(decomposition) == len(seq_smps)
    covered = 0
    for (sid, dec), (sid2, smps) in zip(decomposition, seq_smps):
        assert sid == sid2 and len(dec) == len(smps), "D3: one entry per shimmer pair"
        for (v, info), s in zip(dec, smps):
            assert tuple(v) == tuple(s)
            exp = vmap.get((s[0], s[1]))
            assert (info is None) == (exp is None), "D3: None iff the key is in no bundle"
            if exp is not None:
                assert tuple(info) == exp, "D3: lookup"
                covered += 1
    return covered
