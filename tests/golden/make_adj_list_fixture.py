#!/usr/bin/env python3
"""Expected MAP-graph adjacency lists of the reference's OWN golden frag_map (tests/golden/test_seqs_frag.mdb = the
reference's pgr-db/test/test_data/test_seqs_frag.mdb, made by its gen_frag_db.py from test_seqs.fa at w=80 k=56 r=4
min_span=64) -> tests/golden/test_seqs_adj_list.json.

`seq_db::frag_map_to_adj_list` (pgr-db/src/seq_db.rs:876-945) is a sort and a 2-point stencil over the frag_map: its output
is order-deterministic and independent of petgraph / FxHashMap iteration -- the one part of the MAP-graph path (SURVEY.md
section 8f rank 3) that CAN be pinned on a reference-held artefact.  (The reference's own test of it, pgr-db/src/lib.rs:326-340,
needs an AGC file and asserts nothing.)  This script is a third, deliberately plain reading of those 70 lines, working on the
bytes of the golden .mdb only -- it imports neither the product nor oracle/:

    records  = every (sid, bgn, end, (h0, h1, orient)) of the map                       rs:881-889
    sort     = lexicographic on that tuple (Rust's derived Ord on tuples / ShmmrGraphNode)  rs:893
    kept[i]  = len(frag_map[key_i]) >= min_count  or  sid_i in keeps                     rs:895-921
    for consecutive i, i+1 both kept, same sid, end_i == bgn_{i+1}:                      rs:923-944
        emit (sid, v, w) and (sid, rev(w), rev(v)),  rev((h0, h1, o)) = (h0, h1, 1 - o)

Run from the repo root: python tests/golden/make_adj_list_fixture.py"""
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(0, None), (2, None), (16, None), (16, [0, 5, 17]), (10 ** 6, [3])]


def read_mdb(path):
    """seq_db.rs:1291-1326: "mdb" | 5 x u32 | n_keys u64 | n_keys x { h0 u64 | h1 u64 | n u64 | n x (4 x u32 | u8) }"""
    d = open(path, "rb").read()
    assert d[:3] == b"mdb"
    (n_keys,) = struct.unpack_from("<Q", d, 23)
    off, fm = 31, {}
    for _ in range(n_keys):
        h0, h1, n = struct.unpack_from("<QQQ", d, off)
        off += 24
        sigs = []
        for _ in range(n):
            frg, sid, bgn, end, o = struct.unpack_from("<IIIIB", d, off)
            off += 17
            sigs.append((frg, sid, bgn, end, o))
        fm[(h0, h1)] = sigs
    assert off == len(d)
    return struct.unpack_from("<5I", d, 3), fm


def adj_list(fm, min_count, keeps):
    out = sorted((sid, bgn, end, (key[0], key[1], o)) for key, sigs in fm.items() for (_frg, sid, bgn, end, o) in sigs)
    if len(out) < 2:
        return []
    keeps = set(keeps or ())
    kept = [len(fm[(v[3][0], v[3][1])]) >= min_count or (keeps and v[0] in keeps) for v in out]
    adj = []
    for i in range(len(out) - 1):
        v, w = out[i], out[i + 1]
        if kept[i] and kept[i + 1] and v[0] == w[0] and v[2] == w[1]:
            adj.append([v[0], list(v[3]), list(w[3])])
            adj.append([v[0], [w[3][0], w[3][1], 1 - w[3][2]], [v[3][0], v[3][1], 1 - v[3][2]]])
    return adj


def main():
    spec, fm = read_mdb(os.path.join(HERE, "test_seqs_frag.mdb"))
    assert spec == (80, 56, 4, 64, 0) and len(fm) == 55 and sum(len(v) for v in fm.values()) == 820
    keys = sorted(fm)
    kid = {k: i for i, k in enumerate(keys)}
    cases = []
    for mc, keeps in CASES:
        adj = adj_list(fm, mc, keeps)
        # compact form: node (h0, h1, o) -> 2 * index of (h0, h1) in `keys` + o
        cases.append({"min_count": mc, "keeps": keeps, "n_edges": len(adj),
                      "adj_list": [[sid, 2 * kid[(v[0], v[1])] + v[2], 2 * kid[(w[0], w[1])] + w[2]] for sid, v, w in adj]})
    out = {"source": "tests/golden/test_seqs_frag.mdb (reference: pgr-db/test/test_data/test_seqs_frag.mdb)",
           "function": "pgr-db/src/seq_db.rs:876-945 frag_map_to_adj_list",
           "keys": [list(k) for k in keys], "node": "2 * index into keys + orientation  <->  (hash0, hash1, orientation)",
           "edge": "[sid, node v, node w], in the reference's output order", "cases": cases}
    with open(os.path.join(HERE, "test_seqs_adj_list.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote test_seqs_adj_list.json:", [(c["min_count"], c["keeps"], c["n_edges"]) for c in cases])


if __name__ == "__main__":
    main()
