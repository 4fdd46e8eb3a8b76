#!/usr/bin/env python3
"""BASELINE.json configs[3] timing: 96 AMY1A-like haplotypes -> SeqIndexDB (48,56,4,12) -> MAP-graph adjacency
list -> principal bundles -> bundle decomposition, stage by stage (GPU box).  Not a bench.py line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "tests")]
import seqgen  # noqa: E402

import pgrtk_amd as P  # noqa: E402


def main():
    n_hap = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
    haps = seqgen.amy1a_like(seed=4, n_hap=n_hap, L=L)
    out = {"haplotypes": n_hap, "bases": sum(map(len, haps))}
    sdb = P.SeqIndexDB()
    for rep in range(2):  # second pass = steady state (allocator warm)
        t = time.perf_counter()
        sdb.load_from_seq_list([("h%d" % i, s) for i, s in enumerate(haps)], w=48, k=56, r=4, min_span=12)
        out["index_build_s"] = time.perf_counter() - t
        t = time.perf_counter()
        adj = P.mapgraph.adj_list_records(sdb.ctx, sdb._ix, 0)
        out["adj_list_s"] = time.perf_counter() - t
        out["adj_pairs"] = len(adj)
        t = time.perf_counter()
        pb = P.mapgraph.principal_bundles_from_adj(sdb.ctx, adj, 8)
        out["principal_bundles_host_s"] = time.perf_counter() - t
        out["bundles"] = len(pb)
        t = time.perf_counter()
        b, dec = P.mapgraph.bundle_decomposition(sdb.ctx, sdb._ix, 0, 8)
        out["decomposition_total_s"] = time.perf_counter() - t
    out["records"] = int(P._ffi.lib().pgr_index_n_records(sdb._ix))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
