#!/usr/bin/env python3
"""A/B of the two forms of the per-query kernel on BASELINE.json configs[2] (10 000 x 10 kbp queries against the 10 Gbp index):
the level-1 form (tile kernel + per-query kernel on the tile segments, no list stage of the batch: pgr_query_prof.path 3) against
the chained form (shimmer pipeline, then the per-query kernel: path 2), same context, same index, alternating; every batch's flat
result compared array for array.     python tools/query_form_ab.py [--reps 7] [--contigs 1000] [--queries 10000]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--contigs", type=int, default=1000)
    ap.add_argument("--contig-len", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--qlen", type=int, default=10_000)
    ap.add_argument("--seed", type=int, default=2)
    a = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import bench
    import pgrtk_amd as P
    ctx = P.Context(0)
    spec = P.make_spec(80, 56, 4, 64)
    ids = list(range(a.contigs))
    batch = P.Batch.synthetic([a.contig_len] * a.contigs, seed=a.seed, ctx=ctx)
    ix = P.Index(spec, ctx=ctx)
    ix.add_resident(batch, sids=ids)
    ix.finalize()
    rng = np.random.default_rng(3)
    cs, offs, qs = bench.make_queries(P, a.seed, ids, a.contigs, a.contig_len, a.queries, a.qlen, rng)
    qb = P.Batch.from_seqs(qs, ctx=ctx)
    res = {}
    for form, opt in (("level-1 form", 0), ("chained form", 1)):
        with ctx.options(no_query_level1=opt):
            res[form] = ix.query_hps_resident_raw(qb, 0.025)
            ix.query_hps_resident_raw(qb, 0.025)
            print("%s: path %d" % (form, ctx.last_query_prof()["path"]))
    same = all(np.array_equal(res["level-1 form"][k], res["chained form"][k]) for k in ("q_off", "t_sid", "t_off", "c_score", "c_off", "hps"))
    print("flat results identical: %s (%d hit pairs)" % (same, len(res["level-1 form"]["hps"])))
    ts = {"level-1 form": [], "chained form": []}
    for _ in range(a.reps):
        for form, opt in (("level-1 form", 0), ("chained form", 1)):
            with ctx.options(no_query_level1=opt):
                dt, _ = ix.time_query_resident(qb, 0.025)
                ts[form].append(dt)
    for form in ts:
        v = sorted(ts[form])
        print("%s: median %.3f ms, best %.3f  (%s)" % (form, v[len(v) // 2] * 1e3, v[0] * 1e3, " ".join("%.3f" % (t * 1e3) for t in ts[form])))
    with ctx.options(debug_times=1):
        ix.time_query_resident(qb, 0.025)


if __name__ == "__main__":
    main()
