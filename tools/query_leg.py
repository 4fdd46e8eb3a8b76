#!/usr/bin/env python3
"""The query leg of bench.py alone (BASELINE.json configs[2]) for profiling:
    rocprofv3 --kernel-trace --stats ... -- python tools/query_leg.py [--reps 5] [--contigs 1000]
builds the index of the synthetic contigs, waits 0.5 s (a visible gap in the kernel trace: tools/summarize_query_profile.py
keeps what comes after it), then runs `reps` resident query batches (pgr_query_hps_resident)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--contigs", type=int, default=1000)
    ap.add_argument("--contig-len", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--chained", action="store_true", help="the chained form (context option no_query_level1): shimmer pipeline, then the per-query kernel")
    a = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import bench
    import pgrtk_amd as P
    ctx = P.Context(0)
    if a.chained:
        ctx.set_option("no_query_level1", 1)
    spec = P.make_spec(80, 56, 4, 64)
    ids = list(range(a.contigs))
    batch = P.Batch.synthetic([a.contig_len] * a.contigs, seed=a.seed, ctx=ctx)
    ix = P.Index(spec, ctx=ctx)
    ix.add_resident(batch, sids=ids)
    ix.finalize()
    rng = np.random.default_rng(3)
    cs, offs, qs = bench.make_queries(P, a.seed, ids, a.contigs, a.contig_len, a.queries, 10_000, rng)
    qb = P.Batch.from_seqs(qs, ctx=ctx)
    ix.query_hps_resident_raw(qb, 0.025)  # warm-up
    ctx.synchronize()
    time.sleep(0.5)
    ts = []
    for _ in range(a.reps):
        dt, n_hps = ix.time_query_resident(qb, 0.025)
        ts.append(dt)
    print("query batches (C entry point): %s ms; %d hit pairs" % (" ".join("%.3f" % (t * 1e3) for t in ts), n_hps))
    p = ctx.last_query_prof()
    print("path %d" % p["path"])
    print("stages of the last batch: " + ", ".join("%s %.3f" % (k, p[k]) for k in ("shmmr_ms", "lookup_ms", "chain_ms", "result_ms", "total_ms")))
    t0 = time.perf_counter()
    r = ix.query_hps_resident_raw(qb, 0.025)
    print("through the Python binding (numpy views of the result block): %.3f ms" % ((time.perf_counter() - t0) * 1e3))


if __name__ == "__main__":
    main()
