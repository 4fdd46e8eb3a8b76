#!/usr/bin/env python3
"""Query batches in flight side by side: N host threads, each with a context of its own (streams, workspaces, pinned blocks) and its
own resident query batch, run pgr_query_hps_resident against ONE finalized index (the index's per-batch hints are atomics;
the reference loops over its queries with rayon, pgr-query.rs:135-165).  A single batch is a chain of a VALU-bound tile kernel, a
dozen latency-bound kernels and 7.5 MB over PCIe: what one batch leaves idle another can use.

    python tools/query_concurrency_probe.py [--threads 1 2 3 4] [--reps 20] [--json out.json]
prints, per thread count, the wall time per batch (all threads' batches / wall) and queries per second; every batch's hit-pair
count is compared with the single-threaded one."""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--contigs", type=int, default=1000)
    ap.add_argument("--contig-len", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--json", default=None)
    ap.add_argument("--plain", action="store_true", help="the other contexts by pgr_ctx_create instead of pgr_ctx_create_beside")
    a = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import bench
    import pgrtk_amd as P
    from pgrtk_amd import _ffi
    L = _ffi.lib()
    ctx0 = P.Context(0)
    spec = P.make_spec(80, 56, 4, 64)
    ids = list(range(a.contigs))
    batch = P.Batch.synthetic([a.contig_len] * a.contigs, seed=a.seed, ctx=ctx0)
    ix = P.Index(spec, ctx=ctx0)
    ix.add_resident(batch, sids=ids)
    ix.finalize()
    del batch
    rng = np.random.default_rng(3)
    cs, offs, qs = bench.make_queries(P, a.seed, ids, a.contigs, a.contig_len, a.queries, 10_000, rng)
    nmax = max(a.threads)
    ctxs = [ctx0] + [P.Context(0) if a.plain else P.Context(beside=ctx0) for _ in range(nmax - 1)]
    qbs = [P.Batch.from_seqs(qs, ctx=c) for c in ctxs]
    args = (0.025, 128, 128, 128, 8, 0, 0, 0)

    def one(t):
        res = _ffi.HpsResult()
        rc = L.pgr_query_hps_resident(ctxs[t].handle, ix._h, qbs[t]._h, C.c_float(args[0]), *args[1:], C.byref(res))
        n = int(res.n_hps) if rc == 0 else -1
        if rc == 0:
            L.pgr_hps_result_free(C.byref(res))
        return n
    ref = one(0)
    for t in range(nmax):
        for _ in range(3):
            assert one(t) == ref
    out = {"queries_per_batch": a.queries, "hit_pairs_per_batch": ref, "reps_per_thread": a.reps, "by_threads": {}}
    for n in a.threads:
        counts = [[] for _ in range(n)]
        start = threading.Barrier(n + 1)

        def work(t):
            start.wait()
            for _ in range(a.reps):
                counts[t].append(one(t))
        th = [threading.Thread(target=work, args=(t,)) for t in range(n)]
        for x in th:
            x.start()
        start.wait()
        t0 = time.perf_counter()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        ok = all(c == ref for cc in counts for c in cc)
        per = dt / (n * a.reps)
        print("%d thread(s): %d batches in %.2f ms = %.3f ms per batch, %.1f M queries/s; every batch's hit pairs as single-threaded: %s"
              % (n, n * a.reps, dt * 1e3, per * 1e3, a.queries / per / 1e6, ok), flush=True)
        out["by_threads"][str(n)] = {"ms_per_batch": per * 1e3, "queries_per_s": a.queries / per, "same_hit_pairs": bool(ok)}
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
