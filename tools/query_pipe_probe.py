#!/usr/bin/env python3
"""Query batches through the pipe (pgr_pipe_submit_query / pgr_pipe_collect_query): ms per batch at 1 and 2 jobs in flight against
the synchronous call, on BASELINE.json configs[2] (10 000 x 10 kbp resident queries against the 1000 x 10 Mbp index).
    python tools/query_pipe_probe.py [--batches 32] [--depths 1 2 3] [--debug]"""
import argparse
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=32)
    ap.add_argument("--depths", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--contigs", type=int, default=1000)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--debug", action="store_true")
    ap.add_argument("--host", action="store_true", help="also: host ASCII in (pgr_batch_from_ascii per batch) through the pipe")
    a = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import bench
    import pgrtk_amd as P
    ctx = P.Context(0)
    spec = P.make_spec(80, 56, 4, 64)
    ids = list(range(a.contigs))
    batch = P.Batch.synthetic([10_000_000] * a.contigs, seed=2, ctx=ctx)
    ix = P.Index(spec, ctx=ctx)
    ix.add_resident(batch, sids=ids)
    ix.finalize()
    del batch
    rng = np.random.default_rng(3)
    cs, offs, qs = bench.make_queries(P, 2, ids, a.contigs, 10_000_000, a.queries, 10_000, rng)
    qbs = [P.Batch.from_seqs(qs, ctx=ctx) for _ in range(3)]
    for _ in range(3):
        ix.time_query_resident(qbs[0], 0.025)
    ts = [ix.time_query_resident(qbs[0], 0.025)[0] for _ in range(5)]
    print("synchronous call: %.3f ms per batch" % (sorted(ts)[2] * 1e3))
    pipe = P.Pipe(spec, ctx=ctx)
    for depth in a.depths:
        calls = []

        def run(k):
            outs = []
            for i in range(k):
                if pipe.in_flight == depth:
                    t1 = time.perf_counter()
                    outs.append(pipe.collect_query(raw=False))
                    calls.append(("collect", i, time.perf_counter() - t1))
                t1 = time.perf_counter()
                pipe.submit_query(qbs[i % 3], ix, 0.025)
                calls.append(("submit", i, time.perf_counter() - t1))
            while pipe.in_flight:
                t1 = time.perf_counter()
                outs.append(pipe.collect_query(raw=False))
                calls.append(("collect", k, time.perf_counter() - t1))
            return outs
        run(6)
        ctx.synchronize()
        if a.debug:
            ctx.set_option("debug_times", 1)
        t0 = time.perf_counter()
        outs = run(a.batches)
        dt = time.perf_counter() - t0
        if a.debug:
            ctx.set_option("debug_times", 0)
        slow = [(w, i, "%.2f ms" % (t * 1e3)) for w, i, t in calls[-(2 * a.batches):] if t > 1.5e-3]
        if slow:
            print("   calls above 1.5 ms: %s" % slow[:12])
        print("%d in flight: %.3f ms per batch (%d batches), %.1f M queries/s; same counts every batch: %s"
              % (depth, dt / a.batches * 1e3, a.batches, a.queries * a.batches / dt / 1e6, len(set(outs)) == 1), flush=True)
    if a.host:
        # host ASCII in: pgr_batch_from_ascii of batch i + 1 while batch i is in flight (what host/pgr_query.cpp does)
        import ctypes as C
        from pgrtk_amd import _ffi
        arrs, ptrs, lens_c, n_s = _ffi.seq_ptrs(qs)
        L = _ffi.lib()

        def stage():
            h = C.c_void_p()
            ctx.check(L.pgr_batch_from_ascii(ctx.handle, n_s, ptrs, lens_c, C.byref(h)))
            return P.Batch(ctx, h, n_s)

        def run_host(k):
            held = []
            for i in range(k):
                b_ = stage()
                if pipe.in_flight == 2:
                    pipe.collect_query(raw=False)
                    held.pop(0)
                pipe.submit_query(b_, ix, 0.025)
                held.append(b_)
            while pipe.in_flight:
                pipe.collect_query(raw=False)
                held.pop(0)
        ts = [ix.time_query_host(qs, 0.025)[0] for _ in range(5)]
        print("host ASCII, synchronous call (pgr_query_hps_batch): %.3f ms per batch" % (sorted(ts)[2] * 1e3))
        run_host(3)
        reps = []
        for _ in range(3):
            ctx.synchronize()
            t0 = time.perf_counter()
            run_host(12)
            reps.append((time.perf_counter() - t0) / 12 * 1e3)
        print("host ASCII through the pipe: %s ms per batch" % " ".join("%.3f" % t for t in reps), flush=True)
    pipe.close()


if __name__ == "__main__":
    main()
