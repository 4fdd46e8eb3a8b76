"""pgr_query_hps_batch (10 000 x 10 kbp host ASCII in, chains out) against a 100 x 10 Mbp index: timing in parts / in one piece and
the library's own timeline of one call (context option debug)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import numpy as np
import bench
import pgrtk_amd as P

ctx = P.default_context(0)
spec = P.make_spec()
n_c = 100
ids = list(range(n_c))
b = P.Batch.synthetic([10_000_000] * n_c, seed=2, ctx=ctx)
ix = P.Index(spec, ctx=ctx)
ix.add_resident(b, sids=ids)
ix.finalize()


class A:
    seed = 2
    contig_len = 10_000_000


rng = np.random.default_rng(3)
cs, offs, qs = bench.make_queries(P, 2, ids, n_c, 10_000_000, 10_000, 10_000, rng)


def med(f, n=9):
    f()
    ts = sorted(f() for _ in range(n))
    return ts[n // 2] * 1e3, ts[0] * 1e3


print("in parts   : median %.3f ms, min %.3f" % med(lambda: ix.time_query_host(qs, 0.025)[0]))
with ctx.options(no_pipeline=1):
    print("one piece  : median %.3f ms, min %.3f" % med(lambda: ix.time_query_host(qs, 0.025)[0]))
qb = P.Batch.from_seqs(qs, ctx=ctx)
print("resident   : median %.3f ms, min %.3f" % med(lambda: ix.time_query_resident(qb, 0.025)[0]))
sys.stdout.flush()
with ctx.options(debug=2, debug_times=1):
    t0 = time.perf_counter()
    ix.time_query_host(qs, 0.025)
    print('traced call %.3f ms' % ((time.perf_counter() - t0) * 1e3))
