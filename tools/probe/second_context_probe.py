"""Is the first pass of the 100 Gbp build slow in ANY context that is not the process's first one?  Three contexts one after the
other in one process (each closed before the next is created, or kept), the same 3-batch build in each, first pass and repeat."""
import os
import sys
import time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
import torch  # noqa: F401,E402
import pgrtk_amd as P  # noqa: E402

keep_alive = "--keep" in sys.argv
n_b, n_c, L = 4, 1000, 10_000_000
spec = P.make_spec()
ctxs = []
for k in range(3):
    ctx = P.Context(0)
    for what in ("first pass", "again"):
        ctx.synchronize()
        t0 = time.perf_counter()
        ix = P.Index(spec, ctx=ctx)
        ix.reserve(int(n_b * n_c * L * 0.00304 * 1.02))
        for bi in range(n_b):
            ids = list(range(bi * n_c, (bi + 1) * n_c))
            b = P.Batch.synthetic([L] * n_c, seed=2, ctx=ctx, contig_ids=ids)
            ix.add_resident(b, sids=ids)
            del b
        t1 = time.perf_counter()
        ix.finalize()
        ctx.synchronize()
        t2 = time.perf_counter()
        del ix
        print("context %d (%s), %s: batches %.3f s, sort %.3f s" % (k, "earlier ones kept" if keep_alive else "earlier ones closed", what, t1 - t0, t2 - t1), flush=True)
    if keep_alive:
        ctxs.append(ctx)
    else:
        ctx.close()
