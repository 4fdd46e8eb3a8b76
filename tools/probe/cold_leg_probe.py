"""Which leg of bench.py makes the target leg's in-process first pass slow?  target_100gbp (no content check, no child process) after
nothing / after the headline steps / after pcie_inclusive / after shapes, each in ONE process in that order."""
import os
import sys
import time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import torch  # noqa: E402
import bench  # noqa: E402
import pgrtk_amd as P  # noqa: E402


class A:
    seed = 2


n_b, n_c, L = 10, 1000, 10_000_000
spec = P.make_spec()
ctx = P.Context(0)


def leg(tag):
    fresh = P.Context(0)
    for what in ("first pass", "again"):
        fresh.synchronize()
        t0 = time.perf_counter()
        ix = P.Index(spec, ctx=fresh)
        ix.reserve(int(n_b * n_c * L * 0.003036 * 1.01) + 4096)
        pipe = P.Pipe(spec, ctx=fresh)
        for bi in range(n_b):
            ids = list(range(bi * n_c, (bi + 1) * n_c))
            b = P.Batch.synthetic([L] * n_c, seed=2, ctx=fresh, contig_ids=ids)
            if pipe.in_flight == 2:
                pipe.collect(want_shmmrs=False)
            pipe.submit(b, sids=ids, index=ix)
            del b
        while pipe.in_flight:
            pipe.collect(want_shmmrs=False)
        t1 = time.perf_counter()
        ix.finalize()
        fresh.synchronize()
        t2 = time.perf_counter()
        pipe.close()
        del ix
        print("%-34s %s: batches %.3f s, sort %.3f s" % (tag, what, t1 - t0, t2 - t1), flush=True)
    fresh.close()


leg("after nothing")
batch = P.Batch.synthetic([L] * n_c, seed=2, ctx=ctx)
buf = torch.empty((32_000_000, 5), dtype=torch.int64, device="cuda:0")
for _ in range(5):
    sh, n = batch.shmmrs_and_recs(spec, buf.data_ptr(), buf.shape[0])
    del sh
leg("after 5 headline steps")
if "--skip-pcie" not in sys.argv:
    bench.pcie_bench(P, ctx, spec, A)
    leg("after pcie_inclusive")
bench.shapes_bench(P, ctx, spec, (80, 56, 4, 64), 16, False)
leg("after shapes")
bench.latency_bench(P, ctx, spec, A)
leg("after latency")
