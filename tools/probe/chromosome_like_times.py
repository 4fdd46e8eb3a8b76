"""host-side timeline (context options debug_times + debug) of one pgr_shmmrs_compute over bench.py's chromosome-like contig"""
import os
import sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import time  # noqa: E402
import torch  # noqa: F401,E402
import bench  # noqa: E402
import pgrtk_amd as P  # noqa: E402

ctx = P.Context(0)
sp = P.make_spec()
s = bench.chromosome_like()
b = P.Batch.from_seqs([s], ctx=ctx)
for _ in range(3):
    b.shmmrs(sp)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    sh = b.shmmrs(sp)
    ts.append(time.perf_counter() - t0)
print("GPU: %s ms; level1_ms %.3f" % (" ".join("%.3f" % (t * 1e3) for t in ts), ctx.last_prof().level1_ms))
with ctx.options(debug_times=1, debug=1):
    b.shmmrs(sp)
