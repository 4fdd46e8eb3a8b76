#!/usr/bin/env python3
"""10^6 x 1 kbp reads through the shimmer pipeline, three times, for a kernel trace:
    rocprofv3 --kernel-trace --stats ... -- python tools/reads_trace.py [n] [L]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd")]
import pgrtk_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000
ctx = P.default_context(0)
sp = P.make_spec()
b = P.Batch.synthetic([L] * n, seed=41, ctx=ctx)
sh = b.shmmrs(sp)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    sh = b.shmmrs(sp)
    ts.append(time.perf_counter() - t0)
print("%d x %d bp: %s ms; %d shimmers" % (n, L, " ".join("%.2f" % (t * 1e3) for t in ts), sh.count))
