import os, sys, time, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import numpy as np, bench, pgrtk_amd as P
from pgrtk_amd import _ffi
ctx = P.default_context(0)
n, L = 104, 10_000_000
seqs = [bench.synth_contig_ascii(2, i, L) for i in range(n)]
sp = P.make_spec()
for _ in range(3): P.time_shmmr_batch(seqs, sp, ctx=ctx)
ts = sorted(P.time_shmmr_batch(seqs, sp, ctx=ctx)[0] for _ in range(5))
print("ascii 1.04 Gbp: median %.2f ms best %.2f" % (ts[2] * 1e3, ts[0] * 1e3))
with ctx.options(debug=2):
    P.time_shmmr_batch(seqs, sp, ctx=ctx)
