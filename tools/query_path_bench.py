import os
"""breakdown of the query leg (configs[2] shape at reduced index size): where the time of
pgr_query_hps_batch goes.  Run under rocprofv3 --kernel-trace --stats for the kernel view."""
import sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "pgr-tk_amd")); sys.path.insert(0, _R)
import numpy as np, pgrtk_amd as P
from bench import synth_substrings
ctx = P.default_context(0)
n, L, nq, ql = 1000, 10_000_000, 10_000, 10_000
sp = P.make_spec()
batch = P.Batch.synthetic([L] * n, seed=2, ctx=ctx)
ix = P.Index(sp, ctx=ctx); ix.add_resident(batch); ix.finalize()
rng = np.random.default_rng(3)
cs = rng.integers(0, n, nq); offs = rng.integers(0, L - ql, nq)
qs = synth_substrings(2, cs, offs, ql)
ix.query_hps_raw(qs[:100], 0.025)
for rep in range(3):
    t0 = time.perf_counter(); r = ix.query_hps_raw(qs, 0.025); dt = time.perf_counter() - t0
    print("query batch: %d x %d bp in %.1f ms (%.0f q/s, %d hit pairs)" % (nq, ql, dt * 1e3, nq / dt, len(r["hps"])))
# the C entry point alone (no numpy copies of the result) and the library's own account of the last call
for rep in range(4):
    dt, n_hps = ix.time_query_host(qs, 0.025)
    p = ctx.last_query_prof()
    print("pgr_query_hps_batch: %.3f ms; staging (host pack + enqueue) %.3f, shimmers %.3f, rest %.3f, path %d" % (
        dt * 1e3, p["stage_ms"], p["shmmr_ms"], p["chain_ms"] + p["lookup_ms"] + p["result_ms"], p["path"]))
