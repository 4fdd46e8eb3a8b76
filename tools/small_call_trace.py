#!/usr/bin/env python3
"""the two small calls of bench.py's `latency` block alone, for a kernel trace:
    rocprofv3 --kernel-trace ... -- python tools/small_call_trace.py
one 10 kbp contig through pgr_shmmr_batch (x20), a 0.3 s pause, one 10 kbp query through pgr_query_hps_batch (x20)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
import bench  # noqa: E402
import pgrtk_amd as P  # noqa: E402

ctx = P.Context(0)
spec = P.make_spec(80, 56, 4, 64)
one = P.PackedSeqs.from_list([bench.synth_contig_ascii(2, 0, 10_000)])
b = P.Batch.synthetic([1_000_000] * 8, seed=2, ctx=ctx)
ix = P.Index(spec, ctx=ctx)
ix.add_resident(b)
ix.finalize()
q = P.PackedSeqs.from_list([bench.synth_contig_ascii(2, 3, 200_000)[50_000:60_000]])
for _ in range(5):
    P.time_shmmr_batch(one, spec, ctx=ctx)
    ix.time_query_host(q, 0.025)
ctx.synchronize()
time.sleep(0.3)
ts = sorted(P.time_shmmr_batch(one, spec, ctx=ctx)[0] for _ in range(20))
print("shmmr call: median %.1f us" % (ts[10] * 1e6))
time.sleep(0.3)
ts = sorted(ix.time_query_host(q, 0.025)[0] for _ in range(20))
print("query call: median %.1f us" % (ts[10] * 1e6))
