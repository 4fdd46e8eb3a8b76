"""Staging (host ASCII -> packed planes in HBM) of 100 Mbp as 10 000 x 10 kbp queries against 1 x 100 Mbp, alternating in one
process (the boxes' host side is noisy: medians of 25)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import bench
import pgrtk_amd as P

ctx = P.default_context(0)
big = bench.synth_contig_ascii(2, 0, 100_000_000)
qs = P.PackedSeqs.from_list([big[i * 10000:(i + 1) * 10000] for i in range(10000)])
mid = P.PackedSeqs.from_list([big[i * 1000000:(i + 1) * 1000000] for i in range(100)])
one = P.PackedSeqs.from_list([big])
res = {"10000 x 10 kbp": [], "100 x 1 Mbp": [], "1 x 100 Mbp": []}
for rep in range(26):
    for name, s in (("10000 x 10 kbp", qs), ("100 x 1 Mbp", mid), ("1 x 100 Mbp", one)):
        ctx.synchronize()
        t0 = time.perf_counter()
        b = P.Batch.from_seqs(s, ctx=ctx)
        t1 = time.perf_counter()
        ctx.synchronize()
        t2 = time.perf_counter()
        del b
        if rep:
            res[name].append((t1 - t0, t2 - t0))
for name, v in res.items():
    a = sorted(x[0] for x in v)
    c = sorted(x[1] for x in v)
    print("%-16s host returns after %.3f ms (min %.3f), device done after %.3f ms" % (name, a[len(a) // 2] * 1e3, a[0] * 1e3, c[len(c) // 2] * 1e3))
