import os, sys, time
ROOT="/root/repo"
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import numpy as np, bench, pgrtk_amd as P
ctx = P.default_context(0)
big = bench.synth_contig_ascii(2, 0, 100_000_000)
qs = P.PackedSeqs.from_list([big[i*10000:(i+1)*10000] for i in range(10000)])
one = P.PackedSeqs.from_list([big])
def best(f, n=7):
    f(); ts=[]
    for _ in range(n):
        ctx.synchronize(); t0=time.perf_counter(); r=f(); t1=time.perf_counter(); ctx.synchronize(); t2=time.perf_counter(); ts.append((t1-t0,t2-t0)); del r
    ts.sort(); return ts[len(ts)//2]
for name, s in (("10000 x 10 kbp", qs), ("1 x 100 Mbp", one)):
    a,b = best(lambda: P.Batch.from_seqs(s, ctx=ctx))
    print("%-16s stage: host returns after %.3f ms, device done after %.3f ms" % (name, a*1e3, b*1e3))
ctx.set_option("debug", 1)
P.Batch.from_seqs(qs, ctx=ctx); ctx.synchronize()
