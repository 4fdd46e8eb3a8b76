#!/usr/bin/env python3
"""bench.py's repeat-rich shape in bench.py's situation -- a context whose workspaces a 10 Gbp step has sized -- with and without
sub-tile islands, in alternation (round 6: the shape read 134 Gbp/s in the bench line against 147 before, while
tools/repeat_like_bench.py in a fresh context read 0.427 against 0.416 ms)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pgr-tk_amd")]
import numpy as np  # noqa: E402
import bench  # noqa: E402
import pgrtk_amd as P  # noqa: E402

ctx = P.Context(0)
sp = P.make_spec()
seqs = bench.repeat_like()
b = P.Batch.from_seqs(seqs, ctx=ctx)
bp = sum(len(q) for q in seqs)


def ab(tag):
    ta, tb = [], []
    for rep in range(13):
        with ctx.options(no_sub_tile_islands=1):
            t0 = time.perf_counter()
            b.shmmrs(sp)
            if rep:
                tb.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        b.shmmrs(sp)
        if rep:
            ta.append(time.perf_counter() - t0)
    print("%s: default %.3f ms (median %.3f) = %.1f Gbp/s; islands of whole tiles %.3f ms (median %.3f)"
          % (tag, min(ta) * 1e3, sorted(ta)[6] * 1e3, bp / min(ta) / 1e9, min(tb) * 1e3, sorted(tb)[6] * 1e3), flush=True)


ab("fresh context")
big = P.Batch.synthetic([10_000_000] * 400, 3, ctx=ctx)
if big is not None:
    big.shmmrs(sp)
    del big
    ab("after a 4 Gbp step has sized the workspaces")
with ctx.options(debug_times=1):
    b.shmmrs(sp)
