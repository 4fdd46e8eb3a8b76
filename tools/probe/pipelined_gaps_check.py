"""three chromosome-like contigs (744 Mbp, gaps, isolated N, lower case) through the PIPELINED host entry point (sub-batches staged
on one thread while the previous ones compute; each sub-batch lists its islands beside its tile kernel) against the resident
path of the same contigs, which bench.py compares with the CPU restatement."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import numpy as np  # noqa: E402
import bench  # noqa: E402
import pgrtk_amd as P  # noqa: E402

ctx = P.default_context(0)
spec = P.make_spec()
s = bench.chromosome_like()
seqs = [s, s[::-1].copy(), s[1000:].copy()]
t0 = time.perf_counter()
got = P.sequence_to_shmmrs_batch(seqs, spec, ctx=ctx)
t1 = time.perf_counter()
print("host entry point, %d Mbp: %.1f ms" % (sum(len(q) for q in seqs) // 10 ** 6, (t1 - t0) * 1e3))
ok = True
for i, q in enumerate(seqs):
    b = P.Batch.from_seqs([q], ctx=ctx)
    sh = b.shmmrs(spec)
    ref = sh.download()[0]
    same = len(ref) == len(got[i]) and np.array_equal(ref["x"], got[i]["x"]) and np.array_equal(ref["y"] & 0xFFFFFFFF, got[i]["y"] & 0xFFFFFFFF)
    print("contig %d: %d shimmers, same as the resident path: %s" % (i, len(got[i]), same))
    ok = ok and same
for opt in ({"no_pre_islands": 1}, {"no_pipeline": 1}, {"island_chunk_min": 4096}):
    with ctx.options(**opt):
        g2 = P.sequence_to_shmmrs_batch(seqs, spec, ctx=ctx)
    print(opt, [len(x) for x in g2])
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
r0 = O.sequence_to_shmmrs(0, seqs[0], O.spec())
print("oracle contig 0:", len(r0))
if len(r0) != len(got[0]):
    a, b2 = r0["y"] & 0xFFFFFFFF, got[0]["y"] & 0xFFFFFFFF
    sa, sb = set(a.tolist()), set(b2.tolist())
    print("only in oracle:", sorted(x >> 1 for x in sa - sb)[:20], "only in host path:", sorted(x >> 1 for x in sb - sa)[:20])
sys.exit(0 if ok else 1)
