import os
import sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "pgr-tk_amd")); sys.path.insert(0, os.path.join(_R, "oracle"))
import numpy as np, pgrtk_amd as P, oracle as O
ctx = P.default_context(0)
import os
n, L = int(os.environ.get("NCONTIG", "64")), 10_000_000
seqs = []
for i in range(n):
    s = O.synth_contig(9, i, L).copy()
    s[1234567] = ord("N")          # one N: the whole contig takes the exact-machine path
    if i % 4 == 0: s[5_000_000:5_300_000] = ord("N")
    seqs.append(s)
b = P.Batch.from_seqs(seqs, ctx=ctx)
sp = P.make_spec()
sh = b.shmmrs(sp)
t0 = time.perf_counter(); sh = b.shmmrs(sp); dt = time.perf_counter() - t0
pr = ctx.last_prof()
print("exact-machine path: %d x %d bp in %.1f ms = %.1f Gbp/s; serial contigs %d; aux %.1f ms" % (n, L, dt*1e3, n*L/dt/1e9, pr.n_serial_contigs, pr.level1_aux_ms))
mm, off = sh.download()
osp = O.spec()
for i in (0, 1, 4):
    ref = O.sequence_to_shmmrs(i, seqs[i], osp)
    got = mm[int(off[i]):int(off[i+1])]
    assert len(ref) == len(got) and np.array_equal(ref["x"], got["x"]) and np.array_equal(ref["y"], got["y"]), i
print("bit exact vs oracle on 3 contigs")
