#!/usr/bin/env python3
"""One reference-chromosome-like contig: 248 Mbp with an 18 Mbp run of N (a centromere gap), forty shorter gaps (50 kbp - 1 Mbp),
a few hundred isolated N and lower-case stretches.  Times pgr_shmmrs_compute (resident input) and compares the whole result
with the CPU restatement (checksum + count).  The gaps are islands of the exact state machine whose seams are corrected one
chunk per round: this is the worst case of that path on realistic input."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle")]
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402

L = 248_000_000
rng = np.random.default_rng(1)
s = O.synth_contig(31, 0, L).copy()
s[121_000_000:139_000_000] = ord("N")
for _ in range(40):
    a = int(rng.integers(0, L - 2_000_000))
    s[a:a + int(rng.integers(50_000, 1_000_000))] = ord("N")
for _ in range(300):
    s[int(rng.integers(0, L))] = ord("N")
for _ in range(200):
    a = int(rng.integers(0, L - 100_000))
    n = int(rng.integers(100, 50_000))
    s[a:a + n] |= 0x20  # lower case (soft-masked repeats): the same bases for the reference's table
ctx = P.default_context(0)
b = P.Batch.from_seqs([s], ctx=ctx)
sp = P.make_spec()
sh = b.shmmrs(sp)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    sh = b.shmmrs(sp)
    ts.append(time.perf_counter() - t0)
p = ctx.last_prof()
if "--rounds" in sys.argv:
    with ctx.options(debug=1, debug_times=1):
        b.shmmrs(sp)
print("GPU: %.1f ms (best of 3: %s), %d shimmers, %.1f Mbp through the exact islands" %
      (min(ts) * 1e3, " ".join("%.1f" % (t * 1e3) for t in ts), sh.count, p.exact_bases / 1e6))
t0 = time.perf_counter()
ref = O.sequence_to_shmmrs(0, s, O.spec())
t_cpu = time.perf_counter() - t0
ok = len(ref) == sh.count and np.array_equal(sh.checksum()[0], O.shmmr_checksum(ref))
print("CPU restatement, one thread: %.2f s, %d shimmers; identical: %s" % (t_cpu, len(ref), ok))
sys.exit(0 if ok else 1)
