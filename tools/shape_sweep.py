#!/usr/bin/env python3
"""Batch shapes beyond the bench's 1000 x 10 Mbp: many short contigs (reads / fragmented assemblies), one very long contig.
Synthetic contigs generated on the device (pgr_batch_synthetic) and, for the check, inside the CPU worker threads
(orc_synth_checksums_threads): every contig's 128-bit content checksum and count compared.  Prints Gbp/s per shape."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle")]
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402

ctx = P.default_context(0)
sp, osp = P.make_spec(), O.spec()
bad = 0
for n, L in ((1_000_000, 1_000), (200_000, 5_000), (20_000, 50_000), (100, 10_000_000), (4, 250_000_000), (1, 1_000_000_000)):
    b = P.Batch.synthetic([L] * n, seed=41, ctx=ctx)
    sh = b.shmmrs(sp)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        sh = b.shmmrs(sp)
        ts.append(time.perf_counter() - t0)
    sums, off = sh.checksum(), sh.offsets()
    n_chk = min(n, max(4, int(2_000_000_000 // L // 16)))  # ~2 Gbp of CPU work at most per shape
    if L >= 250_000_000:
        n_chk = min(n, 2)
    cnt, ref, _ = O.synth_checksums_threads(osp, n_chk, 41, 0, L, 16)
    same = int(sum(int(off[i + 1] - off[i]) == int(cnt[i]) and np.array_equal(sums[i], ref[i]) for i in range(n_chk)))
    bad += n_chk - same
    print("%8d x %10d bp: %8.2f ms = %6.1f Gbp/s, %9d shimmers; %d of %d checked contigs identical" %
          (n, L, min(ts) * 1e3, n * L / min(ts) / 1e9, sh.count, same, n_chk), flush=True)
    del sh, b
sys.exit(1 if bad else 0)
