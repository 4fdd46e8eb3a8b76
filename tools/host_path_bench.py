"""PCIe-inclusive rate of the B1 drop-in (pgr_shmmr_batch: host ASCII in, host MM128 out) and a breakdown of
the query leg.  Not the bench metric (bench.py times resident inputs); quoted in DESIGN.md section 5."""
import sys, time
sys.path.insert(0, "/root/repo/pgr-tk_amd"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, pgrtk_amd as P, oracle as O
ctx = P.default_context(0)
n, L = 100, 10_000_000
seqs = [O.synth_contig(2, i, L) for i in range(n)]
sp = P.make_spec()
P.sequence_to_shmmrs_batch(seqs, sp, ctx=ctx)  # first call grows the workspaces
t0 = time.perf_counter(); out = P.sequence_to_shmmrs_batch(seqs, sp, ctx=ctx); dt = time.perf_counter() - t0
print("pgr_shmmr_batch (host ASCII -> host MM128), %d x %d bp: %.1f ms = %.1f Gbp/s" % (n, L, dt * 1e3, n * L / dt / 1e9))
t0 = time.perf_counter(); b = P.Batch.from_seqs(seqs, ctx=ctx); t1 = time.perf_counter(); sh = b.shmmrs(sp); t2 = time.perf_counter(); mm, off = sh.download(); t3 = time.perf_counter()
print("  staging+H2D+pack %.1f ms (%.1f GB/s of ASCII), compute %.1f ms, D2H %.1f ms" % ((t1 - t0) * 1e3, n * L / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
