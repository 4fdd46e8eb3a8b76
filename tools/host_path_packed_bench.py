"""Where the time of the host-buffer entry points goes (pgr_shmmr_batch / pgr_shmmr_batch_packed): staging alone
(pack or copy into the pinned windows + H2D), compute, download; then the pipelined calls with the library's own
timeline (context option debug).  Quoted in DESIGN.md section 5."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import numpy as np  # noqa: E402,F401
import bench  # noqa: E402
import pgrtk_amd as P  # noqa: E402

ctx = P.default_context(0)
n, L = int(os.environ.get("N_CONTIGS", "104")), 10_000_000
seqs = [bench.synth_contig_ascii(2, i, L) for i in range(n)]
bp = n * L
sp = P.make_spec()
packed, _ = P.pack_ascii(seqs)
bare = P.PackedBases(packed.lens, packed.planes, None)


def best(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        ctx.synchronize()
        t0 = time.perf_counter()
        r = f()
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
        del r
    return min(ts)


for name, mk, bytes_per_bp in (("ASCII  (pack on the host + H2D)", lambda: P.Batch.from_seqs(seqs, ctx=ctx), 0.375),
                               ("packed + validity (copy + H2D)", lambda: P.Batch.from_packed(packed, ctx=ctx), 0.375),
                               ("packed planes only (copy + H2D)", lambda: P.Batch.from_packed(bare, ctx=ctx), 0.25)):
    t = best(mk)
    print("stage %-34s %7.2f ms = %6.1f Gbp/s, %5.1f GB/s on the link" % (name, t * 1e3, bp / t / 1e9, bytes_per_bp * bp / t / 1e9))
b = P.Batch.from_packed(packed, ctx=ctx)
t = best(lambda: b.shmmrs(sp))
print("compute (resident) %7.2f ms = %6.1f Gbp/s" % (t * 1e3, bp / t / 1e9))
sh = b.shmmrs(sp)
t = best(lambda: sh.download())
print("download of %d shimmers into malloc'd memory %7.2f ms" % (sh.count, t * 1e3))
for name, f in (("pgr_shmmr_batch (ASCII)", lambda: P.time_shmmr_batch(seqs, sp, ctx=ctx)),
                ("pgr_shmmr_batch_packed (+ validity)", lambda: P.time_shmmr_batch_packed(packed, sp, ctx=ctx)),
                ("pgr_shmmr_batch_packed (planes only)", lambda: P.time_shmmr_batch_packed(bare, sp, ctx=ctx))):
    f()
    ts = sorted(f()[0] for _ in range(3))
    print("%-40s %7.2f ms = %6.1f Gbp/s" % (name, ts[1] * 1e3, bp / ts[1] / 1e9))
sys.stdout.flush()
ctx.set_option("debug", 1)
print("--- timeline of one pipelined packed call (planes only)", flush=True)
P.time_shmmr_batch_packed(bare, sp, ctx=ctx)
print("--- timeline of one pipelined ASCII call", flush=True)
P.time_shmmr_batch(seqs, sp, ctx=ctx)
