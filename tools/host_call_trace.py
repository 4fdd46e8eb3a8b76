"""One pipelined host call under rocprofv3 (--kernel-trace --memory-copy-trace): 104 x 10 Mbp packed planes (and ASCII) through
pgr_shmmr_batch_packed / pgr_shmmr_batch, three calls each.  tools/summarize_host_trace.py turns the trace into a timeline."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import bench  # noqa: E402
import pgrtk_amd as P  # noqa: E402

ctx = P.default_context(0)
n, L = 104, 10_000_000
seqs = [bench.synth_contig_ascii(2, i, L) for i in range(n)]
sp = P.make_spec()
packed, _ = P.pack_ascii(seqs)
bare = P.PackedBases(packed.lens, packed.planes, None)
for name, f in (("planes", lambda: P.time_shmmr_batch_packed(bare, sp, ctx=ctx)), ("ascii", lambda: P.time_shmmr_batch(seqs, sp, ctx=ctx))):
    f()
    for _ in range(3):
        t0 = time.perf_counter()
        f()
        print("%s call: %.2f ms" % (name, (time.perf_counter() - t0) * 1e3), flush=True)
    time.sleep(0.05)
