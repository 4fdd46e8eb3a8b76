"""latency of the B1 drop-in at the batch sizes the reference feeds (<= 129 contigs per call, seq_db.rs:549-564):
pgr_shmmr_batch at the C ABI for a few batch shapes.  Shows why INTEGRATION.md recommends larger batches."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle")]
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402
from pgrtk_amd import _ffi  # noqa: E402

ctx = P.default_context(0)
sp = P.make_spec()
L_ = _ffi.lib()
for n, L in [(1, 10_000), (129, 10_000), (129, 100_000), (129, 1_000_000), (16, 10_000_000)]:
    seqs = [O.synth_contig(7, i, L) for i in range(n)]
    keep, ptrs, lens, nn = _ffi.seq_ptrs(seqs)

    def call():
        mm, off = C.c_void_p(), C.c_void_p()
        t0 = time.perf_counter()
        rc = L_.pgr_shmmr_batch(ctx.handle, C.byref(sp), nn, ptrs, lens, None, 0, C.byref(mm), C.byref(off))
        dt = time.perf_counter() - t0
        assert rc == 0
        L_.pgr_free(mm)
        L_.pgr_free(off)
        return dt
    call()
    call()
    dts = sorted(call() for _ in range(7))
    print("%4d x %9d bp: %8.3f ms per call  (%7.2f Gbp/s)" % (n, L, dts[3] * 1e3, n * L / dts[3] / 1e9))
