"""PCIe-inclusive rate of the B1 drop-in (pgr_shmmr_batch: host ASCII in, host MM128 out), timed at the C ABI (what a
Rust caller sees) and through the Python convenience (the result becomes a numpy view of the library's buffer).
Not the bench metric (bench.py times resident inputs); quoted in DESIGN.md section 5."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle")]
import numpy as np  # noqa: E402,F401
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402
from pgrtk_amd import _ffi  # noqa: E402

ctx = P.default_context(0)
n, L = 100, 10_000_000
seqs = [O.synth_contig(2, i, L) for i in range(n)]
sp = P.make_spec()
L_ = _ffi.lib()
keep, ptrs, lens, nn = _ffi.seq_ptrs(seqs)


def c_call():
    mm, off = C.c_void_p(), C.c_void_p()
    t0 = time.perf_counter()
    rc = L_.pgr_shmmr_batch(ctx.handle, C.byref(sp), nn, ptrs, lens, None, 0, C.byref(mm), C.byref(off))
    dt = time.perf_counter() - t0
    assert rc == 0
    L_.pgr_free(mm)
    L_.pgr_free(off)
    return dt


c_call()  # first call grows the workspaces
dts = sorted(c_call() for _ in range(3))
print("pgr_shmmr_batch at the C ABI (host ASCII -> host MM128), %d x %d bp: %.1f ms = %.1f Gbp/s (median of 3)" %
      (n, L, dts[1] * 1e3, n * L / dts[1] / 1e9))
t0 = time.perf_counter()
out = P.sequence_to_shmmrs_batch(seqs, sp, ctx=ctx)
dt = time.perf_counter() - t0
print("through the Python convenience (a numpy view of the result buffer): %.1f ms = %.1f Gbp/s" % (dt * 1e3, n * L / dt / 1e9))
t0 = time.perf_counter()
b = P.Batch.from_seqs(seqs, ctx=ctx)
t1 = time.perf_counter()
sh = b.shmmrs(sp)
t2 = time.perf_counter()
mm, off = C.c_void_p(), C.c_void_p()
ctx.check(L_.pgr_shmmrs_download(ctx.handle, sh._h, C.byref(mm), C.byref(off)))
t3 = time.perf_counter()
print("  staging+H2D+pack %.1f ms (%.1f GB/s of ASCII), compute %.1f ms, D2H into malloc'd memory %.1f ms (%d shimmers)" %
      ((t1 - t0) * 1e3, n * L / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3, sh.count))
L_.pgr_free(mm)
L_.pgr_free(off)
