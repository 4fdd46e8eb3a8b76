"""10^6 x 1 kbp reads through the host entry point (pgr_shmmr_batch: host ASCII in, host MM128 out) -- what a caller with a FASTQ
in memory sees.  Prints the C call's time and the context's lap times of the last call."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd")]
import numpy as np  # noqa: E402
import pgrtk_amd as P  # noqa: E402
from pgrtk_amd import _ffi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000
ctx = P.default_context(0)
rng = np.random.default_rng(5)
buf = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n * L)
seqs = P.PackedSeqs(buf, np.arange(n + 1, dtype=np.uint64) * np.uint64(L)) if hasattr(P, "PackedSeqs") else None
keep, ptrs, lens, nn = _ffi.seq_ptrs(seqs)
sp = P.make_spec()
lib = _ffi.lib()


def call():
    mm, off = C.c_void_p(), C.c_void_p()
    t0 = time.perf_counter()
    rc = lib.pgr_shmmr_batch(ctx.handle, C.byref(sp), nn, ptrs, lens, None, 0, C.byref(mm), C.byref(off))
    dt = time.perf_counter() - t0
    assert rc == 0, ctx.last_error()
    lib.pgr_free(mm)
    lib.pgr_free(off)
    return dt


call()
call()
ts = sorted(call() for _ in range(5))
print("pgr_shmmr_batch, %d x %d bp from host ASCII: %.2f ms (median of 5; best %.2f) = %.1f Gbp/s" % (n, L, ts[2] * 1e3, ts[0] * 1e3, n * L / ts[2] / 1e9))
with ctx.options(debug=1):
    t = call()
print('traced call %.2f ms' % (t * 1e3))
