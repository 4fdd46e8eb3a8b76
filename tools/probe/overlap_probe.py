"""Does the HBM-bound list stage of one batch hide behind the VALU-bound tile kernel of another?  Cheapest possible probe:
two contexts (each with its own streams) step through their own resident batch from two host threads at once; the aggregate
time per batch against one context alone says what a real software pipeline inside the library could win at most.
    python tools/probe/overlap_probe.py [contigs] [steps]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
import torch  # noqa: F401,E402
import pgrtk_amd as P  # noqa: E402
from pgrtk_amd import exchange  # noqa: E402

n_c = int(sys.argv[1]) if len(sys.argv) > 1 else 500
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = 10_000_000
spec = P.make_spec(80, 56, 4, 64)


def make(ctx, c0):
    b = P.Batch.synthetic([L] * n_c, seed=2, ctx=ctx, contig_ids=list(range(c0, c0 + n_c)))
    probe = b.shmmrs(spec)
    buf = torch.empty((int(probe.n_pairs * 1.05) + 16, exchange.REC_WORDS), dtype=torch.int64, device="cuda:0")
    del probe
    return b, buf


def run(ctx, b, buf, k):
    for _ in range(k):
        sh = b.shmmrs(spec)
        sh.frag_recs_into(buf.data_ptr(), buf.shape[0])
        del sh


c0, c1 = P.Context(0), P.Context(0)
w0, w1 = make(c0, 0), make(c1, n_c)
run(c0, *w0, 3)
run(c1, *w1, 3)
torch.cuda.synchronize()
t = time.perf_counter()
run(c0, *w0, steps)
one = (time.perf_counter() - t) / steps
th = [threading.Thread(target=run, args=(c, *w, steps)) for c, w in ((c0, w0), (c1, w1))]
t = time.perf_counter()
for x in th:
    x.start()
for x in th:
    x.join()
two = (time.perf_counter() - t) / (2 * steps)
print("one context: %.3f ms per batch of %.1f Gbp (%.1f Gbp/s); two contexts at once: %.3f ms per batch (%.1f Gbp/s), %.1f %% less" %
      (one * 1e3, n_c * L / 1e9, n_c * L / one / 1e9, two * 1e3, n_c * L / two / 1e9, (1 - two / one) * 100))
