"""The index build (pair records -> sorted CSR + lookup tables, pgr_index_finalize) of BASELINE.json configs[1]'s records alone,
for a rocprofv3 kernel trace:  rocprofv3 --kernel-trace --stats -- python tools/index_build_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
import torch  # noqa: E402
import pgrtk_amd as P  # noqa: E402
from pgrtk_amd import exchange  # noqa: E402

ctx = P.default_context(0)
spec = P.make_spec()
n = int(os.environ.get("N_CONTIGS", "1000"))
b = P.Batch.synthetic([10_000_000] * n, seed=2, ctx=ctx)
sh = b.shmmrs(spec)
recs = torch.empty((sh.n_pairs + 16, exchange.REC_WORDS), dtype=torch.int64, device="cuda:0")
np_ = sh.frag_recs_into(recs.data_ptr(), recs.shape[0], sids=list(range(n)))
del b, sh
for rep in range(4):
    ctx.synchronize()
    t0 = time.perf_counter()
    ix = P.Index(spec, ctx=ctx)
    ix.add_records(device_ptr=recs.data_ptr(), n=np_)
    t1 = time.perf_counter()
    ix.finalize()
    t2 = time.perf_counter()
    print("add_records %.2f ms, finalize %.2f ms (%d records, %d keys)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, ix.n_records, ix.n_keys), flush=True)
    del ix
