"""pgr_pipe on the headline batch: ms per batch against the synchronous step, per setting of the back stream's priority.
    python tools/probe/pipe_probe.py [contigs] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
import pgrtk_amd as P  # noqa: E402
from pgrtk_amd import exchange  # noqa: E402

n_c = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = 10_000_000
spec = P.make_spec(80, 56, 4, 64)
SIDS = list(range(n_c)) if os.environ.get("PROBE_SIDS") else None
for prio in [int(v) for v in os.environ.get("PROBE_PRIOS", "1,0,-1").split(",")]:
    os.environ["PGR_BACK_PRIORITY"] = str(prio)
    ctx = P.Context(0)
    print("   options: front_priority %d, pipe_small_list %d, lds_match %d" % (ctx.get_option("front_priority"), ctx.get_option("pipe_small_list"), ctx.get_option("lds_match")), flush=True)
    b = P.Batch.synthetic([L] * n_c, seed=2, ctx=ctx)
    probe = b.shmmrs(spec)
    bufs = [torch.empty((int(probe.n_pairs * 1.05) + 16, exchange.REC_WORDS), dtype=torch.int64, device="cuda:0") for _ in range(2)]
    del probe

    def sync_steps(k):
        for _ in range(k):
            sh = b.shmmrs(spec)
            sh.frag_recs_into(bufs[0].data_ptr(), bufs[0].shape[0])
            del sh
    sync_steps(3)
    ctx.synchronize()
    t = time.perf_counter()
    sync_steps(steps)
    t_sync = (time.perf_counter() - t) / steps
    pipe = P.Pipe(spec, ctx=ctx)
    lv1 = []

    def pipe_steps(k):
        for i in range(k):
            if pipe.in_flight == 2:
                pipe.collect(want_shmmrs=False)
                lv1.append(ctx.last_prof().level1_ms)
            pipe.submit(b, sids=SIDS, rec_ptr=bufs[i & 1].data_ptr(), rec_capacity=bufs[0].shape[0])
        while pipe.in_flight:
            pipe.collect(want_shmmrs=False)
            lv1.append(ctx.last_prof().level1_ms)
    pipe_steps(4)
    ctx.synchronize()
    del lv1[:]
    t = time.perf_counter()
    pipe_steps(steps)
    t_pipe = (time.perf_counter() - t) / steps
    print("back stream priority %s: synchronous %.3f ms per batch, pipelined %.3f ms (tile kernel %.3f ms beside the list stage): %.1f Gbp/s -> %.1f Gbp/s"
          % ({1: "high", 0: "default", -1: "low"}[prio], t_sync * 1e3, t_pipe * 1e3, float(np.mean(lv1)), n_c * L / t_sync / 1e9, n_c * L / t_pipe / 1e9), flush=True)
    pipe.close()
    del b, bufs, pipe, ctx
