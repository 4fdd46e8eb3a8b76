"""the list kernel (fused_select_kernel) persistent + descriptor prefetch (default) against one workgroup per block (option
no_persistent_list): stage times of the headline step (10 Gbp resident) and of a batch of reads"""
import os
import sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import time  # noqa: E402
import torch  # noqa: F401,E402
import pgrtk_amd as P  # noqa: E402

ctx = P.Context(0)
spec = P.make_spec()
for what, lens in (("1000 x 10 Mbp", [10_000_000] * 1000), ("10^6 x 1 kbp", [1000] * 1_000_000), ("10 000 x 10 kbp", [10_000] * 10_000)):
    b = P.Batch.synthetic(lens, seed=2, ctx=ctx)
    ref = None
    for opt in (0, 1, 0, 1):
        with ctx.options(no_persistent_list=opt):
            b.shmmrs(spec)
            ts, l2 = [], []
            for _ in range(5):
                t0 = time.perf_counter()
                sh = b.shmmrs(spec)
                ts.append(time.perf_counter() - t0)
                l2.append(ctx.last_prof().level2_ms)
            cs = sh.checksum()
            if ref is None:
                ref = cs
            same = bool((cs == ref).all())
            del sh
        print("%-16s no_persistent_list %d: call %.3f ms (min), level-2 stage %.3f ms (min), same result %s"
              % (what, opt, min(ts) * 1e3, min(l2), same), flush=True)
    del b
