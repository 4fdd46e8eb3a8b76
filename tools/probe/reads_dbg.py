import os, sys, time
sys.path[:0] = [os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "pgr-tk_amd")]
import pgrtk_amd as P
ctx = P.default_context(0)
sp = P.make_spec()
b = P.Batch.synthetic([1000] * 1_000_000, seed=41, ctx=ctx)
sh = b.shmmrs(sp)
sh = b.shmmrs(sp)
with ctx.options(debug_times=1):
    for _ in range(2):
        t0 = time.perf_counter()
        sh = b.shmmrs(sp)
        print("call %.2f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
p = ctx.last_prof()
print("level1 %.3f aux %.3f level2 %.3f total %.3f" % (p.level1_ms, p.level1_aux_ms, p.level2_ms, p.total_ms))
