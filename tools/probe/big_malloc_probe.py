"""What makes ONE later large hipMalloc in the process take ~2.5 s (tools/probe/cold_leg_probe2.py: it follows the small-batch
entry points)?  Times an 8 GiB torch allocation (a raw hipMalloc; empty_cache returns it) after each candidate."""
import ctypes
import os
import sys
import time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import torch  # noqa: E402
import bench  # noqa: E402
import pgrtk_amd as P  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
spec = P.make_spec()
torch.zeros(1, device="cuda:0")


def big(tag):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = torch.empty(8 << 30, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        del x
        torch.cuda.empty_cache()
    print("%-64s 8 GiB hipMalloc x3: %s s" % (tag, " ".join("%.3f" % t for t in ts)), flush=True)


big("start")
ctx = P.Context(0)
big("after context create")
p = ctypes.c_void_p()
hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(64 << 20), 0)
big("after hipHostMalloc 64 MiB")
hip.hipHostFree(p)
big("after hipHostFree")
ps = []
for i in range(64):
    q = ctypes.c_void_p()
    hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t(8192 << (i % 8)))
    ps.append(q)
big("after 64 small hipMalloc")
for q in ps:
    hip.hipFree(q)
big("after freeing them")
b = P.Batch.synthetic([10_000], seed=2, ctx=ctx)
big("after Batch.synthetic 10 kbp")
sh = b.shmmrs(spec)
big("after resident shmmrs of it (small path kernel)")
del sh, b
seq = bench.synth_contig_ascii(2, 0, 10_000)
one = P.PackedSeqs.from_list([seq])
P.time_shmmr_batch(one, spec, ctx=ctx)
big("after ONE host small shmmr_batch")
P.time_shmmr_batch(one, spec, ctx=ctx)
big("after a second one")
for _ in range(20):
    P.time_shmmr_batch(one, spec, ctx=ctx)
big("after 20 more")
print(ctx.mem_stats(), flush=True)
