"""Finer bisect of cold_leg_probe.py: which call inside bench.latency_bench makes a later fresh context's first pass slow."""
import os
import sys
import time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import torch  # noqa: E402
import bench  # noqa: E402
import pgrtk_amd as P  # noqa: E402

n_b, n_c, L = 10, 1000, 10_000_000
spec = P.make_spec()
ctx = P.Context(0)
torch.zeros(1, device="cuda:0")


def leg(tag):
    fresh = P.Context(0)
    t00 = time.perf_counter()
    x = torch.empty(8 << 30, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    t01 = time.perf_counter()
    del x
    torch.cuda.empty_cache()
    t02 = time.perf_counter()
    for what in ("first pass", "again"):
        fresh.synchronize()
        t0 = time.perf_counter()
        ix = P.Index(spec, ctx=fresh)
        ix.reserve(int(n_b * n_c * L * 0.003036 * 1.01) + 4096)
        tr = time.perf_counter()
        pipe = P.Pipe(spec, ctx=fresh)
        per = []
        for bi in range(n_b):
            ta = time.perf_counter()
            ids = list(range(bi * n_c, (bi + 1) * n_c))
            b = P.Batch.synthetic([L] * n_c, seed=2, ctx=fresh, contig_ids=ids)
            tb = time.perf_counter()
            if pipe.in_flight == 2:
                pipe.collect(want_shmmrs=False)
            tc = time.perf_counter()
            pipe.submit(b, sids=ids, index=ix)
            del b
            per.append("%.0f/%.0f/%.0f" % ((tb - ta) * 1e3, (tc - tb) * 1e3, (time.perf_counter() - tc) * 1e3))
        while pipe.in_flight:
            pipe.collect(want_shmmrs=False)
        t1 = time.perf_counter()
        ix.finalize()
        fresh.synchronize()
        pipe.close()
        del ix
        print("%-30s %s: batches %.3f s (reserve %.3f); 8 GiB torch alloc %.3f s free %.3f s; per batch synth/collect/submit ms: %s"
              % (tag, what, t1 - t0, tr - t0, t01 - t00, t02 - t01, " ".join(per)), flush=True)
    fresh.close()


class A:
    seed = 2


leg("after nothing")
seq = bench.synth_contig_ascii(2, 0, 10_000)
one = P.PackedSeqs.from_list([seq])
for _ in range(10):
    P.time_shmmr_batch(one, spec, ctx=ctx)
leg("after small shmmr_batch x10")
many = P.PackedSeqs.from_list([bench.synth_contig_ascii(2, c, 10_000) for c in range(129)])
for _ in range(10):
    P.time_shmmr_batch(many, spec, ctx=ctx)
leg("after 129-contig batch x10")
b = P.Batch.synthetic([1_000_000] * 8, seed=2, ctx=ctx)
ix = P.Index(spec, ctx=ctx)
ix.add_resident(b)
ix.finalize()
leg("after small index build")
q = P.PackedSeqs.from_list([bench.synth_contig_ascii(2, 3, 200_000)[50_000:60_000]])
for _ in range(10):
    ix.time_query_host(q, 0.025)
leg("after 10 queries")
for _ in range(10):
    ix.query_hps_raw(q, 0.025)
leg("after 10 raw queries")
