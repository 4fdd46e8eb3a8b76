"""BASELINE.json's target on one GPU, cold: 100 Gbp of distinct synthetic contigs into ONE index in a FRESH context (first pass)
and again (steady state), through the synchronous calls and through pgr_pipe_*.
usage: t100_pipe_probe.py [sync|pipe] [batches] [--reserve-gib G] [--json]      (--reserve-gib: pgr_ctx_reserve first, timed apart)"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
import torch  # noqa: F401,E402
import pgrtk_amd as P  # noqa: E402

as_json = "--json" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--json"]
reserve_gib = 0.0
if "--reserve-gib" in argv:
    i = argv.index("--reserve-gib")
    reserve_gib = float(argv[i + 1])
    del argv[i:i + 2]
mode = argv[0] if len(argv) > 0 else "pipe"
n_b = int(argv[1]) if len(argv) > 1 else 10
n_c, L = 1000, 10_000_000
# what fresh device memory costs this process right now (4 GiB allocated and written once): milliseconds on a quiet device, a
# tenth of a second and more when another process or context has just released tens of GB (the driver hands such memory out
# only once it has been cleared)
torch.cuda.init()
torch.cuda.synchronize()
_t = time.perf_counter()
_x = torch.empty(4 << 30, dtype=torch.uint8, device="cuda:0")
_x.zero_()
torch.cuda.synchronize()
first_touch_ms = (time.perf_counter() - _t) * 1e3
del _x
torch.cuda.empty_cache()
t_ctx = time.perf_counter()
ctx = P.Context(0)
spec = P.make_spec()
t_ctx = time.perf_counter() - t_ctx
t_reserve = None
if reserve_gib > 0:  # one block for the whole build, allocated and touched now (include/pgr_hip.h: pgr_ctx_reserve)
    t_reserve = time.perf_counter()
    ctx.reserve(int(reserve_gib * (1 << 30)))
    ctx.synchronize()
    t_reserve = time.perf_counter() - t_reserve


rbuf = [torch.empty((33_000_000, 5), dtype=torch.int64, device="cuda:0") for _ in range(2)] if os.environ.get("T100_RECPTR") else None


def once():
    ctx.synchronize()
    t0 = time.perf_counter()
    ix = P.Index(spec, ctx=ctx)
    if hasattr(ix, "reserve"):
        ix.reserve(int(n_b * n_c * L * 0.00304 * 1.02))
    if mode == "pipe":
        pipe = P.Pipe(spec, ctx=ctx)
        for bi in range(n_b):
            ids = list(range(bi * n_c, (bi + 1) * n_c))
            ta = time.perf_counter()
            if os.environ.get("T100_SAME_BATCH") and bi > 0:
                b = same
            else:
                b = same = P.Batch.synthetic([L] * n_c, seed=2, ctx=ctx, contig_ids=ids)
            tb = time.perf_counter()
            if pipe.in_flight == 2:
                pipe.collect(want_shmmrs=False)
            tc = time.perf_counter()
            if os.environ.get("T100_RECPTR"):
                pipe.submit(b, sids=ids, rec_ptr=rbuf[bi & 1].data_ptr(), rec_capacity=rbuf[0].shape[0])
            else:
                pipe.submit(b, sids=ids, index=ix)
            td = time.perf_counter()
            del b
            if os.environ.get("T100_VERBOSE"):
                print("   batch %d: synthetic %.2f ms, collect %.2f ms, submit %.2f ms (tile kernel of the collected job %.2f ms)" %
                      (bi, (tb - ta) * 1e3, (tc - tb) * 1e3, (td - tc) * 1e3, ctx.last_prof().level1_ms))
        while pipe.in_flight:
            pipe.collect(want_shmmrs=False)
        pipe.close()
    else:
        for bi in range(n_b):
            ids = list(range(bi * n_c, (bi + 1) * n_c))
            b = P.Batch.synthetic([L] * n_c, seed=2, ctx=ctx, contig_ids=ids)
            ix.add_resident(b, sids=ids)
            del b
    t1 = time.perf_counter()
    ix.finalize()
    ctx.synchronize()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, ix.n_records, ix.n_keys, ix


if not as_json:
    print("%s: context created in %.3f s; first touch of 4 GiB of device memory %.1f ms%s" %
          (mode, t_ctx, first_touch_ms, "; pgr_ctx_reserve(%.0f GiB) %.3f s" % (reserve_gib, t_reserve) if t_reserve is not None else ""), flush=True)
res = []
for what in ("fresh context", "again", "again"):
    ctx.mem_stats(reset_peak=True)
    a, b, nr, nk, ix = once()
    cs = ix.records_checksum()
    del ix
    res.append({"what": what, "s": a + b, "batches_s": a, "sort_into_frag_map_s": b, "records": nr, "keys": nk,
                "records_checksum": ["%016x" % cs[0], "%016x" % cs[1]], "peak_device_bytes_of_the_allocator": ctx.mem_stats()[1]})
    if not as_json:
        print("%s, %s: %d Gbp in %.3f s (batches %.3f s, sort into the frag_map %.3f s), %d records, %d keys, checksum %016x %016x"
              % (mode, what, n_b * n_c * L // 10**9, a + b, a, b, nr, nk, cs[0], cs[1]), flush=True)
if as_json:
    import json
    print(json.dumps({"mode": mode, "bp": n_b * n_c * L, "context_create_s": t_ctx, "first_touch_of_4GiB_ms": first_touch_ms,
                      "reserve_gib": reserve_gib, "reserve_s": t_reserve, "arena": ctx.arena_stats(), "passes": res}), flush=True)
