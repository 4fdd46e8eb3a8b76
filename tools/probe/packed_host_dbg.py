"""per-window / per-sub-batch timeline (context option debug = 2) of pgr_shmmr_batch_packed on 1.04 Gbp: pageable planes through
the staging windows, pinned planes straight from the caller's arrays"""
import os
import sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
import torch  # noqa: F401,E402
import bench  # noqa: E402
import pgrtk_amd as P  # noqa: E402
ctx = P.Context(0)
spec = P.make_spec()
seqs = [bench.synth_contig_ascii(2, c, 10_000_000) for c in range(104)]
packed, _ = P.pack_ascii(seqs)
bare = P.PackedBases(packed.lens, packed.planes, None)
for what, pin in (("pageable", False), ("pinned", True)):
    for _ in range(2):
        P.time_shmmr_batch_packed(bare, spec, ctx=ctx)
    cm = P.PinnedArrays(bare.planes) if pin else None
    if cm:
        cm.__enter__()
    for _ in range(2):
        P.time_shmmr_batch_packed(bare, spec, ctx=ctx)
    print("==== %s" % what, file=sys.stderr, flush=True)
    with ctx.options(debug=2):
        dt, n = P.time_shmmr_batch_packed(bare, spec, ctx=ctx)
    print("==== %s: %.2f ms" % (what, dt * 1e3), file=sys.stderr, flush=True)
    if cm:
        cm.__exit__(None, None, None)
