#!/usr/bin/env python3
"""A query through a repeat against a pangenome-like index: 8000 contigs of 20 kbp, each with 25 tandem copies of the same
600-bp unit (so a handful of shimmer-pair keys occur 200 000 times in the index, 25 times per sequence) + random flanks; the
query holds 5 copies of the unit between unique flanks taken from contig 17.  Times pgr_query_hps_batch and the CPU restatement,
compares the chains."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402
import seqgen  # noqa: E402

n_ctg = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
rng = np.random.default_rng(3)
unit = seqgen.rnd(rng, 600)
seqs = [seqgen.rnd(rng, 2500) + unit * 25 + seqgen.rnd(rng, 2500) for _ in range(n_ctg)]
query = seqs[17][500:2500] + unit * 5 + seqs[17][-2500:-300]
ctx = P.default_context(0)
ix = P.Index(P.make_spec(), ctx=ctx)
ix.add_seqs(seqs)
ix.finalize()
ix.query_hps_raw([query], 0.025)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    r = ix.query_hps_raw([query], 0.025)
    ts.append(time.perf_counter() - t0)
p = ctx.last_query_prof()
print("GPU: %.2f ms (%s); %d query pairs, %d signatures looked up, %d hits, %d targets" %
      (min(ts) * 1e3, " ".join("%.2f" % (t * 1e3) for t in ts), p["n_query_pairs"], p["n_signatures"], p["n_hits"], len(r["t_sid"])))
print("  stages: shimmers %.2f ms, lookup + counts %.2f ms, hits + chaining %.2f ms" % (p["shmmr_ms"], p["lookup_ms"], p["chain_ms"]))
oix = O.Index(O.spec())
for sid, s in enumerate(seqs):
    oix.add_seq(sid, s)
t0 = time.perf_counter()
ref = oix.query_fragment_to_hps(query, 0.025)
t_cpu = time.perf_counter() - t0
got = []
for t in range(int(r["q_off"][0]), int(r["q_off"][1])):
    ch = []
    for c in range(int(r["t_off"][t]), int(r["t_off"][t + 1])):
        hp = r["hps"][int(r["c_off"][c]):int(r["c_off"][c + 1])]
        ch.append((float(r["c_score"][c]), [tuple(int(v) for v in h) for h in hp]))
    got.append((int(r["t_sid"][t]), ch))
ok = sorted(got) == sorted(ref)
print("CPU restatement: %.1f ms, %d targets; identical: %s" % (t_cpu * 1e3, len(ref), ok))
sys.exit(0 if ok else 1)
