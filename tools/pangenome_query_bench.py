#!/usr/bin/env python3
"""Queries against a pangenome-like index: H haplotypes of one 500 kbp ancestor (0.1 % substitutions each), queries of 10 / 100 /
400 kbp cut from the ancestor: every query pair hits ~H targets, a query has 10^4 - 10^6 hits in H groups of hundreds of hits
(the per-query LDS grouping does not apply above 4096 hits: the global sort and the wavefront chaining kernels carry these).
Times pgr_query_hps_batch, compares all chains with the CPU restatement for the smaller index."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402
import seqgen  # noqa: E402

rng = np.random.default_rng(11)
anc = np.frombuffer(seqgen.rnd(rng, 500_000), dtype=np.uint8)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
ctx = P.default_context(0)
bad = 0
for H, check in ((100, True), (1000, False)):
    haps = []
    for h in range(H):
        s = anc.copy()
        pos = rng.integers(0, len(s), 500)
        s[pos] = rng.choice(ACGT, 500)
        haps.append(s)
    ix = P.Index(P.make_spec(), ctx=ctx)
    ix.add_seqs(haps)
    ix.finalize()
    oix = None
    if check:
        oix = O.Index(O.spec())
        for sid, s in enumerate(haps):
            oix.add_seq(sid, s)
    for ql in (10_000, 100_000, 400_000):
        q = anc[50_000:50_000 + ql]
        ix.query_hps_raw([q], 0.025)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = ix.query_hps_raw([q], 0.025)
            ts.append(time.perf_counter() - t0)
        p = ctx.last_query_prof()
        line = "H=%4d query %6d bp: %7.2f ms, %7d hits in %4d targets, %d chains" % (H, ql, min(ts) * 1e3, p["n_hits"], len(r["t_sid"]),
                                                                                    len(r["c_score"]))
        if oix is not None:
            t0 = time.perf_counter()
            ref = oix.query_fragment_to_hps(q, 0.025)
            t_cpu = time.perf_counter() - t0
            got = []
            for t in range(int(r["q_off"][0]), int(r["q_off"][1])):
                ch = []
                for c in range(int(r["t_off"][t]), int(r["t_off"][t + 1])):
                    hp = r["hps"][int(r["c_off"][c]):int(r["c_off"][c + 1])]
                    ch.append((float(r["c_score"][c]), [tuple(int(v) for v in h) for h in hp]))
                got.append((int(r["t_sid"][t]), ch))
            same = sorted(got) == sorted(ref)
            bad += 0 if same else 1
            line += "; CPU restatement %.1f ms, identical: %s" % (t_cpu * 1e3, same)
        print(line, flush=True)
sys.exit(1 if bad else 0)
