#!/usr/bin/env python3
"""Randomised parity campaign for the index / query path (GPU box): random repeat-rich databases, random queries
(substrings, reverse complements, chimeras, mutated, unrelated), random count filters / span / gap / orientation
parameters; SeqIndexDB.query_fragments_to_hps compared with the oracle's query_fragment_to_hps, chains and f32 scores
bit exact.   usage: fuzz_query.py [iterations] [seed0] [short]
"short": every query is short enough for the one-wavefront-per-query path (csrc/query_fused.hip) -- the summary says how many
batches took it -- and the same batch is run through the stage-by-stage kernels as well (both must equal the oracle)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402
import seqgen  # noqa: E402

COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def rc(s):
    return s.translate(COMP)[::-1]


SHORT = len(sys.argv) > 3 and sys.argv[3] == "short"
PATHS = [0, 0, 0, 0]  # pgr_query_prof.path of the batches: 0 stage by stage, 1 / 2 the per-query kernel behind the shimmer pipeline, 3 its level-1 form


def one_case(seed, ctx):
    rng = np.random.default_rng(seed)
    si = int(rng.integers(0, 4))
    spec_t = [(80, 56, 4, 64), (48, 56, 4, 12), (24, 24, 2, 8), (31, 21, 3, 16)][si]
    qmax = [30000, 9000, 1200, 2500][si] if SHORT else 60000
    cores = [seqgen.rnd(rng, int(rng.integers(2000, 40000))) for _ in range(int(rng.integers(2, 6)))]
    seqs = []
    for _ in range(int(rng.integers(2, 14))):
        parts = [seqgen.rnd(rng, int(rng.integers(0, 20000)))]
        for j in rng.permutation(len(cores))[: int(rng.integers(1, len(cores) + 1))]:
            c = cores[j]
            c = c if rng.random() < 0.6 else rc(c)
            parts.append(c * int(rng.integers(1, 4)) if rng.random() < 0.2 else c)  # tandem copies
            parts.append(seqgen.rnd(rng, int(rng.integers(0, 6000))))
        seqs.append(b"".join(parts))
    if rng.random() < 0.3:
        seqs.append(seqs[0])
    sdb = P.SeqIndexDB(ctx=ctx)
    sdb.load_from_seq_list([("s%d" % i, s) for i, s in enumerate(seqs)], w=spec_t[0], k=spec_t[1], r=spec_t[2], min_span=spec_t[3])
    oix = O.Index(O.spec(*spec_t))
    for i, s in enumerate(seqs):
        oix.add_seq(i, s)
    oix.finalize()
    queries = []
    for _ in range(int(rng.integers(3, 12)) if not SHORT else int(rng.integers(1, 200))):
        kind = rng.random()
        src = seqs[int(rng.integers(0, len(seqs)))]
        if kind < 0.5 and len(src) > 100:
            a = int(rng.integers(0, len(src) - 50))
            q = src[a:a + int(rng.integers(50, qmax))]
        elif kind < 0.7 and not SHORT:
            q = cores[int(rng.integers(0, len(cores)))] + cores[int(rng.integers(0, len(cores)))]
        elif kind < 0.8:
            q = seqgen.rnd(rng, int(rng.integers(0, 5000)))
        elif SHORT:
            a = int(rng.integers(0, max(1, len(src) - 50)))
            q = (src[a:a + int(rng.integers(50, qmax // 2))]) * 2  # a query that repeats itself: multiplicities > 1
        else:
            q = src
        if rng.random() < 0.5:
            q = rc(q)
        if rng.random() < 0.3 and len(q) > 10:
            qa = bytearray(q)
            for p in rng.integers(0, len(qa), max(1, len(qa) // 2000)):
                qa[p] = b"ACGT"[int(rng.integers(0, 4))]
            q = bytes(qa)
        queries.append(q)
    pen = float(rng.choice([0.025, 0.5, 0.1, 0.0, 1.0]))
    mc, mq, mt = [int(rng.choice([1, 2, 8, 128, 100000])) for _ in range(3)]
    span = int(rng.choice([1, 2, 8, 16, 64]))
    gap = None if rng.random() < 0.6 else int(rng.choice([0, 500, 5000, 100000]))
    ori = bool(rng.random() < 0.3)
    got = sdb.query_fragments_to_hps(queries, pen, mc, mq, mt, span, gap, ori)
    PATHS[int(ctx.last_query_prof()["path"])] += 1
    if SHORT:
        # the form that was NOT taken above: the level-1 form of the per-query kernel (round 6) against the chained one
        with ctx.options(no_query_level1=1):
            got1 = sdb.query_fragments_to_hps(queries, pen, mc, mq, mt, span, gap, ori)
        if got1 != got:
            return "seed %d: the level-1 form differs from the chained form (spec %s pen %g counts %d/%d/%d span %d gap %s oriented %s)" % (
                seed, spec_t, pen, mc, mq, mt, span, gap, ori)
        with ctx.options(no_fused_query=1):
            got2 = sdb.query_fragments_to_hps(queries, pen, mc, mq, mt, span, gap, ori)
        # once more on the same index through the general shimmer pipeline: from the second batch on the per-query kernel is
        # enqueued behind it without a host wait (path 2)
        with ctx.options(no_small_path=1):
            got3 = sdb.query_fragments_to_hps(queries, pen, mc, mq, mt, span, gap, ori)
        PATHS[int(ctx.last_query_prof()["path"])] += 1
        if got3 != got:
            return "seed %d: the chained query path differs (spec %s pen %g counts %d/%d/%d span %d gap %s oriented %s)" % (
                seed, spec_t, pen, mc, mq, mt, span, gap, ori)
        # ... and through the pipe (pgr_pipe_submit_query / _collect_query): the queries in three resident batches, two in flight;
        # every section of every collected result against pgr_query_hps_resident on the same batch
        import ctypes as C
        from pgrtk_amd import _ffi
        L = _ffi.lib()
        spec = P.make_spec(*spec_t)
        third = (len(queries) + 2) // 3
        qbs = [P.Batch.from_seqs(queries[i:i + third], ctx=ctx) for i in range(0, len(queries), third)]
        args = (C.c_float(pen), mc, mq, mt, span, int(gap is not None), int(gap or 0), int(ori))

        def sections(res, nq):
            def raw(ptr, nbytes):
                return C.string_at(C.cast(ptr, C.c_void_p), int(nbytes)) if nbytes else b""
            return (raw(res.q_off, 8 * (nq + 1)), raw(res.t_sid, 4 * res.n_targets), raw(res.t_off, 8 * (res.n_targets + 1) if res.n_targets else 0),
                    raw(res.c_score, 4 * res.n_chains), raw(res.c_off, 8 * (res.n_chains + 1) if res.n_chains else 0), raw(res.hps, 24 * res.n_hps))
        ref_s = []
        for b_ in qbs:
            res = _ffi.HpsResult()
            ctx.check(L.pgr_query_hps_resident(ctx.handle, sdb._ix, b_._h, *args, C.byref(res)))
            ref_s.append(sections(res, b_.n))
            L.pgr_hps_result_free(C.byref(res))
        hp = C.c_void_p()
        ctx.check(L.pgr_pipe_create(ctx.handle, C.byref(spec), C.byref(hp)))
        got_s, flying = [], []
        for b_ in qbs + qbs:
            if len(flying) == 2:
                res = _ffi.HpsResult()
                ctx.check(L.pgr_pipe_collect_query(hp, C.byref(res)))
                got_s.append(sections(res, flying.pop(0).n))
                L.pgr_hps_result_free(C.byref(res))
            ctx.check(L.pgr_pipe_submit_query(hp, b_._h, sdb._ix, *args))
            flying.append(b_)
        while flying:
            res = _ffi.HpsResult()
            ctx.check(L.pgr_pipe_collect_query(hp, C.byref(res)))
            got_s.append(sections(res, flying.pop(0).n))
            L.pgr_hps_result_free(C.byref(res))
        L.pgr_pipe_destroy(hp)
        if got_s != ref_s + ref_s:
            return "seed %d: query batches through the pipe differ from the synchronous call (spec %s pen %g counts %d/%d/%d span %d gap %s oriented %s)" % (
                seed, spec_t, pen, mc, mq, mt, span, gap, ori)
        if got2 != got:
            return "seed %d: the two query paths differ (spec %s pen %g counts %d/%d/%d span %d gap %s oriented %s)" % (
                seed, spec_t, pen, mc, mq, mt, span, gap, ori)
    n_chains = 0
    for qi, q in enumerate(queries):
        try:
            ref = oix.query_fragment_to_hps(q, pen, mc, mq, mt, span, gap, ori)
        except RuntimeError:
            continue  # the reference would not terminate on this input (cyclic predecessor map)
        ref = [(sid, [(sc, [((h[0], h[1], h[2]), (h[3], h[4], h[5])) for h in hps]) for sc, hps in chains]) for sid, chains in ref]
        if ref != got[qi]:
            return "seed %d: spec %s pen %g counts %d/%d/%d span %d gap %s oriented %s query %d (len %d)" % (
                seed, spec_t, pen, mc, mq, mt, span, gap, ori, qi, len(q))
        n_chains += sum(len(c) for _, c in ref)
    sdb.close()
    return None, n_chains


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = P.default_context(0)
    fails, chains = [], 0
    t0 = time.time()
    import signal

    class Stop(Exception):
        pass

    def on_term(*_):  # `timeout` ends the run: report what was done
        raise Stop()
    signal.signal(signal.SIGTERM, on_term)
    done = 0
    try:
      for it in range(iters):
        r = one_case(seed0 + it, ctx)
        done = it + 1
        if isinstance(r, str):
            fails.append(r)
            print("FAIL", r, flush=True)
        else:
            chains += r[1]
    except Stop:
        print("(stopped by SIGTERM after %d of %d cases)" % (done, iters))
        iters = done
    print("fuzz_query%s: %d cases (seeds %d..%d), %d chains compared, %d failures, %.0f s; batches by path: %d stage by stage, %d one "
          "wavefront per query behind the shimmer pipeline (%d of those enqueued behind it without a host wait), %d in the level-1 form" % (
              " short" if SHORT else "", iters, seed0, seed0 + iters - 1, chains, len(fails), time.time() - t0, PATHS[0],
              PATHS[1] + PATHS[2], PATHS[2], PATHS[3]))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
