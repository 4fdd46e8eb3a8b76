#!/usr/bin/env python3
"""Randomised campaign for the software pipeline (pgr_pipe_*, GPU box): per case a random ShmmrSpec and 3-6 batches of the adversarial
sequence mix go through ONE pipe with two jobs in flight -- pair records into an index (running sids, direct placement by the
device cursor; staged now and then: option pipe_staged_records) or into a caller's buffer; shimmer lists and the finalized index
are compared with the CPU oracle, bit exact.  Flagged batches (non-ACGT bytes, palindromic k-mers) finish synchronously at
collect; undersized estimates repeat stages.   usage: fuzz_pipe.py [iterations] [seed0] [max_len]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402
import seqgen  # noqa: E402


def one_case(seed, max_len, ctx, pool):
    rng = np.random.default_rng(seed)
    if rng.random() < 0.5:
        w, k, r, ms = [(80, 56, 4, 64), (48, 56, 4, 12), (80, 56, 1, 64), (24, 24, 12, 24)][int(rng.integers(0, 4))]
    else:
        k, w, r, ms = int(rng.integers(2, 57)), int(rng.integers(2, 129)), int(rng.integers(1, 13)), int(rng.integers(0, 200))
    sketch = bool(rng.random() < 0.1)
    spec, osp = P.make_spec(w, k, r, ms, sketch), O.spec(w, k, r, ms, sketch)
    sets = []
    for _ in range(int(rng.integers(3, 7))):
        seqs = []
        for _ in range(int(rng.integers(1, 12))):
            L = int(np.exp(rng.uniform(0, np.log(max_len)))) if rng.random() < 0.9 else int(rng.integers(0, 3 * (w + k)))
            mode = int(rng.integers(0, seqgen.N_MODES)) if rng.random() < 0.4 else 0
            s = seqgen.adversarial(rng, mode, max(L, 0)) if L > 0 else b""
            if rng.random() < 0.1 and len(s) > 1000:
                a = int(rng.integers(0, len(s) - 500))
                ln = int(rng.integers(1, min(len(s) - a, 100000)))
                s = s[:a] + (b"N" if rng.random() < 0.5 else b"A") * ln + s[a + ln:]
            seqs.append(s)
        sets.append(seqs)
    to_index = bool(rng.random() < 0.6)
    staged = bool(rng.random() < 0.25)
    batches = [P.Batch.from_seqs(s, ctx=ctx) for s in sets]
    ctx.set_option("pipe_staged_records", 1 if staged else 0)
    pipe = P.Pipe(spec, ctx=ctx)
    ix = P.Index(spec, ctx=ctx) if to_index else None
    if ix is not None and rng.random() < 0.5:
        ix.reserve(int(sum(len(q) for s in sets for q in s) * 0.02) + 1024)
    # (a caller's buffer that is too small is an error of the CALLER -- seed 9001412 of round 6's campaign: "output buffer too small for
    # the pair records" from a dense spec and a buffer of a pair per 8 bases --: a pair per base + slack)
    bufs = [torch.zeros((sum(len(q) for q in s) + 64, 5), dtype=torch.int64, device="cuda:0") for s in sets] if not to_index else None
    got = []
    sid0 = 0
    sids = []
    for bi, b in enumerate(batches):
        if pipe.in_flight == 2:
            got.append(pipe.collect())
        if to_index:
            pipe.submit(b, index=ix)
        else:
            pipe.submit(b, sids=list(range(sid0, sid0 + b.n)), rec_ptr=bufs[bi].data_ptr(), rec_capacity=bufs[bi].shape[0])
        sids.append(list(range(sid0, sid0 + b.n)))
        sid0 += b.n
    while pipe.in_flight:
        got.append(pipe.collect())
    pipe.close()
    ctx.set_option("pipe_staged_records", 0)
    refs = [list(pool.map(lambda i, s=s: O.sequence_to_shmmrs(i, s[i], osp), range(len(s)))) for s in sets]
    tag = "seed %d: spec (%d,%d,%d,%d,%s) %s%s" % (seed, w, k, r, ms, sketch, "index" if to_index else "buffer", ", staged" if staged else "")
    all_recs = []
    for bi, (sh, n_pairs) in enumerate(got):
        mm, off = sh.download()
        for i, ref in enumerate(refs[bi]):
            g = mm[int(off[i]):int(off[i + 1])]
            if len(ref) != len(g) or not np.array_equal(ref["x"], g["x"]) or not np.array_equal(ref["y"], g["y"]):
                return "%s batch %d seq %d len %d: %d vs %d shimmers" % (tag, bi, i, len(sets[bi][i]), len(ref), len(g))
        exp = [O.frag_recs(ref, sids[bi][i]) for i, ref in enumerate(refs[bi])]
        exp = np.concatenate(exp) if exp else np.zeros(0, dtype=P.FRAG_REC)
        all_recs.append(exp)
        if n_pairs != len(exp):
            return "%s batch %d: %d vs %d pair records" % (tag, bi, n_pairs, len(exp))
        if not to_index and n_pairs:
            r2 = bufs[bi][:n_pairs].cpu().numpy().view(P.FRAG_REC).reshape(-1)
            if any(not np.array_equal(exp[f], r2[f]) for f in ("h0", "h1", "frg_id", "sid", "bgn", "end", "orient")):
                return "%s batch %d: pair records differ" % (tag, bi)
    if to_index:
        ix.finalize()
        rec = ix.download()
        exp = np.concatenate(all_recs) if all_recs else np.zeros(0, dtype=P.FRAG_REC)
        order = np.lexsort((exp["frg_id"], exp["sid"], exp["h1"], exp["h0"]))
        exp = exp[order]
        if len(rec) != len(exp) or any(not np.array_equal(exp[f], rec[f]) for f in ("h0", "h1", "frg_id", "sid", "bgn", "end", "orient")):
            what = ""
            if len(rec) == len(exp):  # which fields, which rows, what stands there instead
                bad = {f: int((exp[f] != rec[f]).sum()) for f in ("h0", "h1", "frg_id", "sid", "bgn", "end", "orient") if (exp[f] != rec[f]).any()}
                rows = np.where(np.any([exp[f] != rec[f] for f in ("h0", "h1", "frg_id", "sid", "bgn", "end", "orient")], axis=0))[0]
                what = "; fields %s; rows %s; got %s expected %s; pairs per job %s" % (
                    bad, rows[:8].tolist(), [tuple(int(rec[f][i]) for f in ("h0", "h1", "sid", "frg_id", "bgn", "end")) for i in rows[:3]],
                    [tuple(int(exp[f][i]) for f in ("h0", "h1", "sid", "frg_id", "bgn", "end")) for i in rows[:3]], [n for _, n in got])
            return "%s: the finalized index differs (%d vs %d records)%s" % (tag, len(rec), len(exp), what)
    return None, sum(len(q) for s in sets for q in s)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    max_len = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
    ctx = P.default_context(0)
    fails, bases, t0 = [], 0, time.time()
    import signal
    stop = []
    signal.signal(signal.SIGTERM, lambda *_: stop.append(1))
    done = 0
    with ThreadPoolExecutor(16) as pool:
        for it in range(iters):
            if stop:
                print("(stopped by SIGTERM after %d of %d cases)" % (done, iters))
                break
            done = it + 1
            try:
                r = one_case(seed0 + it, max_len, ctx, pool)
            except Exception as e:  # noqa: BLE001
                r = "seed %d: %r" % (seed0 + it, e)
            if isinstance(r, str):
                fails.append(r)
                print("FAIL", r, flush=True)
            else:
                bases += r[1]
    print("fuzz_pipe: %d cases (seeds %d..%d), %.2f Gbp, %d failures, %.0f s" % (done, seed0, seed0 + done - 1, bases / 1e9, len(fails), time.time() - t0))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
