#!/usr/bin/env python3
"""One case of tools/fuzz_pipe.py again, with and without debug_poison, saying WHAT differs in the finalized index.
    python tools/fuzz_pipe_case.py SEED [max_len]"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402
import seqgen  # noqa: E402


def gen(seed, max_len):
    rng = np.random.default_rng(seed)
    if rng.random() < 0.5:
        w, k, r, ms = [(80, 56, 4, 64), (48, 56, 4, 12), (80, 56, 1, 64), (24, 24, 12, 24)][int(rng.integers(0, 4))]
    else:
        k, w, r, ms = int(rng.integers(2, 57)), int(rng.integers(2, 129)), int(rng.integers(1, 13)), int(rng.integers(0, 200))
    sketch = bool(rng.random() < 0.1)
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
    reserve = to_index and rng.random() < 0.5
    return (w, k, r, ms, sketch), sets, to_index, staged, reserve


def run(ctx, spec_t, sets, staged, reserve, depth=2):
    spec = P.make_spec(*spec_t)
    batches = [P.Batch.from_seqs(s, ctx=ctx) for s in sets]
    ctx.set_option("pipe_staged_records", 1 if staged else 0)
    pipe = P.Pipe(spec, ctx=ctx)
    ix = P.Index(spec, ctx=ctx)
    if reserve:
        ix.reserve(int(sum(len(q) for s in sets for q in s) * 0.02) + 1024)
    got = []
    for b in batches:
        if pipe.in_flight == depth:
            got.append(pipe.collect())
        pipe.submit(b, index=ix)
    while pipe.in_flight:
        got.append(pipe.collect())
    pipe.close()
    ctx.set_option("pipe_staged_records", 0)
    npairs = [n for _, n in got]
    ix.finalize()
    return ix.download(), npairs


def main():
    seed = int(sys.argv[1])
    max_len = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    spec_t, sets, to_index, staged, reserve = gen(seed, max_len)
    print("seed %d: spec %s, %d batches of %s sequences (%s bases), to_index %s staged %s reserve %s" % (
        seed, spec_t, len(sets), [len(s) for s in sets], [sum(map(len, s)) for s in sets], to_index, staged, reserve))
    osp = O.spec(*spec_t)
    with ThreadPoolExecutor(16) as pool:
        refs = [list(pool.map(lambda i, s=s: O.sequence_to_shmmrs(i, s[i], osp), range(len(s)))) for s in sets]
    exp, sid0 = [], 0
    for bi, s in enumerate(sets):
        exp += [O.frag_recs(ref, sid0 + i) for i, ref in enumerate(refs[bi])]
        sid0 += len(s)
    exp = np.concatenate(exp) if exp else np.zeros(0, dtype=P.FRAG_REC)
    exp = exp[np.lexsort((exp["frg_id"], exp["sid"], exp["h1"], exp["h0"]))]
    ctx = P.default_context(0)
    for name, opts, depth in (("plain", {}, 2), ("poison", {"debug_poison": 1}, 2), ("poison, one job in flight", {"debug_poison": 1}, 1), ("poison again", {"debug_poison": 1}, 2)):
        with ctx.options(**opts):
            rec, npairs = run(ctx, spec_t, sets, staged, reserve, depth)
        bad = {f: int((exp[f] != rec[f]).sum()) if len(exp) == len(rec) else -1 for f in ("h0", "h1", "frg_id", "sid", "bgn", "end", "orient")}
        print("%-28s %d records (oracle %d), pairs per job %s; differing fields %s" % (name, len(rec), len(exp), npairs, {k: v for k, v in bad.items() if v}))
        if len(exp) == len(rec) and any(bad.values()):
            w = np.where((exp["sid"] != rec["sid"]) | (exp["bgn"] != rec["bgn"]) | (exp["h0"] != rec["h0"]) | (exp["frg_id"] != rec["frg_id"]))[0]
            print("   first differing rows: %s" % [(int(i), int(exp["sid"][i]), int(rec["sid"][i]), int(exp["frg_id"][i]), int(rec["frg_id"][i]), int(exp["bgn"][i]), int(rec["bgn"][i])) for i in w[:6]], "(row, sid exp/got, frg_id exp/got, bgn exp/got)")
            print("   sids with differences: %s" % sorted(set(exp["sid"][w].tolist()))[:20])


if __name__ == "__main__":
    main()
