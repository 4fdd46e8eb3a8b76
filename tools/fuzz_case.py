#!/usr/bin/env python3
"""One case of tools/fuzz_parity.py again, with what is needed to find out why it differs: the sequences of seed S regenerated, the whole
batch and the failing sequence ALONE through the library (default options, no_small_path, the early look forced, poison), every result
against the oracle -- where the lists differ (positions), what the sequence looks like there.
    python tools/fuzz_case.py SEED [max_len]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402
import seqgen  # noqa: E402


def gen(seed, max_len):
    rng = np.random.default_rng(seed)
    if rng.random() < 0.4:
        w, k, r, ms = [(80, 56, 4, 64), (48, 56, 4, 12), (80, 56, 1, 64), (24, 24, 12, 24)][int(rng.integers(0, 4))]
    else:
        k = int(rng.integers(2, 57))
        w = int(rng.integers(2, 129))
        r = int(rng.integers(1, 13))
        ms = int(rng.integers(0, 200))
    sketch = bool(rng.random() < 0.1)
    padding = bool(rng.random() < 0.3)
    n = int(rng.integers(1, 24))
    seqs, modes = [], []
    for _ in range(n):
        L = int(np.exp(rng.uniform(0, np.log(max_len)))) if rng.random() < 0.9 else int(rng.integers(0, 3 * (w + k)))
        mode = int(rng.integers(0, seqgen.N_MODES)) if rng.random() < 0.6 else 0
        s = seqgen.adversarial(rng, mode, max(L, 0)) if L > 0 else b""
        ins = None
        if rng.random() < 0.15 and len(s) > 1000:
            a = int(rng.integers(0, len(s) - 500))
            ln = int(rng.integers(1, min(len(s) - a, 200000)))
            ch = b"N" if rng.random() < 0.5 else b"A"
            s = s[:a] + ch * ln + s[a + ln:]
            ins = (a, ln, ch)
        seqs.append(s)
        modes.append((mode, ins))
    rids = None if rng.random() < 0.5 else [int(x) for x in rng.integers(0, 2 ** 31, n)]
    return (w, k, r, ms, sketch), padding, seqs, rids, modes


def diff(ref, got):
    rp = (ref["y"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)
    gp = (got["y"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)
    only_ref = sorted(set(rp.tolist()) - set(gp.tolist()))
    only_got = sorted(set(gp.tolist()) - set(rp.tolist()))
    return only_ref, only_got


def main():
    seed = int(sys.argv[1])
    max_len = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
    spec_t, padding, seqs, rids, modes = gen(seed, max_len)
    w, k, r, ms, sketch = spec_t
    print("seed %d: spec %s padding %s, %d sequences, rids %s" % (seed, spec_t, padding, len(seqs), "given" if rids else "none"))
    ctx = P.default_context(0)
    spec, osp = P.make_spec(*spec_t), O.spec(*spec_t)
    refs = [O.sequence_to_shmmrs(i if rids is None else rids[i], seqs[i], osp, padding) for i in range(len(seqs))]
    bad = []
    for name, opts in (("default", {}), ("no_small_path", {"no_small_path": 1}), ("early look forced", {"early_sync_bp": 0}),
                       ("poison", {"debug_poison": 1})):
        with ctx.options(**opts):
            got = P.sequence_to_shmmrs_batch(seqs, spec, rids=rids, padding=padding, ctx=ctx)
        wrong = [i for i in range(len(seqs)) if len(refs[i]) != len(got[i]) or not np.array_equal(refs[i]["x"], got[i]["x"]) or not np.array_equal(refs[i]["y"], got[i]["y"])]
        print("  whole batch, %-18s: %d sequences differ %s" % (name, len(wrong), wrong))
        if name == "default":
            bad = wrong
            got_default = got
    for i in bad:
        s = seqs[i]
        only_ref, only_got = diff(refs[i], got_default[i])
        print("sequence %d: len %d, generator mode %s, inserted run %s; %d (oracle) vs %d shimmers" % (i, len(s), modes[i][0], modes[i][1], len(refs[i]), len(got_default[i])))
        print("   positions only in the oracle's list: %s" % only_ref[:20])
        print("   positions only in the library's list: %s" % only_got[:20])
        lo = max(0, min(only_ref + only_got) - 200) if (only_ref or only_got) else 0
        hi = min(len(s), (max(only_ref + only_got) if (only_ref or only_got) else 0) + 200)
        seg = s[lo:hi]
        print("   sequence[%d:%d] non-ACGT %d, lower case %d; first 300 bytes: %r" % (lo, hi, sum(1 for c in seg if c not in b"ACGTacgt"), sum(1 for c in seg if c in b"acgt"), seg[:300]))
        for name, opts in (("default", {}), ("no_small_path", {"no_small_path": 1}), ("early look forced", {"early_sync_bp": 0})):
            with ctx.options(**opts):
                g1 = P.sequence_to_shmmrs_batch([s], spec, rids=None if rids is None else [rids[i]], padding=padding, ctx=ctx)[0]
            same = len(g1) == len(refs[i]) and np.array_equal(g1["x"], refs[i]["x"]) and np.array_equal(g1["y"], refs[i]["y"])
            print("   ALONE, %-18s: %s (%d shimmers)" % (name, "== oracle" if same else "DIFFERS", len(g1)))
        # where does it start to differ: the level-1 list (r = 1, min_span 0: no reduction, the span filter only drops equal neighbours)
        sp1, osp1 = P.make_spec(w, k, 1, 0, sketch), O.spec(w, k, 1, 0, sketch)
        l1g = P.sequence_to_shmmrs_batch([s], sp1, rids=None, padding=False, ctx=ctx)[0]
        l1r = O.sequence_to_shmmrs(0, s, osp1, False)
        a, bb = diff(l1r, l1g)
        print("   level-1 lists (r = 1, min_span 0) ALONE: oracle %d, library %d; only oracle %s, only library %s" % (len(l1r), len(l1g), a[:20], bb[:20]))
        near = [int(p_) for p_ in ((l1r["y"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)) if lo <= p_ <= hi]
        print("   oracle level-1 positions in [%d, %d]: %s" % (lo, hi, near))
        for rr in (2, 3, 4):
            spx, ospx = P.make_spec(w, k, rr, ms, sketch), O.spec(w, k, rr, ms, sketch)
            gx = P.sequence_to_shmmrs_batch([s], spx, rids=None, padding=padding, ctx=ctx)[0]
            rx = O.sequence_to_shmmrs(0, s, ospx, padding)
            a, bb = diff(rx, gx)
            print("   r = %d ALONE: oracle %d, library %d; only oracle %s, only library %s" % (rr, len(rx), len(gx), a[:10], bb[:10]))
        prof = ctx.last_prof()
        print("   last prof: tiles %d, serial contigs %d, islands bases %d" % (prof.n_tiles, prof.n_serial_contigs, getattr(prof, "bases_serial", -1)))


if __name__ == "__main__":
    main()
