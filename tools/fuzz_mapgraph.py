#!/usr/bin/env python3
"""Randomised parity campaign for the MAP-graph path (GPU box): random small pangenomes with structural variation,
random min_count / path_len_cutoff / keeps; adjacency list, weighted DFS, principal bundles, bundles-with-id, the
per-sequence decomposition and the .bed body compared product <-> oracle (tests/test_gpu_08_mapgraph._check_all).
usage: fuzz_mapgraph.py [iterations] [seed0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402
import seqgen  # noqa: E402
from test_gpu_08_mapgraph import _check_all  # noqa: E402

COMP = bytes.maketrans(b"ACGT", b"TGCA")


def pangenome(rng):
    anc = seqgen.rnd(rng, int(rng.integers(5000, 50000)))
    haps = []
    for _ in range(int(rng.integers(2, 12))):
        s = bytearray(anc)
        for _ in range(int(rng.integers(0, 6))):
            if len(s) < 3000:
                break
            a = int(rng.integers(0, len(s) - 2000))
            ln = int(rng.integers(100, min(8000, len(s) - a)))
            seg = bytes(s[a:a + ln])
            op = int(rng.integers(0, 5))
            if op == 0:
                s[a:a + ln] = seg.translate(COMP)[::-1]
            elif op == 1:
                del s[a:a + ln]
            elif op == 2:
                s[a:a] = seg * int(rng.integers(1, 5))
            elif op == 3:
                s[a:a] = seqgen.rnd(rng, ln)
            else:  # translocation
                b = int(rng.integers(0, len(s)))
                s[b:b] = seg
        h = bytes(s)
        if rng.random() < 0.2:
            h = h.translate(COMP)[::-1]
        haps.append(h)
    if rng.random() < 0.2:
        haps.append(haps[0])
    return haps


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    O.build()
    ctx = P.default_context(0)
    fails, bundles = [], 0
    t0 = time.time()
    for it in range(iters):
        seed = seed0 + it
        rng = np.random.default_rng(seed)
        haps = pangenome(rng)
        spec_t = [(24, 24, 2, 8), (16, 12, 2, 4), (31, 21, 3, 8), (48, 56, 4, 12)][int(rng.integers(0, 4))]
        mc = int(rng.integers(0, len(haps) + 2))
        cutoff = int(rng.integers(0, 10))
        keeps = None if rng.random() < 0.6 else [int(x) for x in rng.integers(0, len(haps), int(rng.integers(0, 4)))]
        try:
            bundles += _check_all(O, ctx, haps, spec_t, mc, cutoff, keeps, bed_args=(int(rng.integers(0, 3000)), int(rng.integers(0, 10000))))
        except AssertionError as e:
            msg = "seed %d: spec %s min_count %d cutoff %d keeps %s: %s" % (seed, spec_t, mc, cutoff, keeps, str(e)[:200])
            fails.append(msg)
            print("FAIL", msg, flush=True)
    print("fuzz_mapgraph: %d cases (seeds %d..%d), %d bundles compared, %d failures, %.0f s" % (iters, seed0, seed0 + iters - 1, bundles,
                                                                                             len(fails), time.time() - t0))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
