#!/usr/bin/env python3
"""Randomised parity campaign (GPU box): random ShmmrSpec x adversarial sequence mix x ragged lengths, the whole
sequence_to_shmmrs output of the HIP path compared with the CPU oracle, bit exact.  Not part of the pytest suite (it is
open ended); prints the seed of every failing case.   usage: fuzz_parity.py [iterations] [seed0] [max_len] [general]
("general": the one-workgroup kernel for small batches is switched off -- with a small max_len the batches are batches of reads
and run the level-1 kernel on its one-wavefront tiles)"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402
import seqgen  # noqa: E402


def one_case(seed, max_len, ctx, pool):
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
    seqs = []
    for _ in range(n):
        L = int(np.exp(rng.uniform(0, np.log(max_len)))) if rng.random() < 0.9 else int(rng.integers(0, 3 * (w + k)))
        mode = int(rng.integers(0, seqgen.N_MODES)) if rng.random() < 0.6 else 0
        s = seqgen.adversarial(rng, mode, max(L, 0)) if L > 0 else b""
        if rng.random() < 0.15 and len(s) > 1000:  # a long N run / a long homopolymer inside
            a = int(rng.integers(0, len(s) - 500))
            ln = int(rng.integers(1, min(len(s) - a, 200000)))
            s = s[:a] + (b"N" if rng.random() < 0.5 else b"A") * ln + s[a + ln:]
        seqs.append(s)
    rids = None if rng.random() < 0.5 else [int(x) for x in rng.integers(0, 2 ** 31, n)]
    spec = P.make_spec(w, k, r, ms, sketch)
    got = P.sequence_to_shmmrs_batch(seqs, spec, rids=rids, padding=padding, ctx=ctx)
    osp = O.spec(w, k, r, ms, sketch)
    refs = list(pool.map(lambda i: O.sequence_to_shmmrs(i if rids is None else rids[i], seqs[i], osp, padding), range(n)))
    for i in range(n):
        if len(refs[i]) != len(got[i]) or not np.array_equal(refs[i]["x"], got[i]["x"]) or \
                not np.array_equal(refs[i]["y"], got[i]["y"]):
            return "seed %d: spec (%d,%d,%d,%d,%s) padding %s seq %d len %d: %d vs %d shimmers" % (
                seed, w, k, r, ms, sketch, padding, i, len(seqs[i]), len(refs[i]), len(got[i]))
    return None, sum(map(len, seqs))


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    max_len = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
    ctx = P.default_context(0)
    if len(sys.argv) > 4 and sys.argv[4] == "general":
        ctx.set_option("no_small_path", 1)
    fails, bases = [], 0
    t0 = time.time()
    import signal
    stop = []
    signal.signal(signal.SIGTERM, lambda *_: stop.append(1))  # `timeout` ends the run: finish the case, report what was done
    done = 0
    with ThreadPoolExecutor(16) as pool:
        log = open(os.environ["FUZZ_LOG"], "w") if os.environ.get("FUZZ_LOG") else None
        for it in range(iters):
            if stop:
                print("(stopped by SIGTERM after %d of %d cases)" % (done, iters))
                break
            done = it + 1
            if log:  # the last line names the case a crash happened in
                log.seek(0)
                log.write("%d\n" % (seed0 + it))
                log.flush()
            try:
                r = one_case(seed0 + it, max_len, ctx, pool)
            except Exception as e:  # an error code from the library is a failure of that case, not the end of the campaign
                r = "seed %d: %r" % (seed0 + it, e)
            if isinstance(r, str):
                fails.append(r)
                print("FAIL", r, flush=True)
            else:
                bases += r[1]
    iters = done
    print("fuzz_parity: %d cases (seeds %d..%d), %.2f Gbp, %d failures, %.0f s" % (iters, seed0, seed0 + iters - 1, bases / 1e9,
                                                                                  len(fails), time.time() - t0))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
