#!/usr/bin/env python3
"""Repeat-rich contigs the way assemblies have them: a 171-bp satellite monomer repeated over Mbp with 1-2 % divergence, exact
microsatellites ((AT)n, (CAG)n, period 2-6, 1-50 kbp), homopolymer runs, an exact tandem duplication of a 10 kbp unit; 60 Mbp
in 3 contigs.  Times pgr_shmmrs_compute and compares the whole result with the CPU restatement (ties emit every position,
palindromic k-mers are skipped: the dense-tile and island paths)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle")]
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import pgrtk_amd as P  # noqa: E402

rng = np.random.default_rng(7)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rnd(n):
    return rng.choice(ACGT, int(n))


def satellite(total, unit_len=171, div=0.015):
    unit = rnd(unit_len)
    s = np.tile(unit, total // unit_len + 1)[:total].copy()
    m = rng.random(total) < div
    s[m] = rng.choice(ACGT, int(m.sum()))
    return s


def contig(L):
    parts, n = [], 0
    while n < L:
        r = rng.random()
        if r < 0.35:
            p = rnd(rng.integers(50_000, 2_000_000))
        elif r < 0.6:
            p = satellite(int(rng.integers(200_000, 3_000_000)))
        elif r < 0.8:
            u = rnd(rng.integers(2, 7))
            p = np.tile(u, int(rng.integers(1_000, 50_000)) // len(u) + 1)
        elif r < 0.85:  # reverse-complement symmetric units: every k-mer is its own reverse complement (skipped pushes)
            u = np.frombuffer([b"AT", b"CG", b"ACGT", b"AATT", b"GAATTC"][int(rng.integers(0, 5))], dtype=np.uint8)
            p = np.tile(u, int(rng.integers(1_000, 200_000)) // len(u) + 1)
        elif r < 0.9:
            p = np.full(int(rng.integers(100, 20_000)), ACGT[int(rng.integers(0, 4))], dtype=np.uint8)
        else:
            u = rnd(10_000)
            p = np.tile(u, int(rng.integers(3, 40)))
        parts.append(p)
        n += len(p)
    return np.concatenate(parts)[:L]


seqs = [contig(30_000_000), contig(20_000_000), contig(10_000_000)]
ctx = P.default_context(0)
b = P.Batch.from_seqs(seqs, ctx=ctx)
sp = P.make_spec()
sh = b.shmmrs(sp)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    sh = b.shmmrs(sp)
    ts.append(time.perf_counter() - t0)
p = ctx.last_prof()
print("GPU: %.3f ms (%s), %d shimmers from %d level-1 minimizers, %.1f Mbp through the exact islands" %
      (min(ts) * 1e3, " ".join("%.3f" % (t * 1e3) for t in ts), sh.count, p.n_level1, p.exact_bases / 1e6))
with ctx.options(no_island_relay=1):  # A/B: the round-3 seam correction (one seam per host round)
    b.shmmrs(sp)
    ts3 = []
    for _ in range(3):
        t0 = time.perf_counter()
        b.shmmrs(sp)
        ts3.append(time.perf_counter() - t0)
print("     with the round-3 seam correction (option no_island_relay): %.1f ms" % (min(ts3) * 1e3))
for a in sys.argv[1:]:  # --opt name=value: the call with a context option against without, in alternation, best of 8
    if a.startswith("--opt="):
        k, v = a[6:].split("=")
        ta, tb = [], []
        for rep in range(9):
            with ctx.options(**{k: int(v)}):
                t0 = time.perf_counter()
                sh2 = b.shmmrs(sp)
                if rep:
                    tb.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            b.shmmrs(sp)
            if rep:
                ta.append(time.perf_counter() - t0)
        print("     with %s: %.3f ms against %.3f without (alternating, best of 8); same count: %s" % (a[6:], min(tb) * 1e3, min(ta) * 1e3, sh2.count == sh.count))
        if "--times" in sys.argv:
            for o in ({k: int(v)}, {}):
                print("--- laps with %s" % (o or "the defaults"), file=sys.stderr, flush=True)
                with ctx.options(debug_times=1, **o):
                    b.shmmrs(sp)
if "--rounds" in sys.argv:  # the rounds of the island path on stderr
    sys.stderr.flush()
    with ctx.options(debug=1):
        b.shmmrs(sp)
    with ctx.options(debug=1, no_island_relay=1):
        print("--- option no_island_relay", file=sys.stderr, flush=True)
        b.shmmrs(sp)
sums, off = sh.checksum(), sh.offsets()
ok = True
t0 = time.perf_counter()
for i, s in enumerate(seqs):
    ref = O.sequence_to_shmmrs(i, s, O.spec())
    same = int(off[i + 1] - off[i]) == len(ref) and np.array_equal(sums[i], O.shmmr_checksum(ref))
    ok = ok and same
    print("  contig %d: %d bp, %d shimmers, identical: %s" % (i, len(s), len(ref), same))
print("CPU restatement, one thread: %.2f s" % (time.perf_counter() - t0))
sys.exit(0 if ok else 1)
