#!/usr/bin/env python3
"""A genome-like batch at full batch size: a dozen chromosome-sized contigs (~2 Gbp) carrying what a reference assembly carries --
a centromere (a gap of several Mbp of N beside alpha-satellite arrays of a 171-bp unit with 2 % divergence), tens of shorter gaps,
telomeres ((TTAGGG)n behind 10 kbp of N), isolated N, soft-masked (lower-case) interspersed repeats over half of the sequence,
microsatellites at ~1 per 8 kbp with a heavy-tailed length (so that a few hundred per chromosome are (AT)n / (TA)n / (CG)n /
(ACGT)n arrays longer than k: every k-mer inside is its own reverse complement and the push is skipped, shmmrutils.rs:477-480),
poly-A tails, tandem and segmental duplications.

The shapes of bench.py (one 248 Mbp contig, 60 Mbp of repeats) are latency cases: a handful of host rounds on a batch that takes
half a millisecond.  This is the THROUGHPUT case for real input: what sequence_to_shmmrs costs per base when a batch has thousands of
islands.  Times pgr_shmmrs_compute on the resident batch, reports the islands' share, and compares every contig with the CPU
restatement (128-bit checksums, oracle on --threads threads).

  python tools/genome_like_bench.py [--scale 1.0] [--threads 16] [--rounds] [--json out.json] [--no-check]"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle")]
import numpy as np  # noqa: E402

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
CHROM_MBP = (248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133)  # chr1..chr12 of a human assembly, rounded


def tile_unit(u, n):
    return np.tile(u, n // len(u) + 1)[:n]


def genome_like_contig(O, c, L, seed=97):
    """one chromosome-like contig; returns (ascii bytes, what was planted)"""
    rng = np.random.default_rng(seed * 1000 + c)
    s = O.synth_contig(seed, c, L).copy()
    n = {"microsatellites": 0, "palindromic_arrays_over_k": 0, "gaps": 0, "satellite_bp": 0}
    # segmental duplications (10 - 200 kbp blocks copied elsewhere, 1 % divergence) and tandem duplications
    for _ in range(max(1, L // 8_000_000)):
        ln = min(int(rng.integers(10_000, 200_000)), L // 10)
        a, b = int(rng.integers(0, L - ln)), int(rng.integers(0, L - ln))
        blk = s[a:a + ln].copy()
        m = rng.random(ln) < 0.01
        blk[m] = rng.choice(ACGT, int(m.sum()))
        s[b:b + ln] = blk
    for _ in range(max(1, L // 20_000_000)):
        u = min(int(rng.integers(2_000, 12_000)), L // 100)
        a = int(rng.integers(0, L - 40 * u))
        s[a:a + 20 * u] = tile_unit(s[a:a + u].copy(), 20 * u)
    # microsatellites: ~1 per 8 kbp, unit of 1-6 bases, length 12 + exponential (mean 14) with a heavy tail (2 %: up to 2 kbp)
    n_ms = L // 8_000
    pos = rng.integers(1_000, L - 4_000, n_ms)
    ln = (12 + rng.exponential(14.0, n_ms)).astype(np.int64)
    tail = rng.random(n_ms) < 0.02
    ln[tail] = rng.integers(60, 2_000, int(tail.sum()))
    kind = rng.random(n_ms)
    pal_units = [b"AT", b"TA", b"CG", b"ACGT", b"AATT", b"TTAA", b"GAATTC"]
    for p, l, k in zip(pos.tolist(), ln.tolist(), kind.tolist()):
        if k < 0.30:
            u = ACGT[[0]] if k < 0.2 else ACGT[[3]]  # poly-A / poly-T
        elif k < 0.55:
            u = np.frombuffer([b"AC", b"GT", b"AG", b"CT"][int(k * 1e4) % 4], dtype=np.uint8)
        elif k < 0.70:
            u = np.frombuffer(pal_units[int(k * 1e4) % len(pal_units)], dtype=np.uint8)
            if l >= 56 + len(u):
                n["palindromic_arrays_over_k"] += 1
        else:
            u = rng.choice(ACGT, int(rng.integers(3, 7)))
        s[p:p + l] = tile_unit(u, l)
    n["microsatellites"] = int(n_ms)
    # centromere: satellite arrays (171-bp unit, 2 % divergence, a 12-unit higher-order repeat) on both sides of a gap
    cen = int(L * (0.35 + 0.2 * rng.random()))
    gap = min(int(rng.integers(1_000_000, 6_000_000)) if c else 18_000_000, L // 8)  # (the clips: scaled-down contigs of the tests)
    for side in (0, 1):
        total = min(int(rng.integers(500_000, 2_500_000)), L // 12)
        u0 = rng.choice(ACGT, 171)
        hor = np.concatenate([np.where(rng.random(171) < 0.2, rng.choice(ACGT, 171), u0) for _ in range(12)])
        arr = tile_unit(hor, total).copy()
        m = rng.random(total) < 0.02
        arr[m] = rng.choice(ACGT, int(m.sum()))
        a = cen - total if side == 0 else cen + gap
        a = max(0, min(L - total, a))
        s[a:a + total] = arr
        n["satellite_bp"] += total
    s[cen:min(L, cen + gap)] = ord("N")
    # other gaps, telomeres, isolated N
    n_gaps = int(rng.integers(15, 45))
    for _ in range(n_gaps):
        a = int(rng.integers(0, max(1, L - 1_100_000)))
        s[a:a + min(int(rng.integers(1_000, 1_000_000) if rng.random() < 0.3 else rng.integers(100, 50_000)), L // 40)] = ord("N")
    n["gaps"] = n_gaps + 3
    tel = np.frombuffer(b"TTAGGG", dtype=np.uint8)
    s[:10_000] = ord("N")
    s[10_000:18_000] = tile_unit(np.frombuffer(b"CCCTAA", dtype=np.uint8), 8_000)
    s[L - 10_000:] = ord("N")
    s[L - 18_000:L - 10_000] = tile_unit(tel, 8_000)
    for _ in range(L // 1_000_000):
        s[int(rng.integers(0, L))] = ord("N")
    # soft masking: interspersed repeats (300 bp - 6 kbp) over ~half of the sequence; the reference's table maps both cases alike
    n_mask = L // 4_000
    a = rng.integers(0, L - 6_000, n_mask)
    ln = np.where(rng.random(n_mask) < 0.7, rng.integers(280, 320, n_mask), rng.integers(500, 6_000, n_mask))
    for p, l in zip(a.tolist(), ln.tolist()):
        s[p:p + l] |= 0x20
    return s, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="contig lengths x this")
    ap.add_argument("--contigs", type=int, default=len(CHROM_MBP))
    ap.add_argument("--threads", type=int, default=min(16, len(os.sched_getaffinity(0))))
    ap.add_argument("--rounds", action="store_true", help="one more pass with the call's lap times on stderr")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--json", default=None)
    ap.add_argument("--opt", action="append", default=[], help="context option name=value for an A/B pass")
    ap.add_argument("--pipe", type=int, default=0, help="K > 0: the same batch K times through pgr_pipe_* (two jobs in flight), lists only")
    ap.add_argument("--pipe-index", action="store_true", help="... with the pair records into an index (staged once a job is flagged)")
    args = ap.parse_args()
    import oracle as O
    import pgrtk_amd as P
    lens = [int(m * 1_000_000 * args.scale) for m in CHROM_MBP[:args.contigs]]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(args.threads) as ex:
        made = list(ex.map(lambda cl: genome_like_contig(O, cl[0], cl[1]), enumerate(lens)))
    seqs = [m[0] for m in made]
    planted = {k: int(sum(m[1][k] for m in made)) for k in made[0][1]}
    bp = int(sum(lens))
    n_N = int(sum(int((q == ord("N")).sum()) for q in seqs))
    print("generated %d contigs, %.3f Gbp in %.1f s: %s, %d N" % (len(seqs), bp / 1e9, time.perf_counter() - t0, planted, n_N), flush=True)
    ctx = P.default_context(0)
    t0 = time.perf_counter()
    b = P.Batch.from_seqs(seqs, ctx=ctx)
    t_in = time.perf_counter() - t0
    sp = P.make_spec()
    sh = b.shmmrs(sp)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        sh = b.shmmrs(sp)
        ts.append(time.perf_counter() - t0)
    p = ctx.last_prof()
    out = {"contigs": len(seqs), "bp": bp, "planted": planted, "non_acgt_bytes": n_N, "ms": min(ts) * 1e3, "ms_reps": [t * 1e3 for t in ts],
           "Gbp_per_s": bp / min(ts) / 1e9, "shimmers": int(sh.count), "level1_minimizers": int(p.n_level1),
           "Mbp_through_exact_islands": p.exact_bases / 1e6, "level1_tile_ms": p.level1_ms, "level1_aux_ms": p.level1_aux_ms,
           "level2_ms": p.level2_ms, "host_ascii_to_resident_s": t_in}
    print("GPU: %.2f ms (%s) = %.1f Gbp/s; %d shimmers; %.1f Mbp through the exact islands; tile kernel %.2f ms, aux %.2f, list stage %.2f"
          % (out["ms"], " ".join("%.2f" % t for t in out["ms_reps"]), out["Gbp_per_s"], sh.count, out["Mbp_through_exact_islands"],
             p.level1_ms, p.level1_aux_ms, p.level2_ms), flush=True)
    for kv in args.opt:
        k, v = kv.split("=")
        # A/B in alternation (the first call of either kind sizes workspaces): best of 6 each
        ta, tb, same = [], [], True
        for rep in range(7):
            with ctx.options(**{k: int(v)}):
                t0 = time.perf_counter()
                sh2 = b.shmmrs(sp)
                if rep:
                    tb.append(time.perf_counter() - t0)
                same = same and sh2.count == sh.count and bool(np.array_equal(sh2.checksum(), sh.checksum()))
                pb = ctx.last_prof()
                exact_b = pb.exact_bases / 1e6
            t0 = time.perf_counter()
            b.shmmrs(sp)
            if rep:
                ta.append(time.perf_counter() - t0)
        print("  with %s: %.2f ms (%.1f Mbp through the exact islands) against %.2f ms without, alternating, best of 6; same result: %s"
              % (kv, min(tb) * 1e3, exact_b, min(ta) * 1e3, same))
        out.setdefault("options", {})[kv] = {"ms": min(tb) * 1e3, "default_ms": min(ta) * 1e3, "Mbp_through_exact_islands": exact_b, "same_result": bool(same)}
    if args.pipe:
        sync_sum = sh.checksum()

        def pipe_pass(k, opts):
            with ctx.options(**opts):
                pipe = P.Pipe(sp, ctx=ctx)
                ix = P.Index(sp, ctx=ctx) if args.pipe_index else None
                same, got = True, []
                t0 = time.perf_counter()
                for i in range(k):
                    if pipe.in_flight == 2:
                        got.append(pipe.collect())
                    pipe.submit(b, index=ix)
                while pipe.in_flight:
                    got.append(pipe.collect())
                ctx.synchronize()
                t = time.perf_counter() - t0
                for r, _ in got[-2:]:
                    same = same and r.count == sh.count and bool(np.array_equal(r.checksum(), sync_sum))
                n_rec = None
                if ix is not None:
                    ix.finalize()
                    n_rec = ix.n_records
                pipe.close()
            return t, same, n_rec
        pipe_pass(3, {})
        for name, opts in (("fix stream, stage 1 only", {}), ("fix stream, list stage twice", {"no_stage1_only": 1}),
                           ("no_fix_stream", {"no_fix_stream": 1, "no_stage1_only": 1})):
            t, same, n_rec = pipe_pass(args.pipe, opts)
            print("pipe (%s): %d batches in %.2f ms = %.2f ms per batch = %.1f Gbp/s; last two jobs identical to the synchronous call: %s%s"
                  % (name, args.pipe, t * 1e3, t * 1e3 / args.pipe, bp * args.pipe / t / 1e9, same,
                     "" if n_rec is None else "; index records %d" % n_rec), flush=True)
            out.setdefault("pipe", {})[name] = {"batches": args.pipe, "ms_per_batch": t * 1e3 / args.pipe, "Gbp_per_s": bp * args.pipe / t / 1e9,
                                                "content_match_vs_synchronous_call": bool(same), "index_records": n_rec}
    if args.pipe:
        for kv in args.opt:  # the pipe with each A/B option, against the pipe without, in alternation
            k, v = kv.split("=")
            ta, tb, same = [], [], True
            for rep in range(3):
                t, sm, _ = pipe_pass(args.pipe, {k: int(v)})
                tb.append(t)
                same = same and sm
                t, sm, _ = pipe_pass(args.pipe, {})
                ta.append(t)
                same = same and sm
            print("pipe with %s: %.2f ms per batch against %.2f without (best of 3 x %d batches each, alternating); identical to the synchronous call: %s"
                  % (kv, min(tb) * 1e3 / args.pipe, min(ta) * 1e3 / args.pipe, args.pipe, same), flush=True)
            out["pipe"]["with " + kv] = {"ms_per_batch": min(tb) * 1e3 / args.pipe, "default_ms_per_batch": min(ta) * 1e3 / args.pipe,
                                         "content_match_vs_synchronous_call": bool(same)}
    if args.rounds:
        with ctx.options(debug_times=1):
            b.shmmrs(sp)
        print("--- and with the islands' rounds described (debug: slower)", file=sys.stderr, flush=True)
        with ctx.options(debug=1, debug_times=1):
            b.shmmrs(sp)
    ok = None
    if not args.no_check:
        t0 = time.perf_counter()
        sums, off = sh.checksum(), sh.offsets()

        def one(i):
            ref = O.sequence_to_shmmrs(i, seqs[i], O.spec())
            return len(ref), O.shmmr_checksum(ref)
        with ThreadPoolExecutor(args.threads) as ex:
            refs = list(ex.map(one, range(len(seqs))))
        bad = [i for i, (cnt, cs) in enumerate(refs) if int(off[i + 1] - off[i]) != cnt or not np.array_equal(sums[i], cs)]
        ok = not bad
        out.update(content_match=bool(ok), contigs_checked=len(seqs), cpu_s=time.perf_counter() - t0, cpu_threads=args.threads)
        print("CPU restatement on %d threads: %.1f s; contigs identical: %d of %d%s" %
              (args.threads, out["cpu_s"], len(seqs) - len(bad), len(seqs), "" if ok else " -- DIFFERENT: %s" % bad), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)
    sys.exit(0 if ok in (None, True) else 1)


if __name__ == "__main__":
    main()
