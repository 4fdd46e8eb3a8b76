#!/usr/bin/env python3
"""bench.py -- SHIMMER indexing throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (shmmrutils::sequence_to_shmmrs at k=56,w=80,r=4,min_span=64 + the
shimmer-pair records) over one batch of synthetic contigs that is already resident in HBM as 2-bit packed planes.
N=1 workload = BASELINE.json configs[1]: 1000 x 10 Mbp.

    python bench.py [--gpus N --steps K --warmup W] [--strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1, default (weak scaling): every rank runs the same per-GPU workload on its own contigs.
N > 1, --strong: ONE set of --contigs contigs is partitioned over the ranks by exchange.shard_contigs (greedy,
length balanced: the partitioner of SURVEY.md section 8e).  In both modes the per-rank shimmer lists are all-gathered
over RCCL inside the timed region (libpgrhip's own exchange entry points, pgr_exchange_*; --exchange torch uses
torch.distributed instead).

After the timed region rank 0 (N=1 only) adds: the content check of ALL contigs against the CPU restatement of the
reference (128-bit checksum per contig), the CPU baseline taken from that same run, the query leg (BASELINE.json
configs[2]) with its own roofline and CPU baseline, small-call latencies, and the PCIe-inclusive throughput of the
host-buffer entry point.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))

ALGO_BYTES_PER_BP = 0.2986  # BASELINE.md section 5 / SURVEY 8d: 0.25 B packed input + 16 B x 0.003035 final MM128
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
N_SIMD = 256 * 4            # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32
CLOCK_GHZ = 2.4
PCIE_PEAK_GBPS = 63.0       # PCIe Gen5 x16, one direction


def committed(*parts):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", *parts)))
    except Exception:
        return None


def valu_issue(bp_per_launch, launch_ms):
    """The bound that holds for the integer-hash kernel: VALU issue.  From tracked files only:
    profiles/traffic.json (SQ_INSTS_VALU of the committed PMC pass of this workload), profiles/r02_ubench/valu_cycles.json
    (measured cycles per wave64 instruction per opcode) and profiles/r02_tile/isa_histogram.json (opcode mix of the
    kernel in the shipped code object, tools/isa_histogram.py)."""
    t = committed("traffic.json")
    if not t or t.get("bp_per_launch") != bp_per_launch or launch_ms <= 0 or "valu_wave_insts_per_launch" not in t:
        return None
    insts = float(t["valu_wave_insts_per_launch"])
    ach = insts / (launch_ms * 1e-3)
    peak2 = N_SIMD * CLOCK_GHZ * 1e9 / 2.0  # the guide's figure: a wave64 VALU instruction issues over 2 cycles
    out = {"wave64_valu_insts_per_launch": insts, "valu_insts_per_bp": insts * 64.0 / bp_per_launch,
           "achieved_Ginst_per_s": ach / 1e9, "peak_Ginst_per_s": peak2 / 1e9, "frac_of_peak": ach / peak2,
           "peak_note": "MI355X_MICROARCH.md: SIMD-32, 2 cycles per wave64 VALU instruction, 1024 SIMDs x 2.4 GHz"}
    h = committed(t.get("isa_histogram", os.path.join("r02_tile", "isa_histogram.json")))
    if h and h.get("mean_cycles_per_valu_inst"):
        # cycle-weighted lower bound: the hardware's instruction count priced with the measured cost of the kernel's own
        # opcode mix (only add/sub/and/or/xor/not/lshr issue in 2 cycles; shifts left, 64-bit ops, f64 min/max, bfi ... take 4)
        bound_ms = insts * h["mean_cycles_per_valu_inst"] / N_SIMD / (CLOCK_GHZ * 1e9) * 1e3
        out.update({"mean_cycles_per_inst_of_the_kernels_mix": h["mean_cycles_per_valu_inst"],
                    "cycle_weighted_bound_ms": bound_ms, "frac_of_cycle_weighted_bound": min(1.0, bound_ms / launch_ms),
                    "bound_sources": ["profiles/traffic.json", "profiles/r02_ubench/valu_cycles.json",
                                      "profiles/" + t.get("isa_histogram", "r02_tile/isa_histogram.json")]})
    return out


def synth_contig_ascii(seed, contig, length):
    """the BASELINE.md section 4 generator restated with numpy (one splitmix64 per 32 bases) -> ACGT bytes"""
    import numpy as np
    n_words = (length + 31) // 32
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) ^ (np.uint64(contig) * np.uint64(0x9E3779B97F4A7C15)) ^ np.arange(n_words, dtype=np.uint64))
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    sh = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    codes = ((z[:, None] >> sh) & np.uint64(3)).astype(np.uint8).reshape(-1)[:length]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[codes]


def synth_substrings(seed, contigs, offsets, length):
    """the same generator for arbitrary (contig, offset) windows"""
    import numpy as np
    out = []
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    with np.errstate(over="ignore"):
        for c, o in zip(contigs, offsets):
            i = np.arange(o, o + length, dtype=np.uint64)
            z = np.uint64(seed) ^ (np.uint64(c) * np.uint64(0x9E3779B97F4A7C15)) ^ (i >> np.uint64(5))
            z = z + np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            code = (z >> (np.uint64(2) * (i & np.uint64(31)))) & np.uint64(3)
            out.append(acgt[code.astype(np.int64)])
    return out


def make_queries(P, seed, contig_ids, n_contigs, contig_len, nq, qlen, rng):
    """nq substrings of length qlen at random (contig, offset), every second one reverse complemented"""
    import numpy as np
    cs = rng.integers(0, n_contigs, nq)
    offs = rng.integers(0, max(1, contig_len - qlen), nq)
    qs = synth_substrings(seed, [contig_ids[int(c)] for c in cs], offs, qlen)
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    return cs, offs, P.PackedSeqs.from_list([comp[q][::-1] if i & 1 else q for i, q in enumerate(qs)])


def chains_of(r, qi):
    """per-query content of a flat result: [(sid, [(score bits, [hit pairs])])]"""
    out = []
    for t in range(int(r["q_off"][qi]), int(r["q_off"][qi + 1])):
        chains = []
        for c in range(int(r["t_off"][t]), int(r["t_off"][t + 1])):
            hp = r["hps"][int(r["c_off"][c]):int(r["c_off"][c + 1])]
            chains.append((r["c_score"][c].tobytes(), [tuple(int(v) for v in h) for h in hp]))
        out.append((int(r["t_sid"][t]), chains))
    return out


def query_bench(P, ctx, batch, spec, args, contig_ids):
    """BASELINE.json configs[2]: index = the resident contigs; 10 000 x 10 kbp substrings at random (contig, offset),
    half of them reverse-complemented; pgr-query defaults (penalty 0.025, counts 128, span 8)."""
    import numpy as np
    rng = np.random.default_rng(3)
    nq, qlen = args.queries, 10_000
    t0 = time.perf_counter()
    ix = P.Index(spec, ctx=ctx)
    ix.add_resident(batch, sids=contig_ids)
    ix.finalize()
    t_build = time.perf_counter() - t0
    cs, offs, qs = make_queries(P, args.seed, contig_ids, len(contig_ids), args.contig_len, nq, qlen, rng)

    def med3(f):
        f()  # warm-up (grows the workspaces once)
        reps = []
        for _ in range(3):  # the batch is a few ms: report the median of three
            t0 = time.perf_counter()
            r = f()
            reps.append(time.perf_counter() - t0)
        return sorted(reps)[1], reps, r
    # (a) the metric's convention: inputs resident in HBM when the timed region starts
    qb = P.Batch.from_seqs(qs, ctx=ctx)
    ctx.synchronize()
    # timed: the C entry point + pgr_hps_result_free, i.e. what a compiled host pays; the same call through the Python binding
    # (numpy views of the result block, released with the last view) is reported next to it (python_binding_s)
    t_py, _, r = med3(lambda: ix.query_hps_resident_raw(qb, 0.025))
    t_res, reps_res, _ = med3(lambda: ix.time_query_resident(qb, 0.025)[0])
    prof = ctx.last_query_prof()
    # (b) the drop-in boundary: host ASCII in, host chains out
    _, _, r2 = med3(lambda: ix.query_hps_raw(qs, 0.025))
    t_host, reps_host, _ = med3(lambda: ix.time_query_host(qs, 0.025)[0])
    same = all(np.array_equal(r[k], r2[k]) for k in ("q_off", "t_sid", "t_off", "c_score", "c_off", "hps"))
    # self-consistency: the best chain of every query lies on its source contig at its source offset
    ok = 0
    for qi in range(nq):
        best = None
        for t in range(int(r["q_off"][qi]), int(r["q_off"][qi + 1])):
            for c in range(int(r["t_off"][t]), int(r["t_off"][t + 1])):
                n_hp = int(r["c_off"][c + 1] - r["c_off"][c])
                if best is None or n_hp > best[0]:
                    best = (n_hp, int(r["t_sid"][t]), c)
        if best is not None and best[1] == contig_ids[int(cs[qi])]:
            h = r["hps"][int(r["c_off"][best[2]])]
            tb = int(h["tb"])
            if offs[qi] <= tb <= offs[qi] + qlen:
                ok += 1
    n_hps = int(len(r["hps"]))
    algo = 24.0 * n_hps + 0.25 * prof["query_bases"] + 17.0 * prof["n_signatures"]
    qk = committed("r02_query", "summary.json") or {}
    qpmc = (committed("r02_query", "pmc_summary.json") or {}).get("per_query_batch", {})
    out = {
        "workload": "BASELINE.json configs[2]: %d x %d bp queries (50%% reverse complement) against the %d x %d bp index, "
                    "penalty 0.025, max counts 128, max_aln_span 8" % (nq, qlen, len(contig_ids), args.contig_len),
        "index_build_s": t_build, "index_records": ix.n_records, "index_keys": ix.n_keys,
        "query_s": t_res, "query_s_reps": reps_res, "queries_per_s": nq / t_res, "hit_pairs": n_hps,
        "hit_pairs_per_s": n_hps / t_res, "chains": int(len(r["c_score"])),
        "queries_with_best_chain_on_source": ok,
        "inputs": "queries resident in HBM as 2-bit planes when the clock starts (pgr_query_hps_resident); the clock stops "
                  "when the chains are in host memory and the result has been released again (C entry point, no numpy copies)",
        "python_binding_s": t_py,
        "pcie_inclusive": {"query_s": t_host, "query_s_reps": reps_host, "queries_per_s": nq / t_host,
                           "hit_pairs_per_s": n_hps / t_host, "same_result_as_resident": bool(same),
                           "ascii_upload_bytes": int(prof["query_bases"]),
                           "ascii_upload_ms_at_pcie_peak": prof["query_bases"] / (PCIE_PEAK_GBPS * 1e9) * 1e3,
                           "note": "pgr_query_hps_batch: host ASCII in, host chains out"},
        "counts": {k: int(prof[k]) for k in ("n_query_pairs", "n_signatures", "n_hits", "n_groups", "n_chains", "n_hps")},
        "stage_ms": {k: float(prof[k]) for k in ("shmmr_ms", "lookup_ms", "chain_ms", "result_ms", "total_ms")},
        "roofline": {
            "bound": "hbm", "achieved": algo / t_res / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": algo / t_res / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": algo,
            "traffic": qpmc.get("hbm_bytes"),  # HBM bytes per batch by PMC (profiles/r02_query/pmc_summary.json), all kernels
            "algorithmic_bytes_formula": "24 B x hit pairs emitted + 0.25 B x query bases + 17 B x looked-up signatures "
                                         "(SURVEY.md 8d); signatures counted on the device",
            "dominant_kernel": qk.get("dominant_kernel"), "dominant_kernel_ms": qk.get("dominant_kernel_ms"),
            "kernel_ms_total": qk.get("kernel_ms_total"), "launches": qk.get("launches"),
            "what_holds": "latency: the batch moves ~37 MB (5 us of HBM time); %d kernel launches, 4 host round trips (one per "
                          "data-dependent buffer size), the serial critical path of the chaining DP and the PCIe download of "
                          "the chains take the rest" % (qk.get("launches") or 0),
        },
    }
    return out, ix


def query_cpu_baseline(P, ctx, spec, spec_t, args, contig_ids, cores):
    """the CPU restatement on a BOUNDED sample of configs[2]: index of the first S contigs (built by the checker itself,
    thread pool), 512 queries cut from them, one task per query on `cores` threads (= the rayon loop of
    pgr-query.rs:135-138).  The same queries go through the GPU against a GPU index of the same S contigs and the chain
    CONTENT (targets, chains, hit pairs, f32 scores) must be identical."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    S = min(64, len(contig_ids))
    nq, qlen = 8192, 10_000
    sp = O.spec(*spec_t)
    oix = O.Index(sp)
    t0 = time.perf_counter()
    if contig_ids[:S] == list(range(contig_ids[0], contig_ids[0] + S)):
        oix.add_synth_threads(S, contig_ids[0], args.seed, contig_ids[0], args.contig_len, cores)
    else:
        for c in contig_ids[:S]:
            oix.add_seq(c, O.synth_contig(args.seed, c, args.contig_len))
    oix.finalize()
    t_ix = time.perf_counter() - t0
    rng = np.random.default_rng(33)
    cs, offs, qs = make_queries(P, args.seed, contig_ids[:S], S, args.contig_len, nq, qlen, rng)
    qlist = [qs.buf[int(qs.off[i]):int(qs.off[i + 1])] for i in range(nq)]
    ref, dt = None, None
    for _ in range(3):  # a batch is ~0.1 s: the best of three
        ref, d = O.query_batch_threads(oix, qlist, 0.025, cores)
        dt = d if dt is None else min(dt, d)
    gb = P.Batch.synthetic([args.contig_len] * S, seed=args.seed, ctx=ctx, contig_ids=contig_ids[:S])
    gix = P.Index(spec, ctx=ctx)
    gix.add_resident(gb, sids=contig_ids[:S])
    gix.finalize()
    r = gix.query_hps_raw(qs, 0.025)
    n_same = 0
    for qi in range(nq):
        want = [(sid, [(np.float32(sc).tobytes(), [tuple(h) for h in hps]) for sc, hps in chains]) for sid, chains in ref[qi]]
        n_same += int(chains_of(r, qi) == want)
    return {
        "value": nq / dt, "unit": "queries/s", "hit_pairs_per_s": sum(len(h) for q in ref for _, ch in q for _, h in ch) / dt,
        "cores": cores, "kind": "port",
        "sample": "%d queries x %d bp against an index of %d x %d bp contigs of the same workload (best of 3 runs: %.3f s, one "
                  "task per query on %d threads; index built by the checker in %.1f s)" % (nq, qlen, S, args.contig_len, dt, cores, t_ix),
        "queries_compared": nq, "queries_with_identical_chains": n_same, "content_match": n_same == nq,
    }


def query_bench_dist(P, ctx, spec, args, gathered, world, rank, dist, torch, local_rank, all_ids):
    """BASELINE.json configs[2] on N GPUs: every rank builds the (replicated) index of ALL ranks' contigs from the
    all-gathered MM128 lists (pgr_index_add_shmmrs derives the pair records), the queries are sharded round robin, every
    rank chains its own share; value = all queries / slowest rank."""
    import numpy as np
    rng = np.random.default_rng(3)
    nq, qlen = args.queries, 10_000
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix = P.Index(spec, ctx=ctx)
    for part in gathered:  # one rank's list after the other (a sequence never straddles two ranks)
        if int(part.shape[0]):
            g = part if part.is_cuda else part.to("cuda:%d" % local_rank)
            g = g.contiguous()
            ix.add_shmmrs(device_ptr=g.data_ptr(), n=int(g.shape[0]))
    ix.finalize()
    t_build = time.perf_counter() - t0
    cs, offs, qs_all = make_queries(P, args.seed, all_ids, len(all_ids), args.contig_len, nq, qlen, rng)
    mine = np.arange(rank, nq, world)
    qs = P.PackedSeqs.from_list([qs_all.buf[int(qs_all.off[i]):int(qs_all.off[i + 1])] for i in mine])
    qb = P.Batch.from_seqs(qs, ctx=ctx)
    r = ix.query_hps_resident_raw(qb, 0.025)  # content (and warm-up)
    reps = []
    for _ in range(3):  # timed like the single-GPU leg: the C entry point + release of the result
        dist.barrier()
        reps.append(ix.time_query_resident(qb, 0.025)[0])
    ok = 0
    for i in range(len(mine)):
        best = None
        for t in range(int(r["q_off"][i]), int(r["q_off"][i + 1])):
            for c in range(int(r["t_off"][t]), int(r["t_off"][t + 1])):
                n_hp = int(r["c_off"][c + 1] - r["c_off"][c])
                if best is None or n_hp > best[0]:
                    best = (n_hp, int(r["t_sid"][t]))
        ok += int(best is not None and best[1] == all_ids[int(cs[mine[i]])])
    dev = ("cuda:%d" % local_rank) if args.backend == "nccl" else "cpu"
    t = torch.tensor([sorted(reps)[1], t_build], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    agg = torch.tensor([float(len(r["hps"])), float(ok)], dtype=torch.float64, device=dev)
    dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    t_q = float(t[0].item())
    return {
        "workload": "BASELINE.json configs[2] on %d GPUs: %d x %d bp queries sharded round robin, every rank holds the index of "
                    "all %d contigs built from the all-gathered shimmer lists; queries resident in HBM" % (world, nq, qlen, len(all_ids)),
        "index_build_s": float(t[1].item()), "index_records": ix.n_records, "query_s": t_q, "queries_per_s": nq / t_q,
        "hit_pairs": int(agg[0].item()), "hit_pairs_per_s": float(agg[0].item()) / t_q,
        "queries_with_best_chain_on_source": int(agg[1].item()),
    }


def effective_cpus():
    """CPUs this process may really use: scheduler affinity capped by the cgroup CPU quota (a container that sees
    256 CPUs but has a 16-CPU quota runs 16 threads' worth of work)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, (q + per // 2) // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(spec_t, n_contigs, contig_len, seed, contig0, gpu_counts, gpu_sums, cores):
    """the oracle (CPU restatement of the reference, one task per contig like rayon par_iter) over ALL contigs of the
    workload, generated inside the workers; every contig's 128-bit content checksum must equal the GPU's.
    Checker + baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    sp = O.spec(*spec_t)
    t0 = time.perf_counter()
    counts, sums, busy = O.synth_checksums_threads(sp, n_contigs, seed, contig0, contig_len, cores)
    wall = time.perf_counter() - t0
    busy_wall = float(busy.sum()) / cores  # the threaded section's duration without the time spent generating contigs
    ok_counts = bool(np.array_equal(counts, np.asarray(gpu_counts, dtype=np.uint64)))
    n_match = int(np.sum(np.all(sums == np.asarray(gpu_sums, dtype=np.uint64), axis=1)))
    bp = n_contigs * contig_len
    return {
        "value": bp / busy_wall / 1e9, "unit": "Gbp/s", "cores": cores, "kind": "port",
        "sample": "all %d x %d bp synthetic contigs of the workload, one task per contig on %d threads (= the CPUs this "
                  "container may use: affinity capped by the cgroup quota; the host shows %d): %.1f s wall of which %.1f s per "
                  "thread inside sequence_to_shmmrs (the rest generates the contigs)" %
                  (n_contigs, contig_len, cores, os.cpu_count() or 0, wall, busy_wall),
        "per_core_Mbp_per_s": bp / float(busy.sum()) / 1e6,
        "contigs_checked": int(n_contigs), "contigs_with_identical_checksum": n_match,
        "content_match": bool(n_match == n_contigs and ok_counts), "counts_match_gpu": ok_counts,
        "check": "128-bit order-sensitive checksum of (x, pos, strand) per contig, GPU (pgr_shmmrs_checksum) vs CPU",
    }


def latency_bench(P, ctx, spec, args):
    """the regime of the reference's real callers: one small call at a time (seq_db.rs:549-564 batches of <= 129 contigs,
    ext.rs:252-282 single queries)"""
    import numpy as np
    seq = synth_contig_ascii(args.seed, 0, 10_000)
    one = P.PackedSeqs.from_list([seq])

    def med(f, n=60):
        for _ in range(5):
            f()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2] * 1e3, ts[len(ts) // 10] * 1e3
    m1, p1 = med(lambda: P.time_shmmr_batch(one, spec, ctx=ctx))
    m1py, p1py = med(lambda: P.sequence_to_shmmrs_batch(one, spec, ctx=ctx))
    # single 10 kbp query against an index of 8 x 1 Mbp
    b = P.Batch.synthetic([1_000_000] * 8, seed=args.seed, ctx=ctx)
    ix = P.Index(spec, ctx=ctx)
    ix.add_resident(b)
    ix.finalize()
    q = P.PackedSeqs.from_list([synth_contig_ascii(args.seed, 3, 200_000)[50_000:60_000]])
    m2, p2 = med(lambda: ix.time_query_host(q, 0.025))
    m2py, p2py = med(lambda: ix.query_hps_raw(q, 0.025))
    return {"shmmr_batch_one_10kbp_contig_ms": {"median": m1, "p10": p1, "with_python_unpacking": {"median": m1py, "p10": p1py}},
            "query_hps_batch_one_10kbp_query_ms": {"median": m2, "p10": p2, "with_python_unpacking": {"median": m2py, "p10": p2py}},
            "note": "host ASCII in, host result out: the C entry point + release of the result, called through ctypes; "
                    "with_python_unpacking adds the binding's copies into numpy arrays; index of 8 x 1 Mbp for the query"}


def pcie_bench(P, ctx, spec, args):
    """B1 as the reference would call it: host ASCII in, host MM128 out (pgr_shmmr_batch).  Never the bench `value`."""
    import numpy as np
    n = 52  # 520 Mbp: above the 512 Mbp threshold of the pipelined (sub-batched, double-buffered) path
    seqs = [synth_contig_ascii(args.seed, c, 10_000_000) for c in range(n)]
    P.time_shmmr_batch(seqs, spec, ctx=ctx)  # warm-up: pinned windows, workspaces
    reps, n_sh = [], 0
    for _ in range(3):  # the C entry point + release of the result (no numpy copies: those belong to the Python binding)
        dt, n_sh = P.time_shmmr_batch(seqs, spec, ctx=ctx)
        reps.append(dt)
    t = sorted(reps)[1]
    bp = n * 10_000_000
    return {"value": bp / t / 1e9, "unit": "Gbp/s", "bp": bp, "s": t, "s_reps": reps, "shimmers": int(n_sh),
            "ascii_GB_per_s": bp / t / 1e9, "pcie_peak_GB_per_s": PCIE_PEAK_GBPS,
            "note": "pgr_shmmr_batch: 1 byte per base crosses PCIe; sub-batches staged on a copy stream while the previous "
                    "one computes"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)  # the first launches run at ramping clocks
    ap.add_argument("--contigs", type=int, default=1000, help="contigs per GPU (weak scaling) / in total (--strong)")
    ap.add_argument("--contig-len", type=int, default=10_000_000)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--strong", action="store_true",
                    help="partition ONE set of --contigs contigs over the ranks with exchange.shard_contigs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exchange", action="store_true", help="N>1: skip the all-gather of the shimmer lists")
    ap.add_argument("--exchange", default="abi", choices=["abi", "torch"],
                    help="N>1: abi = libpgrhip's pgr_exchange_* (RCCL linked by the library), torch = torch.distributed")
    ap.add_argument("--queries", type=int, default=10_000, help="query leg (after the timed region); 0 = off")
    ap.add_argument("--no-extras", action="store_true", help="skip the latency and PCIe-inclusive legs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--single-device", action="store_true",
                    help="plumbing test on a 1-GPU box: every rank uses cuda:0 (use with --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="plumbing test: run the process group + exchange code path even with one rank")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" %
                  (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    import pgrtk_amd as P
    from pgrtk_amd import exchange
    ctx = P.Context(local_rank)
    spec_t = (80, 56, 4, 64)
    spec = P.make_spec(*spec_t)
    if args.strong:
        all_lens = [args.contig_len] * args.contigs
        contig_ids = exchange.shard_contigs(all_lens, world)[rank]  # global ids of this rank's shard, file order
    else:
        contig_ids = list(range(rank * args.contigs, (rank + 1) * args.contigs))  # every rank has its own contigs
    lens = [args.contig_len] * len(contig_ids)
    batch = P.Batch.synthetic(lens, seed=args.seed, ctx=ctx, contig_ids=contig_ids)  # inputs resident in HBM
    bp_per_step = batch.total_bases

    # ---- set-up (untimed): one probe pass sizes the output buffers, so that no step allocates
    probe = batch.shmmrs(spec)
    dev = "cuda:%d" % local_rank
    n_max = max(len(exchange.shard_contigs([args.contig_len] * args.contigs, world)[r]) for r in range(world)) if args.strong \
        else args.contigs
    cap_mm = int(probe.count * 1.05 * n_max / max(1, len(contig_ids))) + 1024
    rec_buf = torch.empty((int(probe.n_pairs * 1.05) + 16, exchange.REC_WORDS), dtype=torch.int64, device=dev)
    xch = None
    mm_bufs = out_bufs = None
    do_exchange = use_dist and not args.no_exchange
    use_abi = do_exchange and args.exchange == "abi" and args.backend == "nccl"
    if do_exchange:
        # double buffered: the all-gather of step i overlaps the kernels of step i+1
        mm_bufs = [torch.empty((cap_mm, exchange.MM_WORDS), dtype=torch.int64, device=dev) for _ in range(2)]
        gdev = dev if args.backend == "nccl" else "cpu"
        out_bufs = [torch.empty((world * cap_mm, exchange.MM_WORDS), dtype=torch.int64, device=gdev) for _ in range(2)]
        if use_abi:
            # ncclUniqueId from rank 0 through the process group.  If the library's own communicator cannot be created on
            # ANY rank (all ranks agree through an all-reduce), everybody falls back to torch.distributed's all-gather
            try:
                xch = exchange.AbiExchange(ctx, rank, world, dist)
                ok = 1
            except Exception as e:  # noqa: BLE001
                print("rank %d: pgr_exchange_create failed (%r): torch.distributed all-gather instead" % (rank, e), file=sys.stderr)
                xch, ok = None, 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if xch is not None:
                    xch.close()
                xch, use_abi = None, False
    del probe
    if use_dist:
        dist.barrier()  # creates the communicator here, not inside the first timed step
        torch.cuda.synchronize()
    state = {"i": 0, "pending": None}

    def finish_pending():
        if state["pending"] is not None:
            parts, counts = state["pending"].wait(concat=False)  # per-rank views into the gather buffer: no copy
            state["gathered"] = parts
            state["pending"] = None

    def step():
        sh = batch.shmmrs(spec)
        slot = state["i"] & 1
        state["i"] += 1
        n = sh.frag_recs_into(rec_buf.data_ptr(), rec_buf.shape[0], sids=contig_ids)  # the per-GPU index shard
        if mm_bufs is not None:
            # what travels: the final MM128 lists with global sequence ids (16 B per shimmer; the pair records
            # are adjacent shimmers and are re-derived by the receiver, pgr_index_add_shmmrs)
            cnt = sh.count
            finish_pending()  # step i-1's lists have arrived everywhere (and its buffer slot is free again)
            sh.copy_into(mm_bufs[slot].data_ptr(), mm_bufs[slot].shape[0], rids=contig_ids)
            if use_abi:
                state["pending"] = xch.allgather_async(mm_bufs[slot], cnt, out_bufs[slot], cap_mm)
            else:
                local = mm_bufs[slot][:cnt]
                state["pending"] = exchange.PendingAllgather(local if args.backend == "nccl" else local.cpu(),
                                                             out=out_bufs[slot])
        p = ctx.last_prof()
        state["sh"] = sh
        state["n_pairs"] = n
        return p.level1_ms, p.level1_aux_ms, p.level2_ms, p.total_ms, p.bases_tiled, p.n_level1

    def sync():
        finish_pending()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    profs = [step() for _ in range(args.steps)]
    sync()
    dt = time.perf_counter() - t0
    total_bp = bp_per_step
    if use_dist:
        tdev = ("cuda:%d" % local_rank) if args.backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        t = torch.tensor([float(bp_per_step)], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_bp = int(t.item())

    # query leg on N GPUs (after the timed region; a failure here must not cost the headline line)
    dist_query = None
    if use_dist and args.queries > 0 and state.get("gathered") is not None:
        try:
            g = state["gathered"]
            if args.strong:
                all_ids = list(range(args.contigs))
            else:
                all_ids = list(range(world * args.contigs))
            dist_query = query_bench_dist(P, ctx, spec, args, g, world, rank, dist, torch, local_rank, all_ids)
        except Exception as e:  # noqa: BLE001
            dist_query = {"error": repr(e)[:300]}
    if rank == 0:
        k = max(1, args.steps)
        l1_ms = sum(p[0] for p in profs) / k
        aux_ms = sum(p[1] for p in profs) / k
        l2_ms = sum(p[2] for p in profs) / k
        tot_ms = sum(p[3] for p in profs) / k
        bases_tiled = profs[-1][4] if profs else 0
        achieved = ALGO_BYTES_PER_BP * bases_tiled / (l1_ms * 1e-3) / 1e9 if l1_ms > 0 else 0.0
        sh = state["sh"]
        tr = committed("traffic.json") or {}
        out = {
            "metric": "Gbp/s SHIMMER-indexed (k=56,w=80,r=4)",
            "value": total_bp * args.steps / dt / 1e9,
            "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / max(1, args.steps) * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[1]: sequence_to_shmmrs HIP kernels, %s synthetic contigs (seed %d), "
                            "ShmmrSpec k=56 w=80 r=4 min_span=64, 2-bit packed input resident in HBM, output = final MM128 "
                            "lists + shimmer-pair records%s" %
                            (("%d x %d bp in total, partitioned over the ranks by shard_contigs" % (args.contigs, args.contig_len))
                             if args.strong else ("%d x %d bp per GPU" % (args.contigs, args.contig_len)), args.seed,
                             "" if world == 1 else (", per-GPU shimmer lists (pair endpoints, 16 B each) all-gathered over RCCL "
                                                    "(%s)" % ("pgr_exchange_*" if use_abi else "torch.distributed")
                                                    if do_exchange else ", no exchange")),
                "bp_per_gpu_per_step": bp_per_step, "bp_per_step_all_gpus": total_bp,
                "parallelism": "contig-sharded x%d" % world,
                "final_shimmers_per_gpu": sh.count, "pair_records_per_gpu": state["n_pairs"],
            },
            "roofline": {
                "bound": "valu", "kernel": "level1_tile_kernel",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": tr.get("hbm_bytes_per_launch") if tr.get("bp_per_launch") == bases_tiled else None,
                "algorithmic_bytes_per_bp": ALGO_BYTES_PER_BP, "bp_per_launch": bases_tiled,
                "avg_launch_ms": l1_ms,
                "valu_issue": valu_issue(bases_tiled, l1_ms),
                "note": "achieved / peak / frac are the HBM-roofline figures of the metric (algorithmic bytes / launch time "
                        "against 8 TB/s).  What binds the kernel is VALU issue (two 64-bit mix hashes per position): see "
                        "valu_issue.frac_of_cycle_weighted_bound and DESIGN.md section 5",
            },
            "stage_ms": {"level1_tile": l1_ms, "level1_tail_serial": aux_ms, "level2": l2_ms, "compute_total": tot_ms},
        }
        cores = effective_cpus()
        if dist_query is not None:
            out["query"] = dist_query
        if world == 1 and args.queries > 0 and dist_query is None:
            try:
                out["query"], _ix = query_bench(P, ctx, batch, spec, args, contig_ids)
                del _ix
                if not args.no_cpu_baseline:
                    out["query"]["cpu_baseline"] = query_cpu_baseline(P, ctx, spec, spec_t, args, contig_ids, cores)
            except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
                out.setdefault("query", {})["error"] = repr(e)[:300]
        if world == 1 and not args.no_cpu_baseline:
            try:
                _mm_off = [int(v) for v in sh.offsets()]
                gpu_counts = [_mm_off[i + 1] - _mm_off[i] for i in range(len(contig_ids))]
                out["cpu_baseline"] = cpu_baseline(spec_t, len(contig_ids), args.contig_len, args.seed, contig_ids[0],
                                                   gpu_counts, sh.checksum(), cores)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_extras:
            try:
                out["latency"] = latency_bench(P, ctx, spec, args)
            except Exception as e:  # noqa: BLE001
                out["latency"] = {"error": repr(e)[:300]}
            try:
                out["pcie_inclusive"] = pcie_bench(P, ctx, spec, args)
            except Exception as e:  # noqa: BLE001
                out["pcie_inclusive"] = {"error": repr(e)[:300]}
        try:  # RCCL prints a version banner through C stdio (block buffered on a pipe): push it out BEFORE the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        if xch is not None:
            xch.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
