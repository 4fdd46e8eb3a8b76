#!/usr/bin/env python3
"""bench.py -- SHIMMER indexing throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (shmmrutils::sequence_to_shmmrs at k=56,w=80,r=4,min_span=64 + the
shimmer-pair records) over one batch of synthetic contigs that is already resident in HBM as 2-bit packed planes.
N=1 workload = BASELINE.json configs[1]: 1000 x 10 Mbp.

    python bench.py [--gpus N --steps K --warmup W] [--strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1, default (weak scaling): every rank runs the same per-GPU workload on its own contigs.
N > 1, --strong: ONE set of --contigs contigs is partitioned over the ranks by exchange.shard_contigs (greedy,
length balanced: the partitioner of SURVEY.md section 8e).
N > 1, what a step is: shimmers + pair records of the rank's contigs, then the MERGE inside the timed region -- the key
space is cut into N ranges (splitters from a pooled sample), every pair record travels to the rank that owns its range
(one variable all-to-all over RCCL: libpgrhip's pgr_exchange_shard_records; --exchange torch uses torch.distributed) and
each rank sorts its range into its shard of the frag_map (pgr_index_finalize): `value` is defined on that sum,
`exchange_ms` / `merge_ms` say how it splits.  At N = 1 there is nothing to merge: the step is BASELINE configs[1]
(shimmers + pair records) and the index build is timed beside it (`index_build_ms`, `value_incl_index_build`).

After the timed region: EVERY rank checks the shimmer lists of its own contigs against the CPU restatement of the
reference (128-bit checksum per contig) and the exchanged record set against what was sent (order-independent checksum
summed over the ranks, key ranges disjoint and ordered); rank 0 reports the CPU baseline from its own run.  N = 1 adds
the query leg (BASELINE.json configs[2]) with its own roofline and CPU baseline, real-genome shapes, small-call
latencies, the PCIe-inclusive throughput of the host entry points and the time to index 100 Gbp.  ONE JSON line, rank 0.
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
QUERY_PROFILE = "r06_query"  # profiles/<dir> whose counters / kernel trace the query leg's replayed fields come from


def committed(*parts):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", *parts)))
    except Exception:
        return None


def traffic_detail(tr, bp_per_launch, n_final, n_level1):
    """roofline.traffic next to what it is made of: the read side with the MEASURED FETCH_SIZE factors of
    this kernel's pattern (profiles/r05_calib), the write side split into the level-1 list (an intermediate, 12 B per level-1
    minimizer: not algorithmic) and what the metric counts as output (16 B per final shimmer, written by the list stage)."""
    if not tr or not tr.get("hbm_bytes_per_launch") or not bp_per_launch:
        return {}
    algo = ALGO_BYTES_PER_BP * bp_per_launch
    d = {"traffic_low": tr.get("hbm_bytes_per_launch_low"),
         "traffic_over_algorithmic": tr["hbm_bytes_per_launch"] / algo,
         "traffic_low_over_algorithmic": (tr["hbm_bytes_per_launch_low"] / algo) if tr.get("hbm_bytes_per_launch_low") else None,
         "read_side": {"algorithmic_bytes": 0.25 * bp_per_launch, "pmc_fetch_size_bytes": tr.get("fetch_size_bytes_raw"),
                       "bytes_requested": tr.get("read_bytes_requested"), "bytes_distinct": tr.get("read_bytes_distinct"),
                       "factors": tr.get("correction")},
         "write_side": {"pmc_write_size_bytes": tr.get("write_size_bytes"),
                        "level1_list_bytes": 12 * n_level1 if n_level1 else None,
                        "algorithmic_final_output_bytes": 16 * n_final if n_final else None,
                        "note": "the tile kernel writes the level-1 list (12 B per minimizer) and the segment tables; the final "
                                "MM128 lists (16 B per shimmer, the algorithmic output) are written by the list stage behind it"}}
    return d


def valu_issue(bp_per_launch, launch_ms):
    """The bound that holds for the integer-hash kernel: VALU issue.  From tracked files only: profiles/traffic.json
    (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU2 and GRBM_GUI_ACTIVE of the committed PMC passes of this workload),
    profiles/r03_ubench/valu_cycles.json (measured cycles per wave64 instruction of EVERY opcode the kernel uses) and
    profiles/r03_tile/isa_histogram.json (opcode mix of the kernel in the shipped code object, tools/isa_histogram.py).
    Two independent estimates of how busy the SIMDs' issue ports are: from the counters alone, and from the opcode mix."""
    t = committed("traffic.json")
    if not t or t.get("bp_per_launch") != bp_per_launch or launch_ms <= 0 or "valu_wave_insts_per_launch" not in t:
        return None
    insts = float(t["valu_wave_insts_per_launch"])
    ach = insts / (launch_ms * 1e-3)
    peak2 = N_SIMD * CLOCK_GHZ * 1e9 / 2.0  # the guide's figure: a wave64 VALU instruction issues over 2 cycles
    out = {"wave64_valu_insts_per_launch": insts, "valu_insts_per_bp": insts * 64.0 / bp_per_launch,
           "achieved_Ginst_per_s": ach / 1e9, "peak_Ginst_per_s": peak2 / 1e9, "frac_of_peak": ach / peak2,
           "source": "REPLAYED counters: the instruction counts come from the committed PMC passes of this workload "
                     "(profiles/traffic.json <- profiles/%s/pmc_summary.json), not from this run; the launch time they are "
                     "divided by (avg_launch_ms) is measured live.  Instruction counts are a property of the code object: "
                     "profile_consistency says whether the shipped kernel is still the profiled one" % t.get("profile"),
           "peak_note": "MI355X_MICROARCH.md: SIMD-32, 2 cycles per wave64 VALU instruction, 1024 SIMDs x 2.4 GHz; measured on this "
                        "chip (profiles/r03_ubench): 2.4 cycles only for add/sub/and/or/xor/not/mov/right shifts, 4.15-4.27 for every "
                        "other opcode (left shifts, 64-bit ops, multiplies, compares, selects, f64 min/max)"}
    vc = committed(t.get("valu_cycles", os.path.join("r03_ubench", "valu_cycles.json"))) or {}
    slot = vc.get("class_full")
    if slot and t.get("valu2_wave_insts_per_launch") is not None and t.get("gui_active_cycles_per_launch"):
        # counters alone: every VALU instruction holds its SIMD's issue port for one slot, except those the hardware counts as
        # issued through the second path (SQ_ACTIVE_INST_VALU2); slot = cycles per instruction of the micro-kernels that have
        # no second-path instruction.  Checked on the micro-kernels themselves: 0.92 - 1.03 where the truth is 1.
        need = slot * (insts - float(t["valu2_wave_insts_per_launch"])) / N_SIMD
        have = float(t["gui_active_cycles_per_launch"]) / 8.0
        out["busy_by_counters"] = {"issue_slot_cycles": slot, "slots_needed_cycles_per_simd": need, "elapsed_cycles_per_simd": have,
                                   "valu_busy": min(1.0, need / have),
                                   "counters": "SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU2, GRBM_GUI_ACTIVE (profiles/%s/pmc_summary.json)" % t.get("profile")}
    h = committed(t.get("isa_histogram", os.path.join("r03_tile", "isa_histogram.json")))
    if h and h.get("mean_cycles_per_valu_inst"):
        # cycle-weighted lower bound: the hardware's instruction count priced with the measured cost of the kernel's own
        # opcode mix (every opcode measured by itself: no assigned costs)
        bound_ms = insts * h["mean_cycles_per_valu_inst"] / N_SIMD / (CLOCK_GHZ * 1e9) * 1e3
        out.update({"mean_cycles_per_inst_of_the_kernels_mix": h["mean_cycles_per_valu_inst"],
                    "cycle_weighted_bound_ms": bound_ms, "frac_of_cycle_weighted_bound": min(1.0, bound_ms / launch_ms),
                    "bound_sources": ["profiles/traffic.json", "profiles/" + t.get("valu_cycles", "r03_ubench/valu_cycles.json"),
                                      "profiles/" + t.get("isa_histogram", "r03_tile/isa_histogram.json")]})
    return out


def profile_consistency(tr, live_ms, bp_per_launch):
    """does the committed profile still describe the kernel that just ran?  The rocprofv3 average launch duration of the
    dominant kernel in profiles/<profile>/kernel_stats.csv against this run's HIP-event average (5 % tolerance: the boxes'
    clocks differ by 1-2 %), and the code object's VALU instruction count against the committed opcode histogram"""
    import csv
    out = {"profile": tr.get("profile"), "live_avg_launch_ms": live_ms}
    try:
        rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", tr["profile"], "kernel_stats.csv"))))
        row = [r for r in rows if tr["kernel"] in r["Name"] and "80, 56, false" in r["Name"]][0]
        ms = float(row["AverageNs"]) / 1e6
        out.update({"committed_avg_launch_ms": ms, "drift": live_ms / ms - 1.0,
                    "ok": bool(tr.get("bp_per_launch") == bp_per_launch and abs(live_ms / ms - 1.0) <= 0.05)})
    except Exception as e:  # noqa: BLE001
        out.update({"ok": False, "error": repr(e)[:200]})
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


def synth_contig_ascii_long(seed, contig, length, piece=1 << 24):
    """the same for chromosome-sized contigs: `piece` bases at a time (the word-wise form above holds 8 bytes per base while it works)"""
    import numpy as np
    out = np.empty(length, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    sh = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    with np.errstate(over="ignore"):
        for o in range(0, length, piece):
            n = min(piece, length - o)
            w = np.arange(o // 32, (o + n + 31) // 32, dtype=np.uint64)
            z = (np.uint64(seed) ^ (np.uint64(contig) * np.uint64(0x9E3779B97F4A7C15)) ^ w) + np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            out[o:o + n] = acgt[((z[:, None] >> sh) & np.uint64(3)).astype(np.uint8).reshape(-1)[:n]]
    return out


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
    # the same batch through the chained form of the per-query kernel (round 5's path: the shimmer pipeline's list stage over the whole
    # batch, then the per-query kernel; context option no_query_level1), same context, same index, result compared array for array
    chained_form = None
    try:
        with ctx.options(no_query_level1=1):
            _, _, r_ch = med3(lambda: ix.query_hps_resident_raw(qb, 0.025))
            t_ch, reps_ch, _ = med3(lambda: ix.time_query_resident(qb, 0.025)[0])
            p_ch = int(ctx.last_query_prof()["path"])
        chained_form = {"query_s": t_ch, "query_s_reps": reps_ch, "path_id": p_ch,
                        "same_result_as_the_level1_form": bool(all(np.array_equal(r[k], r_ch[k]) for k in ("q_off", "t_sid", "t_off", "c_score", "c_off", "hps"))),
                        "what": "the same batch with context option no_query_level1: 19 launches instead of 8 (profiles/r05_query vs r06_query)"}
        del r_ch
    except Exception as ex:  # noqa: BLE001
        chained_form = {"error": repr(ex)}
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
    # (c) two batches in flight through the context's pipe: pgr_pipe_submit_query / _collect_query -- the tiles of batch i + 1 on the
    # context's stream beside everything behind the tiles of batch i on the pipe's back stream (which is tried against the context's
    # stream until the two do not share a hardware queue).  The reference's rayon loop over the queries (pgr-query.rs:135-165).
    # (Two CONTEXTS on two host threads do the same when the runtime places them well, and run one after the other -- or far worse --
    # when it does not: tools/query_concurrency_probe.py, profiles/r05_query/two_batches_in_flight.txt; not part of this line.)
    piped = None
    try:
        qb_b = P.Batch.from_seqs(qs, ctx=ctx)
        pipe = P.Pipe(spec, ctx=ctx)
        ref_counts = (int(len(r["t_sid"])), int(len(r["c_score"])), n_hps)
        k_batches = 32

        def run_pipe(k, raw_last):
            outs = []
            for i in range(k):
                if pipe.in_flight == 2:
                    outs.append(pipe.collect_query(raw=False))
                pipe.submit_query(qb if i & 1 == 0 else qb_b, ix, 0.025)
            last = None
            while pipe.in_flight:
                if raw_last and pipe.in_flight == 1:
                    last = pipe.collect_query(raw=True)
                else:
                    outs.append(pipe.collect_query(raw=False))
            return outs, last
        run_pipe(4, False)
        reps_p, outs = [], []
        for _ in range(3):  # (the median of three: a long-lived process now and then loses tens of ms in one call while the driver clears
            ctx.synchronize()  # memory released earlier -- DESIGN 5, the cold pass; tools/query_pipe_probe.py shows the calls)
            t0 = time.perf_counter()
            o, _ = run_pipe(k_batches, False)
            ctx.synchronize()
            reps_p.append(time.perf_counter() - t0)
            outs += o
        dtp = sorted(reps_p)[1]
        _, last = run_pipe(3, True)
        same_last = all(np.array_equal(last[k], r[k]) for k in ("q_off", "t_sid", "t_off", "c_score", "c_off", "hps"))
        pipe.close()
        piped = {"batches": k_batches, "ms_per_batch": dtp / k_batches * 1e3, "ms_per_batch_reps": [t / k_batches * 1e3 for t in reps_p],
                 "queries_per_s": nq * k_batches / dtp,
                 "every_batch_same_counts": bool(all(o == ref_counts for o in outs)), "a_batch_same_content": bool(same_last),
                 "what": "pgr_pipe_submit_query / pgr_pipe_collect_query on one context, two batches in flight, %d batches of the same "
                         "%d resident queries; the clock stops when the last batch's chains are in host memory and released" % (k_batches, nq)}
        del qb_b
    except Exception as ex:  # noqa: BLE001
        piped = {"error": repr(ex)}
    algo = 24.0 * n_hps + 0.25 * prof["query_bases"] + 17.0 * prof["n_signatures"]
    qk = committed(QUERY_PROFILE, "summary.json") or {}
    qpmc = (committed(QUERY_PROFILE, "pmc_summary.json") or {}).get("per_query_batch", {})
    fused = int(prof.get("path", 0)) in (1, 2, 3)
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
        "chained_form": chained_form,
        "pipelined": piped,
        "pcie_inclusive": {"query_s": t_host, "query_s_reps": reps_host, "queries_per_s": nq / t_host,
                           "hit_pairs_per_s": n_hps / t_host, "same_result_as_resident": bool(same),
                           "ascii_upload_bytes": int(prof["query_bases"]),
                           "ascii_upload_ms_at_pcie_peak": prof["query_bases"] / (PCIE_PEAK_GBPS * 1e9) * 1e3,
                           "note": "pgr_query_hps_batch: host ASCII in, host chains out"},
        "counts": {k: int(prof[k]) for k in ("n_query_pairs", "n_signatures", "n_hits", "n_groups", "n_chains", "n_hps")},
        "stage_ms": {k: float(prof[k]) for k in ("shmmr_ms", "lookup_ms", "chain_ms", "result_ms", "total_ms")},
        "path": ("the level-1 form of the per-query kernel (csrc/query_fused.hip, round 6): the tile kernel of the queries, then one "
                 "wavefront per query reads ITS level-1 minimizers from the tile segments and runs both reductions, min_span, the pairs, "
                 "lookup, count filters, grouping and the chaining DP -- no list stage of the batch, 8 launches, the call has ONE "
                 "synchronization (stage_ms.shmmr_ms holds the device time of all of it)" if int(prof.get("path", 0)) == 3 else
                 "one wavefront per query behind the shimmers (csrc/query_fused.hip): lookup, count filters, grouping, chaining "
                 "DP in one kernel" + ("; enqueued behind the shimmer pipeline without a host wait in between: "
                                       "stage_ms.shmmr_ms holds the device time of both, the call has ONE "
                                       "synchronization" if int(prof.get("path", 0)) == 2 else
                                       "; stage_ms.chain_ms holds all of it, download included") if fused else
                 "one kernel per stage over the whole batch (csrc/index.hip)"),
        "path_id": int(prof.get("path", 0)),
        "roofline": {
            "bound": "hbm", "achieved": algo / t_res / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": algo / t_res / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": algo,
            "traffic": qpmc.get("hbm_bytes"),  # HBM bytes per batch by PMC, all kernels
            "source": {"achieved, frac, algorithmic_bytes": "measured in this run",
                       "traffic": "REPLAYED from profiles/%s/pmc_summary.json (PMC passes of the same batch), not from this run" % QUERY_PROFILE,
                       "dominant_kernel, dominant_kernel_ms, kernel_ms_total, launches":
                           "REPLAYED from profiles/%s/summary.json (rocprofv3 --kernel-trace of the same batch), not from this run" % QUERY_PROFILE},
            "algorithmic_bytes_formula": "24 B x hit pairs emitted + 0.25 B x query bases + 17 B x looked-up signatures "
                                         "(SURVEY.md 8d); signatures counted on the device",
            "dominant_kernel": qk.get("dominant_kernel"), "dominant_kernel_ms": qk.get("dominant_kernel_ms"),
            "kernel_ms_total": qk.get("kernel_ms_total"), "launches": qk.get("launches"),
            "what_holds": ("latency and PCIe, not HBM: the batch's algorithmic bytes take microseconds at the roofline; the time is the "
                           "chain of dependent launches (stage_ms, measured here), one host wait and the chains' way back over PCIe.  "
                           "Per-kernel times: profiles/%s/kernel_stats.csv" % QUERY_PROFILE) if fused else
                          ("latency: one kernel per stage over the whole batch, four host round trips (one per data-dependent "
                           "buffer size), the serial critical path of the chaining DP and the PCIe download of the chains"),
        },
    }
    return out, ix


def query_cpu_baseline(P, ctx, spec, spec_t, args, contig_ids, cores, gpu_index=None):
    """the CPU restatement on the benchmark's OWN index: the frag_map of all the workload's contigs built by the checker
    itself (thread pool, one task per contig), 8192 queries of configs[2]'s kind cut from them, one task per query on `cores`
    threads (= the rayon loop of pgr-query.rs:135-138).  The same queries go through the GPU index of the same contigs and
    the chain CONTENT (targets, chains, hit pairs, f32 scores) must be identical.  Workloads beyond 12 Gbp (not the
    default) fall back to an index of the first 64 contigs on both sides."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    full = len(contig_ids) * args.contig_len <= 12_000_000_000 and gpu_index is not None
    S = len(contig_ids) if full else min(64, len(contig_ids))
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
    n_rec_cpu = int(len(oix.records())) if not full else None
    del oix
    if full:
        gix = gpu_index
    else:
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
        "index": "the benchmark's own: all %d x %d bp contigs, %d records on the GPU side" % (S, args.contig_len, gix.n_records) if full
                 else "a sample: the first %d contigs (%s records)" % (S, n_rec_cpu),
        "sample": "%d queries x %d bp against the CPU restatement's index of %d x %d bp contigs (best of 3 runs: %.3f s, one "
                  "task per query on %d threads; index built by the checker in %.1f s)" % (nq, qlen, S, args.contig_len, dt, cores, t_ix),
        "queries_compared": nq, "queries_with_identical_chains": n_same, "content_match": n_same == nq,
    }


def query_bench_dist(P, ctx, spec, args, shard, xch, exchange, world, rank, dist, torch, local_rank, all_ids):
    """BASELINE.json configs[2] on N GPUs: the replicated query index is the concatenation of the finalized key-range
    shards (all-gathered in rank order: already sorted, no sort), the queries are sharded round robin, every rank chains
    its own share; value = all queries / slowest rank."""
    import numpy as np
    rng = np.random.default_rng(3)
    nq, qlen = args.queries, 10_000
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix = xch.allgather_index(shard) if xch is not None else exchange.allgather_index_torch(shard)
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
                    "all %d contigs = the key-range shards all-gathered in rank order; queries resident in HBM" %
                    (world, nq, qlen, len(all_ids)),
        "replicate_index_s": float(t[1].item()), "index_records": ix.n_records, "index_keys": ix.n_keys, "query_s": t_q,
        "queries_per_s": nq / t_q, "hit_pairs": int(agg[0].item()), "hit_pairs_per_s": float(agg[0].item()) / t_q,
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


def cpu_baseline(spec_t, contig_ids, contig_len, seed, gpu_counts, gpu_sums, cores, budget_s=None):
    """the oracle (CPU restatement of the reference, one task per contig like rayon par_iter) over the contigs `contig_ids`
    of the workload, generated inside the workers; every contig's 128-bit content checksum must equal the GPU's.
    budget_s: bound the CPU work to about that many seconds on `cores` threads (a prefix of the contigs is checked then).
    Checker + baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    sp = O.spec(*spec_t)
    n_all = len(contig_ids)
    n = n_all
    if budget_s is not None:  # ~66 Mbp/s per thread (measured, reported below as per_core_Mbp_per_s)
        n = max(1, min(n_all, int(budget_s * cores * 60e6 / max(1, contig_len))))
    ids = np.asarray(contig_ids[:n], dtype=np.uint64)
    t0 = time.perf_counter()
    counts, sums, busy = O.synth_checksums_ids_threads(sp, ids, seed, contig_len, cores)
    wall = time.perf_counter() - t0
    busy_wall = float(busy.sum()) / cores  # the threaded section's duration without the time spent generating contigs
    ok_counts = bool(np.array_equal(counts, np.asarray(gpu_counts[:n], dtype=np.uint64)))
    n_match = int(np.sum(np.all(sums == np.asarray(gpu_sums[:n], dtype=np.uint64), axis=1)))
    bp = n * contig_len
    return {
        "value": bp / busy_wall / 1e9, "unit": "Gbp/s", "cores": cores, "kind": "port",
        "sample": "%d of this rank's %d x %d bp synthetic contigs, one task per contig on %d threads (= the CPUs this "
                  "process may use: affinity capped by the cgroup quota%s; the host shows %d): %.1f s wall of which %.1f s per "
                  "thread inside sequence_to_shmmrs (the rest generates the contigs)" %
                  (n, n_all, contig_len, cores, "" if budget_s is None else ", shared between the ranks of this node",
                   os.cpu_count() or 0, wall, busy_wall),
        "per_core_Mbp_per_s": bp / float(busy.sum()) / 1e6,
        "contigs_checked": int(n), "contigs_with_identical_checksum": n_match,
        "content_match": bool(n_match == n and ok_counts), "counts_match_gpu": ok_counts,
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
    many = P.PackedSeqs.from_list([synth_contig_ascii(args.seed, c, 10_000) for c in range(129)])  # seq_db.rs:561: 129 contigs per batch
    m129, p129 = med(lambda: P.time_shmmr_batch(many, spec, ctx=ctx))
    # single 10 kbp query against an index of 8 x 1 Mbp
    b = P.Batch.synthetic([1_000_000] * 8, seed=args.seed, ctx=ctx)
    ix = P.Index(spec, ctx=ctx)
    ix.add_resident(b)
    ix.finalize()
    q = P.PackedSeqs.from_list([synth_contig_ascii(args.seed, 3, 200_000)[50_000:60_000]])
    m2, p2 = med(lambda: ix.time_query_host(q, 0.025))
    m2py, p2py = med(lambda: ix.query_hps_raw(q, 0.025))
    return {"shmmr_batch_one_10kbp_contig_ms": {"median": m1, "p10": p1, "with_python_unpacking": {"median": m1py, "p10": p1py}},
            "shmmr_batch_129_x_10kbp_contigs_ms": {"median": m129, "p10": p129},
            "query_hps_batch_one_10kbp_query_ms": {"median": m2, "p10": p2, "with_python_unpacking": {"median": m2py, "p10": p2py}},
            "path": "batches of short clean contigs: one kernel launch + one synchronization (csrc/small.hip, one workgroup per contig; "
                    "round 2: ~15 dependent device operations, 0.13 ms for one 10 kbp contig)",
            "note": "host ASCII in, host result out: the C entry point + release of the result, called through ctypes; "
                    "with_python_unpacking adds the binding's copies into numpy arrays; index of 8 x 1 Mbp for the query"}


def pcie_bench(P, ctx, spec, args):
    """B1 as a host calls it: host bases in, host MM128 out.  Three forms of input: ASCII (pgr_shmmr_batch: the library's
    CPU packer fills the pinned windows, 0.375 B per base cross PCIe), packed planes + validity plane
    (pgr_shmmr_batch_packed, 0.375 B per base, no packing inside the call) and packed planes alone (0.25 B per base).
    Never the bench `value`."""
    import numpy as np
    n = 104  # 1.04 Gbp: the pipelined (sub-batched, double-buffered) path
    seqs = [synth_contig_ascii(args.seed, c, 10_000_000) for c in range(n)]
    bp = n * 10_000_000

    def med3(f):
        f()  # warm-up: pinned windows, workspaces
        reps = [f() for _ in range(3)]
        return sorted(r[0] for r in reps)[1], [r[0] for r in reps], reps[0][1]
    # the C entry points + release of the result (no numpy copies: those belong to the Python binding)
    t_a, reps_a, n_sh = med3(lambda: P.time_shmmr_batch(seqs, spec, ctx=ctx))
    tp = []
    for _ in range(3):
        t0 = time.perf_counter()
        packed, n_bad = P.pack_ascii(seqs)
        tp.append(time.perf_counter() - t0)
    t_pack = sorted(tp)[1]
    bare = P.PackedBases(packed.lens, packed.planes, None)
    # pageable host arrays: the library copies them through its pinned staging windows
    t_pp, reps_pp, n_sh_pp = med3(lambda: P.time_shmmr_batch_packed(packed, spec, ctx=ctx))
    t_bp, reps_bp, _ = med3(lambda: P.time_shmmr_batch_packed(bare, spec, ctx=ctx))
    # the host's packed buffers pinned once (pgr_host_register: a host that streams its sequences through long-lived buffers):
    # the DMA engine reads them where they lie
    import torch
    with P.PinnedArrays(packed.planes, packed.valid):
        t_p, reps_p, n_sh_p = med3(lambda: P.time_shmmr_batch_packed(packed, spec, ctx=ctx))
        t_b, reps_b, n_sh_b = med3(lambda: P.time_shmmr_batch_packed(bare, spec, ctx=ctx))
        # what this link gives a plain copy of the same planes out of the same pinned memory (the ceiling of the route)
        dst = torch.empty(packed.planes.size, dtype=torch.int64, device="cuda:%d" % ctx.device)
        src_t = torch.from_numpy(packed.planes.view(np.int64))
        tl = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dst.copy_(src_t, non_blocking=True)
            torch.cuda.synchronize()
            tl.append(time.perf_counter() - t0)
        link = packed.planes.nbytes / sorted(tl)[1] / 1e9
        del dst
    return {"value": bp / t_p / 1e9, "unit": "Gbp/s", "bp": bp, "s": t_p, "s_reps": reps_p, "shimmers": int(n_sh_p),
            "input": "host-packed 2-bit planes + validity plane in pinned host memory (pgr_host_register once, then "
                     "pgr_shmmr_batch_packed): the planes cross the link straight from the caller's buffer, the validity plane is "
                     "read on the host and crosses only where it says something (here: nowhere) -- 0.25 B per base on the link",
            "link_GB_per_s": 0.25 * bp / t_p / 1e9, "pcie_peak_GB_per_s": PCIE_PEAK_GBPS,
            "measured_link_GB_per_s": link, "frac_of_measured_link": 0.25 * bp / t_p / 1e9 / link,
            "packed_planes_only": {"value": bp / t_b / 1e9, "s": t_b, "s_reps": reps_b, "link_GB_per_s": 0.25 * bp / t_b / 1e9,
                                   "frac_of_measured_link": 0.25 * bp / t_b / 1e9 / link,
                                   "note": "no validity plane passed (every base is ACGT): 0.25 B per base, pinned host memory"},
            "pageable_host_memory": {"planes_and_validity": {"value": bp / t_pp / 1e9, "s": t_pp, "s_reps": reps_pp},
                                     "planes_only": {"value": bp / t_bp / 1e9, "s": t_bp, "s_reps": reps_bp},
                                     "note": "the same arrays not pinned: the library's threads copy them into its pinned staging "
                                             "windows (non-temporal stores) beside the DMA"},
            "ascii_input": {"value": bp / t_a / 1e9, "s": t_a, "s_reps": reps_a,
                            "note": "pgr_shmmr_batch: ASCII in; the library's CPU packer writes the pinned windows, 0.375 B per "
                                    "base cross PCIe (round 2: the ASCII bytes crossed, 40 Gbp/s)"},
            "host_packer": {"s": t_pack, "GB_per_s": bp / t_pack / 1e9, "threads": effective_cpus(), "non_acgt_bytes": int(n_bad),
                            "note": "pgr_pack_ascii over the same contigs on the CPUs this process may use; NOT inside the packed "
                                    "entry points' time (a host that stores or decodes its sequences 2-bit packed never runs it)"},
            "same_shimmer_count_all_inputs": bool(n_sh == n_sh_p == n_sh_b == n_sh_pp),
            "note": "sub-batches staged on a copy stream while the previous one computes; results (16 B per shimmer) come back "
                    "through pinned windows"}


def chromosome_like(seed=31, L=248_000_000):
    """one reference-chromosome-like contig: an 18 Mbp run of N (centromere gap), forty gaps of 50 kbp - 1 Mbp, 300 isolated
    N, 200 lower-case (soft-masked) stretches"""
    import numpy as np
    rng = np.random.default_rng(1)
    s = np.concatenate([synth_substrings(seed, [0], [o], min(16_000_000, L - o))[0] for o in range(0, L, 16_000_000)])
    s[int(L * 0.488):int(L * 0.488) + 18_000_000] = ord("N")
    for _ in range(40):
        a = int(rng.integers(0, L - 2_000_000))
        s[a:a + int(rng.integers(50_000, 1_000_000))] = ord("N")
    for _ in range(300):
        s[int(rng.integers(0, L))] = ord("N")
    for _ in range(200):
        a = int(rng.integers(0, L - 100_000))
        s[a:a + int(rng.integers(100, 50_000))] |= 0x20  # the same bases for the reference's table
    return s


def repeat_like(seed=7, lens=(30_000_000, 20_000_000, 10_000_000)):
    """repeat-rich contigs the way assemblies have them: 171-bp satellite arrays with 1.5 % divergence, microsatellites,
    reverse-complement symmetric units ((AT)n, (ACGT)n: every k-mer is its own reverse complement), homopolymers, tandem
    duplications of a 10 kbp unit"""
    import numpy as np
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rnd(n):
        return rng.choice(acgt, int(n))

    def contig(L):
        parts, n = [], 0
        while n < L:
            r = rng.random()
            if r < 0.35:
                p = rnd(rng.integers(50_000, 2_000_000))
            elif r < 0.6:
                total = int(rng.integers(200_000, 3_000_000))
                p = np.tile(rnd(171), total // 171 + 1)[:total].copy()
                m = rng.random(total) < 0.015
                p[m] = rng.choice(acgt, int(m.sum()))
            elif r < 0.8:
                u = rnd(rng.integers(2, 7))
                p = np.tile(u, int(rng.integers(1_000, 50_000)) // len(u) + 1)
            elif r < 0.85:
                u = np.frombuffer([b"AT", b"CG", b"ACGT", b"AATT", b"GAATTC"][int(rng.integers(0, 5))], dtype=np.uint8)
                p = np.tile(u, int(rng.integers(1_000, 200_000)) // len(u) + 1)
            elif r < 0.9:
                p = np.full(int(rng.integers(100, 20_000)), acgt[int(rng.integers(0, 4))], dtype=np.uint8)
            else:
                p = np.tile(rnd(10_000), int(rng.integers(3, 40)))
            parts.append(p)
            n += len(p)
        return np.concatenate(parts)[:L]
    return [contig(L) for L in lens]


def shapes_bench(P, ctx, spec, spec_t, cores, check):
    """what real genomes look like next to the i.i.d. ACGT of the headline: gaps, satellites, a million short reads.  Inputs
    resident in HBM; every result compared with the CPU restatement (whole-contig 128-bit checksums)."""
    import numpy as np
    O = None
    if check:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O  # checker only

    def run(batch):
        batch.shmmrs(spec)
        ts = []
        sh = None
        for _ in range(6):  # (a flagged batch's host-side lists take a few calls to stop growing: best of 6)
            t0 = time.perf_counter()
            sh = batch.shmmrs(spec)
            ts.append(time.perf_counter() - t0)
        return min(ts), sh, ctx.last_prof()
    out = {}
    # 1. chromosome-like
    s = chromosome_like()
    b = P.Batch.from_seqs([s], ctx=ctx)
    t, sh, prof = run(b)
    e = {"bp": int(len(s)), "ms": t * 1e3, "Gbp_per_s": len(s) / t / 1e9, "shimmers": sh.count,
         "Mbp_through_exact_islands": prof.exact_bases / 1e6, "non_acgt_bytes": int((s == ord("N")).sum()),
         "what": "one 248 Mbp contig: an 18 Mbp run of N, forty gaps of 50 kbp - 1 Mbp, 300 isolated N, lower-case stretches"}
    if O is not None:
        t0 = time.perf_counter()
        ref = O.sequence_to_shmmrs(0, s, O.spec(*spec_t))
        e["cpu_one_thread_s"] = time.perf_counter() - t0
        e["content_match"] = bool(len(ref) == sh.count and np.array_equal(sh.checksum()[0], O.shmmr_checksum(ref)))
    out["chromosome_like"] = e
    del b, sh, s
    # 2. repeat-rich
    seqs = repeat_like()
    b = P.Batch.from_seqs(seqs, ctx=ctx)
    t, sh, prof = run(b)
    bp = sum(len(q) for q in seqs)
    e = {"bp": int(bp), "ms": t * 1e3, "Gbp_per_s": bp / t / 1e9, "shimmers": sh.count, "level1_minimizers": int(prof.n_level1),
         "Mbp_through_exact_islands": prof.exact_bases / 1e6,
         "what": "60 Mbp in 3 contigs: satellite arrays, microsatellites, (AT)n / (ACGT)n, homopolymers, tandem duplications"}
    if O is not None:
        sums, off = sh.checksum(), sh.offsets()
        ok = True
        for i, q in enumerate(seqs):
            ref = O.sequence_to_shmmrs(i, q, O.spec(*spec_t))
            ok = ok and int(off[i + 1] - off[i]) == len(ref) and bool(np.array_equal(sums[i], O.shmmr_checksum(ref)))
        e["content_match"] = bool(ok)
    out["repeat_like"] = e
    del b, sh, seqs
    # 3. a million short reads
    n, L = 1_000_000, 1_000
    b = P.Batch.synthetic([L] * n, seed=41, ctx=ctx)
    t, sh, prof = run(b)
    e = {"bp": n * L, "ms": t * 1e3, "Gbp_per_s": n * L / t / 1e9, "shimmers": sh.count, "what": "10^6 x 1 kbp reads"}
    if O is not None:
        n_chk = 125_000
        cnt, ref, _ = O.synth_checksums_threads(O.spec(*spec_t), n_chk, 41, 0, L, cores)
        sums, off = sh.checksum(), sh.offsets()
        same = int(np.sum((off[1:n_chk + 1] - off[:n_chk] == cnt) & np.all(sums[:n_chk] == ref, axis=1)))
        e["reads_checked"] = n_chk
        e["content_match"] = bool(same == n_chk)
    out["short_reads"] = e
    del b, sh
    # 4. a genome-like batch at batch size: what a reference assembly costs per base (tools/genome_like_bench.py has the generator)
    try:
        from concurrent.futures import ThreadPoolExecutor
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import genome_like_bench as G
        class Base:  # (the generator's base sequence: the synthetic contigs of BASELINE.md section 4, numpy form, 16 Mbp at a time)
            @staticmethod
            def synth_contig(seed, c, L):
                return synth_contig_ascii_long(seed, c, L)
        lens = [m * 1_000_000 for m in G.CHROM_MBP]
        with ThreadPoolExecutor(cores) as ex:
            made = list(ex.map(lambda cl: G.genome_like_contig(Base, cl[0], cl[1]), enumerate(lens)))
        seqs = [m[0] for m in made]
        bp = int(sum(lens))
        b = P.Batch.from_seqs(seqs, ctx=ctx)
        t, sh, prof = run(b)
        e = {"bp": bp, "contigs": len(seqs), "ms": t * 1e3, "Gbp_per_s": bp / t / 1e9, "shimmers": sh.count,
             "Mbp_through_exact_islands": prof.exact_bases / 1e6, "level1_tile_ms": prof.level1_ms,
             "planted": {k: int(sum(m[1][k] for m in made)) for k in made[0][1]},
             "what": "12 chromosome-sized contigs (133 - 248 Mbp): centromere gaps beside alpha-satellite arrays, tens of shorter gaps, telomeres, "
                     "isolated N, soft-masked repeats over half of the sequence, a microsatellite per 8 kbp (a few hundred (AT)n-like arrays "
                     "longer than k per contig), duplications; pgr_shmmrs_compute on the resident batch"}
        sums, off = sh.checksum(), sh.offsets()
        # the same batch through the pipe (two jobs in flight: a flagged job's islands and list stage beside the next job's tiles)
        pipe = P.Pipe(spec, ctx=ctx)
        k, got = 8, []
        for rep in range(2):
            del got[:]
            ctx.synchronize()
            t0 = time.perf_counter()
            for i in range(k):
                if pipe.in_flight == 2:
                    got.append(pipe.collect()[0])
                pipe.submit(b)
            while pipe.in_flight:
                got.append(pipe.collect()[0])
            ctx.synchronize()
            tp = time.perf_counter() - t0
        same = all(g.count == sh.count and bool(np.array_equal(g.checksum(), sums)) for g in got[-2:])
        pipe.close()
        e["pipelined"] = {"batches": k, "ms_per_batch": tp * 1e3 / k, "Gbp_per_s": bp * k / tp / 1e9, "content_match_vs_synchronous_call": bool(same),
                          "what": "pgr_pipe_submit / pgr_pipe_collect, the same resident batch %d times, lists only" % k}
        del got
        if check:
            def one(i):
                ref = O.sequence_to_shmmrs(i, seqs[i], O.spec(*spec_t))
                return len(ref), O.shmmr_checksum(ref)
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                refs = list(ex.map(one, range(len(seqs))))
            e["cpu_s"] = time.perf_counter() - t0
            e["cpu_threads"] = cores
            e["contigs_checked"] = len(seqs)
            e["content_match"] = bool(all(int(off[i + 1] - off[i]) == c and np.array_equal(sums[i], cs) for i, (c, cs) in enumerate(refs)))
        out["genome_like"] = e
    except Exception as ex:  # noqa: BLE001
        out["genome_like"] = {"error": repr(ex)}
    return out


def target_100gbp(P, ctx, spec, args, spec_t=(80, 56, 4, 64), cores=1, check=True):
    """BASELINE.json's target as a number: wall time to SHIMMER-index 100 Gbp of DISTINCT synthetic contigs (10 batches of
    1000 x 10 Mbp, contig ids 0 .. 9999) into ONE index on this GPU -- generation of the synthetic input on the device,
    shimmers, pair records, and the sort into the frag_map included (target: under 60 s on 8 GPUs).
    In a FRESH context (what a one-shot pgr-mdb run sees: pgr-mdb.rs:53-111), after the bench's own context has given its
    cached blocks back; batches through pgr_pipe_* (two in flight), the index's append buffer reserved from the spec's density.
    Content: every batch's shimmer lists stay on the device until the clock has stopped; >= 64 contigs sampled from ids
    1000 .. 9999 (and 8 from 0 .. 999) are compared with the CPU restatement by their 128-bit checksums, and the index's
    order-independent record checksum must equal the sum over the batches of the checksums of the pair records derived from
    those lists."""
    import numpy as np
    import torch
    n_b, n_c, L = 10, 1000, 10_000_000
    # (what the build takes from the device: the caching allocator's peak 63.5e9 bytes + the contexts' grow-only workspaces -- level-1
    # buffers of two lanes, segment tables -- 21e9, measured as `fallback_bytes` of a 56 GiB arena in round 6, + the per-query kernel's
    # key table, 32 B x 304 M keys)
    RESERVE_GIB = 96
    ctx.synchronize()
    # (NOT ctx.trim().  In a process that has allocated and released many GiB, every so-many-th large hipMalloc blocks for up to
    # seconds while the driver clears released memory (tools/probe/big_malloc_probe.py: torch alone shows it; the gaps and the
    # waits both grow with the number of large allocations) -- the fresh context's first 12-13 GB allocation below may be that
    # one.  The one-shot number is therefore taken in a process of its own, FIRST, before this process releases anything more
    # (the child of an earlier version of this leg, started right after the in-process leg had given 53 GB back, took 1.8 s).)
    free_b, total_b = torch.cuda.mem_get_info(ctx.device)
    held_b = ctx.mem_stats()[0]
    fresh_process = None
    try:  # the one-shot case (pgr-mdb.rs:53-111: one process per file list): the same build in a process of its own
        import subprocess
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "t100_pipe_probe.py"), "pipe", str(n_b), "--json",
                            "--reserve-gib", str(RESERVE_GIB)], capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        j = json.loads(line)
        fresh_process = {"first_pass": j["passes"][0], "repeat": j["passes"][-1], "context_create_s": j["context_create_s"],
                                "pgr_ctx_reserve_s": j.get("reserve_s"), "reserved_GiB": j.get("reserve_gib"), "arena": j.get("arena"),
                                "first_touch_of_4GiB_ms": j.get("first_touch_of_4GiB_ms"),
                                "same_records_as_this_process": j["passes"][0]["records"],
                                "what": "tools/t100_pipe_probe.py pipe %d --json in a new process (while this one still holds its memory)" % n_b}
    except Exception as e:  # noqa: BLE001
        fresh_process = {"error": repr(e)[:300]}
    fresh = P.Context(ctx.device)
    # the build's device memory in ONE block, allocated and touched before the clock starts (pgr_ctx_reserve: what a one-shot host does
    # first; round 5 measured 2.96 s for the first pass in this process, one multi-GB hipMalloc of it blocking for 2.3 s)
    t_res = time.perf_counter()
    fresh.reserve(int(RESERVE_GIB * (1 << 30)))
    fresh.synchronize()
    t_res = time.perf_counter() - t_res
    M = (1 << 64) - 1

    def once(keep):
        fresh.synchronize()
        fresh.mem_stats(reset_peak=True)
        t0 = time.perf_counter()
        calls = []

        def timed(name, f):
            ts = time.perf_counter()
            r = f()
            calls.append((name, time.perf_counter() - ts))
            return r
        ix = timed("pgr_index_create", lambda: P.Index(spec, ctx=fresh))
        # pair records per base at (80, 56, 4, 64), SURVEY 8
        timed("pgr_index_reserve", lambda: ix.reserve(int(n_b * n_c * L * 0.003036 * 1.01) + 4096))
        pipe = timed("pgr_pipe_create", lambda: P.Pipe(spec, ctx=fresh))
        kept = []
        for bi in range(n_b):
            ids = list(range(bi * n_c, (bi + 1) * n_c))
            b = timed("pgr_batch_synthetic of batch %d" % bi, lambda: P.Batch.synthetic([L] * n_c, seed=args.seed, ctx=fresh, contig_ids=ids))
            if pipe.in_flight == 2:
                kept.append(timed("pgr_pipe_collect in front of batch %d" % bi, lambda: pipe.collect(want_shmmrs=keep)))
            timed("pgr_pipe_submit of batch %d" % bi, lambda: pipe.submit(b, sids=ids, index=ix))
            del b
        slowest_call.append(max(calls, key=lambda c: c[1]))
        while pipe.in_flight:
            kept.append(pipe.collect(want_shmmrs=keep))
        t1 = time.perf_counter()
        ix.finalize()
        fresh.synchronize()
        t2 = time.perf_counter()
        pipe.close()
        return t0, t1, t2, ix, kept, fresh.mem_stats()[1]
    slowest_call = []
    t0, t1, t2, ix, _, peak0 = once(False)
    n_rec, n_keys = ix.n_records, ix.n_keys
    del ix
    r0, r1, r2, ix, kept, peak1 = once(check)
    bp = n_b * n_c * L
    out = {"bp": bp, "s": t2 - t0, "Gbp_per_s": bp / (t2 - t0) / 1e9, "batches_s": t1 - t0, "sort_into_frag_map_s": t2 - t1,
           "context": "fresh context inside the bench process (created for this leg, beside the bench's own)",
           "device_memory_free_before_this_leg_bytes": int(free_b), "device_memory_total_bytes": int(total_b),
           "held_by_the_benchs_own_context_bytes": int(held_b),
           "peak_device_bytes_of_the_allocator": peak0,
           "pgr_ctx_reserve": {"GiB": RESERVE_GIB, "s": t_res, "arena_after_both_passes": fresh.arena_stats(),
                               "note": "outside `s`: one hipMalloc + one first touch in front of the build; nothing of the build "
                                       "then asks the runtime for device memory (fallback_calls counts what did)"},
           "slowest_enqueueing_call": {"call": slowest_call[0][0], "s": slowest_call[0][1],
                                       "note": "every call of the loop timed on the host: a first pass of seconds instead of 0.25 s is "
                                               "ONE of them waiting in a multi-GB hipMalloc while the driver clears memory this process "
                                               "released earlier (every so-many-th large hipMalloc of a process that keeps allocating and "
                                               "freeing GiBs: profiles/r05_target/big_malloc_probe.txt, torch alone shows it; DESIGN 5)"},
           "repeat": {"s": r2 - r0, "Gbp_per_s": bp / (r2 - r0) / 1e9, "batches_s": r1 - r0, "sort_into_frag_map_s": r2 - r1,
                      "peak_device_bytes_of_the_allocator": peak1,
                      "note": "the same again in that context: every buffer comes from its cache"},
           "index_records": n_rec, "index_keys": n_keys, "n_gpus": 1,
           "target": "BASELINE.json: >= 100 Gbp SHIMMER-indexed in under 60 s on 8 x MI355X",
           "what": "%d batches of %d x %d bp distinct synthetic contigs generated on the device, through pgr_pipe_submit / "
                   "pgr_pipe_collect into one index (sorted CSR at the end)" % (n_b, n_c, L)}
    if check:
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O
            rng = np.random.default_rng(100)
            sample = sorted(set(int(v) for v in rng.choice(np.arange(n_c, n_b * n_c), 64, replace=False)) |
                            set(int(v) for v in rng.choice(n_c, 8, replace=False)))
            cnt, ref, _ = O.synth_checksums_ids_threads(O.spec(*spec_t), sample, args.seed, L, cores)
            ok_lists, n_pairs_all, acc = True, 0, [0, 0]
            tmp = None
            for bi, (sh, n_pairs) in enumerate(kept):
                sums, off = sh.checksum(), sh.offsets()
                for j, cid in enumerate(sample):
                    if bi * n_c <= cid < (bi + 1) * n_c:
                        c = cid - bi * n_c
                        ok_lists = ok_lists and int(off[c + 1] - off[c]) == int(cnt[j]) and bool(np.array_equal(sums[c], ref[j]))
                if tmp is None or tmp.shape[0] < sh.n_pairs:
                    tmp = torch.empty((int(sh.n_pairs * 1.05) + 16, 5), dtype=torch.int64, device="cuda:%d" % fresh.device)
                n = sh.frag_recs_into(tmp.data_ptr(), tmp.shape[0], sids=list(range(bi * n_c, (bi + 1) * n_c)))
                ok_lists = ok_lists and n == n_pairs
                cs = P.records_checksum(tmp.data_ptr(), n, ctx=fresh)
                acc = [(acc[0] + cs[0]) & M, (acc[1] + cs[1]) & M]
                n_pairs_all += n
            ics = ix.records_checksum()
            out["contigs_checked"] = len(sample)
            out["contigs_checked_with_id_1000_or_above"] = sum(1 for c in sample if c >= n_c)
            out["content_match"] = bool(ok_lists)
            out["index_holds_exactly_the_records_of_the_checked_lists"] = bool(n_pairs_all == ix.n_records and [int(ics[0]), int(ics[1])] == acc)
            out["check"] = ("128-bit order-sensitive checksum per sampled contig, GPU lists of the timed build vs CPU restatement; "
                            "order-independent 128-bit checksum of the index's records vs the sum over the batches of the checksums "
                            "of the pair records of those lists")
        except Exception as e:  # noqa: BLE001
            out["content_check_error"] = repr(e)[:300]
    del kept, ix
    fresh.close()
    if fresh_process is not None:
        if "same_records_as_this_process" in fresh_process:
            fresh_process["same_records_as_this_process"] = bool(fresh_process["same_records_as_this_process"] == n_rec)
        out["fresh_process"] = fresh_process
    return out


def pipelined_leg(P, ctx, batch, spec, rec_buf, contig_ids, steps, warmup, torch, sync_state):
    """The same step through pgr_pipe_*: two batches in flight, the list stage + pair records of batch i on the back stream beside
    the tiles of batch i + 1 (the software-pipelined loop of load_index_from_reader, seq_db.rs:541-571).  >= 4 back-to-back
    batches; the last two jobs' content against the synchronous step's (128-bit checksums of lists and records)."""
    import numpy as np
    bufs = [rec_buf, torch.empty_like(rec_buf)]
    pipe = P.Pipe(spec, ctx=ctx)
    steps = max(4, steps)
    lv1 = []

    def run(k, keep):
        last = []
        for i in range(k):
            if pipe.in_flight == 2:
                sh, n = pipe.collect()
                lv1.append(ctx.last_prof().level1_ms)
                last.append((sh, n, (i - 2) & 1))
                last = last[-keep:] if keep else []
            pipe.submit(batch, sids=contig_ids, rec_ptr=bufs[i & 1].data_ptr(), rec_capacity=rec_buf.shape[0])
        j = k - pipe.in_flight
        while pipe.in_flight:
            sh, n = pipe.collect()
            lv1.append(ctx.last_prof().level1_ms)
            last.append((sh, n, j & 1))
            j += 1
        return last[-keep:] if keep else []
    run(max(2, warmup), 0)
    torch.cuda.synchronize()
    ctx.synchronize()
    del lv1[:]
    t0 = time.perf_counter()
    last = run(steps, 2)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ok = True
    ref_sh, ref_n = sync_state["sh"], sync_state["n_pairs"]
    ref_sum = ref_sh.checksum()
    ref_rec = P.records_checksum(rec_buf.data_ptr(), ref_n, ctx=ctx) if (steps & 1) == 0 else None  # (bufs[0] was rewritten by the pipe with the same records)
    for sh, n, slot in last:
        ok = ok and n == ref_n and sh.count == ref_sh.count and bool(np.array_equal(sh.checksum(), ref_sum))
        rec = P.records_checksum(bufs[slot].data_ptr(), n, ctx=ctx)
        if ref_rec is None:
            ref_rec = rec
        ok = ok and rec == ref_rec
    pipe.close()
    bp = batch.total_bases
    return {"value_pipelined": bp * steps / dt / 1e9, "ms_per_step_pipelined": dt / steps * 1e3, "batches": steps, "in_flight": 2,
            "level1_tile_ms_beside_the_list_stage": float(np.mean(lv1)) if lv1 else None,
            "content_match_vs_synchronous_step": bool(ok),
            "what": "pgr_pipe_submit / pgr_pipe_collect over %d back-to-back batches (the same resident 10 Gbp batch), shimmer lists "
                    "+ pair records per batch as in `value`; tiles on the context's stream, list stage + pair records on a second "
                    "stream behind an event" % steps}


def overlapped_exchange_leg(P, exchange, ctx, xch, use_abi, spec, args, contig_ids, lens, rec_buf, torch, dist, dev, tdev, world, ref_shard_cs,
                            ref_n_shard, result):
    """N > 1: the exchange + shard sort of step i BESIDE the tile kernels of step i + 1.  The compute moves to a second context of
    this process (its own streams, its own copy of the resident batch); the exchange and the index keep the first one and run
    on a worker thread: sample -> splitters -> partition -> all-to-all -> sort of the rank's key range, while the main thread
    is inside the next step's shimmer pipeline.  Every rank issues its collectives in the same order (one worker at a time).
    Content: the last step's shard must have the checksum and the size of the timed loop's shard.  Fills `result` (the caller
    gives this leg a deadline: a collective that never completes must not cost the headline line)."""
    import threading
    ctx2 = P.Context(ctx.device)
    batch2 = P.Batch.synthetic(lens, seed=args.seed, ctx=ctx2, contig_ids=contig_ids)
    bufs = [rec_buf, torch.empty_like(rec_buf)]
    box = {}

    def merge(buf, n):
        try:
            torch.cuda.set_device(ctx.device)  # (a new thread starts on device 0)
            ix = P.Index(spec, ctx=ctx)
            if use_abi:
                xch.shard_records(buf.data_ptr(), n, ix)
            else:
                exchange.shard_records_torch(ctx, buf.data_ptr(), n, ix)
            ix.finalize()
            box["shard"] = ix
        except Exception as e:  # noqa: BLE001
            box["error"] = repr(e)[:300]

    def run(k):
        worker = None
        for i in range(k):
            sh = batch2.shmmrs(spec)
            n = sh.frag_recs_into(bufs[i & 1].data_ptr(), bufs[i & 1].shape[0], sids=contig_ids)
            del sh
            if worker is not None:
                worker.join()
            if "error" in box:
                raise RuntimeError(box["error"])
            worker = threading.Thread(target=merge, args=(bufs[i & 1], n))
            worker.start()
        if worker is not None:
            worker.join()
        if "error" in box:
            raise RuntimeError(box["error"])
    run(max(2, args.warmup))
    ctx.synchronize()
    ctx2.synchronize()
    dist.barrier()
    k = max(4, args.steps)
    t0 = time.perf_counter()
    run(k)
    ctx.synchronize()
    ctx2.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=tdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    shard = box["shard"]
    same = [int(v) for v in shard.records_checksum()] == [int(v) for v in ref_shard_cs] and int(shard.n_records) == int(ref_n_shard)
    ok = torch.tensor([1 if same else 0], dtype=torch.int32, device=tdev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    bp = torch.tensor([float(batch2.total_bases)], dtype=torch.float64, device=tdev)
    dist.all_reduce(bp, op=dist.ReduceOp.SUM)
    result.update({"value_overlapped": float(bp.item()) * k / float(t.item()) / 1e9, "ms_per_step_overlapped": float(t.item()) / k * 1e3,
                   "steps": k, "content_match_vs_timed_loop": bool(int(ok.item())),
                   "what": "the same step, the merge of step i (record all-to-all by key range + sort of the rank's range) on a worker "
                           "thread and the first context beside the shimmer pipeline of step i + 1 on a second context; slowest rank, "
                           "all ranks' bases / time, the last merge inside the clock"})
    del batch2, box
    ctx2.close()


def self_spawn(n):
    """re-run this command line under torch.distributed.run with n ranks on this node; returns its exit code"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher: starting %s" % (n, " ".join(cmd[1:9])), file=sys.stderr, flush=True)
    return subprocess.call(cmd)


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
    ap.add_argument("--no-exchange", action="store_true", help="N>1: skip the merge (no record exchange, no index shards)")
    ap.add_argument("--exchange", default="abi", choices=["abi", "torch"],
                    help="N>1: abi = libpgrhip's pgr_exchange_* (RCCL linked by the library), torch = torch.distributed")
    ap.add_argument("--queries", type=int, default=10_000, help="query leg (after the timed region); 0 = off")
    ap.add_argument("--no-extras", action="store_true", help="skip the latency and PCIe-inclusive legs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--single-device", action="store_true",
                    help="plumbing test on a 1-GPU box: every rank uses cuda:0 (use with --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="plumbing test: run the process group + exchange code path even with one rank")
    ap.add_argument("--two-calls", action="store_true", help="the step as two C calls (pgr_shmmrs_compute + pgr_shmmrs_to_frag_recs_device), as in round 4")
    ap.add_argument("--no-pipelined-leg", action="store_true", help="skip the pgr_pipe_* leg (value_pipelined): profiling runs of the synchronous step")
    ap.add_argument("--no-overlap-leg", action="store_true", help="N>1: skip the leg that runs step i's merge beside step i+1's tiles")
    ap.add_argument("--exchange-timeout", type=int, default=120,
                    help="N>1: seconds the library waits in ncclCommInitRank / for a collective before it aborts its communicator "
                         "(context option exchange_timeout_s); the bench then falls back to the torch.distributed transport")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one process per GPU, rendezvous on 127.0.0.1)
        sys.exit(self_spawn(args.gpus))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world), file=sys.stderr)
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
    ctx.set_option("exchange_timeout_s", args.exchange_timeout)
    ctx.set_option("exchange_collective_timeout_s", args.exchange_timeout)  # (the bench would rather fall back than wait half an hour)
    if args.force_dist:
        # one rank through the library's REAL communicator (the default of one rank is copies without RCCL): this process has
        # PyTorch's librccl.so mapped already, the library takes that copy
        ctx.set_option("exchange_rccl_world1", 1)
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

    # ---- set-up (untimed): one probe pass sizes the record buffer, so that no step allocates it
    probe = batch.shmmrs(spec)
    dev = "cuda:%d" % local_rank
    rec_buf = torch.empty((int(probe.n_pairs * 1.05) + 16, exchange.REC_WORDS), dtype=torch.int64, device=dev)
    del probe
    xch = None
    create_ok = 0
    do_exchange = use_dist and not args.no_exchange
    use_abi = do_exchange and args.exchange == "abi" and args.backend == "nccl"
    if use_abi:
        # ncclUniqueId from rank 0 through the process group.  If the library's own communicator cannot be created on
        # ANY rank (all ranks agree through an all-reduce), everybody uses torch.distributed as the transport instead
        try:
            xch = exchange.AbiExchange(ctx, rank, world, dist)
            ok = 1
        except Exception as e:  # noqa: BLE001
            print("rank %d: pgr_exchange_create failed (%r): torch.distributed transport instead" % (rank, e), file=sys.stderr)
            xch, ok = None, 0
        create_ok = ok
    if use_dist:
        dist.barrier()  # creates the communicator here, not inside the first timed step
        torch.cuda.synchronize()
    state = {}
    fallback = {"note": None}

    def agree_or_fall_back(ok, what):
        """all ranks learn whether the library's own RCCL transport worked on EVERY rank; if not, all switch to
        torch.distributed together (a rank that timed out has aborted its communicator and says so in `what`)"""
        nonlocal xch, use_abi
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if xch is not None:
                xch.close()
            xch, use_abi = None, False
            fallback["note"] = what

    def step():
        # shimmer lists + this rank's pair records: ONE call and one wait (pgr_shmmrs_compute_recs; --two-calls: the round-4 step,
        # pgr_shmmrs_compute then pgr_shmmrs_to_frag_recs_device -- same kernels but for the records' offsets, which the fused call
        # derives on the device)
        if args.two_calls:
            sh = batch.shmmrs(spec)
            n = sh.frag_recs_into(rec_buf.data_ptr(), rec_buf.shape[0], sids=contig_ids)
        else:
            sh, n = batch.shmmrs_and_recs(spec, rec_buf.data_ptr(), rec_buf.shape[0], sids=contig_ids)
        p = ctx.last_prof()
        x_ms = m_ms = 0.0
        if do_exchange:
            # the merge (seq_db.rs:605-612 is ONE map): key-range shards.  Every record travels to the rank that owns its
            # range of first hashes (sample -> splitters -> stable partition -> one variable all-to-all), then each rank
            # sorts its range: the shards in rank order are the single-process CSR
            t0 = time.perf_counter()
            ix = P.Index(spec, ctx=ctx)
            if use_abi:
                got, spl = xch.shard_records(rec_buf.data_ptr(), n, ix)
            else:
                got, spl = exchange.shard_records_torch(ctx, rec_buf.data_ptr(), n, ix)
            t1 = time.perf_counter()
            ix.finalize()
            t2 = time.perf_counter()
            x_ms, m_ms = (t1 - t0) * 1e3, (t2 - t1) * 1e3
            state.update(shard=ix, splitters=spl, received=got)
        state.update(sh=sh, n_pairs=n)
        return p.level1_ms, p.level1_aux_ms, p.level2_ms, p.total_ms, p.bases_tiled, p.n_level1, x_ms, m_ms

    def sync():
        torch.cuda.synchronize()
        ctx.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    if use_abi or (do_exchange and args.exchange == "abi" and args.backend == "nccl"):
        agree_or_fall_back(bool(create_ok), "pgr_exchange_create did not succeed on every rank within %d s: torch.distributed "
                                            "transport for the whole run" % args.exchange_timeout)
    if use_abi:
        # the first exchange over the library's communicator, outside the clock and under its watchdog: a collective that
        # does not complete aborts the communicator on that rank, and every rank falls back together
        try:
            step()
            ok = 1
        except Exception as e:  # noqa: BLE001
            print("rank %d: first pgr_exchange_shard_records failed (%r): torch.distributed transport instead" % (rank, e),
                  file=sys.stderr)
            ok = 0
        agree_or_fall_back(bool(ok), "the first pgr_exchange_shard_records did not complete on every rank within %d s: "
                                     "torch.distributed transport for the whole run" % args.exchange_timeout)
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    profs = [step() for _ in range(args.steps)]
    sync()
    dt = time.perf_counter() - t0
    total_bp = bp_per_step
    tdev = dev if args.backend == "nccl" else "cpu"
    k = max(1, args.steps)
    x_ms = sum(p[6] for p in profs) / k
    m_ms = sum(p[7] for p in profs) / k
    if use_dist:
        t = torch.tensor([dt, x_ms, m_ms], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, x_ms, m_ms = float(t[0].item()), float(t[1].item()), float(t[2].item())
        t = torch.tensor([float(bp_per_step)], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_bp = int(t.item())

    # ---- after the timed region: every rank proves what it computed and what it received
    cores = effective_cpus()
    sh = state["sh"]
    my_cpu = None
    if not args.no_cpu_baseline:
        try:
            _mm_off = [int(v) for v in sh.offsets()]
            gpu_counts = [_mm_off[i + 1] - _mm_off[i] for i in range(len(contig_ids))]
            # N ranks share the node's CPUs: each checks its own contigs on cores / N threads, ~25 s of CPU work at most
            my_cores = cores if world == 1 else max(1, cores // world)
            my_cpu = cpu_baseline(spec_t, contig_ids, args.contig_len, args.seed, gpu_counts, sh.checksum(), my_cores,
                                  budget_s=None if world == 1 else 25.0)
        except Exception as e:  # noqa: BLE001
            my_cpu = {"error": repr(e)[:300]}
    exch_check = None
    if do_exchange:
        try:
            sent = P.records_checksum(rec_buf.data_ptr(), state["n_pairs"], ctx=ctx)
            shard = state["shard"]
            mine = {"rank": rank, "sent": sent, "n_sent": int(state["n_pairs"]), "shard": shard.records_checksum(),
                    "n_shard": int(shard.n_records), "n_keys": int(shard.n_keys), "key_range": shard.key_range(),
                    "splitters": state["splitters"],
                    "cpu": None if my_cpu is None else {k2: my_cpu.get(k2) for k2 in
                                                        ("contigs_checked", "contigs_with_identical_checksum", "content_match",
                                                         "counts_match_gpu", "error")}}
            allv = [None] * world
            dist.all_gather_object(allv, mine)
            M = (1 << 64) - 1
            s_sent = [sum(v["sent"][i] for v in allv) & M for i in (0, 1)]
            s_shard = [sum(v["shard"][i] for v in allv) & M for i in (0, 1)]
            spl = allv[0]["splitters"]
            ranges_ok = all(v["splitters"] == spl for v in allv)
            for r, v in enumerate(allv):
                if v["n_shard"]:
                    lo, hi = v["key_range"]
                    ranges_ok = ranges_ok and (r == 0 or lo >= spl[r - 1]) and (r == world - 1 or hi < spl[r])
            n_tot = sum(v["n_shard"] for v in allv)
            exch_check = {
                "what": "key-range sharded merge: every pair record went to the rank owning its range of first hashes",
                "transport": "pgr_exchange_shard_records (RCCL behind the C ABI)" if use_abi else "torch.distributed (%s)" % args.backend,
                "rccl_ranks_in_the_librarys_communicator": xch.world if (use_abi and xch is not None and xch.uses_rccl) else 0,
                "exchange_fallback": fallback["note"],
                "records_sent_all_ranks": sum(v["n_sent"] for v in allv), "records_in_shards": n_tot,
                "checksum_of_sent_records": ["%016x" % x for x in s_sent], "checksum_of_shard_records": ["%016x" % x for x in s_shard],
                "content_match": bool(s_sent == s_shard and n_tot == sum(v["n_sent"] for v in allv)),
                "key_ranges_disjoint_and_ordered": bool(ranges_ok),
                "records_per_shard": [v["n_shard"] for v in allv], "keys_per_shard": [v["n_keys"] for v in allv],
                "largest_shard_over_mean": (max(v["n_shard"] for v in allv) * world / n_tot) if n_tot else None,
                "check": "order-independent 128-bit checksum (pgr_records_checksum) of the records every rank produced, summed "
                         "over the ranks, == the same over the records of the finalized shards; shard key ranges against the splitters",
                "ranks_cpu_check": [v["cpu"] for v in allv],
            }
        except Exception as e:  # noqa: BLE001
            exch_check = {"error": repr(e)[:300]}

    # query leg on N GPUs (a failure here must not cost the headline line)
    dist_query = None
    if do_exchange and args.queries > 0 and state.get("shard") is not None:
        try:
            all_ids = list(range(args.contigs)) if args.strong else list(range(world * args.contigs))
            dist_query = query_bench_dist(P, ctx, spec, args, state["shard"], xch if use_abi else None, exchange, world, rank, dist,
                                          torch, local_rank, all_ids)
        except Exception as e:  # noqa: BLE001
            dist_query = {"error": repr(e)[:300]}
    # the merge of step i beside the tiles of step i + 1 -- last thing before the line, under a deadline (a collective that hangs
    # in here must not cost the headline: the line is printed all the same and the process leaves without the usual teardown)
    overlapped, leg_hung = None, False
    if do_exchange and state.get("shard") is not None and not args.no_overlap_leg:
        import threading
        overlapped = {}
        ref_cs, ref_n = state["shard"].records_checksum(), state["shard"].n_records

        def leg():
            try:
                overlapped_exchange_leg(P, exchange, ctx, xch, use_abi, spec, args, contig_ids, lens, rec_buf, torch, dist, dev, tdev, world,
                                        ref_cs, ref_n, overlapped)
            except Exception as e:  # noqa: BLE001
                overlapped["error"] = repr(e)[:300]
        th = threading.Thread(target=leg, daemon=True)
        th.start()
        th.join(timeout=max(90.0, 4.0 * args.exchange_timeout if use_abi else 90.0))
        if th.is_alive():
            leg_hung = True
            overlapped = {"error": "the overlapped leg did not finish within its deadline on rank %d" % rank}
    if rank == 0:
        l1_ms = sum(p[0] for p in profs) / k
        aux_ms = sum(p[1] for p in profs) / k
        l2_ms = sum(p[2] for p in profs) / k
        tot_ms = sum(p[3] for p in profs) / k
        bases_tiled = profs[-1][4] if profs else 0
        achieved = ALGO_BYTES_PER_BP * bases_tiled / (l1_ms * 1e-3) / 1e9 if l1_ms > 0 else 0.0
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
                             "" if not do_exchange else
                             "; then the merge inside the timed step: pair records all-to-all by key range (%s), every rank "
                             "sorts its range into its shard of the frag_map" %
                             ("pgr_exchange_shard_records over RCCL" if use_abi else "torch.distributed " + args.backend)),
                "bp_per_gpu_per_step": bp_per_step, "bp_per_step_all_gpus": total_bp,
                "parallelism": "contig-sharded x%d%s" % (world, ", key-range sharded index" if do_exchange else ""),
                "final_shimmers_per_gpu": sh.count, "pair_records_per_gpu": state["n_pairs"],
            },
            "roofline": {
                "bound": "valu", "kernel": "level1_tile_kernel",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": tr.get("hbm_bytes_per_launch") if tr.get("bp_per_launch") == bases_tiled else None,
                "level1_minimizers": int(profs[-1][5]) if profs else None,
                **(traffic_detail(tr, bases_tiled, sh.count, int(profs[-1][5]) if profs else None) if tr.get("bp_per_launch") == bases_tiled else {}),
                "source": {"achieved, frac, avg_launch_ms": "measured in this run (HIP events around the kernel on its own stream)",
                           "traffic": "REPLAYED from profiles/traffic.json <- profiles/%s/pmc_summary.json (separate --pmc passes of "
                                      "this command), not from this run" % tr.get("profile")},
                "profile_consistency": profile_consistency(tr, l1_ms, bases_tiled),
                "algorithmic_bytes_per_bp": ALGO_BYTES_PER_BP, "bp_per_launch": bases_tiled,
                "avg_launch_ms": l1_ms,
                "valu_issue": valu_issue(bases_tiled, l1_ms),
                "note": "achieved / peak / frac are the HBM-roofline figures of the metric (algorithmic bytes / launch time "
                        "against 8 TB/s).  What binds the kernel is VALU issue (two 64-bit mix hashes per position): see "
                        "valu_issue and DESIGN.md section 5",
            },
            "stage_ms": {"level1_tile": l1_ms, "level1_tail_serial": aux_ms, "level2": l2_ms, "compute_total": tot_ms},
        }
        if overlapped is not None:
            out["overlapped"] = overlapped
            if "value_overlapped" in overlapped:
                out["value_overlapped"] = overlapped["value_overlapped"]
        if do_exchange:
            out["exchange_ms"] = x_ms   # sample + splitters + partition + counts + all-to-all (max over ranks, mean over steps)
            out["merge_ms"] = m_ms      # sort of the rank's key range -> CSR + lookup tables (pgr_index_finalize)
            out["value_definition"] = "all ranks' bases / (shimmers + pair records + exchange_ms + merge_ms), slowest rank"
            out["exchange"] = exch_check
        if my_cpu is not None:
            out["cpu_baseline"] = my_cpu
            if exch_check and "ranks_cpu_check" in exch_check:
                rc_ = [c for c in exch_check["ranks_cpu_check"] if c]
                out["cpu_baseline"]["contigs_checked_all_ranks"] = sum(int(c.get("contigs_checked") or 0) for c in rc_)
                out["cpu_baseline"]["content_match_all_ranks"] = bool(rc_ and all(c.get("content_match") for c in rc_))
        if dist_query is not None:
            out["query"] = dist_query
        if world == 1 and not do_exchange and not args.no_pipelined_leg:
            try:
                out["pipelined"] = pipelined_leg(P, ctx, batch, spec, rec_buf, contig_ids, args.steps, args.warmup, torch, state)
                out["value_pipelined"] = out["pipelined"]["value_pipelined"]
            except Exception as e:  # noqa: BLE001
                out["pipelined"] = {"error": repr(e)[:300]}
        if world == 1 and not do_exchange:
            try:  # what the N > 1 merge costs when there is nothing to exchange: the same records into an index
                reps = []
                for _ in range(3):
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    ixb = P.Index(spec, ctx=ctx)
                    ixb.add_records(device_ptr=rec_buf.data_ptr(), n=state["n_pairs"])
                    ixb.finalize()
                    reps.append(time.perf_counter() - t0)
                    del ixb
                ib = sorted(reps)[1] * 1e3
                out["index_build_ms"] = ib
                out["value_incl_index_build"] = total_bp / ((dt / max(1, args.steps)) + ib * 1e-3) / 1e9
            except Exception as e:  # noqa: BLE001
                out["index_build_ms"] = {"error": repr(e)[:300]}
        if world == 1 and args.queries > 0 and dist_query is None:
            try:
                out["query"], _ix = query_bench(P, ctx, batch, spec, args, contig_ids)
                if not args.no_cpu_baseline:
                    out["query"]["cpu_baseline"] = query_cpu_baseline(P, ctx, spec, spec_t, args, contig_ids, cores, gpu_index=_ix)
                del _ix
            except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
                out.setdefault("query", {})["error"] = repr(e)[:300]
        if world == 1 and not args.no_extras:
            for name, fn in (("latency", lambda: latency_bench(P, ctx, spec, args)),
                             ("pcie_inclusive", lambda: pcie_bench(P, ctx, spec, args)),
                             ("shapes", lambda: shapes_bench(P, ctx, spec, spec_t, cores, not args.no_cpu_baseline)),
                             ("target_100Gbp", lambda: target_100gbp(P, ctx, spec, args, spec_t, cores, not args.no_cpu_baseline))):
                try:
                    out[name] = fn()
                except Exception as e:  # noqa: BLE001
                    out[name] = {"error": repr(e)[:300]}
        try:  # RCCL prints a version banner through C stdio (block buffered on a pipe): push it out BEFORE the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if leg_hung:  # (a collective of the overlapped leg is stuck on this rank: no barrier, no teardown -- the line is out)
        sys.stdout.flush()
        os._exit(0)
    if use_dist:
        dist.barrier()
        state.clear()
        if xch is not None:
            xch.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
