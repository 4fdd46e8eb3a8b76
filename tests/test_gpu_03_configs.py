"""BASELINE.json's configs at FULL size, each against the CPU restatement (oracle/), collected right behind the core parity files
so that a driver run with -x reaches them before any process plumbing:

  configs[0]  pgr-mdb on 10 x 1 Mbp synthetic contigs, seed 1 (the reference's own CPU-runnable case)
  configs[1]  sequence_to_shmmrs on 1000 x 10 Mbp (pgr-db/src/shmmrutils.rs:657-669) -- the headline workload
  configs[2]  SeqIndexDB build + query_fragment_to_hps, 10 000 x 10 kbp queries against the 10 Gbp index (aln.rs:147-242)
  configs[3]  the 96-haplotype AMY1A-like region: index + query (here) and MAP-graph / principal bundles (test_gpu_08_mapgraph.py)
  configs[4]  one GPU's 3525 x 10 Mbp slice of the 94 x 3 Gbp build
"""
import hashlib
import json
import os
import time

import numpy as np
import pytest

import procutil
import seqgen
from test_gpu_01_index_query import _build_pair, _oracle_hps_to_tuples, revcomp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "pgr-tk_amd", "bin")


def _canonical_mdb_hash(m):
    """sha256 over the canonically sorted content of an .mdb (keys ascending, per-key signatures in file order)"""
    h = hashlib.sha256()
    for key in sorted(m):
        h.update(np.array(key, dtype="<u8").tobytes())
        h.update(np.array(m[key], dtype="<u4").tobytes())
    return h.hexdigest()


def test_config1_pgr_mdb_synthetic_10x1mbp_seed1(oracle, gpu_ctx, tmp_path):
    """BASELINE.json configs[0] / BASELINE.md section 4 row 1: pgr-mdb on 10 x 1 Mbp synthetic contigs, seed 1,
    ShmmrSpec (80, 56, 4, 64).  The C++ host program with `--synthetic 10x1000000 --seed 1` (contigs generated on the
    device), the Python CLI with the same flags, and the C++ program on the FASTA that `--write-fasta` produced all write
    the same .mdb, and its content equals the frag_map of the CPU restatement (pgr-db/src/seq_db.rs:541-615: index-only
    path, per-contig fragment ids) built from the oracle's own generator.  Reported: the oracle's Gbp/s on 1 thread and
    on all CPUs the process may use, and the content hash."""
    from pgrtk_amd import cli
    N, L, SEED = 10, 1_000_000, 1
    fa = str(tmp_path / "synth.fa")
    p_cpp, p_py, p_fa = (str(tmp_path / n) for n in ("cpp", "py", "fa"))
    r = procutil.run_bounded([os.path.join(BIN, "pgr-mdb"), "--synthetic", "%dx%d" % (N, L), "--seed", str(SEED), "--write-fasta", fa,
                        p_cpp], timeout=180)
    assert r.returncode == 0, r.stderr
    cli.main(["mdb", "--synthetic", "%dx%d" % (N, L), "--seed", str(SEED), p_py])
    lst = tmp_path / "list.txt"
    lst.write_text(fa + "\n")
    r = procutil.run_bounded([os.path.join(BIN, "pgr-mdb"), str(lst), p_fa], timeout=180)
    assert r.returncode == 0, r.stderr
    mdb = open(p_cpp + ".mdb", "rb").read()
    assert mdb == open(p_py + ".mdb", "rb").read() == open(p_fa + ".mdb", "rb").read()
    assert open(p_cpp + ".midx").read() == open(p_py + ".midx").read()
    midx = [l.split("\t") for l in open(p_cpp + ".midx").read().splitlines()]
    assert midx == [[str(c), str(L), "synth_%d_%d" % (SEED, c), "synthetic:%dx%d:seed=%d" % (N, L, SEED)] for c in range(N)]
    assert [l.split("\t")[:3] for l in open(p_fa + ".midx").read().splitlines()] == [m[:3] for m in midx]
    # the FASTA holds the generator's bytes
    recs = oracle.read_fasta(fa)
    assert len(recs) == N and all(s == oracle.synth_contig(SEED, c, L).tobytes() for c, (_, s) in enumerate(recs))

    # the CPU restatement's frag_map of the same contigs: 1 thread, then all CPUs (one task per contig = rayon par_iter)
    sp = oracle.spec(80, 56, 4, 64)
    rates = {}
    import bench
    n_cpu = bench.effective_cpus()  # scheduler affinity capped by the cgroup quota (the GPU boxes show 256 CPUs and grant 16)
    for threads in (1, n_cpu):
        oix = oracle.Index(sp)
        t0 = time.perf_counter()
        oix.add_synth_threads(N, 0, SEED, 0, L, threads)
        rates[threads] = N * L / (time.perf_counter() - t0) / 1e9
    ref = oix.records()
    spec_t, m = oracle.read_mdb(p_cpp + ".mdb")
    assert spec_t == (80, 56, 4, 64, 0)
    exp = {}
    for rr in ref:  # sorted by (h0, h1, sid, frg_id): per-key order = insertion order of seq_db.rs:605-612
        exp.setdefault((int(rr["h0"]), int(rr["h1"])), []).append((int(rr["frg_id"]), int(rr["sid"]), int(rr["bgn"]),
                                                                   int(rr["end"]), int(rr["orient"])))
    assert m == exp and sum(len(v) for v in m.values()) == len(ref) > 25000
    print("\nconfigs[0]: 10 x 1 Mbp seed 1 -> %d pair records, %d keys; oracle %.4f Gbp/s on 1 thread, %.4f Gbp/s on %d threads; "
          ".mdb canonical content sha256 %s" % (len(ref), len(m), rates[1], rates[n_cpu], n_cpu, _canonical_mdb_hash(m)))


def test_config2_full_size(oracle, gpu_ctx):
    """BASELINE.json configs[1] at full size: 1000 x 10 Mbp (seed 2), the final shimmer lists of ALL contigs have the
    checksums the CPU restatement computes for the same contigs (128 bits per contig, order sensitive)"""
    import bench
    import pgrtk_amd as P
    n, L, seed = 1000, 10_000_000, 2
    spec = P.make_spec(80, 56, 4, 64)
    batch = P.Batch.synthetic([L] * n, seed=seed, ctx=gpu_ctx)
    sh = batch.shmmrs(spec)
    off = sh.offsets()
    gpu_counts = (off[1:] - off[:-1]).astype(np.uint64)
    gpu_sums = sh.checksum()
    cores = bench.effective_cpus()
    counts, sums, _ = oracle.synth_checksums_threads(oracle.spec(80, 56, 4, 64), n, seed, 0, L, cores)
    assert np.array_equal(counts, gpu_counts)
    bad = np.nonzero(~np.all(sums == gpu_sums, axis=1))[0]
    assert bad.size == 0, "contigs with a different shimmer list: %s" % bad[:10]
    assert 2.9e7 < int(gpu_counts.sum()) < 3.2e7  # ~3.0e7 final shimmers


def test_config3_full_size(oracle, gpu_ctx):
    """BASELINE.json configs[2] at full size: GPU index of the 1000 x 10 Mbp contigs, 10 000 x 10 kbp queries (half reverse
    complemented).  The CPU restatement builds the index of a 64-contig subset; 1024 more queries cut from that subset go
    through both, chain for chain (targets, chains, hit pairs, f32 score bits).  Every query of the big batch finds its
    source contig."""
    import bench
    import pgrtk_amd as P
    n, L, seed, S = 1000, 10_000_000, 2, 64
    spec = P.make_spec(80, 56, 4, 64)
    ids = list(range(n))
    batch = P.Batch.synthetic([L] * n, seed=seed, ctx=gpu_ctx)
    ix = P.Index(spec, ctx=gpu_ctx)
    ix.add_resident(batch, sids=ids)
    ix.finalize()
    del batch
    assert 2.8e7 < ix.n_records < 3.2e7
    rng = np.random.default_rng(3)
    cs, offs, qs = bench.make_queries(P, seed, ids, n, L, 10_000, 10_000, rng)
    r = ix.query_hps_raw(qs, 0.025)
    ok = 0
    for qi in range(10_000):
        sids = r["t_sid"][int(r["q_off"][qi]):int(r["q_off"][qi + 1])]
        ok += int(int(cs[qi]) in set(int(v) for v in sids))
    assert ok >= 9_990, ok  # (a 10 kbp window holds >= 2 shimmer pairs of its source with near certainty)
    # oracle on a subset
    cores = bench.effective_cpus()
    oix = oracle.Index(oracle.spec(80, 56, 4, 64))
    oix.add_synth_threads(S, 0, seed, 0, L, cores)
    oix.finalize()
    rng2 = np.random.default_rng(31)
    cs2, offs2, qs2 = bench.make_queries(P, seed, ids[:S], S, L, 1024, 10_000, rng2)
    qlist = [qs2.buf[int(qs2.off[i]):int(qs2.off[i + 1])] for i in range(1024)]
    ref, _ = oracle.query_batch_threads(oix, qlist, 0.025, cores)
    r2 = ix.query_hps_raw(qs2, 0.025)
    n_same = 0
    for qi in range(1024):
        want = [(sid, [(np.float32(sc).tobytes(), [tuple(h) for h in hps]) for sc, hps in chains]) for sid, chains in ref[qi]]
        got = [(sid, ch) for sid, ch in bench.chains_of(r2, qi) if sid < S]  # the full index may add other targets
        n_same += int(got == want)
    assert n_same == 1024, n_same


def test_pangenome_config4_index_and_query(oracle, gpu_ctx):
    """BASELINE.json configs[3] input (96 AMY1A-like haplotypes, tandem 10 kbp copies, 0.1 % SNPs) at the
    pgr-pbundle-decomp spec (48,56,4,12): frag_map records and hit chains equal the oracle's.  Repeats
    make every key occur ~96 x copies times, which is what the count filters of aln.rs:203-222 cut on."""
    import ctypes as C
    from pgrtk_amd import _ffi
    haps = seqgen.amy1a_like(seed=4, n_hap=96, L=200_000)
    spec_t = (48, 56, 4, 12)
    sdb, oix = _build_pair(oracle, gpu_ctx, haps, spec_t)
    ref = oix.records()
    p, n = C.c_void_p(), C.c_uint64()
    gpu_ctx.check(_ffi.lib().pgr_index_download(gpu_ctx.handle, sdb._ix, C.byref(p), C.byref(n)))
    got = _ffi.take(p, int(n.value), _ffi.FRAG_REC)
    assert len(got) == len(ref) > 96 * 1000
    for f in ("h0", "h1", "frg_id", "sid", "bgn", "end", "orient"):
        assert np.array_equal(ref[f], got[f]), f
    # queries: a unique flank, the repeat unit, and a reverse-complemented slice across the repeat boundary
    h0 = haps[0]
    queries = [h0[20_000:45_000], h0[100_000:112_000], revcomp(h0[90_000:125_000])]
    n_chains = 0
    for q in queries:
        for cap in (128, 4096):
            got_h = sdb.query_fragment_to_hps(q, 0.025, cap, cap, cap, 8)
            ref_h = _oracle_hps_to_tuples(oix.query_fragment_to_hps(q, 0.025, cap, cap, cap, 8))
            assert got_h == ref_h
            n_chains += sum(len(c) for _, c in ref_h)
    assert n_chains > 96


def test_config5_slice_one_gpus_share(oracle, gpu_ctx):
    """BASELINE.json configs[4] (94 x 3 Gbp over 8 GPUs) as ONE GPU's share: 3525 x 10 Mbp (seed 5) streamed in 4 resident
    batches into one index.  Density, sortedness, record count == sum(shimmers - 1), and content == the CPU checker on
    48 sampled contigs (128-bit checksums of their shimmer lists)."""
    import json
    import time
    import pgrtk_amd as P
    n_total, L, seed = 3525, 10_000_000, 5
    sp = P.make_spec()
    ix = P.Index(sp, ctx=gpu_ctx)
    rng = np.random.default_rng(55)
    sample = sorted(int(v) for v in rng.choice(n_total, 48, replace=False))
    sums, counts, n_shmmr, n_pairs = {}, {}, 0, 0
    t_shmmr = 0.0
    t0 = time.perf_counter()
    for b0 in range(0, n_total, 900):
        ids = list(range(b0, min(n_total, b0 + 900)))
        batch = P.Batch.synthetic([L] * len(ids), seed=seed, contig0=b0, ctx=gpu_ctx)
        t1 = time.perf_counter()
        sh = batch.shmmrs(sp)
        t_shmmr += time.perf_counter() - t1
        cs, off = sh.checksum(), sh.offsets()
        for c in sample:
            if b0 <= c < b0 + len(ids):
                sums[c] = cs[c - b0].copy()
                counts[c] = int(off[c - b0 + 1] - off[c - b0])
        n_shmmr += sh.count
        n_pairs += sh.n_pairs
        del sh
        ix.add_resident(batch, sids=ids)
        batch.close()
    ix.finalize()
    t_all = time.perf_counter() - t0
    bp = n_total * L
    assert ix.n_records == n_pairs == n_shmmr - n_total
    assert 0.0029 < n_shmmr / bp < 0.0032  # 0.003035 final shimmers per base
    recs = ix.download()
    key = recs["h0"].astype(np.uint64)
    assert bool(np.all(key[1:] >= key[:-1]))
    same = (recs["h0"][1:] == recs["h0"][:-1]) & (recs["h1"][1:] == recs["h1"][:-1])
    assert bool(np.all(recs["h1"][1:][recs["h0"][1:] == recs["h0"][:-1]] >= recs["h1"][:-1][recs["h0"][1:] == recs["h0"][:-1]]))
    assert bool(np.all(recs["sid"][1:][same] >= recs["sid"][:-1][same]))
    # content: the checker generates the sampled contigs itself
    osp = oracle.spec()
    for c in sample:
        ref = oracle.sequence_to_shmmrs(0, oracle.synth_contig(seed, c, L), osp)
        assert len(ref) == counts[c] and np.array_equal(oracle.shmmr_checksum(ref), sums[c]), c
    line = {"workload": "configs[4] slice of one GPU: %d x %d bp, seed %d, 4 resident batches into one index" % (n_total, L, seed),
            "bp": bp, "shimmers": n_shmmr, "records": int(ix.n_records), "keys": int(ix.n_keys),
            "shmmr_s": t_shmmr, "total_s_incl_generation_and_sort": t_all, "Gbp_per_s_shimmers": bp / t_shmmr / 1e9,
            "contigs_content_checked": len(sample)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "config5_slice.json"), "w") as f:
        json.dump(line, f)
    print(json.dumps(line))
