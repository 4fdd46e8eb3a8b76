"""Round-4 parity tests (run with -m gpu on a MI355X): BASELINE.json configs[0] through the pgr-mdb counterparts
(`--synthetic NxL --seed S`), the remaining SeqIndexDB methods of row H3."""
import hashlib
import os
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "pgr-tk_amd", "bin")


@pytest.fixture(scope="module")
def gpu_ctx():
    import pgrtk_amd as P
    return P.default_context(0)


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
    r = subprocess.run([os.path.join(BIN, "pgr-mdb"), "--synthetic", "%dx%d" % (N, L), "--seed", str(SEED), "--write-fasta", fa,
                        p_cpp], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    cli.main(["mdb", "--synthetic", "%dx%d" % (N, L), "--seed", str(SEED), p_py])
    lst = tmp_path / "list.txt"
    lst.write_text(fa + "\n")
    r = subprocess.run([os.path.join(BIN, "pgr-mdb"), str(lst), p_fa], capture_output=True, text=True, timeout=600)
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


def test_seqindexdb_remaining_methods(oracle, gpu_ctx, golden_dir, tmp_path):
    """pgr-tk/src/lib.rs:24 pgr_lib_version, :1337 write_midx_to_text_file, :1374 write_frag_and_index_files -- the
    last one is the call tests/golden's `test_seqs_frag.mdb` was made with (gen_frag_db.py): its .mdb/.midx half
    reproduces the golden files' content."""
    import pgrtk_amd as P
    assert P.pgr_lib_version().startswith("pgr-hip ")
    fa = os.path.join(golden_dir, "test_seqs.fa")
    sdb = P.SeqIndexDB(ctx=gpu_ctx)
    sdb.load_from_fastx(fa)
    prefix = str(tmp_path / "test_seqs_frag")
    sdb.write_frag_and_index_files(prefix)
    gspec, g = oracle.read_mdb(os.path.join(golden_dir, "test_seqs_frag.mdb"))
    spec2, g2 = oracle.read_mdb(prefix + ".mdb")
    assert (spec2, g2) == (gspec, g) and os.path.getsize(prefix + ".mdb") == 15291
    ref_midx = [l.split("\t")[:3] for l in open(os.path.join(golden_dir, "test_seqs_frag.midx")).read().splitlines()]
    assert [l.split("\t")[:3] for l in open(prefix + ".midx").read().splitlines()] == ref_midx
    a, b = str(tmp_path / "a.idx"), str(tmp_path / "b.idx")
    sdb.write_mapg_idx(a)
    sdb.write_midx_to_text_file(b)
    assert open(a).read() == open(b).read() and open(a).read().startswith("K\t80\t56\t4\t64\tfalse\n")
    # an index-file backend holds no sequences: the reference's `seq_db.is_some()` is false and nothing is written
    mdb = P.SeqIndexDB(ctx=gpu_ctx)
    mdb.load_from_mdb_index(prefix)
    mdb.write_frag_and_index_files(str(tmp_path / "none"))
    assert not os.path.exists(str(tmp_path / "none.mdb"))


def test_bench_self_spawns_its_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with NO torchrun around it (the shape of the driver's N = 1 command with another N) starts
    its own ranks and prints a gradeable line: 2 ranks on this box's one GPU (gloo transport), merge inside the value,
    every rank's contigs checked, the exchange verified, the transport named."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device", "--steps", "2",
           "--warmup", "1", "--contigs", "30", "--contig-len", "2000000", "--queries", "200"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.split("\n") if ln.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["bp_per_step_all_gpus"] == 120_000_000
    assert line["merge_ms"] > 0 and line["exchange_ms"] > 0
    ex = line["exchange"]
    assert ex["content_match"] is True and ex["key_ranges_disjoint_and_ordered"] is True
    assert ex["transport"].startswith("torch.distributed") and ex["exchange_fallback"] is None
    assert ex["rccl_ranks_in_the_librarys_communicator"] == 0  # (gloo on one device: the library's RCCL path is not taken)
    assert line["cpu_baseline"]["content_match_all_ranks"] is True


def test_exchange_watchdog_times_out_instead_of_hanging(gpu_ctx):
    """a rank whose peers never arrive: ncclCommInitRank for world = 2 with only this rank present would block for ever;
    with the context option exchange_timeout_s the call comes back with an error that names the timeout (the bench then
    falls back to the torch.distributed transport).  On its own context: the worker thread stays inside RCCL."""
    import ctypes as C
    import pgrtk_amd as P
    from pgrtk_amd._ffi import lib
    ctx = P.Context(0)
    ctx.set_option("exchange_timeout_s", 3)
    assert ctx.get_option("exchange_timeout_s") == 3
    idb = np.zeros(128, dtype=np.uint8)
    ctx.check(lib().pgr_exchange_unique_id(ctx.handle, idb.ctypes.data))
    h = C.c_void_p()
    t0 = time.perf_counter()
    rc = lib().pgr_exchange_create(ctx.handle, idb.ctypes.data, 0, 2, C.byref(h))
    dt = time.perf_counter() - t0
    assert rc != 0 and not h.value and 2.5 < dt < 30
    assert b"did not return within 3 s" in lib().pgr_last_error(ctx.handle)
    with pytest.raises(KeyError):
        ctx.get_option("no_such_option")
    with pytest.raises(P.PgrError):
        ctx.set_option("no_such_option", 1)


def test_index_sorts_non_canonical_external_records(gpu_ctx):
    """pgr_index_add_records takes records from outside (other GPUs, other programs): nothing says h0 <= h1 there.  The
    radix passes are bounded by max(h0, h1) over ALL records (round 3 used max(h1): a record whose h0 exceeded every h1
    lost its high bits and the CSR came out mis-sorted).  Appended in (sid, frg_id) order (one-key sort + run fix-ups) and
    in random order (four-field sort): both must equal numpy's lexicographic order."""
    import pgrtk_amd as P
    rng = np.random.default_rng(99)
    n = 50_000
    recs = np.zeros(n, dtype=P.FRAG_REC)
    recs["h0"] = rng.integers(0, 1 << 40, n, dtype=np.uint64)
    recs["h1"] = rng.integers(0, 1 << 30, n, dtype=np.uint64)   # mostly h0 > h1: not canonical
    recs["h0"][::97] = rng.integers(1 << 54, 1 << 56, len(recs["h0"][::97]), dtype=np.uint64)  # above every h1, up to 56 bits
    recs["h0"][5] = (1 << 63) + 12345                            # beyond the library's own 56-bit hashes
    recs["h0"][1000:1040] = 777                                  # a run of equal h0 with mixed h1
    recs["sid"] = np.sort(rng.integers(0, 50, n)).astype(np.uint32)
    recs["frg_id"] = np.arange(n, dtype=np.uint32)
    recs["bgn"] = rng.integers(0, 1 << 20, n)
    recs["end"] = recs["bgn"] + 100
    want = np.sort(recs, order=["h0", "h1", "sid", "frg_id"])
    for order in (np.arange(n), rng.permutation(n)):
        ix = P.Index(P.make_spec(), ctx=gpu_ctx)
        ix.add_records(recs[order])
        ix.finalize()
        got = ix.download()
        for f in ("h0", "h1", "sid", "frg_id", "bgn", "end"):
            assert np.array_equal(got[f], want[f]), f


def test_eight_ranks_strong_scaling_plumbing_on_one_device():
    """world = 8 before an 8-GPU node ever runs it: `bench.py --gpus 8 --strong` (self-spawned, gloo, every rank on this box's one
    GPU) partitions ONE contig set with the greedy partitioner, every rank computes its shard, the records travel by key range
    through an 8 x 8 all-to-all, eight shards are sorted and all-gathered into the replicated query index.  Every rank's
    contigs are checked against the CPU restatement, the exchanged record set against what was sent.  (The same command at
    full size -- 1000 x 10 Mbp, 30.4 M records -- is kept under profiles/r04_dist/.)"""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--single-device", "--strong", "--steps", "1",
           "--warmup", "1", "--contigs", "67", "--contig-len", "1500000", "--queries", "240"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.split("\n") if ln.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["config"]["bp_per_step_all_gpus"] == 67 * 1_500_000
    ex = line["exchange"]
    assert ex["content_match"] is True and ex["key_ranges_disjoint_and_ordered"] is True and len(ex["records_per_shard"]) == 8
    assert ex["records_sent_all_ranks"] == ex["records_in_shards"] == sum(ex["records_per_shard"]) > 250_000
    assert ex["largest_shard_over_mean"] < 1.25
    cb = line["cpu_baseline"]
    assert cb["content_match_all_ranks"] is True and cb["contigs_checked_all_ranks"] == 67
    q = line["query"]
    assert "error" not in q and q["queries_with_best_chain_on_source"] >= 236 and q["index_records"] == ex["records_in_shards"]



def _same(ref, got, what=""):
    assert len(ref) == len(got), "%s: %d vs %d shimmers" % (what, len(ref), len(got))
    assert np.array_equal(ref["x"], got["x"]) and np.array_equal(ref["y"], got["y"]), what


def test_one_wavefront_tiles_for_batches_of_short_contigs(oracle, gpu_ctx):
    """csrc/level1.hip: a batch whose mean contig length is <= 2048 (reads) runs the level-1 kernel with ONE wavefront per
    tile of 1024 positions; a contig of up to ext - 16 positions is one tile whose extended range starts at position 0.
    Lengths around every boundary of that geometry (one tile / two tiles, tile core, k, w), several specs incl. the sketch
    variant, non-ACGT bytes and palindromic k-mers (islands of the exact machine on the short geometry), a long contig in
    the batch; compared with the oracle and with the same call on the 4096-position tiles."""
    import pgrtk_amd as P
    import seqgen
    rng = np.random.default_rng(404)
    for spec_t in ((80, 56, 4, 64, False), (48, 56, 4, 12, False), (33, 31, 3, 8, False), (128, 56, 2, 64, False), (17, 9, 2, 0, False),
                   (80, 56, 1, 64, False), (80, 21, 2, 16, True)):
        w, k, r, ms, sk = spec_t
        tc = (1024 - 2 * ((1 if sk else w) - 1)) // 64 * 64
        lens = [0, 1, k - 1, k, k + 1, k + w - 2, k + w - 1, k + w, 2 * w + k, 3 * w, 500, 999, 1000, 1001, 1007, 1008, 1009, 1023, 1024,
                1025, tc - 1, tc, tc + 1, 2 * tc - w, 2 * tc - 1, 2 * tc, 2 * tc + 1, 2 * tc + w + k, 3 * tc, 3000, 4017, 5000]
        lens = [n for n in lens if n >= 0]
        seqs = [seqgen.rnd(rng, n) for n in lens] * 3
        # adversarial short contigs: N runs, lower case, bytes 0..3, low complexity, palindromic stretches, tandem repeats
        for mode in range(seqgen.N_MODES):
            for L in (300, 900, 1008, 1100, 1700, 2600):
                seqs.append(seqgen.adversarial(rng, mode, L))
        seqs.append(seqgen.rnd(rng, 400) + b"N" * 700 + seqgen.rnd(rng, 500))
        seqs.append(b"N" * 1008)
        seqs.append(b"AT" * 504)
        seqs.append(b"A" * 1000)
        seqs.append(seqgen.rnd(rng, 30_000))  # one long contig among the reads (mean stays below 2048)
        seqs.append(seqgen.rnd(rng, 9_000) + b"N" * 3000 + b"AT" * 90 + seqgen.rnd(rng, 9_000))
        assert sum(len(s) for s in seqs) / len(seqs) <= 2048
        rids = [int(v) for v in rng.integers(0, 2 ** 31, len(seqs))]
        spec = P.make_spec(w, k, r, ms, sketch=sk) if sk else P.make_spec(w, k, r, ms)
        osp = oracle.spec(w, k, r, ms, sketch=sk) if sk else oracle.spec(w, k, r, ms)
        with gpu_ctx.options(no_small_path=1):
            short = P.sequence_to_shmmrs_batch(seqs, spec, rids=rids, ctx=gpu_ctx)
            with gpu_ctx.options(no_short_tiles=1):
                long_ = P.sequence_to_shmmrs_batch(seqs, spec, rids=rids, ctx=gpu_ctx)
        for i, s in enumerate(seqs):
            ref = oracle.sequence_to_shmmrs(rids[i], s, osp)
            _same(ref, short[i], "one-wavefront tiles, spec %s contig %d len %d" % (spec_t, i, len(s)))
            _same(ref, long_[i], "4096-position tiles, spec %s contig %d len %d" % (spec_t, i, len(s)))
    # resident batch of 20 000 reads of ragged length (past the small path's 4096 contigs): whole-contig checksums
    n = 20_000
    lens = [int(v) for v in rng.integers(700, 1400, n)]
    b = P.Batch.synthetic(lens, seed=43, ctx=gpu_ctx)
    spec_t = (80, 56, 4, 64)
    sh = b.shmmrs(P.make_spec(*spec_t))
    with gpu_ctx.options(no_short_tiles=1):
        sh4 = b.shmmrs(P.make_spec(*spec_t))
    assert sh.count == sh4.count and np.array_equal(sh.checksum(), sh4.checksum()) and np.array_equal(sh.offsets(), sh4.offsets())
    sums, off = sh.checksum(), sh.offsets()
    import bench
    for c in range(0, n, 97):
        ref = oracle.sequence_to_shmmrs(c, bench.synth_contig_ascii(43, c, lens[c]), oracle.spec(*spec_t))
        assert int(off[c + 1] - off[c]) == len(ref) and np.array_equal(sums[c], oracle.shmmr_checksum(ref)), c


def test_many_dense_short_contigs_take_the_retry_paths(oracle, gpu_ctx):
    """20 000 low-complexity reads (homopolymers, (AC)n, two-letter noise: ties emit every position, shmmrutils.rs:516-527): the
    level-1 overflow region and the list stage's grid are sized from the density of random sequence plus a small allowance per
    contig, so this batch overflows both and runs again with the true counts -- still bit exact, and the next call (which
    remembers what this one needed) as well."""
    import pgrtk_amd as P
    rng = np.random.default_rng(91)
    seqs = []
    for i in range(20_000):
        L = int(rng.integers(150, 1200))
        m = i % 4
        if m == 0:
            seqs.append(b"A" * L)
        elif m == 1:
            seqs.append((b"AC" * (L // 2 + 1))[:L])
        elif m == 2:
            seqs.append(bytes(rng.choice(np.frombuffer(b"AC", dtype=np.uint8), L)))
        else:
            seqs.append(bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), L)))
    spec_t = (80, 56, 4, 64)
    for attempt in range(2):
        got = P.sequence_to_shmmrs_batch(seqs, P.make_spec(*spec_t), ctx=gpu_ctx)
        for i in range(0, len(seqs), 7):
            ref = oracle.sequence_to_shmmrs(i, seqs[i], oracle.spec(*spec_t))
            _same(ref, got[i], "attempt %d contig %d len %d" % (attempt, i, len(seqs[i])))


def test_island_options_agree_with_the_oracle(oracle, gpu_ctx):
    """the exact islands' variants (context options): chunks shorter than a tile / the round-3 minimum of 4096
    positions, islands listed beside the tile kernel / behind it, the state relay on / off -- on contigs with gaps, isolated N,
    palindromic arrays and low-complexity stretches, through the host entry point (the packer counts the non-ACGT bytes: the
    islands are listed while the tiles run) and as a resident batch; every variant bit-identical to the oracle."""
    import pgrtk_amd as P
    import seqgen
    rng = np.random.default_rng(606)

    def contig(L):
        parts, n = [], 0
        while n < L:
            r = rng.random()
            if r < 0.55:
                p = seqgen.rnd(rng, int(rng.integers(2_000, 60_000)))
            elif r < 0.7:
                p = b"N" * int(rng.integers(1, 30_000))
            elif r < 0.8:
                p = (b"AT", b"ACGT", b"GAATTC")[int(rng.integers(0, 3))] * int(rng.integers(50, 3000))
            elif r < 0.9:
                p = seqgen.rnd(rng, int(rng.integers(100, 9000)), b"AC")
            else:
                p = seqgen.rnd(rng, 500) + b"N" + seqgen.rnd(rng, 700)
            parts.append(p)
            n += len(p)
        return b"".join(parts)[:L]
    seqs = [contig(700_000), contig(150_000), seqgen.rnd(rng, 50_000), b"N" * 40_000 + seqgen.rnd(rng, 30_000) + b"N" * 9_000, contig(20_000)]
    spec_t = (80, 56, 4, 64)
    spec, osp = P.make_spec(*spec_t), oracle.spec(*spec_t)
    refs = [oracle.sequence_to_shmmrs(i, s, osp) for i, s in enumerate(seqs)]
    variants = [{}, {"no_pre_islands": 1}, {"island_chunk_min": 4096}, {"island_chunk_min": 1024, "no_island_relay": 1},
                {"island_chunk_min": 32768}, {"no_short_tiles": 1, "no_small_path": 1},
                # round 5: the first round of the islands around non-ACGT bytes beside the tile kernel (default) / behind it on
                # its stream / only when the flags are in; a batch with a palindromic array drops the early round and starts over
                {"early_islands_in_stream": 1}, {"no_early_islands": 1}]
    for opt in variants:
        with gpu_ctx.options(**dict(opt, no_small_path=1)):
            got = P.sequence_to_shmmrs_batch(seqs, spec, ctx=gpu_ctx)  # host entry point
            b = P.Batch.from_seqs(seqs, ctx=gpu_ctx)
            sh = b.shmmrs(spec)
        for i in range(len(seqs)):
            _same(refs[i], got[i], "host entry, options %s, contig %d" % (opt, i))
        sums, off = sh.checksum(), sh.offsets()
        for i in range(len(seqs)):
            assert int(off[i + 1] - off[i]) == len(refs[i]) and np.array_equal(sums[i], oracle.shmmr_checksum(refs[i])), (opt, i)
