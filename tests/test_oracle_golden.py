"""CPU: the oracle (oracle/pgr_oracle.c) against the reference's golden vectors."""
import os

import numpy as np


def _tuples(mm):
    return [(int(m["x"]) >> 8, (int(m["y"]) & 0xFFFFFFFF) >> 1, int(m["y"]) & 1) for m in mm]


def test_u64hash_kats(oracle):
    assert oracle.u64hash(0) == 0x77CFA1EEF01BCA90
    assert oracle.u64hash(1) == 0x5BCA7C69B794F8CE
    assert oracle.u64hash(0xAD12CF59) == 0x7964DE3EC629529F
    assert oracle.u64hash(2**56 - 1) == 0x3DB304C875DE23B5
    assert oracle.u64hash(2**64 - 1) == 0x1F89206E3F8EC794


def test_golden_mdb_g1(oracle, golden_dir, test_seqs):
    """G1: test_seqs_frag.mdb <= test_seqs.fa through load_from_fastx (gen_frag_db.py):
    820 signatures / 55 keys, global fragment ids, per-key order."""
    gspec, g = oracle.read_mdb(os.path.join(golden_dir, "test_seqs_frag.mdb"))
    assert gspec == (80, 56, 4, 64, 0)
    assert len(g) == 55 and sum(len(v) for v in g.values()) == 820
    ix = oracle.Index(oracle.spec(80, 56, 4, 64))
    for sid, (_, s) in enumerate(test_seqs):
        ix.add_seq(sid, s, fastx_ids=True)
    m = {}
    for x in ix.records():
        m.setdefault((int(x["h0"]), int(x["h1"])), []).append(
            (int(x["frg_id"]), int(x["sid"]), int(x["bgn"]), int(x["end"]), int(x["orient"])))
    assert m == g
    assert ix.n_keys() == 55


def test_golden_midx_g2(golden_dir, test_seqs):
    lines = open(os.path.join(golden_dir, "test_seqs_frag.midx")).read().splitlines()
    assert len(lines) == len(test_seqs) == 66
    for sid, (line, (name, s)) in enumerate(zip(lines, test_seqs)):
        f = line.split("\t")
        assert int(f[0]) == sid and int(f[1]) == len(s) and f[2] == name.decode() and f[3] == "test_seqs.fa"


def test_aggregate_kats(oracle, test_seqs):
    tot = [0, 0, 0]
    for _, s in test_seqs:
        l1 = oracle.level1(s)
        tot[0] += len(l1)
        tot[1] += len(oracle.reduce_shmmr(oracle.reduce_shmmr(l1, 4), 4))
        tot[2] += len(oracle.sequence_to_shmmrs(0, s, oracle.spec()))
    assert tot == [5837, 1254, 886]
    s0 = oracle.sequence_to_shmmrs(0, test_seqs[0][1], oracle.spec())
    assert _tuples(s0)[:4] == [(0x27595DEA8651, 104, 0), (0x138582F2620DB, 285, 0), (0x2180672735B4A, 351, 0),
                               (0x1281CB09ECAA8, 642, 0)]
    assert len(s0) == 14


def test_boundary_condition_g3(oracle, golden_dir):
    """pgr-db/src/lib.rs:342-363: spec (24,24,12,24), padding=true -> exactly 2 shimmers"""
    seqs = [l.strip() for l in open(os.path.join(golden_dir, "boundary_condition_seqs.txt")) if not l.startswith("#")]
    sp = oracle.spec(24, 24, 12, 24)
    outs = [oracle.sequence_to_shmmrs(0, s, sp, padding=True) for s in seqs]
    assert [len(o) for o in outs] == [2, 2]
    assert _tuples(outs[0]) == [(0x89DD27E5C04, 24, 0), (0x56027107815C, 944, 0)]
    assert _tuples(outs[1]) == [(0x89DD27E5C04, 24, 0), (0x56027107815C, 941, 0)]


def test_rc_match_g4(oracle, golden_dir):
    """pgr-db/src/lib.rs:166-180: hashes of a sequence == reversed hashes of its reverse complement"""
    recs = oracle.read_fasta(os.path.join(golden_dir, "test_rev.fa"))
    for sp in (oracle.spec(80, 56, 4, 64, True), oracle.spec(80, 56, 4, 64, False)):
        a = oracle.sequence_to_shmmrs(0, recs[0][1], sp)
        b = oracle.sequence_to_shmmrs(0, recs[1][1], sp)
        assert len(a) > 0
        assert list(a["x"] >> np.uint64(8)) == list((b["x"] >> np.uint64(8))[::-1])


def test_sparse_aln_runs_g5(oracle, golden_dir):
    """aln.rs:458-485 loads test_hits and runs sparse_aln(hp, 8, 0.5, None, false) without asserting values;
    we additionally check structural invariants (every hit in exactly one chain, chains ordered by qb)."""
    h = np.loadtxt(os.path.join(golden_dir, "test_hits"), dtype=np.uint32)
    a = np.zeros(len(h), dtype=oracle.HITPAIR)
    for i, n in enumerate(["qb", "qe", "qo", "tb", "te", "to"]):
        a[n] = h[:, i]
    chains = oracle.sparse_aln(a, 8, 0.5)
    seen = set()
    for score, hps in chains:
        assert len(hps) >= 1
        assert all(hps[i][0] <= hps[i + 1][0] for i in range(len(hps) - 1))
        for hp in hps:
            assert hp not in seen
            seen.add(hp)
    assert len(seen) == len({tuple(int(v) for v in r) for r in h})


# ---------------------------------------------------------------------------------------------------------------
# Chaining has no expected output in the reference (aln.rs:484): the strongest pin available without rustc is two
# restatements written separately that agree.  oracle/pgr_oracle.c (C, arrays) vs oracle/aln_second_reading.py (pure
# Python from the Rust text, dicts / sets keyed by value, numpy float32).
def _chains_equal(c_chains, py_chains):
    """same chains in the same extraction order, scores bit-identical as f32"""
    assert len(c_chains) == len(py_chains)
    for (sc, hc), (sp, hp) in zip(c_chains, py_chains):
        assert np.float32(sc).tobytes() == np.float32(sp).tobytes(), (sc, sp)
        assert [tuple(h) for h in hc] == [h[0] + h[1] for h in hp]


def _load_test_hits(golden_dir):
    h = np.loadtxt(os.path.join(golden_dir, "test_hits"), dtype=np.uint32)
    return [((int(r[0]), int(r[1]), int(r[2])), (int(r[3]), int(r[4]), int(r[5]))) for r in h]


def test_sparse_aln_two_independent_readings_on_test_hits(oracle, golden_dir):
    import aln_second_reading as A2
    hits = _load_test_hits(golden_dir)
    flat = [a + b for a, b in hits]
    # the reference's own call (aln.rs:481), pgr-query's defaults, and two runs that exercise max_gap / orientation
    for max_span, penalty, max_gap, oriented in [(8, 0.5, None, False), (8, 0.025, None, False), (3, 0.1, 50000, True),
                                                 (16, 0.025, 2000, False)]:
        c = oracle.sparse_aln(flat, max_span, penalty, max_gap, oriented)
        p = A2.sparse_aln(hits, max_span, penalty, max_gap, oriented)
        _chains_equal(c, p)
        # and as canonically sorted sets (what survives any tie-break order)
        assert sorted((tuple(map(tuple, hc))) for _, hc in c) == sorted(tuple(h[0] + h[1] for h in hp) for _, hp in p)


def test_sparse_aln_two_independent_readings_on_random_groups(oracle):
    import aln_second_reading as A2
    rng = np.random.default_rng(20260928)
    for case in range(200):
        n = int(rng.integers(2, 60))
        colinear = case % 3 != 0
        hits = []
        q = int(rng.integers(1, 500))
        for _ in range(n):
            ql = int(rng.integers(20, 400))
            if colinear:
                t = q + int(rng.integers(-300, 300)) + 100000
            else:
                t = int(rng.integers(1, 200000))
            o = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            hits.append(((q, q + ql, o[0]), (max(1, t), max(1, t) + ql + int(rng.integers(-5, 6)) + 5, o[1])))
            if rng.random() < 0.15:  # the same query interval on another target interval
                hits.append(((q, q + ql, o[0]), (max(1, t) + 5000, max(1, t) + 5000 + ql, o[1])))
            if rng.random() < 0.7:
                q += int(rng.integers(0, 600))
        rng.shuffle(hits)
        hits = [tuple(map(tuple, h)) for h in hits]
        if len(set(hits)) < len(hits):  # duplicated hit pairs can make the predecessor map cyclic: defined separately
            hits = list(dict.fromkeys(hits))
        if len(hits) < 2:
            continue
        max_span = int(rng.integers(1, 12))
        penalty = float(rng.choice([0.0, 0.025, 0.5, 1.5]))
        max_gap = None if rng.random() < 0.5 else int(rng.integers(100, 5000))
        oriented = bool(rng.integers(0, 2))
        c = oracle.sparse_aln([a + b for a, b in hits], max_span, penalty, max_gap, oriented)
        p = A2.sparse_aln(hits, max_span, penalty, max_gap, oriented)
        _chains_equal(c, p)


def test_query_fragment_to_hps_two_independent_readings(oracle, test_seqs):
    """raw_query_fragment + count filters + chaining (seq_db.rs:1200-1228, aln.rs:147-242): the C oracle's index and
    query against the dict-of-lists reading, on the reference's own test sequences (repeat rich: the count filters fire)"""
    import aln_second_reading as A2
    sp = oracle.spec(80, 56, 4, 64)
    seqs = [s for _, s in test_seqs]
    ix = oracle.Index(sp)
    frag_map = {}
    for sid, s in enumerate(seqs):
        ix.add_seq(sid, s)
        for r in oracle.frag_recs(oracle.sequence_to_shmmrs(sid, s, sp), sid):
            frag_map.setdefault((int(r["h0"]), int(r["h1"])), []).append(
                (int(r["frg_id"]), int(r["sid"]), int(r["bgn"]), int(r["end"]), int(r["orient"])))
    ix.finalize()
    rng = np.random.default_rng(7)
    n_checked = 0
    for case in range(24):
        src = seqs[int(rng.integers(0, len(seqs)))]
        a = int(rng.integers(0, max(1, len(src) - 1500)))
        q = src[a:a + int(rng.integers(800, 3000))]
        kw = [dict(), dict(max_count=2, query_max_count=2, target_max_count=2), dict(max_aln_span=2, oriented=True),
              dict(max_gap=500, target_max_count=1)][case % 4]
        ref = ix.query_fragment_to_hps(q, 0.025, **kw)
        mm = oracle.sequence_to_shmmrs(0, q, sp)
        shm = [(int(x) >> 8, (int(y) & 0xFFFFFFFF) >> 1) for x, y in zip(mm["x"], mm["y"])]
        got = A2.query_fragment_to_hps(A2.raw_query_fragment(frag_map, shm), 0.025, **kw)
        assert sorted(got) == [sid for sid, _ in ref]
        for sid, chains in ref:
            _chains_equal([(sc, hps) for sc, hps in chains], got[sid])
            n_checked += len(chains)
    assert n_checked > 100
