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
