"""CPU: host-side logic of the pgr-query counterpart (range merging, pgr-bin/src/bin/pgr-query.rs:167-285)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))


def _hp(qb, tb, ln=100, qo=0, to=0):
    return ((qb, qb + ln, qo), (tb, tb + ln, to))


def test_chains_to_regions_merge_and_quirks():
    from pgrtk_amd.cli import chains_to_regions
    fwd1 = [_hp(0, 1000), _hp(200, 1200), _hp(400, 1400)]
    fwd2 = [_hp(600, 50000), _hp(800, 50200), _hp(1000, 50400)]
    far = [_hp(1200, 900000), _hp(1400, 900200), _hp(1600, 900400)]
    short = [_hp(0, 5), _hp(10, 15)]  # <= 2 hit pairs: dropped (pgr-query.rs:176)
    rev = [_hp(0, 7000, qo=0, to=1), _hp(200, 6800, qo=0, to=1), _hp(400, 6600, qo=0, to=1)]
    res = chains_to_regions([(3, [(10.0, fwd1), (9.0, fwd2), (8.0, far), (1.0, short)])], 100000)
    assert list(res) == [3]
    r = res[3]
    assert [(x[0], x[1], x[3], len(x[4])) for x in r] == [(1000, 50500, 0, 6), (900000, 900500, 0, 3)]
    # orientation vote uses running counters that are not reset per chain (quirk kept from the reference)
    res = chains_to_regions([(1, [(5.0, fwd1 + fwd1), (4.0, rev)])], 100000)
    assert [x[3] for x in res[1]] == [0]  # 6 forward vs 3 reverse after the second chain: still "forward"
    res = chains_to_regions([(1, [(4.0, rev), (5.0, fwd1)])], 100000)
    assert [(x[3], len(x[4])) for x in res[1]] == [(1, 6)]  # 3 reverse, then 3 vs 3 -> not f > r -> reverse again; merged
    assert chains_to_regions([(9, [(1.0, short)])], 100000) == {}


def test_reverse_complement():
    from pgrtk_amd.cli import reverse_complement
    assert reverse_complement(b"ACGTNacgt") == b"acgtNACGT"


def _norm(x):
    """JSON round trip: tuples become lists"""
    if isinstance(x, (list, tuple)):
        return [_norm(v) for v in x]
    return x


def test_query_sdb_and_merge_regions_golden():
    """pgrtk.query_sdb / merge_regions (pgr-tk/pgrtk/__init__.py:130-221, 270-328): vectors generated from the
    reference's own functions by tests/golden/make_query_sdb_fixture.py"""
    import json
    import pgrtk_amd as P
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "query_sdb_cases.json")))
    assert len(cases["merge_regions"]) == 60 and len(cases["query_sdb"]) == 40
    for c in cases["merge_regions"]:
        rgns = [(r[0], r[1], r[2], r[3], list(r[4])) for r in c["rgns"]]
        assert _norm(P.merge_regions(rgns, tol=c["tol"])) == c["out"]

    class Canned:
        def __init__(self, r):
            self.r = r

        def query_fragment_to_hps(self, *a):
            return self.r

    n_regions = 0
    for c in cases["query_sdb"]:
        r = [(sid, [(sc, [((h[0][0], h[0][1], h[0][2]), (h[1][0], h[1][1], h[1][2])) for h in aln]) for sc, aln in alns])
             for sid, alns in c["r"]]
        got = P.query_sdb(Canned(r), b"ACGT", merge_range_tol=c["tol"])
        assert _norm([[sid, v] for sid, v in got.items()]) == c["out"]
        n_regions += sum(len(v) for v in got.values())
    assert n_regions > 40
    assert P.u8_to_string(P.string_to_u8("ACGTN")) == "ACGTN"
    # group_smps_by_principle_bundle_id (:391-467): the product's and the oracle's restatements against the
    # reference's own function
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mapgraph as og
    n_parts = 0
    for c in cases["group_smps"]:
        smps = [((s[0][0], s[0][1], s[0][2], s[0][3], s[0][4]), None if s[1] is None else (s[1][0], s[1][1], s[1][2]))
                for s in c["smps"]]
        got = P.group_smps_by_principle_bundle_id(smps, c["len_cutoff"], c["merge_length"])
        assert _norm(got) == c["out"]
        assert _norm(og.group_smps_by_principle_bundle_id(smps, c["len_cutoff"], c["merge_length"])) == c["out"]
        n_parts += len(got)
    assert n_parts > 60
    for c in cases["rc"]:
        assert P.rc(c["seq"]) == c["rc"] and P.rc_byte_seq(list(c["seq"].encode())) == c["rc_bytes"]


def test_pdb_bincode_varint_kat():
    """`.pdb` = "PDB:0.5" + bincode 2 standard (little endian, varint) of the tuple pgr-pbundle-decomp.rs:362-378 encodes.
    Known answers written by hand from the format description: < 251 one byte; 251 + u16; 252 + u32; 253 + u64; u8 raw."""
    sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
    from pgrtk_amd import pdb
    bundles = [(250, 251, [(65535, 65536, 1)]), (0, 0, [])]
    vmap = {((1 << 32) - 1, 1 << 32): (3, 0, 300)}
    got = pdb.encode(48, 56, 4, 12, 8, 0, bundles, vmap)
    want = (b"PDB:0.5" + bytes([48, 56, 4, 12, 8, 0]) + bytes([2])            # header, 2 bundles
            + bytes([250]) + b"\xfb\xfb\x00" + bytes([1])                       # id 250, order 251 -> fb + u16, 1 vertex
            + b"\xfb\xff\xff" + b"\xfc\x00\x00\x01\x00" + bytes([1])           # 65535 -> fb ffff ; 65536 -> fc + u32 ; dir
            + bytes([0, 0, 0])                                                   # second bundle: id 0, order 0, no vertices
            + bytes([1])                                                         # one map entry
            + b"\xfc\xff\xff\xff\xff" + b"\xfd\x00\x00\x00\x00\x01\x00\x00\x00"  # key: 2^32-1 -> fc + u32 ; 2^32 -> fd + u64
            + bytes([3, 0]) + b"\xfb\x2c\x01")                                  # (bundle 3, direction 0, position 300)
    assert got == want
    assert pdb.decode(got) == (48, 56, 4, 12, 8, 0, bundles, vmap)


def test_output_names_follow_path_with_extension():
    """pgr-query.rs:291-302 / pgr-pbundle-decomp.rs:294-359 build every output name with Path::with_extension(prefix)"""
    from pgrtk_amd.cli import with_extension as w
    assert w("out", "000.hit") == "out.000.hit"
    assert w("out.v1", "000.hit") == "out.000.hit"          # an existing extension is replaced
    assert w("dir.x/out", "bed") == "dir.x/out.bed"          # dots in directories do not count
    assert w(".out", "pdb") == ".out.pdb"                    # a leading dot is not an extension
    assert w("a/.out.b", "ctg.summary.tsv") == "a/.out.ctg.summary.tsv"


def test_synthetic_generators_agree_with_the_oracle(oracle, tmp_path):
    """pgr-mdb --synthetic: the host forms of the counter-based generator of BASELINE.md
    section 4 -- cli.synthetic_contig and the C++ program's --write-fasta -- produce the oracle's bytes.  (The FASTA is
    written before the program asks for a GPU; without one it then fails loudly: there is no CPU path.)"""
    import subprocess
    from pgrtk_amd import cli
    for seed, c, n in [(1, 0, 1000), (2, 7, 33), (5, 123456, 100001), (1, 9, 1), (0, 0, 64)]:
        assert cli.synthetic_contig(seed, c, n) == oracle.synth_contig(seed, c, n).tobytes()
    assert cli.parse_synthetic("10x1000000") == (10, 1000000)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pgr-tk_amd", "bin", "pgr-mdb")
    fa = str(tmp_path / "s.fa")
    r = subprocess.run([exe, "--synthetic", "4x1234", "--seed", "3", "--write-fasta", fa, str(tmp_path / "p")],
                       capture_output=True, text=True, timeout=120)
    recs = oracle.read_fasta(fa)
    assert [n for n, _ in recs] == [b"synth_3_%d" % c for c in range(4)]
    assert all(s == oracle.synth_contig(3, c, 1234).tobytes() for c, (_, s) in enumerate(recs))
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except ImportError:
        has_gpu = False
    if not has_gpu:
        assert r.returncode != 0 and "no CPU fallback" in r.stderr
    r = subprocess.run([exe, "--synthetic", "4by12", str(tmp_path / "p")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2


def test_pgr_query_accepts_claps_kebab_case_and_names_what_it_does_not_support(tmp_path):
    """clap 4 derives `--fastx-file` / `--frg-file` from the struct fields (pgr-bin/src/bin/pgr-query.rs:26-30): the C++ program and the
    Python CLI take the kebab-case spellings; `--frg-file` (the reference's own sequence store, not built here) is refused with a
    message that says so, an unknown option is not mistaken for a path -- all before either asks for a GPU."""
    import subprocess
    from pgrtk_amd import cli
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pgr-tk_amd", "bin", "pgr-query")
    for flag in ("--frg-file", "--frg_file"):
        r = subprocess.run([exe, "db", "q.fa", str(tmp_path / "o"), flag], capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and "not supported" in r.stderr and "--fastx-file" in r.stderr, r.stderr
        assert cli.main(["query", "db", "q.fa", str(tmp_path / "o"), flag]) == 2
    r = subprocess.run([exe, "db", "q.fa", str(tmp_path / "o"), "--no-such-option"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "unknown option --no-such-option" in r.stderr
    r = subprocess.run([exe, "db", "q.fa"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "--fastx-file" in r.stderr  # (the usage line spells it clap's way)
