"""GPU parity: sequence_to_shmmrs through the C ABI (libpgrhip.so) vs the CPU oracle, bit exact.

Reference path under test: shmmrutils::sequence_to_shmmrs (pgr-db/src/shmmrutils.rs:657-669) as
batched by get_shmmrs_from_seqs (pgr-db/src/seq_db.rs:456-469), and seq_to_index records
(seq_db.rs:360-418).
"""
import os

import numpy as np
import pytest

import seqgen

pytestmark = pytest.mark.gpu


def _assert_same(ref, got, what=""):
    assert len(ref) == len(got), "%s: %d vs %d shimmers" % (what, len(ref), len(got))
    assert np.array_equal(ref["x"], got["x"]), what + ": x differs"
    assert np.array_equal(ref["y"], got["y"]), what + ": y differs"


def _check_batch(oracle, gpu_ctx, seqs, spec_t, padding=False, rids=None, what=""):
    import pgrtk_amd as P
    w, k, r, ms, sk = spec_t
    # three routes to the same answer: the host entry point and the resident path (batches of short clean contigs take the
    # one-workgroup-per-contig kernel of csrc/small.hip there, everything else the general pipeline), and the resident path
    # with that kernel switched off (always the general pipeline: tiles, tails, islands, fused list stage)
    got = P.sequence_to_shmmrs_batch(seqs, P.make_spec(w, k, r, ms, sk), rids=rids, padding=padding, ctx=gpu_ctx)
    batch = P.Batch.from_seqs(seqs, ctx=gpu_ctx)
    mm, off = batch.shmmrs(P.make_spec(w, k, r, ms, sk), rids=rids, padding=padding).download()
    with gpu_ctx.options(no_small_path=1):
        mm_g, off_g = batch.shmmrs(P.make_spec(w, k, r, ms, sk), rids=rids, padding=padding).download()
    osp = oracle.spec(w, k, r, ms, sk)
    assert len(got) == len(seqs) and len(off) == len(seqs) + 1 == len(off_g)
    for i, s in enumerate(seqs):
        rid = i if rids is None else rids[i]
        ref = oracle.sequence_to_shmmrs(rid, s, osp, padding)
        tag = "seq %d (len %d) spec %s pad %s" % (i, len(s), spec_t, padding)
        _assert_same(ref, got[i], "%s %s" % (what, tag))
        _assert_same(ref, mm[int(off[i]):int(off[i + 1])], "%s (resident) %s" % (what, tag))
        _assert_same(ref, mm_g[int(off_g[i]):int(off_g[i + 1])], "%s (general pipeline) %s" % (what, tag))
    return got


def test_native_library_loaded(gpu_ctx):
    """the HIP extension is the thing that runs (the driver also records the loaded .so)"""
    from pgrtk_amd import _ffi
    assert _ffi.lib().pgr_version().startswith(b"pgr-hip")
    maps = open("/proc/self/maps").read()
    assert "libpgrhip.so" in maps


def test_golden_fixture_seqs(oracle, gpu_ctx, test_seqs):
    """test_seqs.fa at the north-star spec: per-contig MM128 lists == oracle (== golden .mdb, see below)"""
    seqs = [s for _, s in test_seqs]
    got = _check_batch(oracle, gpu_ctx, seqs, (80, 56, 4, 64, False), what="test_seqs.fa")
    assert sum(len(g) for g in got) == 886


def test_golden_mdb_through_gpu(oracle, gpu_ctx, test_seqs, golden_dir):
    """G1 end to end: GPU pair records, renumbered with the FASTX-backend global fragment ids
    (seq_db.rs:189-357: Prefix +1, pairs, Suffix +1), must equal test_seqs_frag.mdb exactly."""
    import pgrtk_amd as P
    _, g = oracle.read_mdb(os.path.join(golden_dir, "test_seqs_frag.mdb"))
    seqs = [s for _, s in test_seqs]
    recs = P.frag_recs_batch(seqs, P.make_spec(80, 56, 4, 64), ctx=gpu_ctx)
    shm = P.sequence_to_shmmrs_batch(seqs, P.make_spec(80, 56, 4, 64), ctx=gpu_ctx)
    m = {}
    base = 0
    for sid, (rr, sh) in enumerate(zip(recs, shm)):
        if len(sh) == 0:
            base += 2
            continue
        for x in rr:
            assert int(x["sid"]) == sid
            m.setdefault((int(x["h0"]), int(x["h1"])), []).append(
                (base + 1 + int(x["frg_id"]), sid, int(x["bgn"]), int(x["end"]), int(x["orient"])))
        base += len(sh) + 1
    assert m == g


def test_boundary_condition_kat(oracle, gpu_ctx, golden_dir):
    """pgr-db/src/lib.rs:342-363 (w=24,k=24,r=12,min_span=24,padding=true) -> 2 shimmers each"""
    seqs = [l.strip().encode() for l in open(os.path.join(golden_dir, "boundary_condition_seqs.txt"))
            if not l.startswith("#")]
    got = _check_batch(oracle, gpu_ctx, seqs, (24, 24, 12, 24, False), padding=True, what="boundary")
    assert [len(g) for g in got] == [2, 2]


def test_rc_match_kat(oracle, gpu_ctx, golden_dir):
    """pgr-db/src/lib.rs:166-180 with SHMMRSPEC (sketch=true)"""
    recs = oracle.read_fasta(os.path.join(golden_dir, "test_rev.fa"))
    for sk in (True, False):
        got = _check_batch(oracle, gpu_ctx, [recs[0][1], recs[1][1]], (80, 56, 4, 64, sk), what="rc_match")
        assert len(got[0]) > 0
        assert list(got[0]["x"] >> np.uint64(8)) == list((got[1]["x"] >> np.uint64(8))[::-1])


SPECS = [(80, 56, 4, 64, False), (48, 56, 4, 12, False), (24, 24, 12, 24, False), (17, 9, 2, 0, False),
         (128, 56, 12, 64, False), (80, 56, 1, 64, False), (33, 31, 3, 8, False),
         # w < 17: routed to the serial (exact state machine) kernel
         (16, 8, 2, 0, False), (5, 4, 3, 1, False), (1, 5, 1, 0, False), (2, 3, 2, 0, False),
         # sketch variant (shmmrutils.rs:558-655)
         (80, 56, 4, 64, True), (80, 21, 2, 16, True)]


@pytest.mark.parametrize("spec_t", SPECS)
def test_adversarial_sweep(oracle, gpu_ctx, spec_t):
    """random + adversarial inputs: ties, lower case, N runs (leading / internal), palindromic k-mers,
    tandem repeats, bytes 0..3, lengths around every boundary of the state machine; with and without padding"""
    w, k, r, ms, sk = spec_t
    rng = np.random.default_rng(99 + w * 131 + k)
    lens = [0, 1, k - 1, k, k + 1, k + w - 2, k + w - 1, k + w, k + w + 1, 2 * w, 2 * w + 1, 2 * w + k, 3 * w + k,
            500, 1000, 3000, 5000]
    seqs = []
    for it in range(96):
        seqs.append(seqgen.adversarial(rng, it % seqgen.N_MODES, int(rng.choice(lens))))
    _check_batch(oracle, gpu_ctx, seqs, spec_t, padding=False, what="sweep")
    if not sk:
        _check_batch(oracle, gpu_ctx, seqs, spec_t, padding=True, what="sweep+pad")


def test_tile_boundaries(oracle, gpu_ctx):
    """contig lengths around multiples of the tile core (4096 - 2(w-1)) and of the 64-position chunks"""
    rng = np.random.default_rng(5)
    for (w, k, r, ms) in [(80, 56, 4, 64), (48, 56, 4, 12), (128, 56, 4, 64), (17, 17, 2, 4)]:
        lens = []
        for ext in (4096, 8192):  # tile = 16 positions x 256 or 512 lanes, core = ext - 2(w-1)
            tc = ext - 2 * (w - 1)
            for m in (1, 2, 3):
                for d in (-w - k, -w, -k, -2, -1, 0, 1, 2, k, w, w + k):
                    lens.append(m * tc + d)
        lens += [4096, 8192, 4095, 4097, 8191, 8193, 16384, 64 * 100, 64 * 100 + 1, 64 * 100 - 1]
        seqs = [seqgen.rnd(rng, L) for L in lens]
        _check_batch(oracle, gpu_ctx, seqs, (w, k, r, ms, False), what="tiles")


def test_serial_kernel_cases(oracle, gpu_ctx):
    """inputs the closed form cannot do (non-ACGT bytes, palindromic k-mers): exact state machine on the GPU"""
    rng = np.random.default_rng(11)
    base = seqgen.rnd(rng, 20000)
    seqs = []
    for p in (0, 1, 55, 56, 57, 63, 64, 65, 135, 136, 5000, 19998, 19999):  # one N anywhere
        s = bytearray(base)
        s[p] = ord("N")
        seqs.append(bytes(s))
    for run in (1, 2, 63, 64, 65, 127, 128, 129, 1000, 5000):  # N runs across chunk boundaries
        for at in (0, 100, 6400, 6399, 10000):
            seqs.append(base[:at] + b"N" * run + base[at:])
    seqs.append(b"N" * 3000)
    seqs.append(b"N" * 3000 + base[:500])
    seqs.append(base[:9000] + b"AT" * 40 + base[9000:])  # (AT)n >= 28: palindromic 56-mers
    seqs.append(base[:9000] + b"CG" * 100 + base[9000:])
    seqs.append(b"A" * 10000)  # homopolymer: every window is a tie
    seqs.append(b"AC" * 5000)
    _check_batch(oracle, gpu_ctx, seqs, (80, 56, 4, 64, False), what="serial")
    _check_batch(oracle, gpu_ctx, seqs[::3], (48, 56, 4, 12, False), what="serial48")
    _check_batch(oracle, gpu_ctx, seqs[::5], (80, 56, 4, 64, True), what="serial-sketch")


def test_chunked_exact_machine_seams(oracle, gpu_ctx):
    """long contigs on the exact-machine path: the 32 kbp chunks must agree with the sequential oracle across
    every seam (N runs over seams, palindromes next to seams, N-only chunks, scattered N, low complexity)"""
    import pgrtk_amd as P
    rng = np.random.default_rng(17)
    CS = 32768
    base = seqgen.rnd(rng, 300000)
    seqs = []
    s = bytearray(base)  # single N far from any seam: everything else must be unaffected
    s[1000] = ord("N")
    seqs.append(bytes(s))
    for at in (CS - 1, CS, CS + 1, CS - 56, CS - 80, CS - 136, 2 * CS - 300, 3 * CS + 7):  # N at / around seams
        s = bytearray(base)
        s[at] = ord("N")
        seqs.append(bytes(s))
    for run, at in ((500, CS - 250), (70000, 20000), (3 * CS, CS // 2), (40, 2 * CS - 20), (CS, CS)):  # N runs over seams
        seqs.append(base[:at] + b"N" * run + base[at:200000])
    seqs.append(b"N" * 100000 + base[:50000])  # leading N run longer than 3 chunks
    seqs.append(base[:50000] + b"N" * 100000)  # trailing
    for at in (CS - 60, CS - 10, CS + 30, 2 * CS - 100):  # palindromic 56-mers ((AT)n) next to seams
        seqs.append(base[:at] + b"AT" * 45 + base[at:150000])
    s = bytearray(base)  # scattered N every ~5 kbp
    for p in range(777, len(s), 5003):
        s[p] = ord("n")
    seqs.append(bytes(s))
    seqs.append((b"ACGTTGCA" * 20000)[:150000])  # tandem repeat: ties everywhere (tile path, dense slots)
    seqs.append(b"A" * 70000 + base[:70000] + b"N" + b"C" * 70000)  # homopolymers + N
    for spec_t in [(80, 56, 4, 64, False), (48, 56, 4, 12, False), (16, 8, 2, 0, False), (80, 56, 4, 64, True)]:
        sub = seqs if spec_t[0] == 80 and not spec_t[4] else seqs[::3]
        _check_batch(oracle, gpu_ctx, sub, spec_t, what="chunked")
    prof = gpu_ctx.last_prof()
    assert prof.n_serial_contigs > 0


def test_exact_islands_are_local(oracle, gpu_ctx):
    """a few irregular spots in long contigs: only islands of tiles around them take the exact kernel, the rest
    stays on the closed-form tile kernel, and the stitched lists are bit exact (island edges by position,
    inner seams by emission step, right edge verified by a probe)"""
    import pgrtk_amd as P
    rng = np.random.default_rng(23)
    L = 1_000_000
    seqs = []
    for spots in ([500_000], [10], [L - 5], [7990, 8010], [100_000, 108_050, 116_100], [250_000, 750_000],
                  list(range(50_000, 950_000, 90_000))):
        s = bytearray(seqgen.rnd(rng, L))
        for p in spots:
            s[p] = ord("N")
        seqs.append(bytes(s))
    s = bytearray(seqgen.rnd(rng, L))  # an N run of 3 tiles + a palindromic (AT)n + a lower-case stretch
    s[300_000:325_000] = b"N" * 25_000
    s[600_000:600_090] = b"AT" * 45
    s[700_000:710_000] = bytes(s[700_000:710_000]).lower()
    seqs.append(bytes(s))
    seqs.append(seqgen.rnd(rng, L))  # a clean contig in the same batch
    s = bytearray(seqgen.rnd(rng, 30_000))  # short contigs: island == whole contig
    s[15_000] = ord("N")
    seqs.append(bytes(s))
    for spec_t in [(80, 56, 4, 64, False), (48, 56, 4, 12, False), (80, 56, 4, 64, True)]:
        _check_batch(oracle, gpu_ctx, seqs, spec_t, what="islands")
        prof = gpu_ctx.last_prof()
        assert prof.n_serial_contigs >= 9
        assert prof.exact_bases < 0.2 * sum(len(x) for x in seqs), prof.exact_bases  # islands, not whole contigs


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_island_fuzz(oracle, gpu_ctx, seed):
    """random irregularities (single N, N runs of random length, (AT)n / (CG)n palindromes, homopolymers, lower case,
    bytes 0..3) at random places of 300 kbp contigs, random specs: tile kernel + exact islands == oracle"""
    rng = np.random.default_rng(1000 + seed)
    seqs = []
    for _ in range(14):
        L = int(rng.integers(60_000, 300_000))
        s = bytearray(seqgen.rnd(rng, L))
        for _ in range(int(rng.integers(0, 7))):
            kind = int(rng.integers(0, 6))
            p = int(rng.integers(0, L))
            if kind == 0:
                s[p] = ord("N")
            elif kind == 1:
                n = int(rng.choice([2, 60, 200, 5000, 20000, 70000]))
                s[p:p + n] = b"N" * len(s[p:p + n])
            elif kind == 2:
                rep = (b"AT" if rng.random() < 0.5 else b"CG") * int(rng.integers(28, 90))
                s[p:p + len(rep)] = rep[:len(s[p:p + len(rep)])]
            elif kind == 3:
                n = int(rng.integers(100, 3000))
                s[p:p + n] = bytes([int(rng.choice(list(b"ACGT")))]) * len(s[p:p + n])
            elif kind == 4:
                n = int(rng.integers(10, 5000))
                s[p:p + n] = bytes(s[p:p + n]).lower()
            else:
                n = int(rng.integers(1, 200))
                s[p:p + n] = bytes(int(v) for v in rng.integers(0, 4, len(s[p:p + n])))
        seqs.append(bytes(s))
    specs = [(80, 56, 4, 64, False), (48, 56, 4, 12, False), (31, 24, 3, 8, False), (128, 56, 12, 64, False),
             (80, 56, 4, 64, True)]
    for spec_t in (specs[seed % len(specs)], specs[0]):
        _check_batch(oracle, gpu_ctx, seqs, spec_t, what="island-fuzz seed %d" % seed)
        _check_batch(oracle, gpu_ctx, seqs[:5], spec_t, padding=not spec_t[4], what="island-fuzz+pad")


def test_ragged_and_empty(oracle, gpu_ctx):
    import pgrtk_amd as P
    sp = P.make_spec()
    assert P.sequence_to_shmmrs_batch([], sp, ctx=gpu_ctx) == []
    rng = np.random.default_rng(3)
    seqs = [b"", seqgen.rnd(rng, 10), b"", seqgen.rnd(rng, 100000), b"", seqgen.rnd(rng, 135), seqgen.rnd(rng, 136),
            b""]
    got = _check_batch(oracle, gpu_ctx, seqs, (80, 56, 4, 64, False), what="ragged")
    assert len(got[0]) == 0 and len(got[3]) > 100


def test_rids(oracle, gpu_ctx):
    rng = np.random.default_rng(4)
    seqs = [seqgen.rnd(rng, 5000) for _ in range(5)]
    _check_batch(oracle, gpu_ctx, seqs, (80, 56, 4, 64, False), rids=[7, 7, 0xFFFFFFFE, 0, 123456], what="rids")


def test_bad_spec_is_an_error_not_an_abort(gpu_ctx):
    """the reference asserts (shmmrutils.rs:443-445); across the C ABI this is an error code"""
    import pgrtk_amd as P
    for bad in [(80, 57, 4, 64), (129, 56, 4, 64), (80, 56, 13, 64), (80, 56, 0, 64), (0, 56, 4, 64), (80, 0, 4, 64)]:
        with pytest.raises(P.PgrError) as e:
            P.sequence_to_shmmrs_batch([b"ACGT" * 100], P.make_spec(*bad), ctx=gpu_ctx)
        assert e.value.code == -2
    # the context stays usable
    assert len(P.sequence_to_shmmrs_batch([b"ACGT" * 100], P.make_spec(), ctx=gpu_ctx)) == 1


def test_frag_recs_vs_oracle(oracle, gpu_ctx):
    import pgrtk_amd as P
    rng = np.random.default_rng(8)
    seqs = [seqgen.rnd(rng, L) for L in (200, 3000, 50000, 0, 120000)]
    for qs in (False, True):
        got = P.frag_recs_batch(seqs, P.make_spec(), sids=[5, 6, 7, 8, 9], query_side=qs, ctx=gpu_ctx)
        for i, s in enumerate(seqs):
            sh = oracle.sequence_to_shmmrs(0, s, oracle.spec())
            ref = oracle.frag_recs(sh, 5 + i, query_side=qs)
            assert len(ref) == len(got[i])
            for f in ("h0", "h1", "frg_id", "sid", "bgn", "end", "orient"):
                assert np.array_equal(ref[f], got[i][f]), f


def test_synthetic_device_generator_and_scale(oracle, gpu_ctx):
    """device-side contig generator == oracle generator; 12 x 1 Mbp (+ ragged) bit exact vs the oracle"""
    import pgrtk_amd as P
    lens = [1000000] * 12 + [999999, 31, 32, 33, 0, 4096, 123457]
    b = P.Batch.synthetic(lens, seed=2, contig0=10, ctx=gpu_ctx)
    assert b.total_bases == sum(lens)
    sh = b.shmmrs(P.make_spec())
    mm, off = sh.download()
    osp = oracle.spec()
    for i, L in enumerate(lens):
        s = oracle.synth_contig(2, 10 + i, L)
        ref = oracle.sequence_to_shmmrs(i, s, osp)
        _assert_same(ref, mm[int(off[i]):int(off[i + 1])], "synthetic contig %d" % i)
    prof = gpu_ctx.last_prof()
    assert prof.n_serial_contigs == 0 and prof.n_tiles > 0 and prof.level1_ms > 0
    # same contigs uploaded as ASCII give the same answer (pack kernel == generator)
    seqs = [oracle.synth_contig(2, 10 + i, L) for i, L in enumerate(lens)]
    mm2 = np.concatenate(P.sequence_to_shmmrs_batch(seqs, P.make_spec(), ctx=gpu_ctx))
    assert np.array_equal(mm["x"], mm2["x"]) and np.array_equal(mm["y"], mm2["y"])


def test_size_independent_properties(oracle, gpu_ctx):
    """properties that hold at any size (used at BASELINE.json's full size by bench.py):
    sortedness by position, min_span stencil, hash of reverse complement, idempotent re-run"""
    import pgrtk_amd as P
    lens = [3000000, 2000000]
    b = P.Batch.synthetic(lens, seed=5, ctx=gpu_ctx)
    sp = P.make_spec()
    mm, off = b.shmmrs(sp).download()
    mm_b, off_b = b.shmmrs(sp).download()
    assert np.array_equal(mm, mm_b) and np.array_equal(off, off_b)  # deterministic
    for i in range(len(lens)):
        a = mm[int(off[i]):int(off[i + 1])]
        pos = (a["y"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)
        assert (a["y"] >> np.uint64(32) == i).all()
        assert (np.diff(pos.astype(np.int64)) > 64).all()  # min_span on survivors' unfiltered neighbours implies this
        assert ((a["x"] & np.uint64(0xFF)) == 56).all()
        dens = len(a) / lens[i]
        assert 0.0028 < dens < 0.0033  # 0.003035 / bp on random sequence
    # reverse complement symmetry on a 200 kbp piece
    s = oracle.synth_contig(5, 0, 200000)
    comp = np.zeros(256, dtype=np.uint8)
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    rc = comp[s][::-1].copy()
    f, r = P.sequence_to_shmmrs_batch([s, rc], P.make_spec(sketch=True), ctx=gpu_ctx)
    assert list(f["x"]) == list(r["x"][::-1])


def test_batch_beyond_2_pow_32_bases(gpu_ctx):
    """one resident batch of 6 Gbp (> 2^32 positions): a contig's shimmers do not depend on where it sits in the batch
    (64-bit word / position arithmetic in every kernel): the last contigs equal a small batch of the same contig ids"""
    import pgrtk_amd as P
    spec = P.make_spec()
    n = 600
    b = P.Batch.synthetic([10_000_000] * n, seed=2, ctx=gpu_ctx)
    assert b.total_bases > 2 ** 32
    s = b.shmmrs(spec)
    mm, off = s.download()
    b.close()
    s.close()
    for c0 in (0, 428, n - 3):  # first, the ones straddling base 2^32, last
        b2 = P.Batch.synthetic([10_000_000] * 3, seed=2, contig0=c0, ctx=gpu_ctx)
        s2 = b2.shmmrs(spec)
        mm2, off2 = s2.download()
        lo, hi = int(off[c0]), int(off[c0 + 3])
        assert hi - lo == len(mm2) and np.array_equal(off[c0:c0 + 4] - off[c0], off2)
        assert np.array_equal(mm["x"][lo:hi], mm2["x"])
        assert np.array_equal(mm["y"][lo:hi] & np.uint64(0xFFFFFFFF), mm2["y"] & np.uint64(0xFFFFFFFF))
        assert np.array_equal(mm["y"][lo:hi] >> np.uint64(32), (mm2["y"] >> np.uint64(32)) + np.uint64(c0))
        b2.close()
        s2.close()


def test_pipelined_host_batch(oracle, gpu_ctx):
    """pgr_shmmr_batch cuts host inputs of >= 512 Mbp into sub-batches that are staged by a second thread / stream
    while the previous one computes: same result as the single-batch path (context option no_pipeline), rids and contig order
    kept, contigs with N and empty contigs inside, spot checks against the oracle"""
    import pgrtk_amd as P
    rng = np.random.default_rng(42)
    lens = [int(x) for x in rng.integers(1_000_000, 30_000_000, 40)]
    lens[7] = 0
    lens[20] = 55
    seqs = [oracle.synth_contig(9, i, L).tobytes() if L else b"" for i, L in enumerate(lens)]
    s11 = bytearray(seqs[11])
    s11[100_000:100_400] = b"N" * 400
    s11[5_000_000] = ord("n")
    seqs[11] = bytes(s11)
    assert sum(lens) > 600_000_000
    spec = P.make_spec(80, 56, 4, 64)
    rids = [int(x) for x in rng.integers(0, 2 ** 31, len(seqs))]
    for r in (None, rids):
        a = P.sequence_to_shmmrs_batch(seqs, spec, rids=r, ctx=gpu_ctx)
        with gpu_ctx.options(no_pipeline=1):
            b = P.sequence_to_shmmrs_batch(seqs, spec, rids=r, ctx=gpu_ctx)
        assert len(a) == len(b) == len(seqs)
        for i in range(len(seqs)):
            _assert_same(b[i], a[i], "pipelined vs single batch, contig %d" % i)
        osp = oracle.spec(80, 56, 4, 64)
        for i in (0, 7, 11, 20, len(seqs) - 1):
            _assert_same(oracle.sequence_to_shmmrs(i if r is None else r[i], seqs[i], osp, False), a[i], "contig %d vs oracle" % i)
