"""Round-3 GPU tests: packed input at the ABI, full-size parity of BASELINE.json configs[1] and configs[2] inside the graded
suite, the key-range sharded index build, the N-rank bench line.

Reference paths under test: shmmrutils::sequence_to_shmmrs (pgr-db/src/shmmrutils.rs:657-669), the base table
(:426-436), load_index_from_seq_vec (seq_db.rs:573-615), query_fragment_to_hps (aln.rs:147-242).
"""
import json
import os
import sys

import numpy as np
import pytest

import seqgen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(ref, got, what=""):
    assert len(ref) == len(got), "%s: %d vs %d shimmers" % (what, len(ref), len(got))
    assert np.array_equal(ref["x"], got["x"]) and np.array_equal(ref["y"], got["y"]), what


@pytest.mark.parametrize("spec_t", [(80, 56, 4, 64, False), (48, 56, 4, 12, False), (24, 24, 12, 24, False),
                                    (5, 4, 3, 1, False), (80, 21, 2, 16, True)])
def test_packed_input_equals_ascii_path_and_oracle(oracle, gpu_ctx, spec_t):
    """pgr_shmmr_batch_packed on the output of the library's CPU packer == pgr_shmmr_batch on the ASCII bytes == oracle,
    on the adversarial sweep (N runs, lower case, bytes 0..3, ragged and empty contigs)"""
    import pgrtk_amd as P
    w, k, r, ms, sk = spec_t
    rng = np.random.default_rng(17 + w)
    lens = [0, 1, k - 1, k, k + w - 1, k + w, 2 * w + 1, 31, 32, 33, 500, 1000, 3000, 5000, 70000]
    seqs = [seqgen.adversarial(rng, it % seqgen.N_MODES, int(rng.choice(lens))) for it in range(64)]
    spec = P.make_spec(w, k, r, ms, sk)
    packed, n_bad = P.pack_ascii(seqs)
    assert n_bad == sum(sum(1 for ch in s if ch not in b"ACGTacgt\x00\x01\x02\x03") for s in seqs)
    a = P.sequence_to_shmmrs_batch(seqs, spec, ctx=gpu_ctx)
    b = P.sequence_to_shmmrs_batch_packed(packed, spec, ctx=gpu_ctx)
    osp = oracle.spec(w, k, r, ms, sk)
    for i, s in enumerate(seqs):
        ref = oracle.sequence_to_shmmrs(i, s, osp)
        _same(ref, a[i], "ascii seq %d" % i)
        _same(ref, b[i], "packed seq %d" % i)
    # resident batch from packed planes, with rids and padding
    rids = [1000 + 3 * i for i in range(len(seqs))]
    if not sk:
        c = P.sequence_to_shmmrs_batch_packed(packed, spec, rids=rids, padding=True, ctx=gpu_ctx)
        for i, s in enumerate(seqs):
            _same(oracle.sequence_to_shmmrs(rids[i], s, osp, True), c[i], "packed+pad seq %d" % i)


def test_packed_input_dirty_bits_and_no_validity_plane(oracle, gpu_ctx):
    """what the header promises about packed input: bits past a contig's end and plane bits of invalid positions are
    ignored; valid == NULL means every base is a base"""
    import pgrtk_amd as P
    rng = np.random.default_rng(3)
    spec = P.make_spec()
    osp = oracle.spec()
    clean = [seqgen.rnd(rng, n) for n in (100_000, 33, 64, 777, 250_001)]
    packed, n_bad = P.pack_ascii(clean)
    assert n_bad == 0
    no_valid = P.PackedBases(packed.lens, packed.planes.copy(), None)
    got = P.sequence_to_shmmrs_batch_packed(no_valid, spec, ctx=gpu_ctx)
    for i, s in enumerate(clean):
        _same(oracle.sequence_to_shmmrs(i, s, osp), got[i], "no validity plane, seq %d" % i)
    # garbage in the last word behind the contig's end, in planes and in the validity plane
    dirty_p, dirty_v = packed.planes.copy(), packed.valid.copy()
    w = 0
    for s in clean:
        nw = (len(s) + 31) // 32
        tail = len(s) % 32
        if tail:
            junk = np.uint64((1 << (32 - tail)) - 1)
            dirty_p[w + nw - 1] |= junk | (junk << np.uint64(32))
            dirty_v[w + nw - 1] |= np.uint32(int(junk))
        w += nw
    got = P.sequence_to_shmmrs_batch_packed(P.PackedBases(packed.lens, dirty_p, dirty_v), spec, ctx=gpu_ctx)
    for i, s in enumerate(clean):
        _same(oracle.sequence_to_shmmrs(i, s, osp), got[i], "dirty tail, seq %d" % i)
    # plane bits set where the validity plane says "not a base" (an N packed as a T by a sloppy host)
    noisy = [seqgen.adversarial(rng, 4, 60_000), seqgen.adversarial(rng, 2, 5_000)]
    pk, _ = P.pack_ascii(noisy)
    pl = pk.planes.copy()
    inv = ~pk.valid.astype(np.uint64) & np.uint64(0xFFFFFFFF)
    pl |= inv | (inv << np.uint64(32))
    got = P.sequence_to_shmmrs_batch_packed(P.PackedBases(pk.lens, pl, pk.valid), spec, ctx=gpu_ctx)
    for i, s in enumerate(noisy):
        _same(oracle.sequence_to_shmmrs(i, s, osp), got[i], "noisy planes, seq %d" % i)


def test_packed_input_pipelined_large_batch(oracle, gpu_ctx):
    """>= 512 Mbp of packed host input goes through the sub-batched, double-buffered path (for_each_staged): same result
    as the resident path on the same contigs (checksums), offsets consistent"""
    import bench
    import pgrtk_amd as P
    spec = P.make_spec()
    n, L = 27, 20_000_000
    seqs = [bench.synth_contig_ascii(2, c, L) for c in range(n)]
    seqs[5] = seqs[5][:1_234_567]
    seqs[11] = np.concatenate([seqs[11][:7_000_000], np.full(100_000, ord("N"), dtype=np.uint8), seqs[11][7_100_000:]])
    packed, _ = P.pack_ascii(seqs)
    got = P.sequence_to_shmmrs_batch_packed(packed, spec, ctx=gpu_ctx)
    res = P.Batch.from_seqs(seqs, ctx=gpu_ctx).shmmrs(spec)
    mm, off = res.download()
    assert len(got) == n
    for i in range(n):
        ref = mm[int(off[i]):int(off[i + 1])]
        _same(ref, got[i], "contig %d" % i)
    sp = oracle.spec()
    for i in (5, 11):
        _same(oracle.sequence_to_shmmrs(i, seqs[i].tobytes(), sp), got[i], "oracle contig %d" % i)


def test_shard_partition_is_stable_and_complete(gpu_ctx):
    """pgr_shard_partition with hand-made splitters on 3 and 8 destinations: every record lands in the range
    [splitter[d-1], splitter[d]) of its first hash, inside a destination the append order is kept"""
    import ctypes as C
    import torch
    import pgrtk_amd as P
    from pgrtk_amd import exchange
    from pgrtk_amd._ffi import FRAG_REC, lib
    spec = P.make_spec()
    b = P.Batch.synthetic([3_000_000, 1_000_000], seed=5, ctx=gpu_ctx)
    sh = b.shmmrs(spec)
    recs = torch.zeros((sh.n_pairs, exchange.REC_WORDS), dtype=torch.int64, device="cuda:0")
    n = sh.frag_recs_into(recs.data_ptr(), recs.shape[0])
    host = recs.cpu().numpy().view(FRAG_REC).reshape(-1)[:n]
    for world in (3, 8):
        spl = np.quantile(host["h0"].astype(np.float64), [j / world for j in range(1, world)]).astype(np.uint64)
        spl[0] = host["h0"][7]  # an exact key value as a splitter: records with that hash go to the right
        spl.sort()
        out = torch.zeros_like(recs)
        counts = np.zeros(world, dtype=np.uint64)
        gpu_ctx.check(lib().pgr_shard_partition(gpu_ctx.handle, C.c_void_p(recs.data_ptr()), n, C.c_void_p(spl.ctypes.data), world,
                                                C.c_void_p(out.data_ptr()), C.c_void_p(counts.ctypes.data)))
        got = out.cpu().numpy().view(FRAG_REC).reshape(-1)[:n]
        dest = np.searchsorted(spl, host["h0"], side="right")
        assert [int(c) for c in counts] == [int((dest == d).sum()) for d in range(world)]
        want = np.concatenate([host[dest == d] for d in range(world)])  # stable
        assert got.tobytes() == want.tobytes()


def test_small_call_path_against_oracle_and_general_path(oracle, gpu_ctx, monkeypatch):
    """csrc/small.hip: batches of short clean contigs go through ONE kernel (one workgroup per contig: tiles, tail, both
    reductions, min_span in LDS).  Lengths around every boundary of the state machine and of the tiles, several specs, rids;
    the same calls with the small path switched off take the general pipeline -- three-way equality with the oracle.
    Contigs the kernel hands back (palindromic k-mers, low-complexity lists) and batches with N still come out exact."""
    import pgrtk_amd as P
    rng = np.random.default_rng(77)
    for spec_t in ((80, 56, 4, 64), (48, 56, 4, 12), (24, 24, 12, 24), (33, 31, 3, 8), (80, 56, 1, 64), (128, 56, 12, 64), (17, 9, 2, 0)):
        w, k, r, ms = spec_t
        tc = (4096 - 2 * (w - 1)) // 64 * 64
        lens = [0, 1, k - 1, k, k + 1, k + w - 2, k + w - 1, k + w, 2 * w + k, 3 * w, 500, 1000, 3000, tc - 1, tc, tc + 1, 2 * tc - w,
                2 * tc, 2 * tc + w + k, 10_000, 33_333, 100_000, 131_072]
        seqs = [seqgen.rnd(rng, n) for n in lens]
        rids = [int(v) for v in rng.integers(0, 2 ** 31, len(seqs))]
        spec = P.make_spec(w, k, r, ms)
        osp = oracle.spec(w, k, r, ms)
        small = P.sequence_to_shmmrs_batch(seqs, spec, rids=rids, ctx=gpu_ctx)
        with gpu_ctx.options(no_small_path=1):
            general = P.sequence_to_shmmrs_batch(seqs, spec, rids=rids, ctx=gpu_ctx)
        for i, s in enumerate(seqs):
            ref = oracle.sequence_to_shmmrs(rids[i], s, osp)
            _same(ref, small[i], "small path, spec %s len %d" % (spec_t, len(s)))
            _same(ref, general[i], "general path, spec %s len %d" % (spec_t, len(s)))
    # handed back by the kernel / not eligible: still exact
    spec, osp = P.make_spec(), oracle.spec()
    mixed = [seqgen.rnd(rng, 20_000), seqgen.rnd(rng, 9_000) + b"AT" * 70 + seqgen.rnd(rng, 9_000), b"A" * 60_000, b"ACGTTGCA" * 9000,
             seqgen.rnd(rng, 5_000) + b"N" + seqgen.rnd(rng, 5_000), seqgen.rnd(rng, 131_073)]
    for sub in (mixed[:1], mixed[:2], mixed[2:4], mixed[4:5], mixed[5:], mixed):
        got = P.sequence_to_shmmrs_batch(sub, spec, ctx=gpu_ctx)
        for i, s in enumerate(sub):
            _same(oracle.sequence_to_shmmrs(i, s, osp), got[i], "mixed len %d" % len(s))
    # packed input takes the same path
    clean = [seqgen.rnd(rng, n) for n in (10_000, 777, 56, 40_000)]
    packed, _ = P.pack_ascii(clean)
    for pk in (packed, P.PackedBases(packed.lens, packed.planes, None)):
        got = P.sequence_to_shmmrs_batch_packed(pk, spec, ctx=gpu_ctx)
        for i, s in enumerate(clean):
            _same(oracle.sequence_to_shmmrs(i, s, osp), got[i], "packed small call %d" % i)


def test_small_call_latency_targets(gpu_ctx):
    """the point of the small path: one 10 kbp contig in well under 0.1 ms, the reference's 129-contig batch (seq_db.rs:561) of
    10 kbp contigs in a fraction of a millisecond -- host ASCII in, host MM128 out, timed at the C entry point"""
    import bench
    import pgrtk_amd as P
    spec = P.make_spec()
    one = P.PackedSeqs.from_list([bench.synth_contig_ascii(2, 0, 10_000)])
    many = P.PackedSeqs.from_list([bench.synth_contig_ascii(2, c, 10_000) for c in range(129)])

    def med(seqs):
        for _ in range(5):
            P.time_shmmr_batch(seqs, spec, ctx=gpu_ctx)
        ts = sorted(P.time_shmmr_batch(seqs, spec, ctx=gpu_ctx)[0] for _ in range(40))
        return ts[len(ts) // 2] * 1e3
    t1, t129 = med(one), med(many)
    print("one 10 kbp contig %.3f ms, 129 x 10 kbp %.3f ms" % (t1, t129))
    assert t1 < 0.09 and t129 < 0.3, (t1, t129)  # (measured: see bench.py latency; the assertion leaves room for a slow box)

