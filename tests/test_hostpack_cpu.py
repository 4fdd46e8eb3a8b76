"""The library's CPU packer (pgr_pack_ascii, csrc/hostpack.cpp) against a numpy restatement of the reference's base table
(pgr-db/src/shmmrutils.rs:426-436).  Host code only: runs without a GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ref_pack(seq):
    """numpy restatement: base i of a contig -> bit 31 - (i % 32) of word i // 32"""
    a = np.frombuffer(bytes(seq), dtype=np.uint8)
    code = np.full(256, 4, dtype=np.uint8)
    for ch, c in ((0, 0), (1, 1), (2, 2), (3, 3)):
        code[ch] = c
    for ch, c in zip(b"ACGT", range(4)):
        code[ch] = c
        code[ch | 0x20] = c
    c = code[a]
    nw = (len(a) + 31) // 32
    lo = np.zeros(nw * 32, dtype=np.uint64)
    hi = np.zeros(nw * 32, dtype=np.uint64)
    v = np.zeros(nw * 32, dtype=np.uint64)
    ok = c < 4
    lo[:len(a)] = np.where(ok, c & 1, 0)
    hi[:len(a)] = np.where(ok, c >> 1, 0)
    v[:len(a)] = ok
    sh = (np.uint64(31) - np.arange(32, dtype=np.uint64))[None, :]
    pl = (lo.reshape(nw, 32) << sh).sum(axis=1, dtype=np.uint64)
    ph = (hi.reshape(nw, 32) << sh).sum(axis=1, dtype=np.uint64)
    pv = (v.reshape(nw, 32) << sh).sum(axis=1, dtype=np.uint64)
    return pl | (ph << np.uint64(32)), pv.astype(np.uint32), int((~ok).sum())


def cases():
    rng = np.random.default_rng(5)
    out = [b"", b"A", b"ACGT" * 8, b"ACGT" * 8 + b"N", bytes(range(256)), bytes(range(256)) * 3 + b"acgtn"]
    for n in (31, 32, 33, 63, 64, 65, 1000, 65536 * 32 + 17):  # the last one crosses a job boundary of the packer
        out.append(rng.choice(np.frombuffer(b"ACGTacgtNn\x00\x01\x02\x03*-", dtype=np.uint8), n).tobytes())
    out.append(rng.integers(0, 256, 5000, dtype=np.uint8).tobytes())
    for n in range(0, 200):  # every length around the 32- and 64-byte steps (the AVX-512 path's masked last words), any byte value
        out.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes() if n % 3 else rng.choice(np.frombuffer(b"ACGTN\x00\x03", dtype=np.uint8), n).tobytes())
    return out


def check(P):
    seqs = cases()
    packed, bad = P.pack_ascii(seqs, n_threads=3)
    off = 0
    tot_bad = 0
    for s in seqs:
        pl, pv, nb = ref_pack(s)
        nw = len(pl)
        assert np.array_equal(packed.planes[off:off + nw], pl), len(s)
        assert np.array_equal(packed.valid[off:off + nw], pv), len(s)
        off += nw
        tot_bad += nb
    assert off == packed.planes.size and bad == tot_bad
    assert [int(v) for v in packed.lens] == [len(s) for s in seqs]


def test_pack_ascii_matches_the_base_table():
    sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
    import pgrtk_amd as P
    check(P)


def test_pack_ascii_scalar_path_matches_too():
    """the same with the AVX2 path switched off (PGR_NO_AVX2 is read once per process)"""
    env = dict(os.environ, PGR_NO_AVX2="1", PGR_HOST_THREADS="2")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_hostpack_cpu as t; import pgrtk_amd as P; "
            "t.check(P); print('ok')") % (os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_pack_ascii_avx2_path_matches_too():
    """the same with only the AVX-512 path switched off (on a CPU that has it: the 32-byte AVX2 steps + the byte loop)"""
    env = dict(os.environ, PGR_NO_AVX512="1", PGR_HOST_THREADS="2")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_hostpack_cpu as t; import pgrtk_amd as P; "
            "t.check(P); print('ok')") % (os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_shard_splitters_are_quantiles_of_the_pooled_sample():
    """pgr_shard_splitters (host only): world - 1 ascending splitters at the quantiles of the pooled samples, the same from any
    order of the pool; a record's rank = number of splitters <= its first hash"""
    sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
    from pgrtk_amd import exchange
    rng = np.random.default_rng(9)
    pool = [rng.integers(0, 2 ** 50, n, dtype=np.uint64) for n in (4096, 100, 0, 4096)]
    for world in (1, 2, 3, 8):
        spl = exchange.shard_splitters(pool, world)
        assert len(spl) == world - 1 and list(spl) == sorted(spl)
        spl2 = exchange.shard_splitters(pool[::-1], world)
        assert np.array_equal(spl, spl2)
        allv = np.sort(np.concatenate(pool))
        if world > 1:
            dest = np.searchsorted(spl, allv, side="right")
            sizes = np.bincount(dest, minlength=world)
            assert sizes.max() - sizes.min() <= 2  # the pool itself is cut into equal parts
    assert len(exchange.shard_splitters([np.zeros(0, dtype=np.uint64)], 4)) == 3  # empty pool: still world - 1 splitters


def test_stream_packers_against_the_byte_table(tmp_path):
    """the non-temporal-store packers the staging windows are filled with (pack_words_stream, pack_words_stream_nofence,
    stream_copy: csrc/hostpack.cpp) against a scalar reading of shmmrutils.rs:426-436, through a C++ harness built here with
    g++ (no GPU, no HIP): every byte class, ragged lengths, sub-ranges of a contig's words, unaligned destinations, nothing
    written outside the destination; with the AVX-512 / AVX2 / scalar variants the CPU offers"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "pgr-tk_amd", "csrc")
    exe = str(tmp_path / "hostpack_stream_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", csrc, os.path.join(root, "tests", "hostpack_stream_harness.cpp"),
                    os.path.join(csrc, "hostpack.cpp"), "-o", exe], check=True, timeout=300)
    for env_extra in ({}, {"PGR_NO_AVX512": "1"}, {"PGR_NO_AVX2": "1"}):
        r = subprocess.run([exe, "3000"], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env_extra))
        assert r.returncode == 0, (env_extra, r.stdout, r.stderr)
        assert "3000 cases, 0 failures" in r.stdout
