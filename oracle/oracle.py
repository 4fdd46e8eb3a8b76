"""ctypes wrapper around oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (pgr-tk_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

MM128 = np.dtype([("x", "<u8"), ("y", "<u8")])
FRAG_REC = np.dtype(
    [("h0", "<u8"), ("h1", "<u8"), ("frg_id", "<u4"), ("sid", "<u4"), ("bgn", "<u4"), ("end", "<u4"),
     ("orient", "<u4"), ("_pad", "<u4")]
)
HITPAIR = np.dtype([("qb", "<u4"), ("qe", "<u4"), ("qo", "<u4"), ("tb", "<u4"), ("te", "<u4"), ("to", "<u4")])


class Spec(C.Structure):
    _fields_ = [("w", C.c_uint32), ("k", C.c_uint32), ("r", C.c_uint32), ("min_span", C.c_uint32),
                ("sketch", C.c_uint32)]


class _TargetResult(C.Structure):
    _fields_ = [("sid", C.c_uint32), ("n_chains", C.c_uint32), ("chain_first", C.c_uint32)]


class _HpsResult(C.Structure):
    _fields_ = [
        ("n_targets", C.c_size_t), ("targets", C.POINTER(_TargetResult)),
        ("n_chains", C.c_size_t), ("chain_score", C.POINTER(C.c_float)),
        ("chain_first_hp", C.POINTER(C.c_uint32)), ("chain_n_hp", C.POINTER(C.c_uint32)),
        ("n_hps", C.c_size_t), ("hps", C.c_void_p),
    ]


def build(force=False):
    """(Re)build liboracle.so with gcc.  Building the checker is not using it."""
    src = os.path.join(_HERE, "pgr_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src),
                                                 os.path.getmtime(os.path.join(_HERE, "pgr_oracle.h")))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_u64hash.restype = C.c_uint64
        L.orc_u64hash.argtypes = [C.c_uint64]
        L.orc_level1_minimizers.restype = C.c_size_t
        L.orc_level1_minimizers.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32,
                                            C.POINTER(C.c_void_p)]
        L.orc_reduce_shmmr.restype = C.c_size_t
        L.orc_reduce_shmmr.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
        L.orc_sequence_to_shmmrs.restype = C.c_size_t
        L.orc_sequence_to_shmmrs.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(Spec), C.c_int,
                                             C.POINTER(C.c_void_p)]
        L.orc_shmmrs_to_frag_recs.restype = C.c_size_t
        L.orc_shmmrs_to_frag_recs.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int,
                                              C.POINTER(C.c_void_p)]
        L.orc_index_new.restype = C.c_void_p
        L.orc_index_new.argtypes = [C.POINTER(Spec)]
        L.orc_index_free.argtypes = [C.c_void_p]
        L.orc_index_add_seq.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        L.orc_index_add_seq_fastx_ids.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        L.orc_index_finalize.argtypes = [C.c_void_p]
        L.orc_index_n_keys.restype = C.c_size_t
        L.orc_index_n_keys.argtypes = [C.c_void_p]
        L.orc_index_n_recs.restype = C.c_size_t
        L.orc_index_n_recs.argtypes = [C.c_void_p]
        L.orc_index_recs.restype = C.c_void_p
        L.orc_index_recs.argtypes = [C.c_void_p]
        L.orc_query_fragment_to_hps.restype = C.c_int
        L.orc_query_fragment_to_hps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_uint32,
                                                C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32,
                                                C.c_int, C.POINTER(_HpsResult)]
        L.orc_hps_result_free.argtypes = [C.POINTER(_HpsResult)]
        L.orc_sparse_aln.restype = C.c_int
        L.orc_sparse_aln.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_float, C.c_int, C.c_uint32,
                                     C.c_int, C.POINTER(_HpsResult)]
        L.orc_synth_contig.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p]
        L.orc_shmmr_batch_threads.restype = C.c_uint64
        L.orc_shmmr_batch_threads.argtypes = [C.POINTER(Spec), C.c_uint32, C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_uint64)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_shmmr_checksum.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_synth_checksums_threads.restype = C.c_int
        L.orc_synth_checksums_threads.argtypes = [C.POINTER(Spec), C.c_uint32, C.c_uint64, C.c_uint64, C.c_size_t, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_synth_checksums_ids_threads.restype = C.c_int
        L.orc_synth_checksums_ids_threads.argtypes = [C.POINTER(Spec), C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p,
                                                      C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_index_add_synth_threads.restype = C.c_int
        L.orc_index_add_synth_threads.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_size_t, C.c_int]
        L.orc_query_batch_threads.restype = C.c_int
        L.orc_query_batch_threads.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_float,
                                              C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int,
                                              C.c_int, C.POINTER(_HpsResult), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _as_bytes_array(seq):
    if isinstance(seq, (bytes, bytearray)):
        return np.frombuffer(bytes(seq), dtype=np.uint8)
    if isinstance(seq, str):
        return np.frombuffer(seq.encode(), dtype=np.uint8)
    return np.ascontiguousarray(seq, dtype=np.uint8)


def _take(ptr, n, dtype):
    """copy n records out of a malloc'ed buffer and free it"""
    if n == 0 or not ptr.value:
        if ptr.value:
            lib().orc_free(ptr)
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr.value)
    out = np.frombuffer(buf, dtype=dtype).copy()
    lib().orc_free(ptr)
    return out


def u64hash(v):
    return int(lib().orc_u64hash(C.c_uint64(v)))


def spec(w=80, k=56, r=4, min_span=64, sketch=False):
    return Spec(w, k, r, min_span, 1 if sketch else 0)


def level1(seq, w=80, k=56, rid=0):
    a = _as_bytes_array(seq)
    p = C.c_void_p()
    n = lib().orc_level1_minimizers(rid, a.ctypes.data, a.size, w, k, C.byref(p))
    return _take(p, n, MM128)


def reduce_shmmr(mers, r, padding=False):
    a = np.ascontiguousarray(mers, dtype=MM128)
    p = C.c_void_p()
    n = lib().orc_reduce_shmmr(a.ctypes.data, a.size, r, int(padding), C.byref(p))
    return _take(p, n, MM128)


def sequence_to_shmmrs(rid, seq, sp, padding=False):
    a = _as_bytes_array(seq)
    p = C.c_void_p()
    n = lib().orc_sequence_to_shmmrs(rid, a.ctypes.data, a.size, C.byref(sp), int(padding), C.byref(p))
    if n == (1 << (8 * C.sizeof(C.c_size_t))) - 1:
        raise ValueError("spec rejected (reference would assert)")
    return _take(p, n, MM128)


def frag_recs(shmmrs, sid, query_side=False):
    a = np.ascontiguousarray(shmmrs, dtype=MM128)
    p = C.c_void_p()
    n = lib().orc_shmmrs_to_frag_recs(a.ctypes.data, a.size, sid, int(query_side), C.byref(p))
    return _take(p, n, FRAG_REC)


def synth_contig(seed, contig, length):
    out = np.empty(length, dtype=np.uint8)
    lib().orc_synth_contig(seed, contig, length, out.ctypes.data)
    return out


def shmmr_batch_threads(sp, seqs, n_threads):
    """threaded CPU baseline (one task per contig). returns (total, counts)"""
    arrs = [_as_bytes_array(s) for s in seqs]
    n = len(arrs)
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    lens = (C.c_uint64 * n)(*[a.size for a in arrs])
    counts = (C.c_uint64 * n)()
    tot = lib().orc_shmmr_batch_threads(C.byref(sp), n, ptrs, lens, n_threads, counts)
    return int(tot), np.array(counts[:], dtype=np.uint64)


def synth_checksums_threads(sp, n, seed, contig0, length, n_threads):
    """all n synthetic contigs (generated inside the workers, one task per contig like rayon par_iter) ->
    (counts[n], checksums[n, 2], busy seconds per thread spent inside sequence_to_shmmrs)"""
    counts = np.zeros(max(n, 1), dtype=np.uint64)
    sums = np.zeros((max(n, 1), 2), dtype=np.uint64)
    busy = np.zeros(max(n_threads, 1), dtype=np.float64)
    lib().orc_synth_checksums_threads(C.byref(sp), n, seed, contig0, length, n_threads, counts.ctypes.data, sums.ctypes.data,
                                      busy.ctypes.data)
    return counts[:n], sums[:n], busy


def synth_checksums_ids_threads(sp, ids, seed, length, n_threads):
    """the same for an explicit list of contig ids (one rank's shard of a partitioned contig set)"""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    n = int(ids.size)
    counts = np.zeros(max(n, 1), dtype=np.uint64)
    sums = np.zeros((max(n, 1), 2), dtype=np.uint64)
    busy = np.zeros(max(n_threads, 1), dtype=np.float64)
    lib().orc_synth_checksums_ids_threads(C.byref(sp), n, seed, 0, ids.ctypes.data if n else None, length, n_threads,
                                          counts.ctypes.data, sums.ctypes.data, busy.ctypes.data)
    return counts[:n], sums[:n], busy


def shmmr_checksum(mm):
    a = np.ascontiguousarray(mm, dtype=MM128)
    out = np.zeros(2, dtype=np.uint64)
    lib().orc_shmmr_checksum(a.ctypes.data, a.size, out.ctypes.data)
    return out


def _unpack_hps(res):
    out = []
    hps = None
    if res.n_hps:
        buf = (C.c_char * (res.n_hps * HITPAIR.itemsize)).from_address(res.hps)
        hps = np.frombuffer(buf, dtype=HITPAIR).copy()
    chains = []
    for c in range(res.n_chains):
        f, n = res.chain_first_hp[c], res.chain_n_hp[c]
        chains.append((float(np.float32(res.chain_score[c])), [tuple(int(v) for v in h) for h in hps[f:f + n]]))
    for t in range(res.n_targets):
        tr = res.targets[t]
        out.append((int(tr.sid), chains[tr.chain_first:tr.chain_first + tr.n_chains]))
    return out, chains


def sparse_aln(hits, max_span, penalty, max_gap=None, oriented=False):
    """hits: array-like of (qb,qe,qo,tb,te,to). returns list of (score, [hitpair tuples])"""
    a = np.array([tuple(h) for h in hits], dtype=HITPAIR) if not isinstance(hits, np.ndarray) else hits.copy()
    res = _HpsResult()
    rc = lib().orc_sparse_aln(a.ctypes.data, a.size, max_span, penalty, int(max_gap is not None),
                              int(max_gap or 0), int(oriented), C.byref(res))
    _, chains = _unpack_hps(res)
    lib().orc_hps_result_free(C.byref(res))
    if rc:
        raise RuntimeError("sparse_aln: reference would assert/loop")
    return chains


class Index:
    """frag_map oracle (sorted CSR of fragment signatures)."""

    def __init__(self, sp):
        self.sp = sp
        self._h = C.c_void_p(lib().orc_index_new(C.byref(sp)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_index_free(self._h)
            self._h = None

    def add_seq(self, sid, seq, fastx_ids=False):
        a = _as_bytes_array(seq)
        f = lib().orc_index_add_seq_fastx_ids if fastx_ids else lib().orc_index_add_seq
        if f(self._h, sid, a.ctypes.data, a.size) != 0:
            raise ValueError("spec rejected")

    def add_synth_threads(self, n, sid0, seed, contig0, length, n_threads):
        lib().orc_index_add_synth_threads(self._h, n, sid0, seed, contig0, length, n_threads)

    def finalize(self):
        lib().orc_index_finalize(self._h)

    def records(self):
        self.finalize()
        n = lib().orc_index_n_recs(self._h)
        if n == 0:
            return np.zeros(0, dtype=FRAG_REC)
        p = lib().orc_index_recs(self._h)
        buf = (C.c_char * (n * FRAG_REC.itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=FRAG_REC).copy()

    def n_keys(self):
        self.finalize()
        return lib().orc_index_n_keys(self._h)

    def query_fragment_to_hps(self, seq, penalty, max_count=128, query_max_count=128, target_max_count=128,
                              max_aln_span=8, max_gap=None, oriented=False):
        self.finalize()
        a = _as_bytes_array(seq)
        res = _HpsResult()
        rc = lib().orc_query_fragment_to_hps(self._h, a.ctypes.data, a.size, penalty, max_count,
                                             query_max_count, target_max_count, max_aln_span,
                                             int(max_gap is not None), int(max_gap or 0), int(oriented),
                                             C.byref(res))
        out, _ = _unpack_hps(res)
        lib().orc_hps_result_free(C.byref(res))
        if rc:
            raise RuntimeError("query_fragment_to_hps failed rc=%d" % rc)
        return out


def query_batch_threads(ix, seqs, penalty, n_threads, max_count=128, query_max_count=128, target_max_count=128,
                        max_aln_span=8, max_gap=None, oriented=False):
    """one task per query on n_threads (= the rayon loop of pgr-query.rs:135-138) -> (list of per-query results,
    wall seconds of the threaded section)"""
    import time
    ix.finalize()
    arrs = [_as_bytes_array(s) for s in seqs]
    n = len(arrs)
    ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
    lens = (C.c_uint64 * max(n, 1))(*[a.size for a in arrs])
    results = (_HpsResult * max(n, 1))()
    rcs = (C.c_int * max(n, 1))()
    t0 = time.perf_counter()
    lib().orc_query_batch_threads(ix._h, n, ptrs, lens, penalty, max_count, query_max_count, target_max_count, max_aln_span,
                                  int(max_gap is not None), int(max_gap or 0), int(oriented), n_threads, results, rcs)
    dt = time.perf_counter() - t0
    out = []
    for i in range(n):
        if rcs[i]:
            out.append(None)
        else:
            out.append(_unpack_hps(results[i])[0])
        lib().orc_hps_result_free(C.byref(results[i]))
    return out, dt


def read_fasta(path):
    """minimal FASTA reader with the reference's record semantics (fasta_io.rs:94-106):
    id = header up to the first space, sequence keeps case, newlines stripped."""
    recs = []
    name, chunks = None, []
    opener = open
    if path.endswith(".gz"):
        import gzip
        opener = gzip.open
    with opener(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    recs.append((name, b"".join(chunks)))
                name = line[1:].split(b" ")[0]
                chunks = []
            else:
                chunks.append(line)
    if name is not None:
        recs.append((name, b"".join(chunks)))
    return recs


def read_mdb(path):
    """parse a reference .mdb (seq_db.rs:1291-1326 layout). returns (spec_tuple, dict key->list of sigs)"""
    d = open(path, "rb").read()
    assert d[:3] == b"mdb"
    w, k, r, ms, flag = np.frombuffer(d, dtype="<u4", count=5, offset=3)
    nkeys = int(np.frombuffer(d, dtype="<u8", count=1, offset=23)[0])
    off = 31
    m = {}
    for _ in range(nkeys):
        h0, h1, n = (int(v) for v in np.frombuffer(d, dtype="<u8", count=3, offset=off))
        off += 24
        sigs = []
        for _ in range(n):
            frg, sid, b, e = (int(v) for v in np.frombuffer(d, dtype="<u4", count=4, offset=off))
            o = d[off + 16]
            off += 17
            sigs.append((frg, sid, b, e, o))
        m[(h0, h1)] = sigs
    assert off == len(d)
    return (int(w), int(k), int(r), int(ms), int(flag)), m
