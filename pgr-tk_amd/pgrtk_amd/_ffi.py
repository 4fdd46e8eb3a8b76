"""ctypes binding of libpgrhip.so (include/pgr_hip.h).

The product has no CPU path: if the shared library is missing or no gfx950 device is
usable, importing is fine but creating a context raises loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PGR_HIP_LIB: another build of the same library, for A/B timing of kernel variants)
LIB_PATH = os.environ.get("PGR_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libpgrhip.so")

MM128 = np.dtype([("x", "<u8"), ("y", "<u8")])
FRAG_REC = np.dtype([("h0", "<u8"), ("h1", "<u8"), ("frg_id", "<u4"), ("sid", "<u4"), ("bgn", "<u4"),
                     ("end", "<u4"), ("orient", "<u4"), ("_pad", "<u4")])
HITPAIR = np.dtype([("qb", "<u4"), ("qe", "<u4"), ("qo", "<u4"), ("tb", "<u4"), ("te", "<u4"), ("to", "<u4")])
VERTEX = np.dtype([("h0", "<u8"), ("h1", "<u8"), ("orient", "<u4"), ("count", "<u4")])
ADJ_PAIR = np.dtype([("sid", "<u4"), ("_pad", "<u4"), ("v", VERTEX), ("w", VERTEX)])
DFS_NODE = np.dtype([("node", VERTEX), ("parent", VERTEX), ("has_parent", "<u4"), ("is_leaf", "<u4"), ("rank", "<u4"),
                     ("branch", "<u4"), ("branch_rank", "<u4"), ("_pad", "<u4")])
SMP_BUNDLE = np.dtype([("h0", "<u8"), ("h1", "<u8"), ("bgn", "<u4"), ("end", "<u4"), ("orient", "<u4"), ("sid", "<u4"),
                       ("bundle_id", "<i4"), ("bundle_dir", "<u4"), ("bundle_pos", "<u4"), ("_pad", "<u4")])


class PgrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libpgrhip error %d: %s" % (code, msg))
        self.code = code


class Spec(C.Structure):
    """ShmmrSpec (pgr-db/src/shmmrutils.rs:20-27)"""
    _fields_ = [("w", C.c_uint32), ("k", C.c_uint32), ("r", C.c_uint32), ("min_span", C.c_uint32),
                ("sketch", C.c_uint32)]

    def as_tuple(self):
        return (self.w, self.k, self.r, self.min_span, bool(self.sketch))


class Prof(C.Structure):
    _fields_ = [("level1_ms", C.c_float), ("level1_aux_ms", C.c_float), ("level2_ms", C.c_float),
                ("total_ms", C.c_float), ("n_level1", C.c_uint64), ("n_tiles", C.c_uint64),
                ("n_serial_contigs", C.c_uint64), ("bases_tiled", C.c_uint64), ("exact_bases", C.c_uint64)]


class HpsResult(C.Structure):
    _fields_ = [("n_queries", C.c_uint32), ("q_off", C.POINTER(C.c_uint64)), ("n_targets", C.c_uint64),
                ("t_sid", C.POINTER(C.c_uint32)), ("t_off", C.POINTER(C.c_uint64)), ("n_chains", C.c_uint64),
                ("c_score", C.POINTER(C.c_float)), ("c_off", C.POINTER(C.c_uint64)), ("n_hps", C.c_uint64),
                ("hps", C.c_void_p), ("n_nonterminating", C.c_uint64), ("_owner", C.c_void_p)]


class QueryProf(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("query_bases", C.c_uint64), ("n_query_pairs", C.c_uint64),
                ("n_signatures", C.c_uint64), ("n_hits", C.c_uint64), ("n_groups", C.c_uint64), ("n_chains", C.c_uint64),
                ("n_hps", C.c_uint64), ("stage_ms", C.c_float), ("shmmr_ms", C.c_float), ("lookup_ms", C.c_float),
                ("chain_ms", C.c_float), ("result_ms", C.c_float), ("total_ms", C.c_float),
                ("path", C.c_uint32), ("_pad", C.c_uint32)]


class Bundles(C.Structure):
    _fields_ = [("n_bundles", C.c_uint64), ("b_off", C.POINTER(C.c_uint64)), ("bundle_id", C.POINTER(C.c_uint64)),
                ("mean_ord", C.POINTER(C.c_uint64)), ("n_vertices", C.c_uint64), ("vertices", C.c_void_p)]


# every symbol declared in include/pgr_hip.h: (name, restype, argtypes)
_VP = C.c_void_p
_PVP = C.POINTER(C.c_void_p)
_SIGS = [
    ("pgr_ctx_create", C.c_int, [C.c_int, _PVP]),
    ("pgr_ctx_create_beside", C.c_int, [_VP, _PVP]),
    ("pgr_ctx_destroy", None, [_VP]),
    ("pgr_last_error", C.c_char_p, [_VP]),
    ("pgr_free", None, [_VP]),
    ("pgr_version", C.c_char_p, []),
    ("pgr_host_register", C.c_int, [_VP, C.c_size_t]),
    ("pgr_host_unregister", C.c_int, [_VP]),
    ("pgr_ctx_trim", C.c_int, [_VP]),
    ("pgr_ctx_mem_stats", C.c_int, [_VP, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]),
    ("pgr_ctx_reserve", C.c_int, [_VP, C.c_uint64]),
    ("pgr_debug_take_hip_error", C.c_int, []),
    ("pgr_ctx_arena_stats", C.c_int, [_VP] + [C.POINTER(C.c_uint64)] * 5),
    ("pgr_ctx_set_option", C.c_int, [_VP, C.c_char_p, C.c_int64]),
    ("pgr_ctx_get_option", C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int64)]),
    ("pgr_shmmr_batch", C.c_int, [_VP, C.POINTER(Spec), C.c_uint32, _PVP, C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_uint32), C.c_int, _PVP, _PVP]),
    ("pgr_frag_recs_batch", C.c_int, [_VP, C.POINTER(Spec), C.c_uint32, _PVP, C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint32), C.c_int, _PVP, _PVP]),
    ("pgr_batch_from_ascii", C.c_int, [_VP, C.c_uint32, _PVP, C.POINTER(C.c_uint64), _PVP]),
    ("pgr_packed_words", C.c_uint64, [C.c_uint32, C.POINTER(C.c_uint64)]),
    ("pgr_pack_ascii", C.c_int, [C.c_uint32, _PVP, C.POINTER(C.c_uint64), C.c_int, _VP, _VP, C.POINTER(C.c_uint64)]),
    ("pgr_shmmr_batch_packed", C.c_int, [_VP, C.POINTER(Spec), C.c_uint32, C.POINTER(C.c_uint64), _VP, _VP,
                                         C.POINTER(C.c_uint32), C.c_int, _PVP, _PVP]),
    ("pgr_batch_from_packed", C.c_int, [_VP, C.c_uint32, C.POINTER(C.c_uint64), _VP, _VP, _PVP]),
    ("pgr_index_add_packed", C.c_int, [_VP, _VP, C.c_uint32, C.POINTER(C.c_uint64), _VP, _VP, C.POINTER(C.c_uint32)]),
    ("pgr_batch_synthetic", C.c_int, [_VP, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, _PVP]),
    ("pgr_batch_synthetic_ids", C.c_int, [_VP, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64), _PVP]),
    ("pgr_batch_destroy", None, [_VP]),
    ("pgr_batch_total_bases", C.c_uint64, [_VP]),
    ("pgr_shmmrs_compute", C.c_int, [_VP, _VP, C.POINTER(Spec), C.POINTER(C.c_uint32), C.c_int, _PVP]),
    ("pgr_shmmrs_count", C.c_uint64, [_VP]),
    ("pgr_shmmrs_device_ptr", _VP, [_VP]),
    ("pgr_shmmrs_device_offsets", _VP, [_VP]),
    ("pgr_shmmrs_download", C.c_int, [_VP, _VP, _PVP, _PVP]),
    ("pgr_shmmrs_destroy", None, [_VP]),
    ("pgr_shmmrs_n_pairs", C.c_uint64, [_VP]),
    ("pgr_shmmrs_copy_to_device", C.c_int, [_VP, _VP, _VP, C.c_uint64, C.c_uint32]),
    ("pgr_shmmrs_copy_to_device_rids", C.c_int, [_VP, _VP, _VP, C.c_uint64, C.POINTER(C.c_uint32)]),
    ("pgr_shmmrs_offsets", C.c_int, [_VP, _VP]),
    ("pgr_exchange_unique_id", C.c_int, [_VP, _VP]),
    ("pgr_exchange_create", C.c_int, [_VP, _VP, C.c_int, C.c_int, _PVP]),
    ("pgr_exchange_destroy", None, [_VP]),
    ("pgr_exchange_rank", C.c_int, [_VP]),
    ("pgr_exchange_world", C.c_int, [_VP]),
    ("pgr_exchange_allgather_shmmrs_start", C.c_int, [_VP, _VP, C.c_uint64, _VP, C.c_uint64]),
    ("pgr_exchange_wait", C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    ("pgr_exchange_device_counts", _VP, [_VP]),
    ("pgr_exchange_gather_into_index", C.c_int, [_VP, _VP, C.POINTER(C.c_uint32), _VP, C.POINTER(C.c_uint64)]),
    ("pgr_exchange_shard_records", C.c_int, [_VP, _VP, C.c_uint64, _VP, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("pgr_exchange_allgather_index", C.c_int, [_VP, _VP, _PVP]),
    ("pgr_shard_sample_keys", C.c_int, [_VP, _VP, C.c_uint64, C.c_uint32, _VP, C.POINTER(C.c_uint32)]),
    ("pgr_shard_splitters", C.c_int, [_VP, C.c_uint64, C.c_int, _VP]),
    ("pgr_shard_partition", C.c_int, [_VP, _VP, C.c_uint64, _VP, C.c_int, _VP, _VP]),
    ("pgr_records_checksum", C.c_int, [_VP, _VP, C.c_uint64, _VP]),
    ("pgr_index_records_checksum", C.c_int, [_VP, _VP, _VP]),
    ("pgr_index_device_records", _VP, [_VP]),
    ("pgr_index_key_range", C.c_int, [_VP, _VP, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("pgr_index_add_shmmrs", C.c_int, [_VP, _VP, _VP, C.c_uint64, C.c_int]),
    ("pgr_shmmrs_to_frag_recs_device", C.c_int, [_VP, _VP, C.POINTER(C.c_uint32), C.c_int, _VP, C.c_uint64,
                                                 C.POINTER(C.c_uint64)]),
    ("pgr_shmmrs_compute_recs", C.c_int, [_VP, _VP, C.POINTER(Spec), C.POINTER(C.c_uint32), _VP, C.c_uint64, _PVP, C.POINTER(C.c_uint64)]),
    ("pgr_pipe_create", C.c_int, [_VP, C.POINTER(Spec), _PVP]),
    ("pgr_pipe_submit", C.c_int, [_VP, _VP, C.POINTER(C.c_uint32), _VP, _VP, C.c_uint64]),
    ("pgr_pipe_collect", C.c_int, [_VP, _PVP, C.POINTER(C.c_uint64)]),
    ("pgr_pipe_in_flight", C.c_int, [_VP]),
    ("pgr_pipe_destroy", None, [_VP]),
    ("pgr_ctx_last_prof", C.c_int, [_VP, C.POINTER(Prof)]),
    ("pgr_ctx_synchronize", C.c_int, [_VP]),
    ("pgr_index_create", C.c_int, [_VP, C.POINTER(Spec), _PVP]),
    ("pgr_index_destroy", None, [_VP]),
    ("pgr_index_add_batch", C.c_int, [_VP, _VP, C.c_uint32, _PVP, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    ("pgr_index_add_resident", C.c_int, [_VP, _VP, _VP, C.POINTER(C.c_uint32)]),
    ("pgr_index_reserve", C.c_int, [_VP, _VP, C.c_uint64]),
    ("pgr_index_add_records", C.c_int, [_VP, _VP, _VP, C.c_uint64, C.c_int]),
    ("pgr_index_finalize", C.c_int, [_VP, _VP]),
    ("pgr_index_n_keys", C.c_uint64, [_VP]),
    ("pgr_index_n_records", C.c_uint64, [_VP]),
    ("pgr_index_download", C.c_int, [_VP, _VP, _PVP, C.POINTER(C.c_uint64)]),
    ("pgr_index_write_mdb", C.c_int, [_VP, _VP, C.c_char_p]),
    ("pgr_index_load_mdb", C.c_int, [_VP, C.c_char_p, _PVP]),
    ("pgr_index_spec", C.c_int, [_VP, C.POINTER(Spec)]),
    ("pgr_query_hps_batch", C.c_int, [_VP, _VP, C.c_uint32, _PVP, C.POINTER(C.c_uint64), C.c_float, C.c_uint32,
                                      C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int,
                                      C.POINTER(HpsResult)]),
    ("pgr_query_hps_resident", C.c_int, [_VP, _VP, _VP, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                         C.c_uint32, C.c_int, C.POINTER(HpsResult)]),
    ("pgr_hps_result_free", None, [C.POINTER(HpsResult)]),
    ("pgr_pipe_submit_query", C.c_int, [_VP, _VP, _VP, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                        C.c_uint32, C.c_int]),
    ("pgr_pipe_collect_query", C.c_int, [_VP, C.POINTER(HpsResult)]),
    ("pgr_ctx_last_query_prof", C.c_int, [_VP, C.POINTER(QueryProf)]),
    ("pgr_shmmrs_checksum", C.c_int, [_VP, _VP, _VP]),
    ("pgr_sparse_aln_batch", C.c_int, [_VP, C.c_uint32, _VP, C.POINTER(C.c_uint64), C.c_uint32, C.c_float, C.c_int,
                                       C.c_uint32, C.c_int, C.POINTER(HpsResult)]),
    ("pgr_index_adj_list", C.c_int, [_VP, _VP, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, _PVP,
                                     C.POINTER(C.c_uint64)]),
    ("pgr_index_key_counts", C.c_int, [_VP, _VP, C.c_uint64, _VP, _VP]),
    ("pgr_sort_adj_list_by_weighted_dfs", C.c_int, [_VP, _VP, C.c_uint64, _VP, _PVP, C.POINTER(C.c_uint64)]),
    ("pgr_bundles_free", None, [C.POINTER(Bundles)]),
    ("pgr_principal_bundles", C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32,
                                        C.POINTER(Bundles)]),
    ("pgr_principal_bundles_from_adj_list", C.c_int, [_VP, _VP, C.c_uint64, C.c_uint32, C.POINTER(Bundles)]),
    ("pgr_principal_bundle_decomposition", C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32,
                                                     C.POINTER(Bundles), _PVP, C.POINTER(C.c_uint64), _PVP, _PVP,
                                                     C.POINTER(C.c_uint32)]),
    ("pgr_principal_bundle_projection", C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32,
                                                  C.c_uint32, _PVP, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                                  C.POINTER(Bundles), _PVP, C.POINTER(C.c_uint64), _PVP]),
]
SYMBOLS = [s[0] for s in _SIGS]

_lib = None


def _share_torch_hip_runtime():
    """PyTorch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, no soname); libpgrhip.so links the
    system one.  Two runtimes in one process only work when torch's initialises first -- otherwise torch.cuda finds no
    device later.  When torch is installed, load ITS runtime globally before libpgrhip.so: the library then binds to
    it, the process has one runtime, and the import order of torch and this package stops mattering.
    (PGR_NO_TORCH_PRELOAD=1 keeps the system runtime.)"""
    if os.environ.get("PGR_NO_TORCH_PRELOAD"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(rt):
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # no torch, or an unusual install: the system runtime is used


def lib():
    """load libpgrhip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libpgrhip.so not built: run `python __graft_entry__.py build` "
                              "(make -C pgr-tk_amd); expected at " + LIB_PATH)
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, res, args in _SIGS:
            f = getattr(L, name)  # AttributeError if a declared symbol is missing
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def seq_array(seq):
    if isinstance(seq, np.ndarray):
        return np.ascontiguousarray(seq, dtype=np.uint8)
    if isinstance(seq, str):
        seq = seq.encode()
    return np.frombuffer(bytes(seq), dtype=np.uint8)


class PackedSeqs:
    """n sequences back to back in ONE uint8 buffer: sequence i = buf[off[i]:off[i+1]].  The pointer array the
    C ABI wants is then base + off, computed vectorised (a list of 10 000 Python objects costs milliseconds)."""

    def __init__(self, buf, off):
        self.buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self.off = np.ascontiguousarray(off, dtype=np.uint64)
        assert self.off.ndim == 1 and len(self.off) >= 1 and int(self.off[-1]) <= self.buf.size

    @classmethod
    def from_list(cls, seqs):
        arrs = [seq_array(s) for s in seqs]
        off = np.zeros(len(arrs) + 1, dtype=np.uint64)
        if arrs:
            off[1:] = np.cumsum([a.size for a in arrs], dtype=np.uint64)
        return cls(np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.uint8), off)

    def __len__(self):
        return len(self.off) - 1


def seq_ptrs(seqs):
    """-> (keepalive objects, void**, uint64* lens, n).  PackedSeqs: vectorised; bytes go through a c_char_p
    array (converted at C speed); numpy arrays through their data pointers."""
    n = len(seqs)
    if isinstance(seqs, PackedSeqs):
        base = seqs.buf.__array_interface__["data"][0]
        addr = np.zeros(max(n, 1), dtype=np.uint64)
        lens_np = np.zeros(max(n, 1), dtype=np.uint64)
        addr[:n] = np.uint64(base) + seqs.off[:-1]
        lens_np[:n] = seqs.off[1:] - seqs.off[:-1]
        return (seqs, addr, lens_np), addr.ctypes.data_as(_PVP), lens_np.ctypes.data_as(C.POINTER(C.c_uint64)), n
    if n and all(type(s) is bytes for s in seqs):
        ptrs = C.cast((C.c_char_p * n)(*seqs), _PVP)
        lens_np = np.fromiter(map(len, seqs), dtype=np.uint64, count=n)
        return (seqs, lens_np), ptrs, lens_np.ctypes.data_as(C.POINTER(C.c_uint64)), n
    arrs = [seq_array(s) for s in seqs]
    addr = np.zeros(max(n, 1), dtype=np.uint64)
    lens_np = np.zeros(max(n, 1), dtype=np.uint64)
    for i, a in enumerate(arrs):
        lens_np[i] = a.size
        addr[i] = a.__array_interface__["data"][0] if a.size else 0
    return (arrs, addr, lens_np), addr.ctypes.data_as(_PVP), lens_np.ctypes.data_as(C.POINTER(C.c_uint64)), n


class _FreeOnDel:
    """pgr_free of a library-allocated host buffer when the last numpy view of it is gone"""

    def __init__(self, addr):
        self._addr = addr

    def __del__(self):
        try:
            lib().pgr_free(C.c_void_p(self._addr))
        except Exception:
            pass


def take(ptr, n, dtype):
    """n records of a library-allocated host buffer as a numpy array that owns them: small buffers are copied and released at
    once, buffers of a MB and more become a (writable) view of the library's memory, released with the last view -- the
    shimmers of a 1 Gbp batch are 48 MB, copying them costs a quarter of the whole call"""
    dtype = np.dtype(dtype)
    if not ptr.value:
        return np.zeros(n, dtype=dtype)
    nbytes = n * dtype.itemsize
    if nbytes < (1 << 20):
        out = np.zeros(n, dtype=dtype)
        if n:
            C.memmove(out.ctypes.data, ptr.value, nbytes)
        lib().pgr_free(ptr)
        return out
    buf = (C.c_char * nbytes).from_address(ptr.value)
    buf._owner = _FreeOnDel(ptr.value)  # the array's base keeps the ctypes buffer alive, the buffer keeps the allocation
    return np.frombuffer(buf, dtype=dtype)


class Context:
    """one GPU, one stream (pgr_ctx).  Not thread safe."""

    def __init__(self, device=0, beside=None):
        """beside: another Context -- this one's work runs side by side with that one's (pgr_ctx_create_beside: one context per
        host thread, e.g. two query batches in flight against one index)"""
        self._h = C.c_void_p()
        self.device = int(device if beside is None else beside.device)
        rc = lib().pgr_ctx_create(device, C.byref(self._h)) if beside is None else lib().pgr_ctx_create_beside(beside.handle, C.byref(self._h))
        if rc != 0:
            msg = lib().pgr_last_error(None)
            raise PgrError(rc, msg.decode() if msg else "pgr_ctx_create failed")

    def close(self):
        if self._h:
            lib().pgr_ctx_destroy(self._h)
            self._h = C.c_void_p()

    @property
    def alive(self):
        """the native context still exists.  Objects of a context (Batch, Shmmrs, Pipe, Index, SeqIndexDB, AbiExchange) hold a
        reference to it, so by reference counting it outlives them -- but inside a garbage CYCLE (a test's frame kept by a
        traceback, say) Python runs finalizers in arbitrary order, and pgr_*_destroy on an object of a destroyed context walks
        freed memory (round 6: the HIP "invalid value" that surfaced in an unrelated scan began there).  Their close() looks here
        first: a destroyed context has released every device block of its objects already (pgr_ctx_destroy), only the small host
        structs stay behind."""
        return bool(self._h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def check(self, rc):
        if rc != 0:
            msg = lib().pgr_last_error(self._h)
            raise PgrError(rc, msg.decode() if msg else "?")

    def last_prof(self):
        p = Prof()
        self.check(lib().pgr_ctx_last_prof(self._h, C.byref(p)))
        return p

    def synchronize(self):
        self.check(lib().pgr_ctx_synchronize(self._h))

    def trim(self):
        """give the allocator's cached blocks back to the device"""
        self.check(lib().pgr_ctx_trim(self._h))

    def mem_stats(self, reset_peak=False):
        """-> (bytes the allocator holds now, its high-water mark)"""
        a, b = C.c_uint64(), C.c_uint64()
        self.check(lib().pgr_ctx_mem_stats(self._h, C.byref(a), C.byref(b), int(reset_peak)))
        return int(a.value), int(b.value)

    def reserve(self, n_bytes):
        """one device block, allocated and touched now, that every later device allocation of this context is carved from
        (include/pgr_hip.h: pgr_ctx_reserve)"""
        self.check(lib().pgr_ctx_reserve(self._h, int(n_bytes)))

    def arena_stats(self):
        """-> dict(reserved, used, peak_used, fallback_bytes, fallback_calls) of the reserved arena (all 0 without a reserve)"""
        v = [C.c_uint64() for _ in range(5)]
        self.check(lib().pgr_ctx_arena_stats(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("reserved", "used", "peak_used", "fallback_bytes", "fallback_calls"), (int(x.value) for x in v)))

    def set_option(self, name, value=1):
        """tuning / A-B switch of this context (include/pgr_hip.h: pgr_ctx_set_option)"""
        self.check(lib().pgr_ctx_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int64()
        if lib().pgr_ctx_get_option(self._h, name.encode(), C.byref(v)) != 0:
            raise KeyError(name)
        return int(v.value)

    def options(self, **kw):
        """context manager: set options for the duration of a `with` block, then restore them"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = {k: self.get_option(k) for k in kw}
            try:
                for k, v in kw.items():
                    self.set_option(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(k, v)
        return cm()

    def last_query_prof(self):
        """counts and stage times of the last query batch on this context (pgr_query_prof) as a dict"""
        p = QueryProf()
        self.check(lib().pgr_ctx_last_query_prof(self._h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in QueryProf._fields_}


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
