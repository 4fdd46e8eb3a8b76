"""Thin Python objects over the C ABI: resident batches and the sequence_to_shmmrs hot path.

Names follow the reference (pgr-db/src/shmmrutils.rs, seq_db.rs); the compute is entirely in
libpgrhip.so -- nothing here computes a hash or a minimizer.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import FRAG_REC, MM128, Spec, default_context, lib


def make_spec(w=80, k=56, r=4, min_span=64, sketch=False):
    return Spec(w, k, r, min_span, 1 if sketch else 0)


def _u32_array(vals, n):
    if vals is None:
        return None, None
    a = np.ascontiguousarray(vals, dtype=np.uint32)
    if a.size != n:
        raise ValueError("expected %d ids, got %d" % (n, a.size))
    return a, a.ctypes.data_as(C.POINTER(C.c_uint32))


class PackedBases:
    """contigs as 2-bit planes + validity plane in HOST memory, the layout of pgr_batch_from_packed (include/pgr_hip.h):
    lens[n], planes[words] (low plane | high plane << 32), valid[words] or None (every base valid)"""

    def __init__(self, lens, planes, valid=None):
        self.lens = np.ascontiguousarray(lens, dtype=np.uint64)
        self.planes = np.ascontiguousarray(planes, dtype=np.uint64)
        self.valid = None if valid is None else np.ascontiguousarray(valid, dtype=np.uint32)
        self.n = int(self.lens.size)
        words = int(((self.lens + np.uint64(31)) // np.uint64(32)).sum()) if self.n else 0
        assert self.planes.size >= words and (self.valid is None or self.valid.size >= words)

    def __len__(self):
        return self.n

    @property
    def total_bases(self):
        return int(self.lens.sum()) if self.n else 0

    def _args(self):
        la = self.lens if self.n else np.zeros(1, dtype=np.uint64)
        return (la.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_void_p(self.planes.ctypes.data if self.planes.size else 0),
                C.c_void_p(self.valid.ctypes.data) if self.valid is not None and self.valid.size else None)


class PinnedArrays:
    """context manager: numpy arrays pinned for the DMA engine while the block runs (pgr_host_register / pgr_host_unregister)"""

    def __init__(self, *arrays):
        self.arrays = [a for a in arrays if a is not None and a.size]
        self.done = []

    def __enter__(self):
        for a in self.arrays:
            rc = lib().pgr_host_register(C.c_void_p(a.ctypes.data), a.nbytes)
            if rc != 0:
                self.__exit__(None, None, None)
                raise _ffi.PgrError(rc, (lib().pgr_last_error(None) or b"pgr_host_register failed").decode())
            self.done.append(a)
        return self

    def __exit__(self, *exc):
        for a in self.done:
            lib().pgr_host_unregister(C.c_void_p(a.ctypes.data))
        self.done = []
        return False


def pack_ascii(seqs, n_threads=0):
    """pgr_pack_ascii: the library's threaded CPU packer (no GPU involved) -> (PackedBases, number of non-ACGT bytes)"""
    arrs, ptrs, lens, n = _ffi.seq_ptrs(seqs)
    words = int(lib().pgr_packed_words(n, lens))
    planes = np.empty(max(words, 1), dtype=np.uint64)
    valid = np.empty(max(words, 1), dtype=np.uint32)
    bad = C.c_uint64()
    rc = lib().pgr_pack_ascii(n, ptrs, lens, int(n_threads), C.c_void_p(planes.ctypes.data), C.c_void_p(valid.ctypes.data),
                              C.byref(bad))
    if rc != 0:
        raise _ffi.PgrError(rc, "pgr_pack_ascii: bad arguments")
    lens_np = np.ctypeslib.as_array(lens, shape=(max(n, 1),))[:n].copy()
    return PackedBases(lens_np, planes[:words], valid[:words]), int(bad.value)


class Batch:
    """contigs resident on the GPU as 2-bit planes (pgr_batch)."""

    def __init__(self, ctx, handle, n):
        self.ctx = ctx
        self._h = handle
        self.n = n

    @classmethod
    def from_seqs(cls, seqs, ctx=None):
        ctx = ctx or default_context()
        arrs, ptrs, lens, n = _ffi.seq_ptrs(seqs)
        h = C.c_void_p()
        ctx.check(lib().pgr_batch_from_ascii(ctx.handle, n, ptrs, lens, C.byref(h)))
        return cls(ctx, h, n)

    @classmethod
    def from_packed(cls, packed, ctx=None):
        """H2D of host-packed planes (PackedBases)"""
        ctx = ctx or default_context()
        la, pp, vp = packed._args()
        h = C.c_void_p()
        ctx.check(lib().pgr_batch_from_packed(ctx.handle, packed.n, la, pp, vp, C.byref(h)))
        return cls(ctx, h, packed.n)

    @classmethod
    def synthetic(cls, lens, seed, contig0=0, ctx=None, contig_ids=None):
        """counter-based synthetic contigs generated on the device; contig_ids: explicit global ids (a shard)"""
        ctx = ctx or default_context()
        n = len(lens)
        la = (C.c_uint64 * max(n, 1))(*[int(v) for v in lens])
        h = C.c_void_p()
        if contig_ids is None:
            ctx.check(lib().pgr_batch_synthetic(ctx.handle, n, la, int(seed), int(contig0), C.byref(h)))
        else:
            assert len(contig_ids) == n
            ia = (C.c_uint64 * max(n, 1))(*[int(v) for v in contig_ids])
            ctx.check(lib().pgr_batch_synthetic_ids(ctx.handle, n, la, int(seed), ia, C.byref(h)))
        return cls(ctx, h, n)

    @property
    def total_bases(self):
        return int(lib().pgr_batch_total_bases(self._h))

    def shmmrs(self, spec, rids=None, padding=False):
        keep, rp = _u32_array(rids, self.n)
        h = C.c_void_p()
        self.ctx.check(lib().pgr_shmmrs_compute(self.ctx.handle, self._h, C.byref(spec), rp, int(padding), C.byref(h)))
        return Shmmrs(self.ctx, h, self.n)

    def shmmrs_and_recs(self, spec, rec_ptr, rec_capacity, sids=None):
        """pgr_shmmrs_compute_recs: the shimmer lists and, into caller-owned DEVICE memory, the index-side pair records, in one
        call and one wait -> (Shmmrs, number of records)"""
        keep, sp = _u32_array(sids, self.n)
        h, n_out = C.c_void_p(), C.c_uint64()
        self.ctx.check(lib().pgr_shmmrs_compute_recs(self.ctx.handle, self._h, C.byref(spec), sp, C.c_void_p(rec_ptr), int(rec_capacity),
                                                     C.byref(h), C.byref(n_out)))
        return Shmmrs(self.ctx, h, self.n), int(n_out.value)

    def close(self):
        if self._h:
            if getattr(getattr(self, "ctx", None), "alive", True):  # (_ffi.Context.alive: not on an object of a destroyed context)
                lib().pgr_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Shmmrs:
    """device-resident result of one sequence_to_shmmrs pass over a batch (pgr_shmmrs)."""

    def __init__(self, ctx, handle, n):
        self.ctx = ctx
        self._h = handle
        self.n = n

    @property
    def count(self):
        return int(lib().pgr_shmmrs_count(self._h))

    @property
    def n_pairs(self):
        return int(lib().pgr_shmmrs_n_pairs(self._h))

    @property
    def device_ptr(self):
        return lib().pgr_shmmrs_device_ptr(self._h)

    def download(self):
        """-> (MM128 array, offsets[n+1])"""
        pm, po = C.c_void_p(), C.c_void_p()
        cnt = self.count
        self.ctx.check(lib().pgr_shmmrs_download(self.ctx.handle, self._h, C.byref(pm), C.byref(po)))
        return _ffi.take(pm, cnt, MM128), _ffi.take(po, self.n + 1, np.dtype("<u8"))

    def checksum(self):
        """128-bit content checksum of every contig's list, computed on the GPU -> uint64 array [n, 2]"""
        out = np.zeros((max(self.n, 1), 2), dtype=np.uint64)
        self.ctx.check(lib().pgr_shmmrs_checksum(self.ctx.handle, self._h, out.ctypes.data))
        return out[:self.n]

    def offsets(self):
        """host copy of the n + 1 list offsets"""
        out = np.zeros(self.n + 1, dtype=np.uint64)
        lib().pgr_shmmrs_offsets(self._h, out.ctypes.data)
        return out

    def copy_into(self, device_ptr, capacity, rid_add=0, rids=None):
        """copy the MM128 list into caller-owned DEVICE memory (e.g. a torch tensor for the RCCL all-gather);
        rid_add / rids turn rank-local contig indices into global sequence ids"""
        if rids is not None:
            keep, rp = _u32_array(rids, self.n)
            self.ctx.check(lib().pgr_shmmrs_copy_to_device_rids(self.ctx.handle, self._h, C.c_void_p(device_ptr), capacity, rp))
        else:
            self.ctx.check(lib().pgr_shmmrs_copy_to_device(self.ctx.handle, self._h, C.c_void_p(device_ptr), capacity,
                                                           int(rid_add)))
        return self.count

    def frag_recs_into(self, device_ptr, capacity, sids=None, query_side=False):
        """write the shimmer-pair records into caller-owned DEVICE memory (e.g. a torch tensor)"""
        keep, sp = _u32_array(sids, self.n)
        n_out = C.c_uint64()
        self.ctx.check(lib().pgr_shmmrs_to_frag_recs_device(self.ctx.handle, self._h, sp, int(query_side),
                                                            C.c_void_p(device_ptr), capacity, C.byref(n_out)))
        return int(n_out.value)

    def close(self):
        if self._h:
            if getattr(getattr(self, "ctx", None), "alive", True):  # (_ffi.Context.alive: not on an object of a destroyed context)
                lib().pgr_shmmrs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pipe:
    """pgr_pipe: a software pipeline over resident batches -- submit() enqueues a whole pass (tiles on the context's stream, list
    stage + pair records on the back stream) and returns, collect() hands back the oldest job; two may be in flight.  The loop
    of load_index_from_reader (seq_db.rs:541-571) with the tail of batch i beside the tiles of batch i + 1."""

    def __init__(self, spec, ctx=None):
        self.ctx = ctx or default_context()
        self.spec = spec
        h = C.c_void_p()
        self.ctx.check(lib().pgr_pipe_create(self.ctx.handle, C.byref(spec), C.byref(h)))
        self._h = h
        self._keep = []  # batches (and indexes) of the jobs in flight, oldest first

    @property
    def in_flight(self):
        return int(lib().pgr_pipe_in_flight(self._h))

    def submit(self, batch, sids=None, index=None, rec_ptr=None, rec_capacity=0):
        keep, sp = _u32_array(sids, batch.n)
        self.ctx.check(lib().pgr_pipe_submit(self._h, batch._h, sp, index._h if index is not None else None,
                                             C.c_void_p(rec_ptr) if rec_ptr else None, int(rec_capacity)))
        self._keep.append((batch, index))

    def collect(self, want_shmmrs=True):
        """-> (Shmmrs or None, number of pair records written)"""
        h = C.c_void_p()
        npairs = C.c_uint64()
        n = self._keep[0][0].n if self._keep else 0
        before = self.in_flight
        rc = lib().pgr_pipe_collect(self._h, C.byref(h) if want_shmmrs else None, C.byref(npairs))
        if self._keep and self.in_flight < before:  # (a refused call -- nothing in flight, a query job first -- takes nothing out)
            self._keep.pop(0)
        self.ctx.check(rc)
        return (Shmmrs(self.ctx, h, n) if want_shmmrs else None), int(npairs.value)

    def submit_query(self, batch, index, penalty, max_count=128, max_count_query=128, max_count_target=128, max_aln_span=8,
                     max_gap=None, oriented=False):
        """pgr_pipe_submit_query: a batch of queries (resident) against a finalized index; collect_query() hands back the oldest"""
        self.ctx.check(lib().pgr_pipe_submit_query(self._h, batch._h, index._h, float(penalty), max_count, max_count_query,
                                                   max_count_target, max_aln_span, int(max_gap is not None), int(max_gap or 0),
                                                   int(bool(oriented))))
        self._keep.append((batch, index))

    def collect_query(self, raw=True):
        """-> the result of pgr_query_hps_resident on the oldest query job (numpy views of the result block; raw=False: only
        (n_targets, n_chains, n_hps), the block is released at once -- what a compiled host pays)"""
        res = _ffi.HpsResult()
        batch, index = self._keep[0] if self._keep else (None, None)
        before = self.in_flight
        rc = lib().pgr_pipe_collect_query(self._h, C.byref(res))
        if self._keep and self.in_flight < before:
            self._keep.pop(0)
        self.ctx.check(rc)
        if not raw:
            out = (int(res.n_targets), int(res.n_chains), int(res.n_hps))
            lib().pgr_hps_result_free(C.byref(res))
            return out
        return index._unpack_raw(res, batch.n)

    def close(self):
        if self._h:
            if getattr(getattr(self, "ctx", None), "alive", True):  # (_ffi.Context.alive: not on an object of a destroyed context)
                lib().pgr_pipe_destroy(self._h)
            self._h = C.c_void_p()
            self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def records_checksum(device_ptr, n, ctx=None):
    """order-independent 128-bit content checksum of n pair records at a DEVICE pointer -> (a, b)"""
    ctx = ctx or default_context()
    out = np.zeros(2, dtype=np.uint64)
    ctx.check(lib().pgr_records_checksum(ctx.handle, C.c_void_p(device_ptr), int(n), out.ctypes.data))
    return int(out[0]), int(out[1])


def sequence_to_shmmrs_batch(seqs, spec, rids=None, padding=False, ctx=None):
    """batched shmmrutils::sequence_to_shmmrs (shmmrutils.rs:657-669) -> list of MM128 arrays"""
    ctx = ctx or default_context()
    arrs, ptrs, lens, n = _ffi.seq_ptrs(seqs)
    keep, rp = _u32_array(rids, n)
    pm, po = C.c_void_p(), C.c_void_p()
    ctx.check(lib().pgr_shmmr_batch(ctx.handle, C.byref(spec), n, ptrs, lens, rp, int(padding), C.byref(pm),
                                    C.byref(po)))
    off = _ffi.take(po, n + 1, np.dtype("<u8"))
    mm = _ffi.take(pm, int(off[n]) if n else 0, MM128)
    return [mm[int(off[i]):int(off[i + 1])] for i in range(n)]


def sequence_to_shmmrs_batch_packed(packed, spec, rids=None, padding=False, ctx=None):
    """the same on host-packed input (PackedBases): pgr_shmmr_batch_packed"""
    ctx = ctx or default_context()
    n = packed.n
    la, pp, vp = packed._args()
    keep, rp = _u32_array(rids, n)
    pm, po = C.c_void_p(), C.c_void_p()
    ctx.check(lib().pgr_shmmr_batch_packed(ctx.handle, C.byref(spec), n, la, pp, vp, rp, int(padding), C.byref(pm), C.byref(po)))
    off = _ffi.take(po, n + 1, np.dtype("<u8"))
    mm = _ffi.take(pm, int(off[n]) if n else 0, MM128)
    return [mm[int(off[i]):int(off[i + 1])] for i in range(n)]


def time_shmmr_batch_packed(packed, spec, ctx=None):
    """seconds inside pgr_shmmr_batch_packed + the two pgr_free, and the number of shimmers"""
    import time
    ctx = ctx or default_context()
    n = packed.n
    la, pp, vp = packed._args()
    pm, po = C.c_void_p(), C.c_void_p()
    L, h = lib(), ctx.handle
    t0 = time.perf_counter()
    rc = L.pgr_shmmr_batch_packed(h, C.byref(spec), n, la, pp, vp, None, 0, C.byref(pm), C.byref(po))
    cnt = 0
    if rc == 0:
        cnt = int(C.cast(po, C.POINTER(C.c_uint64))[n])
        L.pgr_free(pm)
        L.pgr_free(po)
    dt = time.perf_counter() - t0
    ctx.check(rc)
    return dt, cnt


def time_shmmr_batch(seqs, spec, ctx=None):
    """seconds spent inside pgr_shmmr_batch + the two pgr_free (what a compiled host pays; pointer arrays built before the
    clock starts, nothing copied into numpy), and the number of shimmers"""
    import time
    ctx = ctx or default_context()
    arrs, ptrs, lens, n = _ffi.seq_ptrs(seqs)
    pm, po = C.c_void_p(), C.c_void_p()
    L, h = lib(), ctx.handle
    t0 = time.perf_counter()
    rc = L.pgr_shmmr_batch(h, C.byref(spec), n, ptrs, lens, None, 0, C.byref(pm), C.byref(po))
    cnt = 0
    if rc == 0:
        cnt = int(C.cast(po, C.POINTER(C.c_uint64))[n])
        L.pgr_free(pm)
        L.pgr_free(po)
    dt = time.perf_counter() - t0
    ctx.check(rc)
    return dt, cnt


def sequence_to_shmmrs(rid, seq, spec, padding=False, ctx=None):
    return sequence_to_shmmrs_batch([seq], spec, rids=[rid], padding=padding, ctx=ctx)[0]


def frag_recs_batch(seqs, spec, sids=None, query_side=False, ctx=None):
    """sequence_to_shmmrs + pair_shmmrs/seq_to_index records (seq_db.rs:360-418) -> list of FRAG_REC arrays"""
    ctx = ctx or default_context()
    arrs, ptrs, lens, n = _ffi.seq_ptrs(seqs)
    keep, sp = _u32_array(sids, n)
    pr, po = C.c_void_p(), C.c_void_p()
    ctx.check(lib().pgr_frag_recs_batch(ctx.handle, C.byref(spec), n, ptrs, lens, sp, int(query_side), C.byref(pr),
                                        C.byref(po)))
    off = _ffi.take(po, n + 1, np.dtype("<u8"))
    recs = _ffi.take(pr, int(off[n]) if n else 0, FRAG_REC)
    return [recs[int(off[i]):int(off[i + 1])] for i in range(n)]


class _HpsOwner:
    """owns one pgr_hps_result: releases the library's block when the last view of it is gone"""

    def __init__(self, res):
        self._res = res

    def __del__(self):
        try:
            lib().pgr_hps_result_free(C.byref(self._res))
        except Exception:
            pass


class Index:
    """ShmmrToFrags on the GPU (pgr_index): sorted CSR of fragment signatures + the query entry point."""

    def __init__(self, spec, ctx=None, _handle=None):
        self.ctx = ctx or default_context()
        self.spec = spec
        self._h = C.c_void_p()
        if _handle is not None:  # an index the library built (pgr_exchange_allgather_index, pgr_index_load_mdb)
            self._h = _handle
        else:
            self.ctx.check(lib().pgr_index_create(self.ctx.handle, C.byref(spec), C.byref(self._h)))

    def records_checksum(self):
        """order-independent 128-bit content checksum of the index's records -> (a, b)"""
        out = np.zeros(2, dtype=np.uint64)
        self.ctx.check(lib().pgr_index_records_checksum(self.ctx.handle, self._h, out.ctypes.data))
        return int(out[0]), int(out[1])

    def key_range(self):
        """(first hash of the first record, first hash of the last record) of a finalized index"""
        lo, hi = C.c_uint64(), C.c_uint64()
        self.ctx.check(lib().pgr_index_key_range(self.ctx.handle, self._h, C.byref(lo), C.byref(hi)))
        return int(lo.value), int(hi.value)

    @property
    def device_records(self):
        return lib().pgr_index_device_records(self._h)

    def reserve(self, n_records):
        self.ctx.check(lib().pgr_index_reserve(self.ctx.handle, self._h, int(n_records)))

    def add_resident(self, batch, sids=None):
        keep, sp = _u32_array(sids, batch.n)
        self.ctx.check(lib().pgr_index_add_resident(self.ctx.handle, self._h, batch._h, sp))

    def add_seqs(self, seqs, sids=None):
        arrs, ptrs, lens, n = _ffi.seq_ptrs(seqs)
        keep, sp = _u32_array(sids, n)
        self.ctx.check(lib().pgr_index_add_batch(self.ctx.handle, self._h, n, ptrs, lens, sp))

    def add_packed(self, packed, sids=None):
        la, pp, vp = packed._args()
        keep, sp = _u32_array(sids, packed.n)
        self.ctx.check(lib().pgr_index_add_packed(self.ctx.handle, self._h, packed.n, la, pp, vp, sp))

    def add_records(self, recs=None, device_ptr=None, n=None):
        """merge pair records computed elsewhere: a host FRAG_REC array, or n records at a DEVICE pointer
        (e.g. the all-gathered torch tensor of the multi-GPU exchange)"""
        if recs is not None:
            a = np.ascontiguousarray(recs, dtype=FRAG_REC)
            self.ctx.check(lib().pgr_index_add_records(self.ctx.handle, self._h, a.ctypes.data, a.size, 0))
        else:
            self.ctx.check(lib().pgr_index_add_records(self.ctx.handle, self._h, C.c_void_p(device_ptr), int(n), 1))

    def add_shmmrs(self, mm=None, device_ptr=None, n=None):
        """merge shimmer lists (MM128, rid = sequence id, one sequence's shimmers contiguous): a host array or
        n elements at a DEVICE pointer (the all-gathered tensor); pair records are derived on the GPU"""
        if mm is not None:
            a = np.ascontiguousarray(mm, dtype=MM128)
            self.ctx.check(lib().pgr_index_add_shmmrs(self.ctx.handle, self._h, a.ctypes.data, a.size, 0))
        else:
            self.ctx.check(lib().pgr_index_add_shmmrs(self.ctx.handle, self._h, C.c_void_p(device_ptr), int(n), 1))

    def finalize(self):
        self.ctx.check(lib().pgr_index_finalize(self.ctx.handle, self._h))

    @property
    def n_keys(self):
        return int(lib().pgr_index_n_keys(self._h))

    @property
    def n_records(self):
        return int(lib().pgr_index_n_records(self._h))

    def download(self):
        p, n = C.c_void_p(), C.c_uint64()
        self.ctx.check(lib().pgr_index_download(self.ctx.handle, self._h, C.byref(p), C.byref(n)))
        return _ffi.take(p, int(n.value), FRAG_REC)

    def query_hps_raw(self, seqs, penalty, max_count=128, max_count_query=128, max_count_target=128, max_aln_span=8,
                      max_gap=None, oriented=False):
        """pgr_query_hps_batch -> flat numpy arrays (q_off, t_sid, t_off, c_score, c_off, hps)"""
        arrs, ptrs, lens, n = _ffi.seq_ptrs(seqs)
        res = _ffi.HpsResult()
        self.ctx.check(lib().pgr_query_hps_batch(self.ctx.handle, self._h, n, ptrs, lens, float(penalty), max_count,
                                                 max_count_query, max_count_target, max_aln_span,
                                                 int(max_gap is not None), int(max_gap or 0), int(bool(oriented)),
                                                 C.byref(res)))
        return self._unpack_raw(res, n)

    def query_hps_resident_raw(self, batch, penalty, max_count=128, max_count_query=128, max_count_target=128,
                               max_aln_span=8, max_gap=None, oriented=False, ctx=None):
        """pgr_query_hps_resident: the queries are a Batch already on the GPU.  ctx: the context the call runs on (the batch's;
        default: the index's) -- a finalized index may be queried from several contexts, one per host thread"""
        res = _ffi.HpsResult()
        ctx = ctx or self.ctx
        ctx.check(lib().pgr_query_hps_resident(ctx.handle, self._h, batch._h, float(penalty), max_count,
                                               max_count_query, max_count_target, max_aln_span,
                                               int(max_gap is not None), int(max_gap or 0), int(bool(oriented)),
                                               C.byref(res)))
        return self._unpack_raw(res, batch.n)

    def time_query_resident(self, batch, penalty, max_count=128, max_count_query=128, max_count_target=128,
                            max_aln_span=8, max_gap=None, oriented=False):
        """seconds spent inside pgr_query_hps_resident + pgr_hps_result_free (what a compiled host pays: the result stays in
        the library's block, nothing is unpacked into numpy arrays), and the number of hit pairs of the result"""
        import time
        res = _ffi.HpsResult()
        L, h, ih, bh = lib(), self.ctx.handle, self._h, batch._h
        a = (float(penalty), max_count, max_count_query, max_count_target, max_aln_span, int(max_gap is not None),
             int(max_gap or 0), int(bool(oriented)))
        t0 = time.perf_counter()
        rc = L.pgr_query_hps_resident(h, ih, bh, *a, C.byref(res))
        n = int(res.n_hps) if rc == 0 else 0
        if rc == 0:
            L.pgr_hps_result_free(C.byref(res))
        dt = time.perf_counter() - t0
        self.ctx.check(rc)
        return dt, n

    def time_query_host(self, seqs, penalty, max_count=128, max_count_query=128, max_count_target=128, max_aln_span=8,
                        max_gap=None, oriented=False):
        """the same for pgr_query_hps_batch (host ASCII in); the pointer arrays are built before the clock starts"""
        import time
        arrs, ptrs, lens, n = _ffi.seq_ptrs(seqs)
        res = _ffi.HpsResult()
        L, h, ih = lib(), self.ctx.handle, self._h
        a = (float(penalty), max_count, max_count_query, max_count_target, max_aln_span, int(max_gap is not None),
             int(max_gap or 0), int(bool(oriented)))
        t0 = time.perf_counter()
        rc = L.pgr_query_hps_batch(h, ih, n, ptrs, lens, *a, C.byref(res))
        nh = int(res.n_hps) if rc == 0 else 0
        if rc == 0:
            L.pgr_hps_result_free(C.byref(res))
        dt = time.perf_counter() - t0
        self.ctx.check(rc)
        return dt, nh

    @staticmethod
    def _unpack_raw(res, n):
        """numpy VIEWS of the library's result block (no copies: the hit pairs of a 10 000-query batch are 7 MB); the block is
        released (pgr_hps_result_free) when the last of the arrays is garbage collected"""
        nt, nc, nh = int(res.n_targets), int(res.n_chains), int(res.n_hps)
        owner = _HpsOwner(res)

        def view(ptr, count, dtype):
            dtype = np.dtype(dtype)
            if count == 0 or not ptr:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (count * dtype.itemsize)).from_address(C.cast(ptr, C.c_void_p).value)
            buf._owner = owner  # the array's base keeps the ctypes buffer alive, the buffer keeps the C block alive
            a = np.frombuffer(buf, dtype=dtype)
            a.flags.writeable = False
            return a
        return {
            "q_off": view(res.q_off, n + 1, "<u8"), "t_sid": view(res.t_sid, nt, "<u4"), "t_off": view(res.t_off, nt + 1, "<u8"),
            "c_score": view(res.c_score, nc, "<f4"), "c_off": view(res.c_off, nc + 1, "<u8"),
            "hps": view(res.hps, nh, _ffi.HITPAIR), "n_nonterminating": int(res.n_nonterminating),
        }

    def close(self):
        if self._h:
            if getattr(getattr(self, "ctx", None), "alive", True):  # (_ffi.Context.alive: not on an object of a destroyed context)
                lib().pgr_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
