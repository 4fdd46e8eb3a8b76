"""Multi-GPU exchange step: contigs are sharded across ranks (one process per GPU); every rank
computes the shimmers of its own contigs, then the per-rank buffers are all-gathered (RCCL over
xGMI via torch.distributed, backend "nccl"; "gloo" on CPU in the tests) so that every rank -- in
particular the one that owns the host-side ShmmrToFrags map -- holds the full set in global
(rank, sid, position) order.  What travels is the final MM128 list (16 B per shimmer, rid =
global sequence id); the shimmer-pair records (40 B per pair) are adjacent shimmers and are
derived on the receiving GPU (pgr_index_add_shmmrs), 2.5x less traffic than shipping records.
Rows are int64 tensors [n, words] (MM_WORDS or REC_WORDS).

Two transports:
  * AbiExchange   -- libpgrhip's own pgr_exchange_* entry points (RCCL loaded and driven by the C library, collective
                     on its own HIP stream): what a Rust / C++ host calls; Python only hands the 128-byte unique id from
                     rank 0 to the other ranks (any host-side channel does: here the torch process group, in
                     host/pgr_mdb.cpp a pipe);
  * PendingAllgather / allgather_records -- torch.distributed (backend "nccl" = RCCL, "gloo" for the CPU tests).
torch is plumbing here (device memory, process group); no compute.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

REC_WORDS = 5  # one pgr_frag_rec = 40 bytes = 5 x int64
MM_WORDS = 2   # one MM128 = 16 bytes = 2 x int64 (what the ranks exchange: pairs are adjacent shimmers)


def shard_contigs(lens, world_size):
    """greedy length-balanced assignment of contigs to ranks.
    Returns a list of index lists, one per rank; ids inside a rank stay in file order."""
    order = sorted(range(len(lens)), key=lambda i: (-int(lens[i]), i))
    load = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda q: (load[q], q))
        shards[r].append(i)
        load[r] += int(lens[i])
    for s in shards:
        s.sort()
    return shards


class _AbiPending:
    def __init__(self, xch, out, cap):
        self.xch, self.out, self.cap = xch, out, cap

    def wait(self, concat=True):
        """-> (gathered, counts).  concat=True: one [sum n, 2] tensor in rank order (a copy when world > 1);
        concat=False: the list of per-rank views into the gather buffer -- no copy, what a step loop wants"""
        from ._ffi import lib
        counts = np.zeros(self.xch.world, dtype=np.uint64)
        self.xch.ctx.check(lib().pgr_exchange_wait(self.xch._h, counts.ctypes.data_as(C.POINTER(C.c_uint64))))
        counts = [int(c) for c in counts]
        parts = [self.out[r * self.cap: r * self.cap + c] for r, c in enumerate(counts)]
        if not concat:
            return parts, counts
        return (parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)), counts


class AbiExchange:
    """pgr_exchange_* through ctypes.  group: a torch.distributed module / process group used ONLY to broadcast the
    ncclUniqueId bytes (host side); the collective itself never touches torch."""

    def __init__(self, ctx, rank, world, dist_mod=None, unique_id=None):
        from ._ffi import lib
        self.ctx, self.rank, self.world = ctx, rank, world
        # one rank: the library copies on its own stream and does not load RCCL (context option exchange_rccl_world1 = 1 asks
        # for a real communicator all the same), so there is no unique id to make
        self.uses_rccl = world > 1 or bool(ctx.get_option("exchange_rccl_world1"))
        if unique_id is None:
            idb = np.zeros(128, dtype=np.uint8)
            if rank == 0 and self.uses_rccl:
                ctx.check(lib().pgr_exchange_unique_id(ctx.handle, idb.ctypes.data))
            if world > 1:
                d = dist_mod or dist
                t = torch.from_numpy(idb)
                if d.get_backend() == "nccl":
                    t = t.cuda()
                d.broadcast(t, src=0)
                idb = t.cpu().numpy()
            unique_id = idb.tobytes()
        self.unique_id = unique_id
        self._h = C.c_void_p()
        idarr = np.frombuffer(unique_id, dtype=np.uint8).copy()
        ctx.check(lib().pgr_exchange_create(ctx.handle, idarr.ctypes.data, rank, world, C.byref(self._h)))

    def allgather_async(self, local, n_local, out, cap_per_rank):
        """local: int64 tensor [>= cap_per_rank, 2] on the GPU holding n_local valid rows; out: [world * cap_per_rank, 2]"""
        from ._ffi import lib
        assert local.is_cuda and out.is_cuda and local.shape[0] >= cap_per_rank and out.shape[0] >= self.world * cap_per_rank
        self.ctx.check(lib().pgr_exchange_allgather_shmmrs_start(self._h, C.c_void_p(local.data_ptr()), int(n_local),
                                                                 C.c_void_p(out.data_ptr()), int(cap_per_rank)))
        return _AbiPending(self, out, cap_per_rank)

    def shard_records(self, recs_ptr, n, index, reuse_splitters=False):
        """pgr_exchange_shard_records: this rank's pair records (DEVICE pointer) travel to the ranks that own their key
        ranges, straight into `index` (finish it with index.finalize()).  -> (records received, splitters)"""
        from ._ffi import lib
        spl = np.zeros(max(self.world - 1, 1), dtype=np.uint64)
        got = C.c_uint64()
        self.ctx.check(lib().pgr_exchange_shard_records(self._h, C.c_void_p(recs_ptr), int(n), index._h, int(reuse_splitters),
                                                        spl.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(got)))
        return int(got.value), [int(v) for v in spl[:self.world - 1]]

    def allgather_index(self, shard):
        """pgr_exchange_allgather_index: the replicated (finalized) index from every rank's finalized shard"""
        from ._ffi import lib
        from .engine import Index
        h = C.c_void_p()
        self.ctx.check(lib().pgr_exchange_allgather_index(self._h, shard._h, C.byref(h)))
        return Index(shard.spec, ctx=self.ctx, _handle=h)

    def close(self):
        from ._ffi import lib
        if self._h:
            if getattr(getattr(self, "ctx", None), "alive", True):  # (_ffi.Context.alive: not on an object of a destroyed context)
                lib().pgr_exchange_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PendingAllgather:
    """an all-gather of one rank-local record tensor in flight (collective on the process group's own
    stream, so the next batch's kernels overlap with it); wait() returns (gathered, counts)."""

    def __init__(self, local, group=None, out=None):
        """out: optional preallocated [>= world * n_max, words] tensor to gather into (callers that exchange every
        step keep two and alternate, so the steady state allocates nothing)"""
        world = dist.get_world_size(group)
        assert local.dtype == torch.int64 and local.dim() == 2
        words = local.shape[1]
        self.local = local
        n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        counts = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(counts, n_local, group=group)
        self.counts = [int(v) for v in counts.cpu()]
        self.n_max = max(self.counts) if self.counts else 0
        self.out = None
        self.work = None
        if self.n_max:
            if local.shape[0] == self.n_max:
                padded = local.contiguous()
            else:
                padded = local.new_zeros((self.n_max, words))
                padded[: local.shape[0]] = local
            self.padded = padded
            if out is not None and out.shape[0] >= world * self.n_max and out.shape[1] == words and out.dtype == local.dtype \
                    and out.device == local.device:
                self.out = out[: world * self.n_max]
            else:
                self.out = local.new_empty((world * self.n_max, words))
            self.work = dist.all_gather_into_tensor(self.out, padded, group=group, async_op=True)

    def wait(self, concat=True):
        if not concat:  # per-rank views (no copy when the counts are equal; the padded buffer is sliced otherwise)
            if self.work is None:
                return [], self.counts
            self.work.wait()
            if self.out.is_cuda:
                torch.cuda.current_stream(self.out.device).synchronize()
            return [self.out[r * self.n_max: r * self.n_max + c] for r, c in enumerate(self.counts)], self.counts
        if self.work is None:
            return self.local.new_zeros((0, self.local.shape[1])), self.counts
        self.work.wait()
        if self.out.is_cuda:
            # the record buffers are rewritten by kernels on libpgrhip's own stream, which torch does not
            # order against: make the completion visible to the host before the caller reuses them
            torch.cuda.current_stream(self.out.device).synchronize()
        if all(c == self.n_max for c in self.counts):
            return self.out, self.counts
        parts = [self.out[r * self.n_max: r * self.n_max + c] for r, c in enumerate(self.counts)]
        return torch.cat(parts, dim=0), self.counts


def allgather_records(local, group=None):
    """local: int64 tensor [n_local, 5] (40-byte records) on this rank's device.
    Returns (gathered [sum n, 5] in rank order, counts list).  Two collectives: the counts
    (8 bytes per rank) and one padded all-gather of the payload."""
    world = dist.get_world_size(group)
    assert local.dtype == torch.int64 and local.dim() == 2
    words = local.shape[1]
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts, n_local, group=group)
    counts_h = [int(v) for v in counts.cpu()]
    n_max = max(counts_h) if counts_h else 0
    if n_max == 0:
        return local.new_zeros((0, words)), counts_h
    if local.shape[0] == n_max:
        padded = local.contiguous()
    else:
        padded = local.new_zeros((n_max, words))
        padded[: local.shape[0]] = local
    out = local.new_empty((world * n_max, words))
    dist.all_gather_into_tensor(out, padded, group=group)
    if all(c == n_max for c in counts_h):
        return out, counts_h
    parts = [out[r * n_max: r * n_max + counts_h[r]] for r in range(world)]
    return torch.cat(parts, dim=0), counts_h


# ---------------------------------------------------------------------------------------------------------------------
# key-range sharded index over torch.distributed (the transport of the CPU / one-GPU tests; on a multi-GPU node
# AbiExchange.shard_records does the same through RCCL inside the library).  The device work -- sampling, stable
# partition, checksums -- is libpgrhip's (pgr_shard_*); torch only moves bytes.
SHARD_SAMPLES = 4096


def shard_splitters(samples_per_rank, world):
    """pgr_shard_splitters on the pooled samples (a list of uint64 arrays, one per rank) -> world - 1 splitters"""
    from ._ffi import lib
    pool = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.uint64) for a in samples_per_rank])
                                if samples_per_rank else np.zeros(0, dtype=np.uint64))
    spl = np.zeros(max(world - 1, 1), dtype=np.uint64)
    rc = lib().pgr_shard_splitters(C.c_void_p(pool.ctypes.data if pool.size else 0), int(pool.size), int(world),
                                   C.c_void_p(spl.ctypes.data))
    if rc != 0:
        raise ValueError("pgr_shard_splitters: bad arguments")
    return spl[:world - 1]


def shard_records_torch(ctx, recs_ptr, n, index, group=None):
    """the steps of pgr_exchange_shard_records with torch.distributed as the transport.  recs_ptr: DEVICE pointer to this
    rank's n pair records.  -> (records received, splitters)"""
    from ._ffi import lib
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device())
    # 1. pooled sample -> splitters
    smp = np.zeros(1 + SHARD_SAMPLES, dtype=np.uint64)
    n_s = C.c_uint32()
    ctx.check(lib().pgr_shard_sample_keys(ctx.handle, C.c_void_p(recs_ptr), int(n), SHARD_SAMPLES,
                                          C.c_void_p(smp.ctypes.data + 8), C.byref(n_s)))
    smp[0] = n_s.value
    t = torch.from_numpy(smp.view(np.int64))
    allt = torch.empty(world * smp.size, dtype=torch.int64)
    if on_gpu:
        t, allt = t.to(dev), allt.to(dev)
    dist.all_gather_into_tensor(allt, t, group=group)
    alls = allt.cpu().numpy().view(np.uint64).reshape(world, smp.size)
    spl = shard_splitters([alls[r, 1:1 + int(alls[r, 0])] for r in range(world)], world)
    # 2. stable partition by destination rank (device)
    part = torch.empty((max(int(n), 1), REC_WORDS), dtype=torch.int64, device=dev)
    counts = np.zeros(world, dtype=np.uint64)
    spl_arg = np.ascontiguousarray(spl) if world > 1 else np.zeros(1, dtype=np.uint64)
    ctx.check(lib().pgr_shard_partition(ctx.handle, C.c_void_p(recs_ptr), int(n), C.c_void_p(spl_arg.ctypes.data), world,
                                        C.c_void_p(part.data_ptr()), C.c_void_p(counts.ctypes.data)))
    # 3. counts, 4. payload
    send_cnt = torch.from_numpy(counts.astype(np.int64))
    recv_cnt = torch.empty(world, dtype=torch.int64)
    if on_gpu:
        send_cnt, recv_cnt = send_cnt.to(dev), recv_cnt.to(dev)
    dist.all_to_all_single(recv_cnt, send_cnt, group=group)
    rc_l = [int(v) for v in recv_cnt.cpu()]
    sc_l = [int(v) for v in counts]
    src = part[:int(n)] if on_gpu else part[:int(n)].cpu()
    out = torch.empty((sum(rc_l), REC_WORDS), dtype=torch.int64, device=src.device)
    dist.all_to_all_single(out, src, output_split_sizes=rc_l, input_split_sizes=sc_l, group=group)
    if on_gpu:
        torch.cuda.current_stream().synchronize()
    got = out if on_gpu else out.to(dev)
    if got.shape[0]:
        index.add_records(device_ptr=got.data_ptr(), n=int(got.shape[0]))
    return int(got.shape[0]), [int(v) for v in spl]


def allgather_index_torch(shard, group=None):
    """the replicated index from the finalized shards over torch.distributed (counterpart of pgr_exchange_allgather_index)"""
    from .engine import Index
    world = dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device())
    n = shard.n_records
    # (the shard's sorted records go through host memory here: this is the test transport; a multi-GPU node uses
    # AbiExchange.allgather_index, device to device)
    mine = torch.from_numpy(np.ascontiguousarray(shard.download()).view(np.int64).reshape(n, REC_WORDS).copy()).to(dev) if n \
        else torch.empty((0, REC_WORDS), dtype=torch.int64, device=dev)
    gathered, counts = PendingAllgather(mine[:n] if on_gpu else mine[:n].cpu(), group=group).wait()
    g = gathered if gathered.is_cuda else gathered.to(dev)
    full = Index(shard.spec, ctx=shard.ctx)
    if int(g.shape[0]):
        g = g.contiguous()
        full.add_records(device_ptr=g.data_ptr(), n=int(g.shape[0]))
    full.finalize()
    return full
