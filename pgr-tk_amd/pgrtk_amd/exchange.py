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
        if unique_id is None:
            idb = np.zeros(128, dtype=np.uint8)
            if rank == 0:
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

    def close(self):
        from ._ffi import lib
        if self._h:
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
