"""One rank of the sharded index build (run by tests/test_gpu_90_dist_plumbing.py, two processes on one GPU box).

    python exchange_worker.py <transport> <rank> <world> <port> <outdir>

transport "abi":   libpgrhip's pgr_exchange_* (RCCL driven by the C library); the 128-byte unique id travels through a
                   file in <outdir> (what host/pgr_mdb.cpp does with a pipe) -- no torch.distributed at all;
transport "gloo":  torch.distributed on CPU tensors (fallback when RCCL refuses two ranks on one device).
Every rank takes its shard of ONE ragged synthetic contig set from exchange.shard_contigs, computes its shimmers with
global sequence ids, all-gathers the lists, builds the replicated index from the gathered lists and writes its sorted
records to <outdir>/records_<rank>.npy.
transport "shard-abi" / "shard-gloo": the key-range sharded build instead -- every rank derives its pair records, the
records travel to the rank that owns their range of first hashes (pgr_exchange_shard_records over RCCL, or the same
steps with torch.distributed as the transport), every rank sorts ITS range and writes it to records_<rank>.npy with a
meta_<rank>.json (counts, checksums, key range, splitters); then the replicated index is rebuilt from the shards
(replicated_<rank>.npy).
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))

LENS = [1_500_000, 40_000, 2_200_000, 700_000, 120, 0, 900_000, 3_100_000, 64_000, 1_000_000, 333_333, 2_000_000]
SEED = 11


def main():
    transport, rank, world, port, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import numpy as np
    import torch
    import pgrtk_amd as P
    from pgrtk_amd import exchange
    torch.cuda.set_device(0)
    ctx = P.Context(0)
    spec = P.make_spec()
    ids = exchange.shard_contigs(LENS, world)[rank]
    batch = P.Batch.synthetic([LENS[i] for i in ids], seed=SEED, ctx=ctx, contig_ids=ids)
    sh = batch.shmmrs(spec)
    if transport.startswith("shard-"):
        return shard_mode(transport[6:], rank, world, port, outdir, ctx, spec, sh, ids)
    cap = 64 * 1024  # agreed capacity per rank (>= every rank's count)
    local = torch.zeros((cap, 2), dtype=torch.int64, device="cuda:0")
    n = sh.copy_into(local.data_ptr(), cap, rids=ids)
    if transport == "abi":
        idfile = os.path.join(outdir, "unique_id.bin")
        if rank == 0:
            import ctypes as C
            from pgrtk_amd._ffi import lib
            idb = np.zeros(128, dtype=np.uint8)
            ctx.check(lib().pgr_exchange_unique_id(ctx.handle, idb.ctypes.data))
            with open(idfile + ".tmp", "wb") as f:
                f.write(idb.tobytes())
            os.rename(idfile + ".tmp", idfile)
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 60:
                raise RuntimeError("no unique id from rank 0")
            time.sleep(0.01)
        uid = open(idfile, "rb").read()
        xch = exchange.AbiExchange(ctx, rank, world, unique_id=uid)
        out = torch.zeros((world * cap, 2), dtype=torch.int64, device="cuda:0")
        gathered, counts = xch.allgather_async(local, n, out, cap).wait()
        xch.close()
    else:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        gathered, counts = exchange.PendingAllgather(local[:n].cpu()).wait()
        gathered = gathered.cuda()
        dist.destroy_process_group()
    assert counts[rank] == n
    ix = P.Index(spec, ctx=ctx)
    g = gathered.contiguous()
    ix.add_shmmrs(device_ptr=g.data_ptr(), n=int(g.shape[0]))
    ix.finalize()
    np.save(os.path.join(outdir, "records_%d.npy" % rank), ix.download())
    print("rank %d: %d of %d shimmers local, %d records" % (rank, n, int(g.shape[0]), ix.n_records))


def _unique_id(ctx, rank, outdir):
    import numpy as np
    idfile = os.path.join(outdir, "unique_id.bin")
    if rank == 0:
        from pgrtk_amd._ffi import lib
        idb = np.zeros(128, dtype=np.uint8)
        ctx.check(lib().pgr_exchange_unique_id(ctx.handle, idb.ctypes.data))
        with open(idfile + ".tmp", "wb") as f:
            f.write(idb.tobytes())
        os.rename(idfile + ".tmp", idfile)
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 60:
            raise RuntimeError("no unique id from rank 0")
        time.sleep(0.01)
    return open(idfile, "rb").read()


def shard_mode(transport, rank, world, port, outdir, ctx, spec, sh, ids):
    import json
    import numpy as np
    import torch
    import pgrtk_amd as P
    from pgrtk_amd import exchange
    n_pairs = sh.n_pairs
    recs = torch.zeros((max(n_pairs, 1), exchange.REC_WORDS), dtype=torch.int64, device="cuda:0")
    n = sh.frag_recs_into(recs.data_ptr(), recs.shape[0], sids=ids)
    assert n == n_pairs
    sent = P.records_checksum(recs.data_ptr(), n, ctx=ctx)
    ix = P.Index(spec, ctx=ctx)
    if transport == "abi":
        xch = exchange.AbiExchange(ctx, rank, world, unique_id=_unique_id(ctx, rank, outdir))
        got, spl = xch.shard_records(recs.data_ptr(), n, ix)
        ix.finalize()
        full = xch.allgather_index(ix)
        xch.close()
    else:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        got, spl = exchange.shard_records_torch(ctx, recs.data_ptr(), n, ix)
        ix.finalize()
        full = exchange.allgather_index_torch(ix)
        dist.destroy_process_group()
    assert got == ix.n_records
    np.save(os.path.join(outdir, "records_%d.npy" % rank), ix.download())
    np.save(os.path.join(outdir, "replicated_%d.npy" % rank), full.download())
    with open(os.path.join(outdir, "meta_%d.json" % rank), "w") as f:
        json.dump({"n_sent": n, "n_shard": ix.n_records, "n_keys": ix.n_keys, "sent": list(sent), "shard": list(ix.records_checksum()),
                   "key_range": list(ix.key_range()), "splitters": spl, "full_keys": full.n_keys}, f)
    print("rank %d: %d records sent, %d in its key range" % (rank, n, ix.n_records))


if __name__ == "__main__":
    main()
