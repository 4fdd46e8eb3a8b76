"""One rank of the sharded index build (run by tests/test_gpu_round2.py, two processes on one GPU box).

    python exchange_worker.py <transport> <rank> <world> <port> <outdir>

transport "abi":   libpgrhip's pgr_exchange_* (RCCL driven by the C library); the 128-byte unique id travels through a
                   file in <outdir> (what host/pgr_mdb.cpp does with a pipe) -- no torch.distributed at all;
transport "gloo":  torch.distributed on CPU tensors (fallback when RCCL refuses two ranks on one device).
Every rank takes its shard of ONE ragged synthetic contig set from exchange.shard_contigs, computes its shimmers with
global sequence ids, all-gathers the lists, builds the replicated index from the gathered lists and writes its sorted
records to <outdir>/records_<rank>.npy.
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


if __name__ == "__main__":
    main()
