"""CPU, world_size 2, gloo: the multi-GPU exchange step (pgrtk_amd/exchange.py) -- contig sharding and
the all-gather of per-rank shimmer-pair record buffers -- is correct by construction.  On the GPU box
the same code runs over RCCL (backend "nccl")."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, counts, q):
    sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
    from pgrtk_amd import exchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = counts[rank]
        # record j of rank r = [r, j, r*1000+j, 7, 9]
        local = torch.zeros((n, exchange.REC_WORDS), dtype=torch.int64)
        if n:
            local[:, 0] = rank
            local[:, 1] = torch.arange(n)
            local[:, 2] = rank * 1000 + torch.arange(n)
            local[:, 3] = 7
            local[:, 4] = 9
        out, cnts = exchange.allgather_records(local)
        pend = exchange.PendingAllgather(local)  # the overlapped form bench.py uses must agree
        out2, cnts2 = pend.wait()
        assert cnts2 == cnts and out2.shape == out.shape and bool((out2 == out).all())
        # the MM128 form (2 words per row) with caller-provided gather buffers, as bench.py exchanges every step:
        # a big enough buffer is used in place, a too small one is replaced
        mm = local[:, :exchange.MM_WORDS].contiguous()
        big = torch.full((world * (max(counts) + 3), exchange.MM_WORDS), -1, dtype=torch.int64)
        g3, c3 = exchange.PendingAllgather(mm, out=big).wait()
        assert c3 == cnts and bool((g3 == out[:, :exchange.MM_WORDS]).all())
        parts, c5 = exchange.PendingAllgather(mm).wait(concat=False)  # the copy-free form of the step loop
        assert c5 == cnts and [int(p.shape[0]) for p in parts] == (cnts if max(counts) else [])
        if max(counts):
            assert bool((torch.cat(parts, dim=0) == g3).all())
        if max(counts):
            assert g3.data_ptr() == big.data_ptr() or len(set(counts)) > 1  # gathered in place (ragged: re-packed)
            small = torch.empty((1, exchange.MM_WORDS), dtype=torch.int64)
            g4, _ = exchange.PendingAllgather(mm, out=small).wait()
            assert bool((g4 == g3).all())
        q.put((rank, out.numpy().copy(), cnts))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("counts", [[5, 5], [3, 8], [0, 4], [0, 0]])
def test_allgather_records_gloo(counts):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, counts, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, out, cnts = q.get(timeout=120)
        res[r] = (out, cnts)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = []
    for r in range(world):
        for j in range(counts[r]):
            exp.append([r, j, r * 1000 + j, 7, 9])
    exp = np.array(exp, dtype=np.int64).reshape(-1, 5)
    for r in range(world):
        out, cnts = res[r]
        assert cnts == counts
        assert out.shape == exp.shape and (out == exp).all()  # rank-major order, identical on every rank


def test_shard_contigs_balanced():
    sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
    from pgrtk_amd import exchange
    rng = np.random.default_rng(0)
    lens = [int(v) for v in rng.integers(1000, 10_000_000, 300)]
    for world in (1, 2, 4, 8):
        shards = exchange.shard_contigs(lens, world)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(len(lens)))  # a partition
        loads = [sum(lens[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(lens)  # greedy LPT bound
        assert all(s == sorted(s) for s in shards)  # file order inside a rank
