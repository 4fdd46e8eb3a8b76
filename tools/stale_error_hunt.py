#!/usr/bin/env python3
"""Which call leaves an error in the HIP runtime's per-thread "last error"?  (include/pgr_hip.h: pgr_debug_take_hip_error.)  Runs the
in-process part of tests/test_gpu_90_dist_plumbing.py::test_key_range_sharded_index_two_ranks_equals_single_process one call at a
time and asks after each.   python tools/stale_error_hunt.py"""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import pgrtk_amd as P  # noqa: E402
from pgrtk_amd import _ffi  # noqa: E402
import exchange_worker as W  # noqa: E402


def ask(what):
    e = int(_ffi.lib().pgr_debug_take_hip_error())
    print("%-60s -> %d" % (what, e), flush=True)


ctx = P.default_context(0)
ask("default_context")
b = P.Batch.synthetic(W.LENS, seed=W.SEED, ctx=ctx)
ask("Batch.synthetic")
ix = P.Index(P.make_spec(), ctx=ctx)
ask("Index()")
ix.add_resident(b)
ask("add_resident")
ix.finalize()
ask("finalize")
want = ix.download()
ask("download (%d records)" % len(want))
nk = ix.n_keys
ask("n_keys")
sh = b.shmmrs(P.make_spec())
ask("shmmrs")
mm, off = sh.download()
ask("shmmrs.download")
del sh
gc.collect()
ask("del shmmrs")
del ix
gc.collect()
ask("del index")
del b
gc.collect()
ask("del batch")
x = torch.zeros(1000, device="cuda:0")
ask("torch.zeros on the device")
del x
ask("del tensor")
