#!/usr/bin/env python3
"""BASELINE.json configs[4] (HPRC-scale: 94 x 3 Gbp as 94 x 300 contigs of 10 Mbp, seed 5, sharded by contig over 8
GPUs): the slice ONE GPU owns -- 3525 contigs = 35.25 Gbp -- streamed through the GPU in batches of 1000 contigs,
every batch appended to the GPU-resident index, one finalize (sort) at the end.  Prints stage times as JSON."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))
import pgrtk_amd as P  # noqa: E402


def main():
    n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 3525   # 94 * 300 / 8
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    per_batch = 1000
    ctx = P.default_context(0)
    spec = P.make_spec(80, 56, 4, 64)
    ix = P.Index(spec, ctx=ctx)
    c0 = rank * n_contigs
    t_synth = t_add = 0.0
    t_all = time.perf_counter()
    done = 0
    while done < n_contigs:
        n = min(per_batch, n_contigs - done)
        t = time.perf_counter()
        batch = P.Batch.synthetic([L] * n, seed=5, contig0=c0 + done, ctx=ctx)
        P._ffi.lib().pgr_ctx_synchronize(ctx.handle)
        t_synth += time.perf_counter() - t
        t = time.perf_counter()
        ix.add_resident(batch, sids=list(range(c0 + done, c0 + done + n)))   # shimmers + pair records, appended
        t_add += time.perf_counter() - t
        batch.close()
        done += n
    t = time.perf_counter()
    ix.finalize()
    t_fin = time.perf_counter() - t
    total = time.perf_counter() - t_all
    bp = n_contigs * L
    print(json.dumps({"workload": "configs[4] slice of one GPU: %d x %d bp (seed 5, contigs %d..%d)" % (n_contigs, L, c0, c0 + n_contigs),
                      "bases": bp, "synth_s": t_synth, "shimmers_and_records_s": t_add, "finalize_sort_s": t_fin,
                      "total_s": total, "records": ix.n_records, "keys": ix.n_keys,
                      "Gbp_per_s_index_build": bp / (t_add + t_fin) / 1e9,
                      "projected_100Gbp_on_8_gpus_s": 100e9 / 8 / (bp / (t_add + t_fin))}))


if __name__ == "__main__":
    main()
