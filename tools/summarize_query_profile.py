#!/usr/bin/env python3
"""rocprofv3 kernel trace of tools/query_leg.py -> profiles/<name>/{kernel_stats.csv, summary.json}

    tools/summarize_query_profile.py gpurun_out/<dir>/q_trace profiles/r03_query [batches]

Only the kernels after the 0.5 s gap (the timed query batches) are counted; numbers are per query batch."""
import collections
import csv
import glob
import json
import os
import sys

import re


def short_name(n):
    n = n.replace("(anonymous namespace)::", "")
    if n.startswith("void "):
        n = n[5:]
    if "rocprim" in n:
        m = re.search(r"(radix_sort_block_sort|radix_sort_onesweep\w*|onesweep\w*|lookback_scan\w*|transform_impl|radix_sort\w*|"
                      r"histogram\w*|scan\w*|init_\w+)", n)
        return "rocprim::" + (m.group(1) if m else "kernel(lambda)")
    return n.split("(")[0].split("<")[0]


src, dst = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # 0: one batch per launch of the tile kernel behind the gap
os.makedirs(dst, exist_ok=True)
rows = list(csv.DictReader(open(glob.glob(os.path.join(src, "*kernel_trace.csv"))[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gap_i, gap = 0, 0
for i in range(1, len(rows)):
    g = int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])
    if g > gap:
        gap, gap_i = g, i
q = rows[gap_i:]
if reps == 0:
    reps = max(1, sum(1 for r in q if "level1_tile_kernel" in r["Kernel_Name"]))
agg = collections.OrderedDict()
for r in q:
    n = short_name(r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += d
span = (int(q[-1]["End_Timestamp"]) - int(q[0]["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in agg.values()) / 1e3
with open(os.path.join(dst, "kernel_stats.csv"), "w") as f:
    f.write("kernel,launches_per_batch,us_per_batch,percent_of_kernel_time\n")
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write('"%s",%.1f,%.2f,%.1f\n' % (n, c / reps, us / reps, 100.0 * us / 1e3 / tot))
dom = max(agg.items(), key=lambda kv: kv[1][1])
out = {"source": "rocprofv3 --kernel-trace of tools/query_leg.py (%d resident query batches of BASELINE.json configs[2])" % reps,
       "launches": round(len(q) / reps, 1), "kernel_ms_total": tot / reps, "wall_ms_per_batch_under_the_profiler": span / reps,
       "dominant_kernel": dom[0], "dominant_kernel_ms": dom[1][1] / 1e3 / reps, "gap_before_timed_batches_ms": gap / 1e6}
json.dump(out, open(os.path.join(dst, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
print(open(os.path.join(dst, "kernel_stats.csv")).read()[:3000])

# ---- counters of the same batches (optional: <src>/../q_pmc_{fetch,write,sq}): per query batch, summed over the kernels
#      behind the 0.5 s gap; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 correction as in summarize_profile.py)
pm = {}
for d in ("q_pmc_fetch", "q_pmc_write", "q_pmc_sq"):
    fs = glob.glob(os.path.join(os.path.dirname(src.rstrip("/")), d, "*counter_collection.csv"))
    if not fs:
        continue
    rr = list(csv.DictReader(open(fs[0])))
    rr.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the gap: largest distance between consecutive dispatches
    gi, g = 0, 0
    for i in range(1, len(rr)):
        dd = int(rr[i]["Start_Timestamp"]) - int(rr[i - 1]["End_Timestamp"])
        if dd > g:
            g, gi = dd, i
    qq = rr[gi:]
    nb = max(1, len(set(r["Dispatch_Id"] for r in qq if "level1_tile_kernel" in r["Kernel_Name"])))
    for r in qq:
        k = short_name(r["Kernel_Name"])
        e = pm.setdefault(k, {})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]) / nb
if pm:
    tot = {}
    for k, e in pm.items():
        for c, v in e.items():
            tot[c] = tot.get(c, 0.0) + v
    hbm = (2.0 * tot.get("FETCH_SIZE", 0.0) + tot.get("WRITE_SIZE", 0.0)) * 1024.0
    top = sorted(pm.items(), key=lambda kv: -(2.0 * kv[1].get("FETCH_SIZE", 0.0) + kv[1].get("WRITE_SIZE", 0.0)))[:12]
    json.dump({"per_query_batch": {"hbm_bytes": hbm, "fetch_size_KiB_raw": tot.get("FETCH_SIZE"), "write_size_KiB": tot.get("WRITE_SIZE"),
                                   "valu_wave_insts": tot.get("SQ_INSTS_VALU"), "waves": tot.get("SQ_WAVES"),
                                   "lds_insts": tot.get("SQ_INSTS_LDS")},
               "correction": "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), as for the tile kernel",
               "kernels_by_hbm_bytes": [{"kernel": k, **{c: v for c, v in e.items()}} for k, e in top]},
              open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    print("pmc_summary.json: %.1f MB of HBM traffic per query batch" % (hbm / 1e6))

# The bench line of the same gpurun command (copied to profiles/<tile profile>/bench.json by summarize_profile.py) was printed BEFORE
# this summary existed: the derived fields of its query block (launches, kernel time, PMC traffic) came from the previous profile.
# Refresh them from this run's trace so that the line and the trace next to it belong to the same build.
bj = os.path.join(os.path.dirname(os.path.abspath(dst)), "r03_tile", "bench.json")
if os.path.exists(bj) and os.path.exists(os.path.join(os.path.dirname(src.rstrip("/")), "bench.json")):
    bench = json.load(open(bj))
    r = bench.get("query", {}).get("roofline")
    if r is not None:
        r.update({"dominant_kernel": out["dominant_kernel"], "dominant_kernel_ms": out["dominant_kernel_ms"],
                  "kernel_ms_total": out["kernel_ms_total"], "launches": out["launches"]})
        if pm:
            r["traffic"] = hbm
        if isinstance(r.get("what_holds"), str):
            import re
            r["what_holds"] = re.sub(r"\d+ kernel launches", "%d kernel launches" % round(out["launches"]), r["what_holds"])
        bench["derived_blocks_query"] = "query.roofline.{launches, kernel_ms_total, dominant_kernel*, traffic} refreshed by tools/summarize_query_profile.py from the trace of this same profile run"
        json.dump(bench, open(bj, "w"))
