#!/usr/bin/env python3
"""gpurun_out/prof_<tag> -> profiles/<name>/ (kernel_stats.csv, pmc_summary.json, bench.json) and
profiles/traffic.json (HBM bytes per launch of the dominant kernel, read by bench.py).

HBM traffic per launch = k x FETCH_SIZE + WRITE_SIZE (KiB -> bytes).  On gfx950 FETCH_SIZE under-reports wide coalesced reads
(MI355X_MICROARCH.md, HBM section: x 2 for a streaming read); for THIS kernel's pattern the factor was measured (profiles/r05_calib):
1.81 (bytes requested) / 1.63 (distinct bytes) -- both are written out.  WRITE_SIZE is taken as is (calibration 1.000; it also
matches the known size of the level-1 list written by the kernel)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles", name)
os.makedirs(dst, exist_ok=True)
shutil.copyfile(glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))[0], os.path.join(dst, "kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench.json")):
    shutil.copyfile(os.path.join(src, "bench.json"), os.path.join(dst, "bench.json"))
out = {}
for d in ["pmc_fetch", "pmc_write", "pmc_sq", "pmc_misc", "pmc_valu", "pmc_valu2"]:
    fs = glob.glob(os.path.join(src, d, "*counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "pgr::" in k:
            out.setdefault(k, {}).update({c: {"launches": len(x), "mean_per_launch": sum(x) / len(x)} for c, x in v.items()})
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
dom = [k for k in out if "level1_tile_kernel" in k]
if dom and "FETCH_SIZE" in out[dom[0]] and "WRITE_SIZE" in out[dom[0]]:
    f = out[dom[0]]["FETCH_SIZE"]["mean_per_launch"] * 1024
    w = out[dom[0]]["WRITE_SIZE"]["mean_per_launch"] * 1024
    bench = json.load(open(os.path.join(dst, "bench.json"))) if os.path.exists(os.path.join(dst, "bench.json")) else {}
    # FETCH_SIZE under-reports wide reads on gfx950 (the guide's correction: x 2).  The factor was MEASURED for this kernel's access
    # pattern (tools/probe/fetch_calib.hip -> profiles/r05_calib/calibration.json, pattern read_tile_like: 4096-position tiles
    # with the halos re-read): bytes REQUESTED by the loads = 1.81 x the counter, DISTINCT bytes = 1.63 x.  The halo re-reads are
    # L2 hits or not, so HBM read traffic lies between the two; `hbm_bytes_per_launch` is the upper figure, `..._low` the lower.
    cal = json.load(open(os.path.join(ROOT, "profiles", "r05_calib", "calibration.json")))["patterns"]["read_tile_like"]
    k_req, k_dist = cal["requested_over_counter"], cal["distinct_over_counter"]
    t = {"kernel": "level1_tile_kernel", "profile": name, "bp_per_launch": bench.get("roofline", {}).get("bp_per_launch"),
         "fetch_size_bytes_raw": f, "write_size_bytes": w,
         "read_bytes_requested": k_req * f, "read_bytes_distinct": k_dist * f,
         "hbm_bytes_per_launch": k_req * f + w, "hbm_bytes_per_launch_low": k_dist * f + w,
         "hbm_bytes_per_launch_by_the_guides_factor_2": 2 * f + w,
         "correction": "FETCH_SIZE x %.3f (requested) / x %.3f (distinct), measured for the tile pattern in profiles/r05_calib; "
                       "WRITE_SIZE as is (calibration: 1.000)" % (k_req, k_dist)}
    if "SQ_INSTS_VALU" in out[dom[0]]:
        t["valu_wave_insts_per_launch"] = out[dom[0]]["SQ_INSTS_VALU"]["mean_per_launch"]
    if "SQ_ACTIVE_INST_VALU2" in out[dom[0]] and "GRBM_GUI_ACTIVE" in out[dom[0]]:
        # counter-only VALU occupancy: every VALU instruction holds the SIMD's issue port for one slot, except the ones the
        # hardware counts as issued through its second path (SQ_ACTIVE_INST_VALU2: 46 % of the instructions of the 2.4-cycle
        # opcodes in the single-opcode micro-kernels, 0 for every other opcode).  Slot length = cycles per instruction of the
        # micro-kernels without any second-path instruction (profiles/r03_ubench: 4.17).
        t["valu2_wave_insts_per_launch"] = out[dom[0]]["SQ_ACTIVE_INST_VALU2"]["mean_per_launch"]
        t["gui_active_cycles_per_launch"] = out[dom[0]]["GRBM_GUI_ACTIVE"]["mean_per_launch"]
    # the opcode histogram of the same build and the measured opcode costs live next to the counters (tools/isa_histogram.py,
    # tools/ubench_table.py); bench.py follows these two keys
    if os.path.exists(os.path.join(dst, "isa_histogram.json")):
        t["isa_histogram"] = name + "/isa_histogram.json"
    else:
        # no histogram of this build: the last committed one stays the reference as long as the hot path's instruction stream is
        # the same (SQ_INSTS_VALU per launch says so).  Round 6: tools/isa_histogram.py over-counts the current code object -- the cold
        # blocks of round 5's palindrome-position report (level1.hip: s_pal, option pal_positions) carry no cold marker and weigh 1 --
        # 1474 against the 1217 instructions per wavefront of r05_tile's histogram; the hardware counts 11.646 G per launch in both.
        try:
            prev = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("isa_histogram")
        except (OSError, ValueError):
            prev = None
        prev = prev or "r05_tile/isa_histogram.json"
        if os.path.exists(os.path.join(ROOT, "profiles", prev)):
            t["isa_histogram"] = prev
    ub = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ubench", "valu_cycles.json")))
    if ub:
        t["valu_cycles"] = os.path.relpath(ub[-1], os.path.join(ROOT, "profiles"))
    json.dump(t, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(t))
    # bench.json was written by the same gpurun command BEFORE these counters were summarized: its derived blocks (PMC traffic,
    # VALU occupancy) were computed from the previous profile's counters.  Recompute them -- with bench.py's own function, from
    # the launch time that run measured -- so that the line and the counters next to it belong to the same build.
    if bench.get("roofline"):
        sys.path.insert(0, ROOT)
        import bench as bench_mod
        r = bench["roofline"]
        r["traffic"] = t["hbm_bytes_per_launch"]
        r.update(bench_mod.traffic_detail(t, r["bp_per_launch"], bench.get("config", {}).get("final_shimmers_per_gpu"),
                                          bench.get("roofline", {}).get("level1_minimizers")))
        vi = bench_mod.valu_issue(r["bp_per_launch"], r["avg_launch_ms"])
        if vi:
            r["valu_issue"] = vi
        bench["derived_blocks"] = "roofline.traffic and roofline.valu_issue recomputed by tools/summarize_profile.py from the counters of this same profile run"
        json.dump(bench, open(os.path.join(dst, "bench.json"), "w"))
for row in csv.DictReader(open(os.path.join(dst, "kernel_stats.csv"))):
    if "pgr::" in row["Name"]:
        print("%-60s calls %3s avg %10.3f ms  %5s%%" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e6, row["Percentage"]))
