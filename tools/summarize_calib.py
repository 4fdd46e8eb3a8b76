#!/usr/bin/env python3
"""gpurun_out/r05_calib/{fetch,write}/**/c_counter_collection.csv + bytes.json (tools/probe/fetch_calib) ->
profiles/r05_calib/calibration.json: bytes the kernel moved / bytes the counter reports, per access pattern."""
import csv
import glob
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05_calib"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r05_calib"
known = json.load(open(os.path.join(src, "bytes.json")))


def counter(kind, name):
    vals = {}
    for f in glob.glob(os.path.join(src, kind, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == name:
                k = r["Kernel_Name"].split("(")[0].split("::")[-1]
                vals.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}  # mean over the launches (one row per dispatch, summed over instances upstream)


def per_dispatch(kind, name):
    """Counter_Value rows come per (dispatch, dimension instance): sum the instances of a dispatch, then average the dispatches"""
    acc = {}
    for f in glob.glob(os.path.join(src, kind, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == name:
                k = r["Kernel_Name"].split("(")[0].split("::")[-1]
                acc.setdefault(k, {}).setdefault(r["Dispatch_Id"], 0.0)
                acc[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(d.values()) / len(d) for k, d in acc.items()}


fetch = per_dispatch("fetch", "FETCH_SIZE")
write = per_dispatch("write", "WRITE_SIZE")
out = {"unit": "the counters are in KiB", "patterns": {}}
for k, v in known.items():
    e = {"bytes_requested": v["requested"]}
    if "distinct" in v:
        e["bytes_distinct"] = v["distinct"]
    if k.startswith("read") and k in fetch:
        e["FETCH_SIZE_bytes"] = fetch[k] * 1024
        e["requested_over_counter"] = v["requested"] / (fetch[k] * 1024)
        if "distinct" in v:
            e["distinct_over_counter"] = v["distinct"] / (fetch[k] * 1024)
    if k.startswith("write") and k in write:
        e["WRITE_SIZE_bytes"] = write[k] * 1024
        e["requested_over_counter"] = v["requested"] / (write[k] * 1024)
    out["patterns"][k] = e
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, "calibration.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
