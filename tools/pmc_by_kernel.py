#!/usr/bin/env python3
"""rocprofv3 --pmc ... --output-format csv -d DIR  ->  the counters summed per kernel (per dispatch average with --per-dispatch)
    python tools/pmc_by_kernel.py DIR [substring ...]"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
want = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    k = re.sub(r"^void ", "", k).split("(")[0][:70]
    if want and not any(w in k for w in want):
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r.get("Dispatch_Id", ""))
for k, v in agg.items():
    n = max(1, len(disp[k]))
    print("%s  dispatches %d" % (k, n))
    for a, b in sorted(v.items()):
        print("    %-22s %16.0f per dispatch" % (a, b / n))
