#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection.csv per kernel (short names)."""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::|tredgpu::|void ", "", r["Kernel_Name"]).split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    if len(sys.argv) < 3 or re.search(sys.argv[2], k):
        print(k, "dispatches", len(n[k]), {a: "%.4g" % b for a, b in sorted(v.items())})
