#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: short kernel name, calls, total ms, avg us."""
import csv
import re
import sys

for row in list(csv.DictReader(open(sys.argv[1])))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    name = re.sub(r"\(anonymous namespace\)::|tredgpu::|void ", "", row["Name"]).split("(")[0]
    print("%-40s calls %5s  total %9.3f ms  avg %10.1f us" % (name[:40], row["Calls"], float(row["TotalDurationNs"]) / 1e6,
                                                              float(row["AverageNs"]) / 1e3))
