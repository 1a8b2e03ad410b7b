#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/profile_round.sh into one JSON (per kernel, per launch).

  python3 tools/pmc_to_json.py gpurun_out/prof_TAG > pmc_summary.json

Per kernel: launches, average duration (kernel_stats.csv), every counter summed over its dispatches / launches,
and hbm_bytes_per_launch = (FETCH_SIZE + WRITE_SIZE) x 1024 (rocprofv3 reports both in KiB;
MI355X_MICROARCH.md / cdna_hip_programming.md section 7).  FETCH_SIZE is NOT doubled: the gfx950 half-count
applies to wide (16 B per lane) coalesced streaming reads, and none of these kernels reads that way
(2-bit packed reads fetched 4 bytes per lane, table gathers); fetch_bytes_if_streaming gives the doubled figure
as the upper bound of the read side."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|tredgpu::|void ", "", name)
    return re.split(r"[<(]", name)[0]


def main(root):
    out = {"how": "rocprofv3 --kernel-trace --pmc <one pass per counter group> -- python3 bench.py --steps 3 --warmup 1 "
                  "--no-cpu-baseline (RANK=0 WORLD_SIZE=1); FETCH_SIZE and WRITE_SIZE in separate passes, KiB x 1024",
           "kernels": {}}
    # which build the counters belong to: the library's own report of its source hash (tredgpu_version)
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from tredparse_amd import _lib
        out["library_version"] = _lib.version()
    except Exception as e:                      # (summarising on a box without the library)
        out["library_version"] = "unavailable: {}".format(e)
    ks = os.path.join(root, "kernel_stats.csv")
    if os.path.exists(ks):
        for r in csv.DictReader(open(ks)):
            k = out["kernels"].setdefault(short(r["Name"]), {})
            k["launches_stats_pass"] = k.get("launches_stats_pass", 0) + int(r["Calls"])
            k["total_ns"] = k.get("total_ns", 0) + int(float(r["TotalDurationNs"]))
        for k in out["kernels"].values():
            k["avg_ms"] = k.pop("total_ns") / k["launches_stats_pass"] / 1e6
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for f in files:
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            rec = out["kernels"].setdefault(k, {})
            n = max(len(disp[k]), 1)
            for c, tot in v.items():
                rec[c + "_per_launch"] = tot / n
            rec.setdefault("launches_pmc_pass", {})[os.path.basename(d)] = n
    for k, rec in out["kernels"].items():
        f, w = rec.get("FETCH_SIZE_per_launch"), rec.get("WRITE_SIZE_per_launch")
        if f is not None and w is not None:
            rec["fetch_bytes_per_launch"] = f * 1024
            rec["write_bytes_per_launch"] = w * 1024
            rec["hbm_bytes_per_launch"] = (f + w) * 1024
            rec["fetch_bytes_if_streaming"] = 2 * f * 1024
        v, wc, busy = rec.get("SQ_INSTS_VALU_per_launch"), rec.get("SQ_WAVE_CYCLES_per_launch"), rec.get("SQ_BUSY_CYCLES_per_launch")
        if v and rec.get("avg_ms"):
            # VALU wave-instructions per SIMD and shader cycle at the nominal 2.4 GHz: x 2 cycles = issue-slot share
            rec["valu_inst_per_simd_cycle_at_2400MHz"] = v / 1024 / (rec["avg_ms"] * 1e-3 * 2.4e9)
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
