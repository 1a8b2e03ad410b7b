#!/usr/bin/env python3
"""Copies the summaries of tools/profile_all.sh from gpurun_out/prof_<round>_*/ into profiles/ under the round's names
(<round>_<leg>_pmc_summary.json, <round>_<leg>_kernel_stats.csv; the headline leg also as <round>_pmc_summary.json, the
name bench.py looks up counter traffic under) and refuses a set taken on more than one build.

  python3 tools/collect_profiles.py r06
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main(rnd):
    versions = {}
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_{}_*".format(rnd)))):
        if not os.path.isdir(d):
            continue
        leg = os.path.basename(d)[len("prof_{}_".format(rnd)):]
        summary = os.path.join(d, "pmc_summary.json")
        if os.path.exists(summary):
            with open(summary) as fp:
                versions[leg] = json.load(fp).get("library_version")
            shutil.copy(summary, os.path.join(ROOT, "profiles", "{}_{}_pmc_summary.json".format(rnd, leg)))
            if leg == "headline":
                shutil.copy(summary, os.path.join(ROOT, "profiles", "{}_pmc_summary.json".format(rnd)))
        stats = os.path.join(d, "kernel_stats.csv")
        if os.path.exists(stats):
            shutil.copy(stats, os.path.join(ROOT, "profiles", "{}_{}_kernel_stats.csv".format(rnd, leg)))
        for extra in ("ubench_valu.txt", "bench_under_rocprof.json"):
            if leg == "headline" and os.path.exists(os.path.join(d, extra)):
                shutil.copy(os.path.join(d, extra), os.path.join(ROOT, "profiles", "{}_{}".format(rnd, extra)))
        for name in sorted(os.listdir(d)):
            if name.startswith("pmc_") and name.endswith(".err") is False and os.path.isdir(os.path.join(d, name)):
                for f in glob.glob(os.path.join(d, name, "**", "*counter_collection.csv"), recursive=True):
                    if os.path.getsize(f) < 4 << 20:
                        shutil.copy(f, os.path.join(ROOT, "profiles", "{}_{}_{}.csv".format(rnd, leg, name)))
    print(json.dumps(versions, indent=1))
    if len(set(versions.values())) > 1:
        raise SystemExit("the summaries of round {} come from more than one build".format(rnd))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r06")
