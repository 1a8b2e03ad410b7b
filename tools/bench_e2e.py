#!/usr/bin/env python3
"""End-to-end rate from BAM files (reported next to the kernel-only rate of bench.py).

Runs the drop-in driver's two halves over the reference's two test BAMs x all loci of the table:
host half (BAM open, sex / read length / depth, read selection, pair lengths: tredparse_amd.tred.collect_sample)
and device half (one Engine.genotype batch for all units + result formatting), and prints one JSON line with
the time split.  The BAMs are tiny (one covered locus each), so this measures the per-sample x locus host
overhead of the Python front end, not inflate throughput.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=64, help="samples per repetition (the two BAMs, repeated)")
    ap.add_argument("--cpus", type=int, default=1, help="host threads scanning BAMs")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--drivers", type=int, default=1,
                    help="independent driver processes sharing the GPU (each with its own --cpus workers); the "
                         "aggregate rate is printed")
    a = ap.parse_args()
    if a.drivers > 1:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--samples", str(a.samples), "--cpus", str(a.cpus),
               "--reps", str(a.reps)]
        t0 = time.perf_counter()
        procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(a.drivers)]
        outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
        wall = time.perf_counter() - t0
        print(json.dumps({"metric": "sample x TRED genotypes/sec end to end from BAM", "unit": "genotypes/s",
                          "value": sum(o["value"] for o in outs), "drivers": a.drivers, "host_workers_per_driver": a.cpus,
                          "per_driver": [round(o["value"]) for o in outs], "wall_seconds_incl_startup": wall,
                          "workload": outs[0]["workload"]}))
        return
    from tredparse_amd import tred
    from tredparse_amd.meta import TREDsRepo
    repo = TREDsRepo("hg38")
    names = list(repo.names)
    bams = [os.path.join(ROOT, "tests", "golden", "bam", b) for b in ("t001.bam", "t002.bam")]
    tasks = [("s{:04d}".format(i), bams[i % 2], repo, names, 300, False, False, False, True, "ERROR")
             for i in range(a.samples)]
    import torch  # noqa: F401  (load PyTorch's HIP runtime before libtredgpu)
    if torch.cuda.is_available():
        torch.cuda.init()
    from tredparse_amd.engine import Engine
    engine = Engine()
    done = []
    tred.run_many(tasks[:4], engine, pool=None, sink=done.append)      # warm-up: caches, HIP context
    t0 = time.perf_counter()
    for _ in range(a.reps):
        tred.run_many(tasks, engine, batch=64, sink=done.append, threads=a.cpus)
    dt = time.perf_counter() - t0
    units = a.reps * len(tasks) * len(names)
    print(json.dumps({"metric": "sample x TRED genotypes/sec end to end from BAM",
                      "value": units / dt, "unit": "genotypes/s", "units": units, "seconds": dt,
                      "host_workers": a.cpus, "samples_per_rep": len(tasks), "reps": a.reps,
                      "workload": "tests/golden/bam t001 / t002 alternating x {} loci, 64 samples per GPU batch".format(len(names))}))


if __name__ == "__main__":
    main()
