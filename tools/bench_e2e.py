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
    import torch  # noqa: F401  (load PyTorch's HIP runtime before libtredgpu)
    if torch.cuda.is_available():
        torch.cuda.init()
    from tredparse_amd import tred
    from tredparse_amd.engine import Engine
    from tredparse_amd.meta import TREDsRepo
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    repo = TREDsRepo("hg38")
    names = list(repo.names)
    bams = [os.path.join(ROOT, "tests", "golden", "bam", b) for b in ("t001.bam", "t002.bam")]
    engine = Engine()
    t_host = t_dev = 0.0
    units = 0
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        collected = []
        for bam in bams:
            arg = (os.path.basename(bam)[:-4], bam, repo, names, 300, False, False, False, True, "ERROR")
            collected.append(tred.collect_sample(arg))
        t1 = time.perf_counter()
        pend = [p for _, ps in collected for p in ps]
        res = engine.genotype([p.caller.unit([sq for _, sq in p.bp.reads]) for p in pend], want_grid=True)
        k = 0
        for result, ps in collected:
            tred.finish_sample(result, ps, res[k:k + len(ps)])
            k += len(ps)
        t2 = time.perf_counter()
        if rep:   # first repetition warms caches / the HIP context
            t_host += t1 - t0
            t_dev += t2 - t1
            units += len(pend)
    print(json.dumps({"metric": "sample x TRED genotypes/sec end to end from BAM (one host process)",
                      "value": units / (t_host + t_dev), "unit": "genotypes/s", "units": units, "reps": reps,
                      "host_collect_s": t_host, "device_and_format_s": t_dev,
                      "workload": "tests/golden/bam t001 + t002 x {} loci".format(len(names))}))


if __name__ == "__main__":
    main()
