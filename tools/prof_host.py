#!/usr/bin/env python3
"""What a driver process costs the HOST per sample, measured without a GPU: real native scans of synthetic BAMs, the real
packing, formatting and file writing -- only the three kernel calls are replaced by arrays of the same shapes and
densities (a third of the reads tagged, marginals of 1-30 entries, ~80 joint entries per unit).  cProfile of the driver
thread and of the writer thread plus their CPU times: the numbers behind DESIGN 6's "interpreter-lock work per sample".

usage: python tools/prof_host.py [synthetic BAMs = 24] [scan threads = 4] [native emit = 1]
"""
import cProfile
import glob
import io
import os
import pstats
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tredparse_amd import _lib, engine as eng, shard, synth_bam, tred   # noqa: E402
from tredparse_amd.meta import TREDsRepo                                  # noqa: E402


from tests.fake_engine import FakeEngine                                   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    native = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
    root = tempfile.mkdtemp(prefix="prof_host_")
    synth_bam.make_bams(root, n, seed=7, workers=shard.usable_cpus())
    bams = sorted(glob.glob(os.path.join(root, "*.bam")))
    repo = TREDsRepo("hg38", sites=os.path.join(root, "no_sites"))
    names = [l["name"] for l in synth_bam.bench_loci()]
    tasks = [(os.path.basename(b)[:-4], b, repo, names, 300, False, False, True, True, "ERROR") for b in bams] * 4
    engine = FakeEngine(odd_units=False)
    os.chdir(root)
    wprof, wcpu = cProfile.Profile(), [0.0]

    def sink(result):
        c0 = time.thread_time()
        wprof.enable()
        tred.write_vcf_json(result, "hg38", repo, names, quiet=True)
        wprof.disable()
        wcpu[0] += time.thread_time() - c0
    def go(some, batch):
        if not native:
            return tred.run_many(some, engine, batch=batch, sink=sink, threads=threads, lazy_details=True, background_sink=True)
        emit = tred.Emitter("hg38", repo, names, workers=2)
        try:
            tred.run_many(some, engine, batch=batch, threads=threads, lazy_details=True, emit=emit)
        finally:
            emit.close()
    go(tasks[:8], 8)
    wcpu[0] = 0.0
    for k in tred.TIMING:
        tred.TIMING[k] = 0.0
    pr = cProfile.Profile()
    t0, c0, p0 = time.perf_counter(), time.thread_time(), time.process_time()
    pr.enable()
    go(tasks, 16)
    pr.disable()
    dt = time.perf_counter() - t0
    k = len(tasks)
    print("samples {}  wall {:.2f} s = {:.2f} ms/sample   driver thread cpu {:.2f} ms/sample   writer thread cpu {:.2f} ms/sample   "
          "process cpu {:.2f} ms/sample".format(k, dt, 1e3 * dt / k, 1e3 * (time.thread_time() - c0) / k, 1e3 * wcpu[0] / k,
                                                1e3 * (time.process_time() - p0) / k))
    print({a: round(b, 3) for a, b in tred.TIMING.items() if b})
    for name, prof in (("DRIVER", pr), ("WRITER", wprof)):
        s = io.StringIO()
        pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(25)
        print(name, s.getvalue()[:5000])
    import shutil
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
