#!/usr/bin/env python3
"""Where ONE driver process of the from-BAM path spends its time with the BGZF blocks inflated on the GPU (run on the GPU
box): cProfile of the driver thread and of the writer thread, their thread CPU times and the process CPU time per sample
(16.7 ms of CPU per 30x sample).

usage: python tools/prof_e2e.py [synthetic BAMs = 96] [scan threads = 3] [pair walks on the GPU = 1]
"""
import cProfile, pstats, os, sys, tempfile, time, io, glob, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tredparse_amd import shard, synth_bam, tred
from tredparse_amd.meta import TREDsRepo
root = tempfile.mkdtemp(prefix="prof_e2e_")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
walk = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
made = synth_bam.make_bams(root, n, seed=7, workers=shard.usable_cpus())   # (process pool FIRST: a process that holds a HIP context must not fork)
import torch
torch.cuda.init()
from tredparse_amd.engine import Engine
bams = sorted(glob.glob(os.path.join(root, "*.bam")))
repo = TREDsRepo("hg38", sites=os.path.join(root, "no_sites"))
names = [l["name"] for l in synth_bam.bench_loci()]
tasks = [(os.path.basename(b)[:-4], b, repo, names, 300, False, False, True, True, "ERROR") for b in bams] * 3
engine = Engine(0)
os.chdir(root)
wprof = cProfile.Profile()
wcpu = [0.0]
def sink(result):
    c0 = time.thread_time()
    wprof.enable()
    tred.write_vcf_json(result, "hg38", repo, names, quiet=True)
    wprof.disable()
    wcpu[0] += time.thread_time() - c0
tred.run_many(tasks[:48], engine, batch=16, sink=sink, threads=threads, inflate_device=0, gpu_walk=walk)   # (warm: HIP context, inflaters' staging)
for k in tred.TIMING: tred.TIMING[k] = 0.0
wcpu[0] = 0.0
wprof = cProfile.Profile()
pr = cProfile.Profile()
t0, c0, p0 = time.perf_counter(), time.thread_time(), time.process_time()
pr.enable()
tred.run_many(tasks, engine, batch=16, sink=sink, threads=threads, lazy_details=True, background_sink=2, inflate_device=0, gpu_walk=walk)
pr.disable()
dt = time.perf_counter() - t0
print("gpu_walk", walk, "seconds", dt, "samples", len(tasks), "ms/sample wall", 1e3 * dt / len(tasks), "driver thread cpu ms/sample", 1e3 * (time.thread_time() - c0) / len(tasks),
      "writer thread cpu ms/sample", 1e3 * wcpu[0] / len(tasks), "process cpu ms/sample", 1e3 * (time.process_time() - p0) / len(tasks), tred.TIMING)
for name, prof in (("DRIVER", pr), ("WRITER", wprof)):
    s = io.StringIO()
    pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(22)
    print(name, s.getvalue()[:4500])
