#!/usr/bin/env python3
"""How the GPU shares the front-end call (decode + pair walks + alternative-locus walks + fetch of the wanted blocks)
between callers: T threads of ONE process, each with an inflater (stream) of its own, against P processes with one
each -- samples per second of the whole job.  Run on the GPU box:

  python tools/conc_probe.py make <dir>                     four synthetic BAMs (before anything touches the GPU)
  python tools/conc_probe.py run <dir> <samples per call> <threads> [seconds = 4]
  python tools/conc_probe.py sweep <dir>                    the table of profiles/r05_conc_probe.json

The bench's drivers are processes because of the interpreter lock on the host side; this says what the DEVICE side of
that choice costs.
"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def prepare(root, m, inf):
    """One call's worth of input: m samples (the four BAMs in turn) planned and filled into inf's staging."""
    from tredparse_amd import _lib, bam_parser, synth_bam
    from tredparse_amd.meta import TREDsRepo
    bams = sorted(glob.glob(os.path.join(root, "*.bam")))
    repo = TREDsRepo("hg38", sites=os.path.join(root, "no_sites"))
    names = [l["name"] for l in synth_bam.bench_loci()]
    loci = [repo[n] for n in names]
    hs = [bam_parser.open_bam(bams[k % len(bams)]) for k in range(m)]
    plans, tabs = [], []
    for f in hs:
        sites, regions = bam_parser._site_arrays(repo, names, loci, f)
        plans.append(f.plan(sites, regions, 150))
        tabs.append((f.plan_walks(sites, 150), f.plan_blocks(), f.plan_alt_walks(sites, regions, 150)))
    n_all = sum(p[0] for p in plans)
    comp, out, coff, ooff = inf.reserve(sum(p[1] for p in plans), sum(p[2] for p in plans), n_all)
    at = cb = ob = c0 = a0 = 0
    tasks, chunks, atasks, achunks, host = [], [], [], [], []
    for f, p, ((t, c), blocks, (ta, ca)) in zip(hs, plans, tabs):
        f.plan_fill(inf.comp_addr, cb, ob, coff[at:at + p[0] + 1], ooff[at:at + p[0] + 1])
        t, c, ta, ca = t.copy(), c.copy(), ta.copy(), ca.copy()
        t["chunk_first"] += c0; t["block_first"] += at; t["block_end"] += at
        c["begin_block"][c["begin_block"] >= 0] += at
        ta["chunk_first"] += a0; ta["block_first"] += at; ta["block_end"] += at
        ca["begin_block"][ca["begin_block"] >= 0] += at
        tasks.append(t); chunks.append(c); atasks.append(ta); achunks.append(ca)
        host.append((at, p[0], blocks[0], blocks[3], len(t)))
        at, cb, ob, c0, a0 = at + p[0], cb + p[1], ob + p[2], c0 + len(c), a0 + len(ca)
    bcoff, bclen, bcrc = (np.concatenate([tb[1][k] for tb in tabs]) for k in range(3))
    return dict(n=n_all, bcoff=bcoff, bclen=bclen, bcrc=bcrc, tasks=np.concatenate(tasks), chunks=np.concatenate(chunks),
                atasks=np.concatenate(atasks), achunks=np.concatenate(achunks), ooff=ooff, host=host, keep=hs)


def one_call(inf, w):
    from tredparse_amd import _lib
    from tredparse_amd.bam_parser import walk_need
    status, crc, res, gp, tp, ares, alt_need = inf.run_walk(w["n"], w["bcoff"], w["bclen"], w["bcrc"], w["tasks"], w["chunks"],
                                                            alt_tasks=w["atasks"], alt_chunks=w["achunks"],
                                                            pool_pairs=_lib.walk_pool_pairs(w["tasks"], w["ooff"]))
    need = np.zeros(w["n"], np.uint8)
    t0 = 0
    for at, n, coffset, hostflags, nt in w["host"]:
        need[at:at + n] = walk_need(coffset, hostflags, res[t0:t0 + nt], alt_need[at:at + n])
        t0 += nt
    inf.fetch_dense(need)
    return int((res["status"] != 0).sum())


def run(root, m, threads, seconds):
    from tredparse_amd import _lib
    infs = [_lib.Inflater(0, host_out=False) for _ in range(threads)]      # (as the product's feeder uses them: dense fetch by kernel)
    work = [prepare(root, m, inf) for inf in infs]
    for inf, w in zip(infs, work):
        one_call(inf, w)                      # warm: allocations, pinned staging
    counts = [0] * threads
    stop = [False]
    go = threading.Barrier(threads + 1)

    def loop(k):
        go.wait()
        while not stop[0]:
            one_call(infs[k], work[k])
            counts[k] += 1
    ts = [threading.Thread(target=loop, args=(k,)) for k in range(threads)]
    for t in ts:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    time.sleep(seconds)
    stop[0] = True
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    return {"samples_per_call": m, "threads": threads, "calls": sum(counts), "seconds": dt, "samples_per_s": sum(counts) * m / dt,
            "ms_per_call": 1e3 * dt * threads / max(1, sum(counts))}


def main():
    cmd, root = sys.argv[1], sys.argv[2]
    if cmd == "make":
        from tredparse_amd import synth_bam
        os.makedirs(root, exist_ok=True)
        synth_bam.make_bams(root, 4, seed=7, workers=4)
        return
    if cmd == "run":
        m, threads = int(sys.argv[3]), int(sys.argv[4])
        seconds = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0
        print(json.dumps(run(root, m, threads, seconds)), flush=True)
        return
    # sweep: (processes, threads per process, samples per call); processes start together and run for the same time
    rows = []
    for procs, threads, m in ((1, 1, 16), (1, 2, 16), (1, 3, 16), (1, 4, 16), (1, 6, 16), (2, 1, 16), (3, 1, 16), (6, 1, 16), (3, 2, 16),
                              (1, 1, 32), (1, 3, 32), (3, 1, 32), (1, 1, 48), (1, 2, 48), (2, 1, 48)):
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "run", root, str(m), str(threads), "4"],
                               stdout=subprocess.PIPE, text=True) for _ in range(procs)]
        outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
        rows.append({"processes": procs, "threads_per_process": threads, "samples_per_call": m,
                     "samples_per_s": round(sum(o["samples_per_s"] for o in outs), 1),
                     "ms_per_call": round(float(np.mean([o["ms_per_call"] for o in outs])), 2)})
        print(json.dumps(rows[-1]), flush=True)
    from tredparse_amd import _lib
    print(json.dumps({"library": _lib.version(), "what": "tredgpu_inflate_walk + tredgpu_inflater_fetch_dense of the wanted blocks, 30x samples, "
                      "callers looping for 4 s", "rows": rows}))


if __name__ == "__main__":
    main()
