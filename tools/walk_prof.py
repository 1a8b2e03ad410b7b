#!/usr/bin/env python3
"""The front end (GPU box) on m synthetic 30x samples per call, three calls: by default as the product drives it (round 6:
tred.run_many with gpu_select -- decode, pair walks, alternative-locus walks, read selection, then the genotyping call over
the reads packed on the device); WALK_PROF_PRODUCT=0: tredgpu_inflate_walk alone as in round 5 -- decode, pair walks (chain /
parse / resolve), alternative-locus walks --, printing the regions walked and the device time of the pair walk's launches.  `rocprofv3 --kernel-trace --stats -- python3 tools/walk_prof.py 16 tredparse_amd/libtredgpu.so <dir>` is how the
kernels' rocprofv3 summaries under profiles/ are taken (tools/profile_all.sh; the BAMs made beforehand with
`python tools/walk_prof.py make <dir>`: a process under the profiler must not fork the workers that write them).

  python tools/walk_prof.py [samples = 16] [library = tredparse_amd/libtredgpu.so] [directory of BAMs made before]
"""
import glob
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "make":
        from tredparse_amd import synth_bam
        os.makedirs(sys.argv[2], exist_ok=True)
        synth_bam.make_bams(sys.argv[2], 4, seed=7, workers=4)
        return
    from tredparse_amd import _lib
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    if len(sys.argv) > 2:
        _lib.LIB_PATH = os.path.abspath(sys.argv[2])
    from tredparse_amd import bam_parser, synth_bam
    from tredparse_amd.meta import TREDsRepo
    if len(sys.argv) > 3:
        root = sys.argv[3]
    else:
        root = tempfile.mkdtemp(prefix="tred_walkprof_")
        synth_bam.make_bams(root, 4, seed=7, workers=4)
    bams = sorted(glob.glob(os.path.join(root, "*.bam")))
    repo = TREDsRepo("hg38", sites=os.path.join(root, "no_sites"))
    names = [l["name"] for l in synth_bam.bench_loci()]
    loci = [repo[n] for n in names]
    if os.environ.get("WALK_PROF_PRODUCT", "1") != "0":
        # round 6: the product's front end as run_many drives it -- decode, walks, read selection (select_kernel), then the
        # genotyping call over the reads packed on the device (pack_selected_kernel + the SW / tally / grid kernels at this
        # batch size) -- three chunks of m samples
        from tredparse_amd import tred
        from tredparse_amd.engine import Engine
        engine = Engine(0)
        tasks = [("w%04d" % i, bams[i % len(bams)], repo, names, 300, False, False, True, True, "ERROR") for i in range(3 * m)]
        for k in tred.TIMING:
            tred.TIMING[k] = 0
        got = []
        tred.run_many(tasks, engine, batch=m, threads=2, lazy_details=True, sink=got.append, inflate_device=0, gpu_walk=True, gpu_select=True)
        print(json.dumps({"library": _lib.version(), "samples": m, "chunks": 3, "results": len(got),
                          "select_samples": int(tred.TIMING["select_samples"]), "select_declined": int(tred.TIMING["select_declined"]),
                          "walk_blocks_fetched": int(tred.TIMING["walk_blocks_fetched"]), "regions": int(tred.TIMING["walk_regions"]),
                          "alt_regions": int(tred.TIMING["walk_alt_regions"]), "blocks": int(tred.TIMING["inflate_blocks"])}))
        tred.release_inflaters()
        return
    inf = _lib.Inflater(0)
    hs = [bam_parser.open_bam(bams[k % len(bams)]) for k in range(m)]
    plans, tabs = [], []
    for f in hs:
        sites, regions = bam_parser._site_arrays(repo, names, loci, f)
        plans.append(f.plan(sites, regions, 150))
        tabs.append((f.plan_walks(sites, 150), f.plan_blocks(), f.plan_alt_walks(sites, regions, 150)))
    n_all = sum(p[0] for p in plans)
    comp, out, coff, ooff = inf.reserve(sum(p[1] for p in plans), sum(p[2] for p in plans), n_all)
    at = cb = ob = c0 = 0
    tasks, chunks, atasks, achunks, a0 = [], [], [], [], 0
    for f, p, ((t, c), _, (ta, ca)) in zip(hs, plans, tabs):
        f.plan_fill(inf.comp_addr, cb, ob, coff[at:at + p[0] + 1], ooff[at:at + p[0] + 1])
        t, c, ta, ca = t.copy(), c.copy(), ta.copy(), ca.copy()
        t["chunk_first"] += c0; t["block_first"] += at; t["block_end"] += at
        c["begin_block"][c["begin_block"] >= 0] += at
        ta["chunk_first"] += a0; ta["block_first"] += at; ta["block_end"] += at
        ca["begin_block"][ca["begin_block"] >= 0] += at
        tasks.append(t); chunks.append(c); atasks.append(ta); achunks.append(ca)
        at, cb, ob, c0, a0 = at + p[0], cb + p[1], ob + p[2], c0 + len(c), a0 + len(ca)
    atasks, achunks = np.concatenate(atasks), np.concatenate(achunks)
    bcoff, bclen, bcrc = (np.concatenate([tb[1][k] for tb in tabs]) for k in range(3))
    tasks, chunks = np.concatenate(tasks), np.concatenate(chunks)
    for _ in range(3):
        status, crc, res, gp, tp, ares, _ = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks, pairs_per_task=8192, alt_tasks=atasks, alt_chunks=achunks)
    ok = res["status"] == 0
    print(json.dumps({"library": _lib.version(), "samples": m, "regions": int(ok.sum()), "regions_declined": int((~ok).sum()),
                      "alt_regions": int((atasks["n_chunks"] >= 0).sum()), "alt_records": int(ares["n"].sum()),
                      "records_per_region_window": float(res["n_window"][ok].mean()), "pairs": int(len(gp) + len(tp)),
                      "pair_walk_ms (chain + parse + resolve launches)": inf.walk_ms()}))


if __name__ == "__main__":
    main()
