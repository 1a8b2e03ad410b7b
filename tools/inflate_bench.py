#!/usr/bin/env python3
"""Throughput of the GPU's batch DEFLATE decoder (tredgpu_inflate_blocks) on the BGZF blocks of a synthetic
30x BAM of the bench (tredparse_amd/synth_bam.py): per-call latency and samples/s against the number of samples in one
launch, the host CPU time of a call, and -- when the BAM layer is built -- the host cost of a scan with and without the
preloaded blocks.  One JSON line.  Run it directly, or behind `rocprofv3 --kernel-trace --stats -- python3` for the
kernel's own duration.

usage: python tools/inflate_bench.py [samples per launch ...]     (default: 1 4 16 28 56)
"""
import glob
import json
import os
import struct
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bgzf_blocks(path):
    """[(deflate payload, crc32, isize)] of every block of the file."""
    raw = open(path, "rb").read()
    pos, blocks = 0, []
    while pos < len(raw):
        xlen = struct.unpack_from("<H", raw, pos + 10)[0]
        bsize = struct.unpack_from("<H", raw, pos + 16)[0] + 1
        blocks.append((raw[pos + 12 + xlen:pos + bsize - 8],) + struct.unpack_from("<II", raw, pos + bsize - 8))
        pos += bsize
    return [b for b in blocks if b[2] > 0]


def lay(inf, blocks):
    n = len(blocks)
    offs = np.zeros(n + 1, np.int64)
    for k, b in enumerate(blocks):
        offs[k + 1] = (offs[k] + len(b[0]) + 3) & ~3
    comp, out, coff, ooff = inf.reserve(int(offs[-1]), int(sum(b[2] for b in blocks)), n)
    for k, b in enumerate(blocks):
        comp[offs[k]:offs[k] + len(b[0])] = np.frombuffer(b[0], np.uint8)
    coff[:] = offs
    ooff[0] = 0
    ooff[1:] = np.cumsum([b[2] for b in blocks])
    return out, ooff


def main():
    from tredparse_amd import _lib, bam_parser, synth_bam
    from tredparse_amd.meta import TREDsRepo
    sizes = [int(a) for a in sys.argv[1:]] or [1, 4, 16, 28, 56]
    root = tempfile.mkdtemp(prefix="tred_inflate_")
    # (workers=1: under rocprofv3 the process is on the GPU before Python starts, and such a process must not fork)
    synth_bam.make_bams(root, 4, seed=7, workers=int(os.environ.get("TRED_BENCH_WORKERS", "4")))
    bams = sorted(glob.glob(os.path.join(root, "*.bam")))
    blocks = bgzf_blocks(bams[0])
    rec = {"blocks_per_sample": len(blocks), "compressed_MB": sum(len(b[0]) for b in blocks) / 1e6,
           "inflated_MB": sum(b[2] for b in blocks) / 1e6, "library": _lib.version(), "launches": []}
    inf = _lib.Inflater(0)
    for m in sizes:
        out, ooff = lay(inf, blocks * m)
        status = inf.run(len(blocks) * m)
        assert (status == 0).all()
        assert all(zlib.crc32(bytes(out[ooff[k]:ooff[k + 1]])) == blocks[k % len(blocks)][1] for k in range(0, len(blocks) * m, 97))
        c0, t0 = time.process_time(), time.perf_counter()
        reps = 5
        for _ in range(reps):
            inf.run(len(blocks) * m)
        dt, dc = (time.perf_counter() - t0) / reps, (time.process_time() - c0) / reps
        dev_ms, kernel_ms = inf.timing()
        t0 = time.perf_counter()
        for _ in range(reps):
            st, sums = inf.run(len(blocks) * m, crc=True)
        dt_crc = (time.perf_counter() - t0) / reps
        assert (st == 0).all() and all(int(sums[k]) == blocks[k % len(blocks)][1] for k in range(len(blocks) * m))
        rec["launches"].append({"samples": m, "blocks": len(blocks) * m, "ms_per_call": dt * 1e3, "host_cpu_ms_per_call": dc * 1e3,
                                "device_ms": dev_ms, "kernel_ms": kernel_ms, "kernel_ms_with_crc": inf.timing()[1], "ms_per_call_with_crc": dt_crc * 1e3,
                                "samples_per_s": m / dt, "output_GBps": m * rec["inflated_MB"] / 1e3 / dt,
                                "kernel_samples_per_s": m / (kernel_ms * 1e-3), "kernel_output_GBps": m * rec["inflated_MB"] / kernel_ms})
    # the host's side of one sample: plain scan vs plan + fill + scan over preloaded blocks
    repo = TREDsRepo("hg38", sites=os.path.join(root, "no_sites"))
    names = [l["name"] for l in synth_bam.bench_loci()]
    loci = [repo[n] for n in names]
    cost = {"plain_scan": 0.0, "open": 0.0, "plan": 0.0, "fill": 0.0, "launch": 0.0, "scan_preloaded": 0.0}
    for rep in range(2):
        for b in bams:
            t = time.process_time(); s0 = bam_parser.scan_sample(b, repo, names); cost["plain_scan"] += (time.process_time() - t) * rep
            t = time.process_time(); f = bam_parser.open_bam(b); rl = f.max_read_len(101); cost["open"] += (time.process_time() - t) * rep
            sites, regions = bam_parser._site_arrays(repo, names, loci, f)
            t = time.process_time(); n, cb, ob = f.plan(sites, regions, rl, extra=bam_parser.y_regions("hg38")); cost["plan"] += (time.process_time() - t) * rep
            comp, out, coff, ooff = inf.reserve(cb, ob, n)
            t = time.process_time(); f.plan_fill(inf.comp_addr, 0, 0, coff, ooff); cost["fill"] += (time.process_time() - t) * rep
            t = time.process_time(); st, sums = inf.run(n, crc=True); cost["launch"] += (time.process_time() - t) * rep
            t = time.process_time()
            f.preload(inf.out_addr, ooff, st, sums)            # (device checksums: the scan does not walk the bytes for the CRC again)
            s1 = bam_parser.scan_sample(b, repo, names, handle=f, readlen=rl)
            hits, misses = f.preload_clear()
            cost["scan_preloaded"] += (time.process_time() - t) * rep
            f.close()
            assert np.array_equal(s0.packed, s1.packed) and np.array_equal(s0.global_lens, s1.global_lens)
    rec["host_cpu_ms_per_sample"] = {k: v / len(bams) * 1e3 for k, v in cost.items()}
    rec["planned_blocks"], rec["block_loads_preloaded"], rec["block_loads_inflated_by_the_scan"] = n, hits, misses
    # the same with the pair walk on the device (tredgpu_inflate_walk): what a call of m samples costs, which blocks come
    # back, and the host's share of a sample
    rec["walk"] = []
    for m in [k for k in sizes if k >= 4][-2:]:
        hs = [bam_parser.open_bam(bams[k % len(bams)]) for k in range(m)]
        plans, tabs = [], []
        c_plan = time.process_time()
        for f in hs:
            sites, regions = bam_parser._site_arrays(repo, names, loci, f)
            plans.append(f.plan(sites, regions, 150, extra=bam_parser.y_regions("hg38")))
            tabs.append((f.plan_walks(sites, 150), f.plan_blocks(), f.plan_alt_walks(sites, regions, 150)))
        c_plan = time.process_time() - c_plan
        n_all = sum(p[0] for p in plans)
        comp, out, coff, ooff = inf.reserve(sum(p[1] for p in plans), sum(p[2] for p in plans), n_all)
        at = cb = ob = c0 = 0
        firsts, tasks, chunks, atasks, achunks, a0 = [], [], [], [], [], 0
        for f, p, ((t, c), _, (ta, ca)) in zip(hs, plans, tabs):
            f.plan_fill(inf.comp_addr, cb, ob, coff[at:at + p[0] + 1], ooff[at:at + p[0] + 1])
            t, c, ta, ca = t.copy(), c.copy(), ta.copy(), ca.copy()
            t["chunk_first"] += c0; t["block_first"] += at; t["block_end"] += at
            c["begin_block"][c["begin_block"] >= 0] += at
            ta["chunk_first"] += a0; ta["block_first"] += at; ta["block_end"] += at
            ca["begin_block"][ca["begin_block"] >= 0] += at
            tasks.append(t); chunks.append(c); firsts.append(at); atasks.append(ta); achunks.append(ca)
            at, cb, ob, c0, a0 = at + p[0], cb + p[1], ob + p[2], c0 + len(c), a0 + len(ca)
        atasks, achunks = np.concatenate(atasks), np.concatenate(achunks)
        bcoff, bclen, bcrc = (np.concatenate([tb[1][k] for tb in tabs]) for k in range(3))
        tasks, chunks = np.concatenate(tasks), np.concatenate(chunks)
        inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks, alt_tasks=atasks, alt_chunks=achunks)
        reps = 3
        t0, c0_ = time.perf_counter(), time.process_time()
        for _ in range(reps):
            status, crc, res, gp, tp, ares, alt_need = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks, alt_tasks=atasks, alt_chunks=achunks)
        dt, dc = (time.perf_counter() - t0) / reps, (time.process_time() - c0_) / reps
        walk_ms, (dev_ms, kernel_ms) = inf.walk_ms(), inf.timing()
        assert (status == 0).all() and (res["status"] == 0).all(), np.unique(res["status"], return_counts=True)
        need = np.zeros(n_all, np.uint8)
        nt, na = len(names), len(atasks) // m
        assert (ares["status"][atasks["n_chunks"] >= 0] == 0).all()
        for k, (p, (_, blk, _)) in enumerate(zip(plans, tabs)):
            need[firsts[k]:firsts[k] + p[0]] = bam_parser.walk_need(blk[0], blk[3], res[k * nt:(k + 1) * nt], alt_need[firsts[k]:firsts[k] + p[0]])
        t0 = time.perf_counter()
        for _ in range(reps):
            copies = inf.fetch(need)
        dt_fetch = (time.perf_counter() - t0) / reps
        c_scan = time.process_time()
        for k, (f, p) in enumerate(zip(hs, plans)):
            a = firsts[k]
            f.preload(inf.out_addr, ooff[a:a + p[0] + 1], np.where(need[a:a + p[0]] != 0, status[a:a + p[0]], 1).astype(np.int32), crc[a:a + p[0]])
            s1 = bam_parser.scan_sample(bams[k % len(bams)], repo, names, handle=f, readlen=150, pe=(res[k * nt:(k + 1) * nt], gp, tp), alt=ares[k * na:(k + 1) * na])
            hits, misses = f.preload_clear()
            assert misses == 0
        c_scan = time.process_time() - c_scan
        s0 = bam_parser.scan_sample(bams[(m - 1) % len(bams)], repo, names, readlen=150)
        assert np.array_equal(s0.packed, s1.packed) and np.array_equal(s0.global_lens, s1.global_lens) and np.array_equal(s0.target_lens, s1.target_lens)
        for f in hs:
            f.close()
        rec["walk"].append({"samples": m, "blocks": n_all, "regions": len(tasks), "alt_regions": int((atasks["n_chunks"] >= 0).sum()),
                            "alt_records": int(ares["n"].sum()), "ms_per_call": dt * 1e3, "host_cpu_ms_per_call": dc * 1e3,
                            "decode_kernel_ms": kernel_ms, "walk_kernel_ms": walk_ms, "pairs": int(len(gp) + len(tp)),
                            "window_records": int(res["n_window"].sum()), "blocks_fetched": int(need.sum()), "fetch_copies": copies,
                            "fetch_ms": dt_fetch * 1e3, "fetched_MB_per_sample": float((np.diff(ooff[:n_all + 1])[need != 0]).sum()) / m / 1e6,
                            "samples_per_s_decode_walk_fetch": m / (dt + dt_fetch),
                            "host_cpu_ms_per_sample": {"plan_and_walk_tables": c_plan / m * 1e3, "preload_and_scan": c_scan / m * 1e3}})
    inf.close()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
