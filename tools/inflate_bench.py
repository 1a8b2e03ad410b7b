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
    inf.close()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
