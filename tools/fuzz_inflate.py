#!/usr/bin/env python3
"""Randomised campaign for the GPU's batch DEFLATE decoder (tredgpu_inflate_blocks) against zlib: streams
of every zlib level / strategy / window over BAM-like records, incompressible bytes, runs, periodic text and the empty
stream, in batches of a few hundred per launch; every fifth stream damaged (bit flips, truncation, garbage).
Intact streams must come back byte-identical with status 0; damaged ones must come back with SOME status and, where
zlib still inflates them to the expected size, with zlib's bytes.  One JSON line.

usage: python tools/fuzz_inflate.py [rounds] [seed]
"""
import json
import os
import struct
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bam_like(rng, n):
    rec = bytearray()
    while len(rec) < n:
        name = b"r%09d" % int(rng.integers(10 ** 9))
        seq = bytes(rng.integers(0, 256, 75, dtype=np.uint8))
        qual = bytes(rng.choice(np.array([2, 11, 25, 37], np.uint8), 150))
        rec += struct.pack("<iiiIIiii", 300, 3, int(rng.integers(1 << 27)), 0x12345678, 0x0990000, 150, 3, 0) + name + b"\0" + seq + qual
    return bytes(rec[:n])


def make(rng):
    n = int(rng.choice([0, 1, 7, 300, 5000, 30000, 65280, 65536]))
    kind = int(rng.integers(5))
    if kind == 0:
        d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
    elif kind == 1:
        d = bytes(rng.integers(0, int(rng.integers(2, 9)), n, dtype=np.uint8))
    elif kind == 2:
        unit = bytes(rng.integers(65, 70, int(rng.integers(1, 40)), dtype=np.uint8))
        d = (unit * (n // len(unit) + 1))[:n]
    elif kind == 3:
        d = bam_like(rng, n)
    else:
        d = bytes(np.repeat(rng.integers(0, 256, max(n // 200, 1), dtype=np.uint8), 200)[:n])
    c = zlib.compressobj(int(rng.choice([0, 1, 3, 6, 9])), zlib.DEFLATED, -int(rng.choice([9, 12, 15])), int(rng.choice([1, 8, 9])),
                         [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED][int(rng.integers(5))])
    step = int(rng.choice([0, 0, 3000]))
    if step:
        p = b"".join(c.compress(d[i:i + step]) + c.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(d), step)) + c.flush()
    else:
        p = c.compress(d) + c.flush()
    return d, p


def campaign(rounds=4, seed=1, per_round=300):
    from tredparse_amd import _lib
    rng = np.random.default_rng(seed)
    inf = _lib.Inflater(0)
    res = {"streams": 0, "intact": 0, "damaged": 0, "mismatches": 0, "damaged_accepted_like_zlib": 0,
           "damaged_accepted_where_zlib_refuses": 0, "damaged_refused": 0, "bytes": 0}
    t0 = time.time()
    for _ in range(rounds):
        datas, payloads, want = [], [], []
        for k in range(per_round):
            d, p = make(rng)
            ok = d
            if k % 5 == 4 and len(p) > 0:
                p = bytearray(p)
                mode = int(rng.integers(3))
                if mode == 0:
                    for _ in range(int(rng.integers(1, 4))):
                        p[int(rng.integers(len(p)))] ^= 1 << int(rng.integers(8))
                elif mode == 1:
                    p = p[:int(rng.integers(1, len(p) + 1))]
                else:
                    p = bytearray(rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8).tobytes())
                p = bytes(p)
                try:
                    got = zlib.decompressobj(-15).decompress(p)
                    ok = got if len(got) == len(d) else None
                except zlib.error:
                    ok = None
                ok = ("damaged", ok)
            datas.append(d); payloads.append(p); want.append(ok)
        order = rng.permutation(per_round)
        n = per_round
        offs = np.zeros(n + 1, np.int64)
        for j, k in enumerate(order):
            offs[j + 1] = (offs[j] + len(payloads[k]) + 3) & ~3
        comp, out, coff, ooff = inf.reserve(int(offs[-1]), int(sum(len(d) for d in datas)), n)
        comp[:] = 0
        for j, k in enumerate(order):
            comp[offs[j]:offs[j] + len(payloads[k])] = np.frombuffer(payloads[k], np.uint8)
        coff[:] = offs
        ooff[0] = 0
        ooff[1:] = np.cumsum([len(datas[k]) for k in order])
        status = inf.run(n)
        for j, k in enumerate(order):
            got = bytes(out[ooff[j]:ooff[j + 1]])
            res["streams"] += 1
            res["bytes"] += len(got)
            w = want[k]
            if not isinstance(w, tuple):
                res["intact"] += 1
                if status[j] == -3:                     # more second-level Huffman tables than the decoder holds: the host's block
                    res["intact_left_to_the_host"] = res.get("intact_left_to_the_host", 0) + 1
                elif status[j] != 0 or got != w:
                    res["mismatches"] += 1
                    print("MISMATCH intact stream", k, status[j], len(w), file=sys.stderr)
            else:
                res["damaged"] += 1
                if status[j] != 0:
                    res["damaged_refused"] += 1
                elif w[1] is not None:
                    res["damaged_accepted_like_zlib"] += 1
                    if got != w[1]:
                        res["mismatches"] += 1
                        print("MISMATCH damaged stream accepted with other bytes", k, file=sys.stderr)
                else:
                    res["damaged_accepted_where_zlib_refuses"] += 1     # (a valid prefix: the host's CRC-32 catches it)
    inf.close()
    res.update(rounds=rounds, seed=seed, seconds=round(time.time() - t0, 1))
    return res


if __name__ == "__main__":
    res = campaign(int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    from tredparse_amd import _lib
    res["library"] = _lib.version()
    print(json.dumps(res))
