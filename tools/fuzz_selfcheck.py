#!/usr/bin/env python3
"""Randomised self-check of the SW kernel on the GPU alone: the production launch (every exact shortcut on: 6-mer strand
filter, exact-score drop, strand exit, steady-state exit, deferred best-cell resolution) against the arg-max over the
per-template dump of the kernel variant that has none of them (whose records tools/fuzz_parity.py and the tests check
against the reference's ssw.c field by field).  No CPU oracle in the loop, so a round takes seconds for tens of
thousands of reads: the batches are those of fuzz_parity.py (random read lengths, error rates, N density, allele
ranges, scorings, --useclippedreads, adversarial units) with more samples per round.  Prints one JSON line.

usage: python tools/fuzz_selfcheck.py [rounds] [seed]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def campaign(rounds=20, seed=1):
    """Runs the campaign and returns its summary (tests/test_fuzz_gpu.py runs a fixed-seed slice of it)."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from fuzz_parity import draw_round
    from oracle import pyoracle as po          # (rc() of the adversarial reads only)
    from tredparse_amd import _lib, synth
    loci = synth.load_loci()
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    n_reads = n_bad = 0
    tags = np.zeros(6, np.int64)
    by_scoring = {}
    t0 = time.time()
    for k in range(rounds):
        b, reads, unit_read_off, unit_ladder, clip, scoring, readlen = draw_round(rng, loci, synth, po, samples=(4, 16))
        ctx.set_ladders(b.ladders)
        n = len(reads)
        packed, woff, rlen = _lib.pack_reads(reads)
        params = _lib.SwParams(scoring[0], scoring[1], scoring[2], scoring[3], 9, int(clip), readlen, 0)
        tag = np.zeros(n, np.uint8); h = np.zeros(n, np.int16); sc = np.zeros(n, np.int16)
        ctx.sw_classify(_lib.MEM_HOST, packed, woff, rlen, n, unit_read_off, unit_ladder, len(unit_ladder), params, tag, h, sc)
        nt = max(2 * l[3] for l in b.ladders)
        dump = np.zeros((n, nt, 6), np.int16)
        t2 = np.zeros(n, np.uint8); h2 = np.zeros(n, np.int16); s2 = np.zeros(n, np.int16)
        ctx.sw_classify(_lib.MEM_HOST, packed, woff, rlen, n, unit_read_off, unit_ladder, len(unit_ladder), params,
                        t2, h2, s2, dump, nt)
        kk = np.arange(nt)
        units, strand = kk // 2 + 1, kk % 2
        score, dtag = dump[:, :, 0].astype(np.int64), dump[:, :, 5].astype(np.int64)
        key = np.where(dtag != _lib.TAG_NONE, (score << 11) | ((511 - units)[None, :] << 1) | (1 - strand)[None, :], -1)
        at = key.argmax(1)
        rows = np.arange(n)
        none = key[rows, at] < 0
        want = np.stack([np.where(none, _lib.TAG_NONE, dtag[rows, at]), np.where(none, 0, units[at]),
                         np.where(none, 0, score[rows, at])], 1)
        too_long = np.asarray(rlen) > 16 * 20          # flagged, not aligned (TREDGPU_TAG_INVALID)
        got = np.stack([tag, h, sc], 1).astype(np.int64)
        bad = np.nonzero(((got != want).any(1) | (tag != t2) | (h != h2) | (sc != s2)) & ~too_long)[0]
        n_reads += n
        n_bad += len(bad)
        tags += np.bincount(np.minimum(tag, 5), minlength=6)[:6]
        by_scoring[str(scoring)] = by_scoring.get(str(scoring), 0) + n
        if len(bad):
            print("MISMATCH round", k, "readlen", readlen, "clip", clip, "scoring", scoring, bad[:5], got[bad[:5]].tolist(),
                  want[bad[:5]].tolist(), file=sys.stderr)
    ctx.close()
    return {"reads": int(n_reads), "mismatches": int(n_bad), "rounds": rounds, "seed": seed,
            "tags_none_full_pref_post_rept_hang": [int(x) for x in tags], "reads_by_scoring": by_scoring,
            "seconds": round(time.time() - t0, 1)}


def main():
    res = campaign(*[int(x) for x in sys.argv[1:3]])
    from tredparse_amd import _lib
    res["library"] = _lib.version()
    print(json.dumps(res))
    return 1 if res["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
