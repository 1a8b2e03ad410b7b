#!/usr/bin/env python3
"""Randomised parity campaign, GPU vs the reference's own compiled ssw.c (oracle/_ref): every read of randomly
drawn batches (read length, coverage, error rates, N density, allele ranges incl. expansions, all 32 loci)
through the pruned kernel path against (tag, h, score) computed alignment by alignment by the reference.
Not part of the test suite (minutes of CPU per million reads); prints one JSON line.  oracle/ is only the
checker here, as in tests/.

usage: python tools/fuzz_parity.py [rounds] [seed]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from oracle import pyoracle as po
    from tredparse_amd import _lib, synth
    loci = synth.load_loci()
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    n_reads = n_bad = n_pairs = n_bad_pairs = 0
    tags = np.zeros(6, np.int64)
    t0 = time.time()
    for k in range(rounds):
        readlen = int(rng.choice([36, 75, 100, 125, 150, 150, 150, 250]))
        p = synth.SynthParams(coverage=float(rng.choice([10, 30, 60])), readlen=readlen,
                              sub=float(rng.choice([0.0, 0.01, 0.03])), indel=float(rng.choice([0.0, 0.001, 0.01])),
                              nrate=float(rng.choice([0.0, 0.005, 0.05])), min_units=int(rng.integers(1, 8)),
                              max_units=int(rng.integers(20, 70)), expanded_max=int(rng.choice([0, 120, 200])),
                              expanded_frac=0.3)
        sel = [l for l in loci if 36 + len(l["repeat"]) * -(-readlen // len(l["repeat"])) <= 511]
        sel = [sel[i] for i in rng.permutation(len(sel))[:int(rng.integers(4, len(sel) + 1))]]
        b = synth.build_batch(int(rng.integers(1 << 30)), sel, int(rng.integers(1, 4)), p, workers=8)
        ctx.set_ladders(b.ladders)
        n = b.n_reads
        reads = [synth.decode(r) for r in b.codes]
        clip = bool(rng.random() < 0.3)
        scoring = [(1, 5, 7, 2)] * 3 + [(2, 3, 5, 2), (1, 4, 6, 1), (1, 1, 2, 1), (3, 5, 7, 2), (1, 9, 12, 3)]
        scoring = scoring[int(rng.integers(len(scoring)))]
        if clip:   # --useclippedreads: ragged lengths exercise the per-read REPT cut-off (bam_parser.py:154-155)
            reads = [r[int(rng.integers(0, max(1, len(r) // 3))):] for r in reads]
        packed, woff, rlen = _lib.pack_reads(reads)
        tag = np.zeros(n, np.uint8); h = np.zeros(n, np.int16); sc = np.zeros(n, np.int16)
        ctx.sw_classify(_lib.MEM_HOST, packed, woff, rlen, n, b.unit_read_off, b.unit_ladder, b.n_units,
                        _lib.SwParams(scoring[0], scoring[1], scoring[2], scoring[3], 9, int(clip), readlen, 0), tag, h, sc)
        cls = po.ref_classify(reads, np.repeat(b.unit_ladder, np.diff(b.unit_read_off)), po.LocusSet(b.ladders), clip=clip,
                              scoring=scoring, threads=0)
        bad = np.nonzero((tag != cls[:, 0]) | (h != cls[:, 1]) | (sc != cls[:, 2]))[0]
        n_reads += n
        n_bad += len(bad)
        # per-template results (score, ref_begin, ref_end, read_begin, read_end) of the first units, dump path
        gsub = 1
        while gsub < b.n_units and b.unit_read_off[gsub] < 120:
            gsub += 1
        m = int(b.unit_read_off[gsub])
        if m:
            ls = po.LocusSet(b.ladders)
            nt = max(2 * l[3] for l in b.ladders)
            dump = np.zeros((m, nt, 6), np.int16)
            packed2, woff2, rlen2 = _lib.pack_reads(reads[:m])
            t2 = np.zeros(m, np.uint8); h2 = np.zeros(m, np.int16); s2 = np.zeros(m, np.int16)
            ctx.sw_classify(_lib.MEM_HOST, packed2, woff2, rlen2, m, b.unit_read_off[:gsub + 1].copy(),
                            b.unit_ladder[:gsub].copy(), gsub,
                            _lib.SwParams(scoring[0], scoring[1], scoring[2], scoring[3], 9, int(clip), readlen, 0),
                            t2, h2, s2, dump, nt)
            pr, pt, where = [], [], []
            for r in range(m):
                lad = int(b.unit_ladder[np.searchsorted(b.unit_read_off, r, side="right") - 1])
                for j, t in enumerate(range(ls.lad_off[lad], ls.lad_off[lad + 1])):
                    pr.append(r); pt.append(t); where.append((r, j))
            want = po.ref_sw_pairs(reads[:m], ls.templates, pr, pt, scoring=scoring, threads=0)
            got = np.array([dump[r, j, :5] for r, j in where], np.int32)
            nb = int((got != want).any(axis=1).sum())
            n_pairs += len(pr)
            n_bad_pairs += nb
            if nb:
                kk = int(np.nonzero((got != want).any(axis=1))[0][0])
                print("PAIR MISMATCH round", k, "readlen", readlen, "scoring", scoring, where[kk], got[kk], want[kk], file=sys.stderr)
            if not (np.array_equal(t2, tag[:m]) and np.array_equal(h2, h[:m]) and np.array_equal(s2, sc[:m])):
                n_bad += 1
                print("DUMP/NON-DUMP PATH DISAGREE round", k, file=sys.stderr)
        tags += np.bincount(tag, minlength=6)[:6]
        if len(bad):
            print("MISMATCH round", k, "readlen", readlen, "clip", clip, "scoring", scoring, bad[:5], tag[bad[:5]], h[bad[:5]], sc[bad[:5]], cls[bad[:5]],
                  file=sys.stderr)
    print(json.dumps({"reads": int(n_reads), "mismatches": int(n_bad), "template_pairs": int(n_pairs), "pair_mismatches": int(n_bad_pairs), "rounds": rounds, "seed": seed,
                      "tags_none_full_pref_post_rept_hang": [int(x) for x in tags], "seconds": round(time.time() - t0, 1)}))
    return 1 if (n_bad or n_bad_pairs) else 0


if __name__ == "__main__":
    sys.exit(main())
