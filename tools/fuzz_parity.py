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


import os
# TREDGPU_FUZZ_READLENS=330,400,460: the campaign at other read lengths (e.g. the 32-rows-per-lane instantiation's)
LONG_READS = [int(x) for x in os.environ.get("TREDGPU_FUZZ_READLENS", "").split(",") if x]


def draw_round(rng, loci, synth, po, samples=(1, 4)):
    """One random batch: (batch, reads, unit_read_off, unit_ladder, clip, scoring, readlen)."""
    readlen = int(rng.choice(LONG_READS if LONG_READS else [36, 75, 100, 125, 150, 150, 150, 250, 300]))
    p = synth.SynthParams(coverage=float(rng.choice([10, 30, 60])), readlen=readlen,
                          sub=float(rng.choice([0.0, 0.01, 0.03])), indel=float(rng.choice([0.0, 0.001, 0.01])),
                          nrate=float(rng.choice([0.0, 0.005, 0.05])), min_units=int(rng.integers(1, 8)),
                          max_units=int(rng.integers(20, 70)), expanded_max=int(rng.choice([0, 120, 200])),
                          expanded_frac=0.3)
    sel = [l for l in loci if 36 + len(l["repeat"]) * -(-readlen // len(l["repeat"])) <= 511]
    sel = [sel[i] for i in rng.permutation(len(sel))[:int(rng.integers(4, len(sel) + 1))]]
    b = synth.build_batch(int(rng.integers(1 << 30)), sel, int(rng.integers(samples[0], samples[1])), p, workers=8)
    reads = [synth.decode(r) for r in b.codes]
    uro, ulad = list(b.unit_read_off), list(b.unit_ladder)
    # adversarial units: pure repeats in any phase / strand, exact template windows, chimeras, N runs, junk
    for _ in range(4):
        lad = int(rng.integers(len(b.ladders)))
        prefix, rep, suffix, mu = b.ladders[lad]
        rep_c = rep.replace("N", "C")
        extra = []
        for _ in range(25):
            kind = int(rng.integers(9))
            u = int(rng.integers(1, mu + 1))
            tmpl = prefix + rep_c * u + suffix
            if kind == 0:
                ph = int(rng.integers(len(rep_c)))
                r = (rep_c * (readlen // len(rep_c) + 2))[ph:ph + readlen]
            elif kind == 1:
                o = int(rng.integers(0, max(1, len(tmpl) - 10)))
                r = tmpl[o:o + readlen]
            elif kind == 2:
                other = b.ladders[int(rng.integers(len(b.ladders)))]
                r = (tmpl[:readlen // 2] + other[0] + other[1].replace("N", "A") * 4 + other[2])[:readlen]
            elif kind == 3:
                r = "".join("ACGT"[i] for i in rng.integers(0, 4, readlen))
            elif kind == 4:
                cut = int(rng.integers(1, readlen))
                r = (tmpl * 3)[:cut] + "N" * (readlen - cut)
            elif kind == 5:
                r = (rep_c * u)[:readlen // 2] + "ACGT"[int(rng.integers(4))] + (rep_c * (mu + 2))[:readlen // 2]
            elif kind == 6:
                r = "N" * int(rng.integers(1, readlen + 1))
            else:
                # indels at the repeat / suffix junction (where an alignment leaves the trunk for the suffix
                # continuation vectors): deletion, insertion, or both back to back
                j = len(prefix) + len(rep_c) * u + int(rng.integers(-4, 5))
                a, c = tmpl[:max(j, 0)], tmpl[max(j, 0):]
                mode = int(rng.integers(3))
                if mode != 1:
                    c = c[int(rng.integers(1, 7)):]
                if mode != 0:
                    a = a + "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(1, 7))))
                r = (a + c)[max(0, len(a) - readlen + int(rng.integers(4, 30))):]
            if rng.random() < 0.5:
                r = po.rc(r)
            if r:
                extra.append(r[:readlen])
        reads += extra
        uro.append(len(reads))
        ulad.append(lad)
    n = len(reads)
    unit_read_off, unit_ladder = np.asarray(uro, np.int32), np.asarray(ulad, np.int32)
    clip = bool(rng.random() < 0.3)
    scoring = [(1, 5, 7, 2)] * 3 + [(2, 3, 5, 2), (1, 4, 6, 1), (1, 1, 2, 1), (3, 5, 7, 2), (1, 9, 12, 3),
                                      (2, 6, 3, 1), (1, 3, 2, 2), (3, 2, 4, 1)]
    scoring = scoring[int(rng.integers(len(scoring)))]
    if clip:   # --useclippedreads: ragged lengths exercise the per-read REPT cut-off (bam_parser.py:154-155)
        reads = [r[int(rng.integers(0, max(1, len(r) // 3))):] for r in reads]
    return b, reads, unit_read_off, unit_ladder, clip, scoring, readlen


def campaign(rounds=12, seed=1):
    """Runs the campaign and returns its summary (tests/test_fuzz_gpu.py runs a fixed-seed slice of it)."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from oracle import pyoracle as po
    from tredparse_amd import _lib, synth
    loci = synth.load_loci()
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    n_reads = n_bad = n_pairs = n_bad_pairs = n_ref_faults = 0
    tags = np.zeros(6, np.int64)
    t0 = time.time()
    for k in range(rounds):
        b, reads, unit_read_off, unit_ladder, clip, scoring, readlen = draw_round(rng, loci, synth, po)
        ctx.set_ladders(b.ladders)
        n = len(reads)
        packed, woff, rlen = _lib.pack_reads(reads)
        tag = np.zeros(n, np.uint8); h = np.zeros(n, np.int16); sc = np.zeros(n, np.int16)
        ctx.sw_classify(_lib.MEM_HOST, packed, woff, rlen, n, unit_read_off, unit_ladder, len(unit_ladder),
                        _lib.SwParams(scoring[0], scoring[1], scoring[2], scoring[3], 9, int(clip), readlen, 0), tag, h, sc)
        cls = po.ref_classify(reads, np.repeat(unit_ladder, np.diff(unit_read_off)), po.LocusSet(b.ladders), clip=clip,
                              scoring=scoring, threads=0)
        # (reads on which the reference's own CIGAR pass faulted -- oracle/ref_driver.c -- carry tag -1: counted, not compared)
        judged = cls[:, 0] >= 0
        n_ref_faults += int((~judged).sum())
        bad = np.nonzero(judged & ((tag != cls[:, 0]) | (h != cls[:, 1]) | (sc != cls[:, 2])))[0]
        n_reads += n
        n_bad += len(bad)
        # per-template results (score, ref_begin, ref_end, read_begin, read_end) of the first units, dump path
        # (the adversarial units at the end of the batch)
        g0 = len(unit_ladder) - 4
        r0 = int(unit_read_off[g0])
        sub_reads = reads[r0:]
        m = len(sub_reads)
        if m:
            ls = po.LocusSet(b.ladders)
            nt = max(2 * l[3] for l in b.ladders)
            dump = np.zeros((m, nt, 6), np.int16)
            packed2, woff2, rlen2 = _lib.pack_reads(sub_reads)
            sub_off = (unit_read_off[g0:] - r0).astype(np.int32)
            t2 = np.zeros(m, np.uint8); h2 = np.zeros(m, np.int16); s2 = np.zeros(m, np.int16)
            ctx.sw_classify(_lib.MEM_HOST, packed2, woff2, rlen2, m, sub_off, unit_ladder[g0:].copy(), 4,
                            _lib.SwParams(scoring[0], scoring[1], scoring[2], scoring[3], 9, int(clip), readlen, 0),
                            t2, h2, s2, dump, nt)
            pr, pt, where = [], [], []
            for r in range(m):
                lad = int(unit_ladder[g0 + np.searchsorted(sub_off, r, side="right") - 1])
                for j, t in enumerate(range(ls.lad_off[lad], ls.lad_off[lad + 1])):
                    pr.append(r); pt.append(t); where.append((r, j))
            want = po.ref_sw_pairs(sub_reads, ls.templates, pr, pt, scoring=scoring, threads=0)
            got = np.array([dump[r, j, :5] for r, j in where], np.int32)
            faulted = want[:, 0] == po.REF_CRASHED
            n_ref_faults += int(faulted.sum())
            got[faulted] = want[faulted]
            nb = int((got != want).any(axis=1).sum())
            n_pairs += len(pr)
            n_bad_pairs += nb
            if nb:
                kk = int(np.nonzero((got != want).any(axis=1))[0][0])
                print("PAIR MISMATCH round", k, "readlen", readlen, "scoring", scoring, where[kk], got[kk], want[kk], file=sys.stderr)
            if not (np.array_equal(t2, tag[r0:]) and np.array_equal(h2, h[r0:]) and np.array_equal(s2, sc[r0:])):
                n_bad += 1
                print("DUMP/NON-DUMP PATH DISAGREE round", k, file=sys.stderr)
        tags += np.bincount(tag, minlength=6)[:6]
        if len(bad):
            print("MISMATCH round", k, "readlen", readlen, "clip", clip, "scoring", scoring, bad[:5], tag[bad[:5]], h[bad[:5]], sc[bad[:5]], cls[bad[:5]],
                  file=sys.stderr)
    ctx.close()
    return {"reads": int(n_reads), "mismatches": int(n_bad), "template_pairs": int(n_pairs),
            "pair_mismatches": int(n_bad_pairs), "reference_faults": int(n_ref_faults), "rounds": rounds, "seed": seed,
            "tags_none_full_pref_post_rept_hang": [int(x) for x in tags], "seconds": round(time.time() - t0, 1)}


def main():
    res = campaign(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    from tredparse_amd import _lib
    res["library"] = _lib.version()
    print(json.dumps(res))
    return 1 if (res["mismatches"] or res["pair_mismatches"]) else 0


if __name__ == "__main__":
    sys.exit(main())
