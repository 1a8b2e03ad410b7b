#!/usr/bin/env python3
"""Which input of the likelihood model explains README's DM1 call of 5|62 for tests/t002.bam, where v0.7.8's code run
over this repository's BAM layer gives 5|66?  (/root/reference/README.md:76-86.)

BUILD-CONTAINER ONLY (reads /root/reference through tools/refshim.py; nothing here travels to the GPU box).

README's FR / PR / RR strings for t002 are reproduced exactly, so the per-read evidence is the same; what pysam
supplies beyond it are (a) the local depth -- `sum(c.n for c in sam.pileup(chr, start, end)) / (end - start + 1)`,
bam_parser.py:404-411 -- and (b) the record set PEextractor pairs up (bam_parser.py:316-369).  This script runs the
REFERENCE's own IntegratedCaller on t002 / DM1 and sweeps those inputs:

  1. depth from 10 to 100 (the window-truncated pileup gives ~29, the all-columns pileup 47.1), with the paired-end
     term as is, and with the paired-end model switched off;
  2. paired-end list variants: duplicates kept, secondary / supplementary records kept or dropped, pair length without
     the soft-clip extension, the spanning rule with and without the 9-base margins, pairs limited to the repeat window,
     the `a, b = reads[:2]` choice made by name order instead of file order, the 1 000-base cut moved;
  3. which records the main window fetch hands to the aligner: with and without the unmapped reads placed at their
     mates' position (a BAI fetch returns them; whether pysam 0.9.1 did depends on its multiple_iterators / until_eof
     defaults), which changes RR.

It prints one table per sweep and, at the end, every combination that yields allele 2 = 62 -- or the statement that
none does.  The conclusion is recorded in DESIGN.md 2 and pinned by tests/golden/pin_pysam.json.

usage: python tools/pin_pysam.py [--json tests/golden/pin_pysam.json]
"""
import argparse
import json
import logging
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import refshim  # noqa: E402


class _Col(object):
    __slots__ = ("n",)

    def __init__(self, n):
        self.n = n


def make_pysam(pileup_mode="all", drop_unmapped_in_fetch=False):
    """The pysam stand-in of tools/gen_golden.py with its two free choices exposed."""
    from tredparse_amd import bamio
    mod = types.ModuleType("pysam")

    class AlignmentFile(bamio.PyAlignmentFile):
        def fetch(self, *a, **k):
            for r in bamio.PyAlignmentFile.fetch(self, *a, **k):
                if drop_unmapped_in_fetch and a and r.is_unmapped:
                    continue
                yield r

        def pileup(self, chrom, start, end):
            cov = {}
            for r in bamio.PyAlignmentFile.fetch(self, chrom, start, end):
                if r.flag & bamio.FUNMAP:
                    continue
                if pileup_mode != "nofilter" and r.flag & (bamio.FSECONDARY | bamio.FQCFAIL | bamio.FDUP):
                    continue
                if r.reference_end is None:
                    continue
                for p in range(r.pos, r.reference_end):
                    if pileup_mode == "truncate" and not (start <= p < end):
                        continue
                    cov[p] = cov.get(p, 0) + 1
            for p in sorted(cov):
                yield _Col(cov[p])
    mod.AlignmentFile = AlignmentFile
    return mod


def load(pysam_mod):
    for name in [m for m in sys.modules if m == "tredparse" or m.startswith("tredparse.") or m in
                 ("ssw", "ssw_wrap", "utils", "bam_parser", "models", "meta", "pysam")]:
        del sys.modules[name]
    return refshim.load_reference(pysam_standin=pysam_mod, full=True)


def caller_for(ref, repo, bam, tred, depth, readlen=None):
    """The reference's own objects for one unit, as tred.py:225-249 builds them."""
    lg = logging.getLogger("pin")
    if readlen is None:
        readlen = ref.bam_parser.BamReadLen(bam, lg).readlen
    ip = ref.utils.InputParams(bam=bam, READLEN=readlen, tredName=tred, repo=repo, maxinsert=300, fullsearch=False,
                               gender="Unknown", depth=depth, clip=False, alts=True, repeatpairs=True, log="ERROR")
    bp = ref.bam_parser.BamParser(ip)
    bp.parse()
    return bp, ref.models.IntegratedCaller(bp, maxinsert=300, fullsearch=False)


def call(caller):
    caller.call()
    return [int(a) for a in caller.alleles]


def pe_variants(ref, bp):
    """(name, global_lens, target_lens) under different readings of PEextractor's record set."""
    from tredparse_amd import bamio
    SPAN, FLANK, ELONG = ref.bam_parser.SPAN, ref.bam_parser.FLANKMATCH, ref.bam_parser.DNAPE_ELONGATE
    f = bamio.PyAlignmentFile(bp.bam)
    start, end = bp.startRepeat, bp.endRepeat
    recs = list(f.fetch(bp.chr, max(start - ELONG, 0), end + ELONG))

    def tlen_of(a, b, clips=True):
        s, e = a.reference_start, b.reference_end
        if clips and a.query_alignment_start > 0:
            s -= a.query_alignment_start
        if clips and b.query_alignment_end < b.query_length:
            e += b.query_length - b.query_alignment_end
        return e - s

    def lists(keep_dup=False, keep_secondary=True, clips=True, margin=FLANK, cut=SPAN, by_name=False, window=None):
        cache = {}
        for x in recs:
            if not x.is_paired or x.is_unmapped:
                continue
            if x.is_duplicate and not keep_dup:
                continue
            if not keep_secondary and (x.flag & (bamio.FSECONDARY | 0x800)):
                continue
            if window is not None and not (x.reference_end > start - window and x.reference_start < end + window):
                continue
            cache.setdefault(x.query_name, []).append(x)
        g, t = [], []
        names = sorted(cache) if by_name else list(cache)
        for n in names:
            reads = cache[n]
            if len(reads) < 2:
                continue
            a, b = reads[:2]
            if not ((not a.is_reverse) and b.is_reverse):
                continue
            tl = tlen_of(a, b, clips)
            if tl >= cut:
                continue
            if a.reference_start < start - margin and b.reference_end > end + margin:
                t.append(tl)
            else:
                g.append(tl)
        return g, t

    out = [("as v0.7.8 reads it", ) + lists()]
    out.append(("duplicates kept", ) + lists(keep_dup=True))
    out.append(("secondary / supplementary dropped", ) + lists(keep_secondary=False))
    out.append(("pair length without the clip extension", ) + lists(clips=False))
    out.append(("spanning rule without the 9-base margins", ) + lists(margin=0))
    out.append(("spanning rule with 18-base margins", ) + lists(margin=2 * FLANK))
    out.append(("pairs cut at 500 instead of 1000 bases", ) + lists(cut=500))
    out.append(("pairs cut at 2000 bases", ) + lists(cut=2000))
    out.append(("pairs within +-1 kb of the repeat only", ) + lists(window=1000))
    out.append(("pairs within +-2 kb of the repeat only", ) + lists(window=2000))
    f.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    if args.json:
        args.json = os.path.abspath(args.json)
    logging.disable(logging.CRITICAL)
    bam = os.path.join(refshim.REF, "tests", "t002.bam")
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    record = {"generator": "tools/pin_pysam.py (the reference's IntegratedCaller on tests/t002.bam, DM1)", "readme": [5, 62]}
    hits = []

    ref = load(make_pysam("all"))
    repo = ref.meta.TREDsRepo(ref="hg38", toy=False, sites=os.path.join(tmp, "sites"))
    xt = repo["DM1"]
    bd = ref.bam_parser.BamDepth(bam, "hg38", logging.getLogger("pin"))
    ws, we = max(0, xt.repeat_start - ref.bam_parser.SPAN), xt.repeat_end + ref.bam_parser.SPAN
    depth_all = bd.region_depth(xt.chr, ws, we)
    ref_t = load(make_pysam("truncate"))
    depth_trunc = ref_t.bam_parser.BamDepth(bam, "hg38", logging.getLogger("pin")).region_depth(xt.chr, ws, we)
    ref_n = load(make_pysam("nofilter"))
    depth_nofilter = ref_n.bam_parser.BamDepth(bam, "hg38", logging.getLogger("pin")).region_depth(xt.chr, ws, we)
    ref = load(make_pysam("all"))
    repo = ref.meta.TREDsRepo(ref="hg38", toy=False, sites=os.path.join(tmp, "sites"))
    print("depth of {}:{}-{}: {:.3f} (every column of the overlapping reads / window: htslib's default pileup), {:.3f} (columns "
          "inside the window only: truncate=True), {:.3f} (all columns, duplicates / secondary / QC-fail records counted too: "
          "the most any pileup over this file can report)".format(xt.chr, ws, we, depth_all, depth_trunc, depth_nofilter))
    record["depth_all_columns"], record["depth_truncated"], record["depth_no_filter"] = depth_all, depth_trunc, depth_nofilter

    # ---- 1. depth sweep -----------------------------------------------------------------------------------------
    bp, c0 = caller_for(ref, repo, bam, "DM1", depth_all)
    base = call(c0)
    counts = {k: dict(v) for k, v in bp.counts.items()}
    print("v0.7.8 as this repository runs it: alleles {}, RR {}, PEDP {}".format(base, counts.get("REPT"), c0.PEDP))
    record["baseline"] = {"alleles": base, "rept": int(bp.rept), "pedp": int(c0.PEDP)}
    sweep = []
    for with_pe in (True, False):
        row = []
        for d in list(range(10, 101, 2)) + [depth_all, depth_trunc]:
            _, c = caller_for(ref, repo, bam, "DM1", float(d))
            if not with_pe:
                c.pemodel = None
            a = call(c)
            row.append((float(d), a))
            if a[1] == 62:
                hits.append({"sweep": "depth", "depth": float(d), "paired_end_term": with_pe, "alleles": a})
        sweep.append({"paired_end_term": with_pe, "calls": row})
        xs = sorted(row)
        print("\ndepth sweep, paired-end term {}:".format("on" if with_pe else "OFF"))
        print("  " + "  ".join("{:g}:{}|{}".format(d, a[0], a[1]) for d, a in xs))
    record["depth_sweep"] = sweep

    # ---- 2. paired-end list variants (at both depths) -------------------------------------------------------------
    print("\npaired-end list variants:")
    pe_rows = []
    for name, g, t in pe_variants(ref, bp):
        for d in (depth_all, depth_trunc):
            _, c = caller_for(ref, repo, bam, "DM1", d)
            pe = types.SimpleNamespace(global_lens=g, target_lens=t, ref=bp.referenceLen,
                                       MINPE=bp.endRepeat - bp.startRepeat + 2 * ref.bam_parser.FLANKMATCH + 2)
            ok = len(g) >= 100 and len(t) >= ref.models.MIN_SPANNING_PAIRS
            try:
                c.pemodel = ref.models.PEMaxLikModel(pe) if ok else None
                a = call(c)
            except Exception as e:
                a = [type(e).__name__]
            pe_rows.append({"variant": name, "depth": d, "n_global": len(g), "n_target": len(t), "alleles": a})
            print("  {:45s} depth {:6.2f}  global {:5d} spanning {:3d}  -> {}".format(name, d, len(g), len(t), a))
            if len(a) == 2 and a[1] == 62:
                hits.append({"sweep": "paired-end list", "variant": name, "depth": d, "alleles": a})
    record["pe_variants"] = pe_rows

    # ---- 3. the window fetch without the unmapped reads ------------------------------------------------------------
    ref_u = load(make_pysam("all", drop_unmapped_in_fetch=True))
    repo_u = ref_u.meta.TREDsRepo(ref="hg38", toy=False, sites=os.path.join(tmp, "sites"))
    print("\nwindow fetch WITHOUT the unmapped reads placed at their mates' position:")
    rows = []
    for d in (depth_all, depth_trunc):
        bp_u, c = caller_for(ref_u, repo_u, bam, "DM1", d)
        a = call(c)
        rr = dict(bp_u.counts.get("REPT", {}))
        rows.append({"depth": d, "rept": rr, "alleles": a})
        print("  depth {:6.2f}: RR {} -> {}".format(d, rr, a))
        if a[1] == 62:
            hits.append({"sweep": "no unmapped reads in fetch", "depth": d, "alleles": a, "rept": rr})
    record["no_unmapped_in_fetch"] = rows

    print("\ncombinations that give allele 2 = 62: {}".format(hits if hits else "NONE"))
    record["hits_62"] = hits
    reach = [h for h in hits if h["sweep"] != "depth" or h["depth"] <= depth_nofilter]
    record["reachable_62"] = reach
    print("of these within reach of a pileup over this file (depth <= {:.1f}): {}".format(depth_nofilter, reach if reach else "NONE"))
    # the inputs of the unit, so that the sweep can be replayed against the oracle and the kernels (tests/test_pin_pysam.py)
    record["unit"] = {"readlen": int(c0.readlen), "period": int(c0.period), "ploidy": int(c0.ploidy), "rept": int(bp.rept),
                      "full": {str(k): int(v) for k, v in bp.counts["FULL"].items()},
                      "pref": {str(k): int(v) for k, v in bp.counts["PREF"].items()},
                      "ref_len": int(bp.referenceLen), "minpe": int(bp.endRepeat - bp.startRepeat + 2 * ref.bam_parser.FLANKMATCH + 2)}
    pe0 = ref.bam_parser.PEextractor(bp)
    record["unit"]["global_lens"], record["unit"]["target_lens"] = [int(x) for x in pe0.global_lens], [int(x) for x in pe0.target_lens]
    if args.json:
        with open(args.json, "w") as fp:
            json.dump(record, fp, indent=1, default=lambda o: o.item() if isinstance(o, np.generic) else str(o))
        print("wrote", args.json)


if __name__ == "__main__":
    main()
