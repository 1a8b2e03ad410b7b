#!/usr/bin/env python3
"""A live check of this repository's BAM layer against pysam itself -- for a maintainer who HAS pysam.

The reference reads BAMs through pysam (tredparse/bam_parser.py:22):
    samfile.fetch(chr, start, end)                         bam_parser.py:206, 226, 333
    sum(c.n for c in samfile.pileup(chr, start, end))      bam_parser.py:404-407
pysam, samtools and htslib are absent from the build image, so the "reference run()" goldens under tests/golden were
generated with tredparse_amd.bamio.PyAlignmentFile standing in for pysam (tools/gen_golden.py): the reference's selection
LOGIC is real there, the fetch / pileup primitive under it is this repository's on both sides of the comparison
("parity unpinned").  This script closes that gap wherever pysam can be imported: for every BAM given
(default: tests/golden/bam/*.bam) and every locus of the site table it compares, region by region,

    fetch    the records of bamio.PyAlignmentFile.fetch and of bamio.NativeAlignmentFile.fetch with pysam's, in order:
             (query name, flag, reference_start, reference_end, next_reference_id, next_reference_start) --
             over the locus' window (repeat +-1000: BamParser.parse), the +-10 kb pair-length region (PEextractor) and
             every alternative region of the locus (the mate rescue);
    pileup   per window: sum(c.n for c in pysam pileup(chr, start, end)) against pileup_depth_sum (both layers), and the
             per-column n against the column sums this layer's rule implies (every column a kept read covers, reads with
             flag & (UNMAP | SECONDARY | QCFAIL | DUP) skipped: htslib's default mask, no truncation).

and prints the first difference of each kind (exit code 1), or "identical" (exit code 0).  Without pysam it says so and
exits 0 (`--require` makes that an error, exit code 2): tests/test_pin_pysam.py asserts the skip path.

usage: python tools/pin_pysam_live.py [--ref hg38] [--require] [--json out.json] [bam ...]
"""
import argparse
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SPAN = 1000
PE_REACH = 10 * SPAN
SKIP_FLAGS = 0x4 | 0x100 | 0x200 | 0x400


def import_pysam():
    try:
        import pysam
        return pysam
    except Exception:            # not installed, or installed against a missing htslib
        return None


def key_of(r):
    end = r.reference_end
    return (r.query_name, int(r.flag), int(r.reference_start), -1 if end is None else int(end), int(r.next_reference_id),
            int(r.next_reference_start))


def regions_of(locus):
    """[(kind, contig, start, end)] the reference queries for one locus (half-open, as handed to fetch / pileup)."""
    out = [("window", locus.chr, max(0, locus.repeat_start - SPAN), locus.repeat_end + SPAN),
           ("pairs", locus.chr, max(0, locus.repeat_start - PE_REACH), locus.repeat_end + PE_REACH)]
    out += [("alt", c, a, b) for c, a, b in locus.alt]
    return out


def column_sums(records, start, end):
    """{column: n} of this layer's pile-up rule over the records a fetch of [start, end) returned."""
    cols = {}
    for r in records:
        if r.flag & SKIP_FLAGS or r.reference_end is None:
            continue
        for c in range(r.reference_start, r.reference_end):
            cols[c] = cols.get(c, 0) + 1
    return cols


def compare_file(pysam, path, repo, names, log):
    """Differences found in one BAM: a list of dicts (empty: identical)."""
    from tredparse_amd import bamio
    diffs = []
    theirs = pysam.AlignmentFile(path, "rb")
    layers = [("PyAlignmentFile", bamio.PyAlignmentFile(path))]
    if bamio._native() is not None:
        layers.append(("NativeAlignmentFile", bamio.NativeAlignmentFile(path)))
    contigs = set(theirs.references)
    checked = 0
    for name in names:
        locus = repo[name]
        for kind, contig, start, end in regions_of(locus):
            c = contig if contig in contigs else (contig[3:] if contig.startswith("chr") and contig[3:] in contigs else None)
            if c is None:
                continue
            want = [key_of(r) for r in theirs.fetch(c, start, end)]
            checked += 1
            for lname, f in layers:
                got_recs = list(f.fetch(c, start, end))
                got = [key_of(r) for r in got_recs]
                if got != want:
                    k = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
                    diffs.append({"bam": path, "locus": name, "kind": "fetch/" + kind, "layer": lname, "region": [c, start, end],
                                  "n_ours": len(got), "n_pysam": len(want), "first_difference_at": k,
                                  "ours": got[k] if k < len(got) else None, "pysam": want[k] if k < len(want) else None})
                if kind != "window":
                    continue
                cols = {col.reference_pos: col.n for col in theirs.pileup(c, start, end)}
                total = sum(cols.values())
                ours_total = f.pileup_depth_sum(c, start, end)
                if ours_total != total:
                    diffs.append({"bam": path, "locus": name, "kind": "pileup/sum", "layer": lname, "region": [c, start, end],
                                  "ours": int(ours_total), "pysam": int(total)})
                mine = column_sums(got_recs, start, end)
                if mine != cols:
                    bad = sorted(k for k in set(mine) | set(cols) if mine.get(k, 0) != cols.get(k, 0))
                    diffs.append({"bam": path, "locus": name, "kind": "pileup/column", "layer": lname, "region": [c, start, end],
                                  "columns_differing": len(bad), "first_column": bad[0], "ours": mine.get(bad[0], 0),
                                  "pysam": cols.get(bad[0], 0)})
    for _, f in layers:
        f.close()
    theirs.close()
    log("{}: {} regions of {} loci checked against pysam {}: {}".format(path, checked, len(names), getattr(pysam, "__version__", "?"),
                                                                     "identical" if not diffs else "{} differences".format(len(diffs))))
    return diffs


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("bams", nargs="*", help="BAM files (default: tests/golden/bam/*.bam)")
    ap.add_argument("--ref", default="hg38", help="genome build of the site table")
    ap.add_argument("--require", action="store_true", help="exit 2 when pysam cannot be imported (default: skip, exit 0)")
    ap.add_argument("--json", help="write the record here")
    args = ap.parse_args(argv)
    pysam = import_pysam()
    rec = {"pysam": getattr(pysam, "__version__", None), "files": [], "differences": []}
    if pysam is None:
        print("pin_pysam_live: pysam cannot be imported here -- skipped (run this where pysam is installed; the goldens' "
              "fetch / pileup primitive stays this repository's own until then)")
        rec["skipped"] = True
        if args.json:
            with open(args.json, "w") as fp:
                json.dump(rec, fp, indent=1)
        return 2 if args.require else 0
    from tredparse_amd.meta import TREDsRepo
    repo = TREDsRepo(args.ref)
    names = sorted(repo.names)
    bams = args.bams or sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "bam", "*.bam")))
    for path in bams:
        rec["files"].append(path)
        rec["differences"] += compare_file(pysam, path, repo, names, print)
    for d in rec["differences"][:20]:
        print("DIFFERENCE", json.dumps(d))
    print("pin_pysam_live: {}".format("identical" if not rec["differences"] else "{} differences (first 20 above)".format(len(rec["differences"]))))
    if args.json:
        with open(args.json, "w") as fp:
            json.dump(rec, fp, indent=1)
    return 1 if rec["differences"] else 0


if __name__ == "__main__":
    sys.exit(main())
