#!/usr/bin/env python3
"""Population report over many tred.py outputs: who is at risk at which locus, and the allele spectrum.

Inputs: the per-sample `<key>.json` files (or `<key>.tred.vcf.gz` files) tred.py writes, or a TSV written by an
earlier run.  Outputs, all next to --tsv:

    <tsv>               one row per sample: SampleKey, inferredGender, and per locus `calls` ("a|b") and `label`
                        (+ any --columns), columns sorted by name
    <tsv>.cases.txt     per locus with at-risk samples: a header line, the locus' parameters, and a table of the
                        cases with their read evidence
    <tsv>.details.txt   tab-separated read-depth evidence (FDP, PDP, RDP, PEDP) of every case
    <tsv>.report.txt    one row per locus: cut-offs, number of pre-risk / risk / carrier samples, allele counts

Same flags and file formats as tredparse's tredreport.py (tredparse/tredreport.py:198-302); implemented on plain
rows (dicts) rather than data frames, VCF input parsed directly (no PyVCF).  Host-side only.
"""
import argparse
import csv
import gzip
import json
import os
import re
import sys
from collections import Counter

from . import __version__
from .meta import BUILDS, TREDsRepo

DETAIL_COLUMNS = ("Locus", "Inheritance", "SampleKey", "Sex", "Calls", "FullReads", "PartialReads", "RepeatReads",
                  "PairedReads")
REPORT_COLUMNS = ("abbreviation", "allele_freq", "cutoff_prerisk", "cutoff_risk", "inheritance", "motif", "n_carrier",
                  "n_prerisk", "n_risk", "title")


def left_truncate_text(values, maxcol=30):
    """Long strings keep their tail: '...' + the last maxcol-3 characters."""
    return [v if not isinstance(v, str) or len(v) <= maxcol else "..." + v[3 - maxcol:] for v in values]


def counts_to_af(counts):
    """Counter of allele sizes -> '{5:12,6:3}' (sorted by size; missing calls are not alleles)."""
    return "{" + ",".join("{}:{}".format(k, counts[k]) for k in sorted(k for k in counts if isinstance(k, int))) + "}"


# ---- loading ------------------------------------------------------------------------------------------------------
def _key_of(path):
    return os.path.basename(path).split(".")[0]


def read_json(path):
    with open(path) as fp:
        row = dict(json.load(fp)["tredCalls"])
    row["SampleKey"] = _key_of(path)        # the file name is the key (early versions left the field empty)
    return row


def read_vcf(path):
    """One sample's tred.vcf(.gz): per record the GB sizes, PP, FR, PR and LABEL of the sample column."""
    row = {"SampleKey": _key_of(path)}
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as fp:
        for line in fp:
            if line.startswith("#") or not line.strip():
                continue
            cells = line.rstrip("\n").split("\t")
            sample = dict(zip(cells[8].split(":"), cells[9].split(":")))
            locus = cells[2]
            a, b = sample["GB"].split("/")
            row[locus + ".1"], row[locus + ".2"] = int(a), int(b)
            row[locus + ".PP"] = float(sample["PP"])
            row[locus + ".FR"], row[locus + ".PR"], row[locus + ".label"] = sample["FR"], sample["PR"], sample["LABEL"]
    return row


def read_tsv(path):
    with open(path) as fp:
        return [dict(r) for r in csv.DictReader(fp, delimiter="\t")]


def load(files, cpus=1):
    reader = read_json if files[0].endswith(".json") else read_vcf
    if cpus > 1 and len(files) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(cpus, len(files))) as ex:
            return list(ex.map(reader, files))
    return [reader(f) for f in files]


# ---- the sample table -----------------------------------------------------------------------------------------------
def add_calls(rows, repo, has_sex):
    """Per locus and sample: integer allele sizes (`.1_`, `.2_`; -1 = missing; '.' for the second allele of a male
    at an X-linked locus) and the `calls` string."""
    def size(v):
        return -1 if v in (None, "") else int(float(v))

    for name in repo.names:
        if not any((name + ".1") in r for r in rows):
            continue
        hemizygous = has_sex and repo[name].is_xlinked
        for r in rows:
            a, b = size(r.get(name + ".1")), size(r.get(name + ".2"))
            if hemizygous and r.get("inferredGender") == "Male":
                b = "."
            r[name + ".1_"], r[name + ".2_"] = a, b
            r[name + ".calls"] = "{}|{}".format(a, b)


def _is_number(v):
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def _column_cells(values):
    """One TSV column as the reference's data frame prints it (tredreport.py:112-141: fillna(-1), to_csv): a column
    of whole numbers without gaps stays integer; numbers with a gap or a fraction are floats throughout (a gap
    reads -1.0); a column holding any text prints every entry as it is and gaps as -1."""
    present = [v for v in values if v is not None]
    if present and all(_is_number(v) for v in present) and \
            (len(present) < len(values) or any(isinstance(v, float) for v in present)):
        return [repr(float(-1 if v is None else v)) for v in values]
    return ["-1" if v is None else str(v) for v in values]


def write_tsv(rows, path, extra_columns, has_sex):
    lead = ["SampleKey"] + (["inferredGender"] if has_sex else [])
    wanted = ("calls", "label") + tuple(extra_columns)
    seen = set()
    for r in rows:
        seen.update(r)
    columns = lead + sorted(c for c in seen if c not in lead and c.rsplit(".", 1)[-1] in wanted and "." in c)
    cells = [_column_cells([r.get(c) for r in rows]) for c in columns]
    with open(path, "w", newline="") as fp:
        w = csv.writer(fp, delimiter="\t", lineterminator="\n")
        w.writerow(columns)
        for i in range(len(rows)):
            w.writerow([col[i] for col in cells])
    print("TSV output written to `{}` (# samples={})".format(path, len(rows)), file=sys.stderr)


def df_to_tsv(rows, tsvfile, extra_columns=(), jsonformat=True, ref="hg38"):
    """Table of samples -> TSV; returns the rows with the per-locus call columns added."""
    add_calls(rows, TREDsRepo(ref), jsonformat)
    write_tsv(rows, tsvfile, extra_columns, jsonformat)
    return rows


# ---- per-locus summary --------------------------------------------------------------------------------------------
def _num(v, default=-1.0):
    try:
        return float(v)
    except (TypeError, ValueError):
        return default


_PLAIN_NUMBER = re.compile(r"^\s*[+-]?[0-9]+\.[0-9]*$")


def _float_cells(values, digits=6):
    """A column of floats as a data frame's text form shows it: fixed notation with `digits` decimals, trailing
    zeros dropped as long as every entry ends in one (one decimal stays); scientific
    notation for the whole column once an entry would otherwise show as 0 (|x| < 10^-digits) or the column is both
    wide and holds |x| > 1e6."""
    def trim(cells):
        def plain(x):
            return _PLAIN_NUMBER.match(x) is not None
        while True:
            numbers = [x for x in cells if plain(x)]
            if not numbers or not all(x.endswith("0") for x in numbers):
                break
            cells = [x[:-1] if plain(x) else x for x in cells]
        return [x + "0" if plain(x) and x.endswith(".") else x for x in cells]

    def render(spec):
        return trim(["NaN" if v != v else spec.format(v) for v in values])
    cells = render("{:." + str(digits) + "f}")
    mags = [abs(v) for v in values if v == v]
    too_long = bool(cells) and max(len(x) for x in cells) > digits + 6
    if any(0 < m < 10 ** -digits for m in mags) or (too_long and any(m > 1e6 for m in mags)):
        cells = render("{:." + str(digits) + "e}")
    return cells


def _table(rows, columns):
    """Fixed-width text table the way the reference prints its case tables (tredreport.py:89-97, a data frame's
    to_string(index=False)): one header line, every cell right-aligned to its column's width, one blank between
    columns; a numeric column's header carries one leading blank, its values the float layout of _float_cells."""
    heads, cols = [], []
    for c in columns:
        values = [r.get(c, "") for r in rows]
        if values and all(_is_number(v) for v in values):
            heads.append(" " + c)
            if any(isinstance(v, float) for v in values):
                cols.append(_float_cells([float(v) for v in values]))
            else:
                cols.append([str(v) for v in values])
        else:
            heads.append(c)
            cols.append([str(v) for v in values])
    widths = [max([len(h)] + [len(x) for x in col]) for h, col in zip(heads, cols)]
    lines = [" ".join(h.rjust(w) for h, w in zip(heads, widths))]
    for i in range(len(rows)):
        lines.append(" ".join(col[i].rjust(w) for col, w in zip(cols, widths)))
    return "\n".join(lines)


def get_tred_summary(rows, name, repo, minPP=.5, casesfw=None, detailsfw=None):
    """(locus, n_prerisk, n_risk, n_carrier, allele spectrum) of one locus; cases and their evidence are appended to
    the two open files.  A case is a sample labelled `risk` with PP above minPP; a carrier is a sample not labelled
    risk whose longer allele is past the disease cut-off (for contraction loci: at or below it, and called)."""
    t = repo[name]
    meta = repo.rows[name]
    label, pp, second = name + ".label", name + ".PP", name + ".2"
    have = [r for r in rows if label in r]
    prerisk = [r for r in have if r[label] == "prerisk"]
    cases = [r for r in have if r[label] == "risk" and _num(r.get(pp)) > minPP]
    if t.is_expansion:
        carriers = [r for r in have if r[label] != "risk" and _num(r.get(second)) >= t.cutoff_risk]
    else:
        carriers = [r for r in have if r[label] != "risk" and 0 < _num(r.get(second)) <= t.cutoff_risk]
    if detailsfw is not None and name != "AR":
        depth_keys = [name + k for k in (".FDP", ".PDP", ".RDP", ".PEDP")]
        for r in cases:
            if all(k in r for k in depth_keys):
                fields = [name, t.inheritance, r["SampleKey"], r.get("inferredGender", ""), r[name + ".calls"]]
                fields += [int(_num(r[k])) for k in depth_keys]
                print("\t".join(str(x) for x in fields), file=detailsfw)
    if cases and casesfw is not None:
        print("[{}] - {}".format(name, meta.get("title", "")), file=casesfw)
        print("rep={} inherit={} cutoff={} n_risk={} n_carrier={} loc={}".format(
            t.repeat, t.inheritance, t.cutoff_risk, len(cases), len(carriers), meta["repeat_location"]), file=casesfw)
        columns = ["SampleKey", "inferredGender", name + ".calls", name + ".FR", name + ".PR", name + ".RR", pp]
        columns = [c for c in columns if any(c in r for r in cases)]
        shown = []
        for r in cases:
            view = {c: r.get(c, "") for c in columns}
            for k in (".FR", ".PR", ".RR"):
                if name + k in view:
                    view[name + k] = left_truncate_text([view[name + k]])[0]
            shown.append(view)
        print(_table(shown, columns), file=casesfw)
        print(file=casesfw)
    spectrum = Counter()
    for r in have:
        for allele in (r[name + ".1_"], r[name + ".2_"]):
            if isinstance(allele, int) and allele != -1:
                spectrum[allele] += 1
    return t, len(prerisk), len(cases), len(carriers), counts_to_af(spectrum)


def main(args):
    p = argparse.ArgumentParser(prog="tredreport.py", description=__doc__.split("\n\n")[0],
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("files", nargs="*", help="tred.py JSON or VCF outputs; none: re-read --tsv")
    p.add_argument("--ref", choices=BUILDS, default="hg38", help="genome build of the calls")
    p.add_argument("--tsv", default="out.tsv", help="sample table to write (or to read when no files are given)")
    p.add_argument("--columns", help="further per-locus fields for the table, comma separated (e.g. PP,FR)")
    p.add_argument("--minPP", type=float, default=.5, help="smallest P(pathological) for a sample to count as a case")
    p.add_argument("--cpus", type=int, default=os.cpu_count() or 1, help="threads reading the input files")
    p.add_argument("--version", action="version", version="%(prog)s " + __version__)
    a = p.parse_args(args)
    repo = TREDsRepo(a.ref)
    if a.files:
        jsonformat = a.files[0].endswith(".json")
        print("Using {} cpus to parse {} {} files".format(min(len(a.files), a.cpus), len(a.files),
                                                          "JSON" if jsonformat else "VCF"), file=sys.stderr)
        rows = load(a.files, a.cpus)
        rows = df_to_tsv(rows, a.tsv, extra_columns=a.columns.split(",") if a.columns else (), jsonformat=jsonformat,
                         ref=a.ref)
    elif os.path.exists(a.tsv):
        rows = read_tsv(a.tsv)
        for r in rows:                           # a table written earlier: split the calls back into sizes
            for name in repo.names:
                if name + ".calls" in r:
                    x, y = r[name + ".calls"].split("|")
                    r[name + ".1_"], r[name + ".2_"] = int(x), (int(y) if y != "." else ".")
                    r.setdefault(name + ".2", y if y != "." else -1)
    else:
        p.print_help()
        sys.exit(1)
    if not rows:
        sys.exit("Dataframe empty - check input files")
    total = Counter()
    report = []
    with open(a.tsv + ".cases.txt", "w") as casesfw, open(a.tsv + ".details.txt", "w") as detailsfw:
        print("\t".join(DETAIL_COLUMNS), file=detailsfw)
        for name in repo.names:
            if not any(name + ".label" in r for r in rows):
                continue
            t, n_prerisk, n_risk, n_carrier, af = get_tred_summary(rows, name, repo, minPP=a.minPP, casesfw=casesfw,
                                                                    detailsfw=detailsfw)
            total.update(prerisk=n_prerisk, risk=n_risk, carrier=n_carrier, loci=int(n_risk > 0))
            meta = repo.rows[name]
            report.append({"abbreviation": name, "title": meta.get("title", ""), "motif": meta.get("motif", meta.get("repeat", "")),
                           "inheritance": t.inheritance, "cutoff_prerisk": t.cutoff_prerisk,
                           "cutoff_risk": t.cutoff_risk, "n_prerisk": n_prerisk, "n_risk": n_risk,
                           "n_carrier": n_carrier, "allele_freq": af})
    print("Outlier cases saved to `{}`".format(a.tsv + ".cases.txt"), file=sys.stderr)
    print("Read count details saved to `{}`".format(a.tsv + ".details.txt"), file=sys.stderr)
    with open(a.tsv + ".report.txt", "w", newline="") as fp:
        w = csv.DictWriter(fp, REPORT_COLUMNS, delimiter="\t", lineterminator="\n")
        w.writeheader()
        w.writerows(report)
    print("Summary report written to `{}` (# samples={})".format(a.tsv + ".report.txt", len(report)), file=sys.stderr)
    print("Summary: n_prerisk={prerisk}, n_risk={risk}, n_carrier={carrier}, n_affected_loci={loci}".format(
        **{k: total[k] for k in ("prerisk", "risk", "carrier", "loci")}), file=sys.stderr)
    return total


if __name__ == "__main__":
    main(sys.argv[1:])
