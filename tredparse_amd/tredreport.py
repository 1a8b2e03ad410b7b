#!/usr/bin/env python3
"""Report significant calls -- counterpart of tredparse/tredreport.py for the JSON files tred.py writes.

JSON files -> one TSV row per sample (df_to_tsv, tredreport.py:110-142), then per locus the
pre-risk / risk / carrier counts, the at-risk cases with their read evidence and the allele
frequencies (get_tred_summary :37-101, main :198-302).  VCF input of the reference (needs PyVCF) is
not carried over; everything is host-side pandas, nothing here touches the GPU.
"""
import argparse
import json
import math
import os.path as op
import sys
from collections import Counter

import pandas as pd

from . import __version__
from .meta import TREDsRepo


def left_truncate_text(a, maxcol=30):  # tredreport.py:31-34
    trim = lambda t: (t if not isinstance(t, str) or len(t) <= maxcol else "..." + t[-(maxcol - 3):])
    return [trim(x) for x in list(a)]


def counts_to_af(counts):  # tredreport.py:104-107
    return "{" + ",".join("{}:{}".format(k, v) for k, v in sorted(counts.items())
                          if not (k == '.' or (isinstance(k, float) and math.isnan(k)))) + "}"


def json_to_df(jsonfiles):  # tredreport.py:172-195 (sample key from the file name)
    rows = []
    for jsonfile in jsonfiles:
        with open(jsonfile) as fp:
            js = json.load(fp)
        d = {'SampleKey': op.basename(jsonfile).split(".")[0]}
        d.update(js['tredCalls'])
        rows.append(d)
    return pd.DataFrame(rows)


def df_to_tsv(df, tsvfile, extra_columns=(), ref="hg38"):  # tredreport.py:110-142
    df = df.fillna(-1)
    dd = ["SampleKey", "inferredGender"]
    repo = TREDsRepo(ref)
    for tred in repo.names:
        tr = repo[tred]
        if tred + ".1" not in df.columns:
            continue
        df[tred + ".1_"] = df[tred + ".1"].astype("int")
        df[tred + ".2_"] = df[tred + ".2"].astype("int").astype(object)
        if tr.is_xlinked:
            df.loc[(df["inferredGender"] == "Male"), tred + ".2_"] = "."
        df[tred + ".calls"] = ["{}|{}".format(a, b) for (a, b) in zip(df[tred + ".1_"], df[tred + ".2_"])]
    all_columns = ["calls", "label"] + list(extra_columns)
    columns = dd + sorted([x for x in df.columns if (x not in dd) and any(x.endswith("." + z) for z in all_columns)])
    tf = df.reindex(columns=columns)
    tf.to_csv(tsvfile, sep='\t', index=False)
    print("TSV output written to `{}` (# samples={})".format(tsvfile, tf.shape[0]), file=sys.stderr)
    return df


def get_tred_summary(df, tred, repo, minPP=.5, casesfw=None, detailsfw=None):  # tredreport.py:37-101
    pf2, label, pp = tred + ".2", tred + ".label", tred + ".PP"
    tr = repo[tred]
    row = repo.rows[tred]
    prerisk = df[df[label] == "prerisk"]
    risk = df[(df[label] == "risk") & (df[pp] > minPP)].copy()
    if tr.is_expansion:
        carrier = df[(df[label] != "risk") & (df[pf2] >= tr.cutoff_risk)]
    else:
        carrier = df[(df[label] != "risk") & (df[pf2] <= tr.cutoff_risk) & (df[pf2] > 0)]
    n_prerisk, n_risk, n_carrier = prerisk.shape[0], risk.shape[0], carrier.shape[0]
    calls = tred + ".calls"
    core = ["SampleKey", "inferredGender", calls]
    columns = core + [tred + ".FR", tred + ".PR", tred + ".RR", pp]
    for k in (".FR", ".PR", ".RR"):
        if tred + k in risk.columns:
            risk[tred + k] = left_truncate_text(risk[tred + k])
    if detailsfw is not None and tred != "AR":
        have = [c for c in (tred + ".FDP", tred + ".PDP", tred + ".RDP", tred + ".PEDP") if c in risk.columns]
        if len(have) == 4:
            for _, r in risk[core + have].iterrows():
                samplekey, sex, call, fdp, pdp, rdp, pedp = r
                print("\t".join(str(x) for x in (tred, tr.inheritance, samplekey, sex, call, int(fdp), int(pdp),
                                                 int(rdp), int(pedp))), file=detailsfw)
    if n_risk and casesfw is not None:
        print("[{}] - {}".format(tred, row.get("title", "")), file=casesfw)
        print("rep={}".format(tr.repeat), "inherit={}".format(tr.inheritance), "cutoff={}".format(tr.cutoff_risk),
              "n_risk={}".format(n_risk), "n_carrier={}".format(n_carrier),
              "loc={}".format(row["repeat_location"]), file=casesfw)
        print(risk[[c for c in columns if c in risk.columns]].to_string(index=False), file=casesfw)
        print(file=casesfw)
    cnt = Counter()
    cnt.update(df[tred + ".1_"])
    cnt.update(x for x in df[tred + ".2_"] if x != ".")
    cnt.pop(-1, None)
    return tr, n_prerisk, n_risk, n_carrier, counts_to_af(cnt)


def main(args):
    p = argparse.ArgumentParser(description=__doc__, prog="tredreport.py",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("files", nargs="*")
    p.add_argument('--ref', choices=("hg38", "hg38_nochr", "hg19", "hg19_nochr"), default='hg38')
    p.add_argument('--tsv', default="out.tsv", help="Path to the tsv file")
    p.add_argument('--columns', help="Columns to extract, use comma to separate")
    p.add_argument('--minPP', default=.5, type=float, help="Minimum Prob(pathological) to report cases")
    p.add_argument('--version', action='version', version="%(prog)s " + __version__)
    args = p.parse_args(args)
    columns = args.columns.split(",") if args.columns else []
    repo = TREDsRepo(args.ref)
    if not args.files:
        sys.exit(not p.print_help())
    if not args.files[0].endswith(".json"):
        sys.exit("only the JSON output of tred.py is supported (the reference's VCF path needs PyVCF)")
    df = df_to_tsv(json_to_df(args.files), args.tsv, extra_columns=columns, ref=args.ref)
    if df.empty:
        sys.exit("Dataframe empty - check input files")
    rows = []
    total = Counter()
    with open(args.tsv + ".cases.txt", "w") as casesfw, open(args.tsv + ".details.txt", "w") as detailsfw:
        print("\t".join("Locus,Inheritance,SampleKey,Sex,Calls,FullReads,PartialReads,RepeatReads,PairedReads".split(',')),
              file=detailsfw)
        for tred in repo.names:
            if tred + ".label" not in df.columns:
                continue
            tr, n_prerisk, n_risk, n_carrier, af = get_tred_summary(df, tred, repo, minPP=args.minPP, casesfw=casesfw,
                                                                    detailsfw=detailsfw)
            total.update(prerisk=n_prerisk, risk=n_risk, carrier=n_carrier, loci=1 if n_risk else 0)
            r = repo.rows[tred]
            rows.append({"abbreviation": tred, "title": r.get("title", ""), "motif": r.get("repeat", ""),
                         "inheritance": tr.inheritance, "cutoff_prerisk": tr.cutoff_prerisk,
                         "cutoff_risk": tr.cutoff_risk, "n_prerisk": n_prerisk, "n_risk": n_risk,
                         "n_carrier": n_carrier, "allele_freq": af})
    pd.DataFrame(rows).to_csv(args.tsv + ".report.txt", sep="\t", index=False)
    print("Summary: n_prerisk={prerisk}, n_risk={risk}, n_carrier={carrier}, n_affected_loci={loci}".format(**total),
          file=sys.stderr)
    return total


if __name__ == '__main__':
    main(sys.argv[1:])
