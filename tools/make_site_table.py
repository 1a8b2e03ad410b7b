#!/usr/bin/env python3
"""Derive this repo's own compact data tables from the reference's data files.

Run in the build container only (reads /root/reference at run time; nothing of the reference's
source text is embedded here).  Outputs (committed, plain data):

  tredparse_amd/data/treds.json   locus table: the columns of tredparse/data/TREDs.meta.csv that the
                                  hot path and its callers consume (meta.py:103-129) and the reporter prints (motif, title) + the ALT regions
                                  of TREDs.alts.csv (meta.py:81-95) + the allele_freq column (used only
                                  by the synthetic generator).
  tredparse_amd/data/model.json   lobSTR step-size / stutter constants parsed the way
                                  models.py:42-84 parses illumina_v3.pcrfree.{stepmodel,stuttermodel}.
"""
import json
import os
import sys

import pandas as pd

REF = os.environ.get("TRED_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tredparse_amd", "data")


def main():
    data = os.path.join(REF, "tredparse", "data")
    df = pd.read_csv(os.path.join(data, "TREDs.meta.csv"), index_col=0, encoding="latin-1")
    alts = pd.read_csv(os.path.join(data, "TREDs.alts.csv"), index_col=0)
    loci = []
    for name, row in df.iterrows():
        a = alts.loc[name] if name in alts.index else None
        def alt_field(col):
            if a is None or pd.isnull(a[col]):
                return ""
            return str(a[col])
        af = row["allele_freq"]
        loci.append({
            "name": name,
            "title": str(row["title"]),
            "gene_name": str(row["gene_name"]),
            "motif": str(row["motif"]),
            "repeat": row["repeat"],
            "repeat_location": row["repeat_location"],
            "repeat_location.hg19": row["repeat_location.hg19"],
            "prefix": row["prefix"],
            "suffix": row["suffix"],
            "inheritance": row["inheritance"],
            "mutation_nature": row["mutation_nature"],
            "cutoff_prerisk": int(row["cutoff_prerisk"]),
            "cutoff_risk": int(row["cutoff_risk"]),
            "alts": alt_field("alts"),
            "alts.hg19": alt_field("alts.hg19"),
            "allele_freq": "" if pd.isnull(af) else str(af),
        })
    with open(os.path.join(OUT, "treds.json"), "w") as fp:
        json.dump({"source": "humanlongevity/tredparse v0.7.8 tredparse/data/TREDs.meta.csv + TREDs.alts.csv",
                   "loci": loci}, fp, indent=1)

    # step model (models.py:46-61): 6 floats, ProbIncrease=, 6 rows "Period<k>Model v..."
    lines = open(os.path.join(data, "illumina_v3.pcrfree.stepmodel")).read().splitlines()
    non_unit = [float(lines[i].strip()) for i in range(6)]
    prob_increase = float(lines[6].split("=")[1])
    step = {}
    for i in range(6):
        toks = lines[7 + i].split()
        assert toks[0] == "Period{}Model".format(i + 1), toks[0]
        step[str(i + 1)] = [float(x) for x in toks[1:]]
    # stutter model (models.py:68-77): skip 6 header lines, then every non-empty row is a weight
    rows = open(os.path.join(data, "illumina_v3.pcrfree.stuttermodel")).read().splitlines()
    weights = [float(r.strip()) for r in rows[6:] if r.strip()]
    with open(os.path.join(OUT, "model.json"), "w") as fp:
        json.dump({"source": "illumina_v3.pcrfree.{stepmodel,stuttermodel} (lobSTR-trained constants)",
                   "non_unit_step_by_period": non_unit, "prob_increase": prob_increase,
                   "step_size_by_period": step, "stutter_weights": weights}, fp, indent=1)
    print("wrote", len(loci), "loci;", {k: len(v) for k, v in step.items()}, weights)


if __name__ == "__main__":
    sys.exit(main())


def chr_y_regions():
    """First 25 rows of chrY.<ref>.unique_ccn.gc (bam_parser.py:413-429 reads at most row 19)."""
    data = os.path.join(REF, "tredparse", "data")
    for ref in ("hg38", "hg19"):
        rows = open(os.path.join(data, "chrY.{}.unique_ccn.gc".format(ref))).read().splitlines()[:25]
        with open(os.path.join(OUT, "chrY.{}.unique_ccn.tsv".format(ref)), "w") as fp:
            for r in rows:
                fp.write("\t".join(r.split()[:4]) + "\n")


if __name__ == "__main__":
    chr_y_regions()
