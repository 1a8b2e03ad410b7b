"""Locus table mirroring tredparse/meta.py (TREDsRepo :29-100, TRED :103-129, get_region :143-150).

The table itself is this repo's own data file tredparse_amd/data/treds.json (derived from the
reference's TREDs.meta.csv / TREDs.alts.csv by tools/make_site_table.py); user loci are read from
``<sites>/*.json`` with the reference's schema (meta.py:44-49, sites/README.md).
"""
import json
import os
from glob import glob

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "hg38"
SITES = "sites"


def get_region(location):
    chr_, location = location.split(":")
    start, end = location.split("-")
    return chr_, int(start), int(end)


class TRED(object):
    def __init__(self, name, row, ref=REF, alt=()):
        self.row = row
        self.name = name
        self.alt = list(alt)
        self.repeat = row["repeat"]
        field = "repeat_location"
        if ref != REF:
            field += "." + ref.split("_")[0]
        repeat_location = row[field]
        if "_nochr" in ref:  # Some reference version do not have chr (meta.py:115-116)
            repeat_location = repeat_location.replace("chr", "")
        self.chr, self.repeat_start, self.repeat_end = get_region(repeat_location)
        self.ref_copy = (self.repeat_end - self.repeat_start + 1) // len(self.repeat)   # py2 int division
        self.prefix = row["prefix"]
        self.suffix = row["suffix"]
        self.cutoff_prerisk = row["cutoff_prerisk"]
        self.cutoff_risk = row["cutoff_risk"]
        self.inheritance = row["inheritance"]
        self.is_xlinked = self.inheritance[0] == 'X'
        self.is_recessive = self.inheritance[-1] == 'R'
        self.is_expansion = row["mutation_nature"] == 'increase'
        self.ploidy = 2

    def __repr__(self):
        return "{} inheritance={} id={}_{}_{}".format(self.name, self.inheritance, self.chr,
                                                      self.repeat_start, self.repeat)

    def __str__(self):
        return ";".join(str(x) for x in (self.name, self.repeat, self.chr, self.repeat_start,
                                         self.repeat_end, self.prefix, self.suffix))


class TREDsRepo(dict):
    def __init__(self, ref=REF, toy=False, sites=SITES):
        self.ref = ref
        with open(os.path.join(HERE, "data", "treds.json")) as fp:
            rows = json.load(fp)["loci"]
        self.names = []
        alts_field = "alts" if ref == REF else "alts." + ref.split("_")[0]
        self.rows = {}
        for row in rows:
            name = row["name"]
            _alts = row.get(alts_field, "")
            regions = [get_region(x) for x in _alts.split("|")] if _alts else []
            self[name] = TRED(name, row, ref=ref, alt=regions)
            self.names.append(name)
            self.rows[name] = row
        for s in sorted(glob("{}/*.json".format(sites))):
            with open(s) as fp:
                user = json.load(fp)
            for name, row in user.items():
                self[str(name)] = TRED(str(name), row, ref=ref, alt=[])
                self.names.append(str(name))
                self.rows[str(name)] = row
        if toy:
            tr = self.get("HD")
            tr.name = "toy"
            tr.chr = "CHR4"
            tr.repeat_start = 1001
            tr.repeat_end = 1057
            self[tr.name] = tr

    def set_ploidy(self, haploid):
        if not haploid:
            return
        for k, v in self.items():
            if v.chr in haploid:
                v.ploidy = 1

    def get_info(self, tredName):
        tr = self.get(tredName)
        info = "END={};MOTIF={};NS=1;REF={};CR={};IH={};RL={};VT=STR".format(
            tr.repeat_end, tr.repeat, tr.ref_copy, tr.cutoff_risk, tr.inheritance,
            tr.ref_copy * len(tr.repeat))
        return tr.chr, tr.repeat_start, tr.ref_copy, tr.repeat, info
