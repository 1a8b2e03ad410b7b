"""The locus table: which repeats are genotyped, where they lie in each reference build, their flanks and the
disease thresholds.

Data come from this package's own ``data/treds.json`` (32 loci, derived once by tools/make_site_table.py) plus any
``<sites>/*.json`` files of user loci in the schema the reference documents (sites/README.md).  The class names are
the reference's (tredparse/meta.py:29-129: ``TREDsRepo`` keyed by locus name, entries with ``repeat``, ``chr``,
``repeat_start``/``repeat_end``, ``prefix``/``suffix``, cut-offs, inheritance flags, ``ploidy``, ``alt``) because
callers and the reporter address loci through them; the implementation is a read-only table of ``Locus`` records.
"""
import glob
import json
import os
import re

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "treds.json")
BUILDS = ("hg38", "hg38_nochr", "hg19", "hg19_nochr")
_REGION = re.compile(r"^([^:]+):(\d+)-(\d+)$")


def get_region(text):
    """'chr4:3074877-3074933' -> ('chr4', 3074877, 3074933)."""
    m = _REGION.match(text.strip())
    if not m:
        raise ValueError("not a region: `{}`".format(text))
    return m.group(1), int(m.group(2)), int(m.group(3))


class Locus(object):
    """One tandem-repeat locus in one reference build."""

    __slots__ = ("name", "row", "repeat", "chr", "repeat_start", "repeat_end", "prefix", "suffix", "cutoff_prerisk",
                 "cutoff_risk", "inheritance", "mutation_nature", "ploidy", "alt")

    def __init__(self, name, row, build="hg38", alt=()):
        assembly, _, style = build.partition("_")
        column = "repeat_location" if assembly == "hg38" else "repeat_location." + assembly
        where = row[column]
        if style == "nochr":                       # builds whose contigs are named 1, 2, ... X
            where = where.replace("chr", "")
        self.name, self.row = name, row
        self.chr, self.repeat_start, self.repeat_end = get_region(where)
        self.repeat, self.prefix, self.suffix = row["repeat"], row["prefix"], row["suffix"]
        self.cutoff_prerisk, self.cutoff_risk = row["cutoff_prerisk"], row["cutoff_risk"]
        self.inheritance, self.mutation_nature = row["inheritance"], row["mutation_nature"]
        self.ploidy = 2
        self.alt = list(alt)

    # what the table encodes in its inheritance / mutation_nature columns (AD, AR, XLD, XLR; increase, decrease)
    is_xlinked = property(lambda self: self.inheritance.startswith("X"))
    is_recessive = property(lambda self: self.inheritance.endswith("R"))
    is_expansion = property(lambda self: self.mutation_nature == "increase")

    @property
    def period(self):
        return len(self.repeat)

    @property
    def ref_copy(self):
        """Whole repeat units of the tract in the reference genome."""
        return (self.repeat_end - self.repeat_start + 1) // self.period

    def __repr__(self):
        return "{} inheritance={} id={}_{}_{}".format(self.name, self.inheritance, self.chr, self.repeat_start, self.repeat)

    def __str__(self):
        fields = (self.name, self.repeat, self.chr, self.repeat_start, self.repeat_end, self.prefix, self.suffix)
        return ";".join(map(str, fields))


TRED = Locus


class TREDsRepo(object):
    """name -> Locus, in table order (``names``)."""

    def __init__(self, ref="hg38", toy=False, sites="sites"):
        if ref not in BUILDS:
            raise ValueError("unknown reference build `{}`".format(ref))
        self.ref = ref
        self.names, self.rows, self._loci = [], {}, {}
        assembly = ref.split("_")[0]
        alts_column = "alts" if assembly == "hg38" else "alts." + assembly
        with open(_DATA) as fp:
            for row in json.load(fp)["loci"]:
                regions = [get_region(r) for r in row.get(alts_column, "").split("|") if r]
                self._add(row["name"], row, regions)
        for path in sorted(glob.glob(os.path.join(sites, "*.json"))):      # user loci: {name: {column: value}}
            with open(path) as fp:
                for name, row in json.load(fp).items():
                    self._add(str(name), row, [])
        if toy:          # the reference's --toy switch: HD moved to a 1 kb artificial contig
            hd = self._loci["HD"]
            hd.name, hd.chr, hd.repeat_start, hd.repeat_end = "toy", "CHR4", 1001, 1057
            self._loci["toy"] = hd

    def _add(self, name, row, regions):
        self._loci[name] = Locus(name, row, build=self.ref, alt=regions)
        self.rows[name] = row
        self.names.append(name)

    # mapping protocol
    def __getitem__(self, name):
        return self._loci[name]

    def __contains__(self, name):
        return name in self._loci

    def __iter__(self):
        return iter(self._loci)

    def __len__(self):
        return len(self._loci)

    def get(self, name, default=None):
        return self._loci.get(name, default)

    def items(self):
        return self._loci.items()

    def values(self):
        return self._loci.values()

    def set_ploidy(self, haploid):
        """--haploid chrX ...: loci on the listed contigs are genotyped with one allele."""
        for locus in self._loci.values():
            if haploid and locus.chr in haploid:
                locus.ploidy = 1

    def get_info(self, name):
        """(chrom, pos, reference copies, motif, INFO column) of the locus' VCF line."""
        t = self._loci[name]
        info = ";".join("{}={}".format(k, v) for k, v in (
            ("END", t.repeat_end), ("MOTIF", t.repeat), ("NS", 1), ("REF", t.ref_copy), ("CR", t.cutoff_risk),
            ("IH", t.inheritance), ("RL", t.ref_copy * t.period))) + ";VT=STR"
        return t.chr, t.repeat_start, t.ref_copy, t.repeat, info
