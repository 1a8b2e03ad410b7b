"""Host mirror of tredparse/models.py's caller interface on top of the GPU likelihood grid.

IntegratedCaller keeps the reference's constructor and attributes (models.py:101-147, 394-415):
``alleles, label, CI, PP, P_h1, P_h2, P_h1h2, PEDP, PEG, PET, P_PEG, P_PET``.  The grid itself
(pdf_spanning .. evaluate, calc_CI, calc_PP, PEMaxLikModel; models.py:149-368, 426-473) runs in
libtredgpu.so; what stays on the host is formatting: sparsify (:304-317), calc_label (:370-392),
mean_std / histogram (:87-98).
"""
import logging
from math import exp

import numpy as np

from .bam_parser import PEextractor, SPAN

SMALL_VALUE = exp(-10)
MIN_SPANNING_PAIRS = 5

# tredgpu_call.status values that stand for an exception of the reference (the locus is dropped by
# tred.py:245-249); see INTEGRATION.md
STATUS_ERRORS = {-2: "LinAlgError: singular KDE covariance", -3: "IndexError: observation outside the 1000-bin pdf",
                 -4: "grid dump capacity", -5: "grid larger than the kernel's limits", -6: "pair length outside [0,1000)",
                 -7: "KeyError: period without a step model", -8: "ValueError: empty grid", -9: "too many distinct sizes"}


class GridError(RuntimeError):
    pass


def mean_std(a):  # models.py:87-91
    if not a:
        return ""
    a = np.array(a)
    return "{:.0f}+/-{:.0f}bp".format(a.mean(), a.std())


def histogram(a, bins=40):  # models.py:94-98
    if not a:
        return ""
    ar, br = np.histogram(a, bins=bins, range=(0, SPAN))
    return ",".join(["{}:{}".format(int(b), a) for (a, b) in zip(ar, br)])


def calc_label(tred, alleles):  # models.py:370-392
    a, b = sorted(alleles)
    label = "ok" if a != -1 else "missing"
    cutoff_prerisk, cutoff_risk = tred.cutoff_prerisk, tred.cutoff_risk
    if tred.is_expansion:
        crit_allele = a if tred.is_recessive else b
        if cutoff_prerisk <= crit_allele < cutoff_risk:
            label = "prerisk"
        elif crit_allele >= cutoff_risk:
            label = "risk"
    else:
        crit_allele = b if tred.is_recessive else a
        if cutoff_prerisk <= crit_allele < cutoff_risk:
            label = "prerisk"
        elif 0 < crit_allele <= cutoff_risk:
            label = "risk"
    return label


def sparsify_marginal(P, epsilon=SMALL_VALUE):
    """models.py:304-317 for a marginal given as a dense array indexed by repeat units."""
    total = float(P.sum())
    return {str(int(k)): float(P[k] / total) for k in np.nonzero(P >= epsilon)[0]}


def sparsify_joint(grid, period, epsilon=SMALL_VALUE):
    """models.py:279-285 + 304-317 from the dumped grid rows {h1, h2, ml1..ml4} (enumeration order)."""
    ml = grid[:, 2] + grid[:, 3] + grid[:, 4] + grid[:, 5]
    mlexp = np.exp(ml - ml.max())
    P = {}
    for (h1, h2), v in zip(grid[:, :2].astype(np.int64), mlexp):
        P[(int(h1), int(h2))] = float(v)      # later duplicates overwrite, as in the reference's dict
    total = sum(P.values())
    return {"{},{}".format(h1 // period, h2 // period): v / total for (h1, h2), v in P.items() if v >= epsilon}


def sparsify_joint_triples(triples, total, period):
    """models.py:279-285 + 304-317 from the kernel's sparse joint output: triples {h1, h2, exp(ml - max)} of the
    distinct pairs >= e^-10 and the sum over all distinct pairs."""
    return {"{},{}".format(int(h1) // period, int(h2) // period): float(v) / total for h1, h2, v in triples}


class IntegratedCaller:
    """Same constructor and result attributes as the reference's IntegratedCaller."""

    def __init__(self, bamParser, score=1.0, gc=.68, maxinsert=300, fullsearch=False, pe=None):
        self.bamParser = bamParser
        self.tred = bamParser.tred
        self.readlen = bamParser.READLEN
        self.period = bamParser.repeatSize
        self.counts = bamParser.counts
        self.rept = bamParser.rept
        self.ploidy = bamParser.ploidy
        self.half_depth = bamParser.depth / 2
        self.maxinsert = maxinsert
        self.fullsearch = fullsearch
        self.logger = logging.getLogger('IntegratedCaller')
        self.pe = pe if pe is not None else PEextractor(bamParser)
        self.PEDP = len(self.pe.target_lens)
        self.PEG = mean_std(self.pe.global_lens)
        self.PET = mean_std(self.pe.target_lens)
        self.P_PEG = histogram(self.pe.global_lens)
        self.P_PET = histogram(self.pe.target_lens)
        self.P_h1 = ""
        self.P_h2 = ""
        self.P_h1h2 = ""

    def unit(self, reads=()):
        from .engine import Unit
        bp = self.bamParser
        return Unit(self.tred, self.readlen, reads, bp.depth, self.ploidy, self.pe.global_lens, self.pe.target_lens,
                    maxinsert=self.maxinsert, fullsearch=self.fullsearch, clip=bp.clip)

    def from_result(self, res):
        """Fill the reference's attributes from an engine.UnitResult (models.py:394-415)."""
        call = res.call
        status = int(call["status"])
        if status < 0:
            raise GridError(STATUS_ERRORS.get(status, "status {}".format(status)))
        if status == 1:      # no evidence: alleles (-1,-1), lik = PP = -1 (models.py:406-408)
            self.alleles = [-1, -1]
            self.lik = self.PP = -1
            self.CI = ""
        else:
            self.alleles = sorted([int(call["h1"]) // self.period, int(call["h2"]) // self.period])
            self.lik = float(call["lik"])
            self.PP = float(call["pp"])
            self.CI = "{}-{}|{}-{}".format(*[int(x) for x in call["ci"]])
            self.P_h1 = sparsify_marginal(res.P_h1)
            self.P_h2 = sparsify_marginal(res.P_h2)
            if getattr(res, "joint", None) is not None:
                self.P_h1h2 = sparsify_joint_triples(res.joint[0], res.joint[1], self.period)
            elif res.grid is not None:
                self.P_h1h2 = sparsify_joint(res.grid, self.period)
        self.label = calc_label(self.tred, self.alleles)

    def call(self, engine=None, **kwargs):
        """Single-unit convenience with the reference's signature: histograms come from bamParser.counts."""
        from .engine import Engine
        engine = engine or Engine()
        full = {int(k): int(v) for k, v in self.counts["FULL"].items()}
        pref = {int(k): int(v) for k, v in self.counts["PREF"].items()}
        res = engine.grid_from_counts(self.unit(), full, pref, int(self.rept))
        self.from_result(res)
