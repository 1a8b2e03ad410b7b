"""Host mirror of tredparse/models.py's caller interface on top of the GPU likelihood grid.

The grid itself (pdf_spanning .. evaluate, calc_CI, calc_PP, PEMaxLikModel; models.py:149-368, 426-473) runs in
libtredgpu.so; what stays on the host is formatting: `format_call` turns a unit's kernel results into the
caller's outputs (alleles, CI, PP, label, sparse marginals and joint), `pair_summary` the pair-length
statistics.  IntegratedCaller keeps the reference's constructor and attributes for single-locus use.
"""
import logging
from collections.abc import Mapping
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


def mean_std(lengths):
    """'346+/-78bp' (population standard deviation); '' for no pairs."""
    if len(lengths) == 0:
        return ""
    v = np.asarray(lengths, np.float64)
    return "%.0f+/-%.0fbp" % (v.mean(), v.std())


def histogram(lengths, bins=40):
    """'0:0,25:3,...': counts of pair lengths in `bins` equal bins over [0, SPAN], keyed by the bin's left edge."""
    if len(lengths) == 0:
        return ""
    counts, edges = np.histogram(np.asarray(lengths), bins=bins, range=(0, SPAN))
    return ",".join("%d:%d" % (left, c) for left, c in zip(edges[:-1], counts))


def calc_label(tred, alleles):
    """ok / prerisk / risk / missing for a pair of allele sizes (repeat units; -1 = no call).
    The decisive allele is the longer one for a dominant expansion disorder and the shorter one for a recessive
    one (both copies must be expanded); for loci whose pathogenic change is a contraction the roles swap and
    "risk" means 0 < allele <= cutoff.  The pre-risk band [cutoff_prerisk, cutoff_risk) is tested first."""
    lo, hi = min(alleles), max(alleles)
    if tred.is_expansion:
        decisive = lo if tred.is_recessive else hi
        at_risk = decisive >= tred.cutoff_risk
    else:
        decisive = hi if tred.is_recessive else lo
        at_risk = 0 < decisive <= tred.cutoff_risk
    if tred.cutoff_prerisk <= decisive < tred.cutoff_risk:
        return "prerisk"
    if at_risk:
        return "risk"
    return "missing" if lo == -1 else "ok"


class SparseDist(Mapping):
    """A sparse distribution of the output (`P_h1`, `P_h2`: keys "a"; `P_h1h2`: keys "a,b") held as arrays.  Reads
    like the dict {key string: probability} (built on first use; == against a dict works); json_text() is the dict as
    tred.to_json prints it, written natively without a Python string and float per entry."""
    __slots__ = ("a", "b", "values", "_dict")

    def __init__(self, a, b, values):
        self.a, self.b, self.values, self._dict = a, b, values, None

    def as_dict(self):
        if self._dict is None:
            if self.b is None:
                keys = map(str, self.a.tolist())
            else:
                keys = ("%d,%d" % k for k in zip(self.a.tolist(), self.b.tolist()))
            self._dict = dict(zip(keys, self.values.tolist()))
        return self._dict

    def __getitem__(self, key):
        return self.as_dict()[key]

    def __iter__(self):
        return iter(self.as_dict())

    def __len__(self):
        return len(self.as_dict())

    def __repr__(self):
        return repr(self.as_dict())

    def json_text(self, depth):
        from . import bamio
        return bamio.sparse_json(self.a, self.b, self.values, depth)


def sparsify_marginal(P, epsilon=SMALL_VALUE, lazy=False):
    """models.py:304-317 for a marginal given as a dense array indexed by repeat units."""
    total = float(P.sum())
    keep = np.nonzero(P >= epsilon)[0]
    d = SparseDist(keep, None, P[keep] / total)
    return d if lazy else d.as_dict()


def sparsify_joint(grid, period, epsilon=SMALL_VALUE):
    """models.py:279-285 + 304-317 from the dumped grid rows {h1, h2, ml1..ml4} (enumeration order)."""
    ml = grid[:, 2] + grid[:, 3] + grid[:, 4] + grid[:, 5]
    mlexp = np.exp(ml - ml.max())
    P = {}
    for (h1, h2), v in zip(grid[:, :2].astype(np.int64), mlexp):
        P[(int(h1), int(h2))] = float(v)      # later duplicates overwrite, as in the reference's dict
    total = sum(P.values())
    return {"{},{}".format(h1 // period, h2 // period): v / total for (h1, h2), v in P.items() if v >= epsilon}


def sparsify_joint_triples(triples, total, period, lazy=False):
    """models.py:279-285 + 304-317 from the kernel's sparse joint output: triples {h1, h2, exp(ml - max)} of the
    distinct pairs >= e^-10 and the sum over all distinct pairs."""
    t = np.asarray(triples)
    d = SparseDist(t[:, 0].astype(np.int64) // period, t[:, 1].astype(np.int64) // period, t[:, 2] / total)
    return d if lazy else d.as_dict()


def format_call(tred, res, lazy=False):
    """The caller's outputs for one unit from an engine.UnitResult, as a dict:
    alleles (units, sorted), lik, PP, CI "lo-hi|lo-hi", label, P_h1, P_h2, P_h1h2 (sparse, keyed by units).
    Raises GridError where the reference's grid raises (the locus is then dropped)."""
    call = res.call
    status = int(call["status"])
    if status < 0:
        raise GridError(STATUS_ERRORS.get(status, "status {}".format(status)))
    period = len(tred.repeat)
    out = {"P_h1": "", "P_h2": "", "P_h1h2": ""}
    if status == 1:                      # no read evidence at all
        out.update(alleles=[-1, -1], lik=-1, PP=-1, CI="")
    else:
        out["alleles"] = sorted((int(call["h1"]) // period, int(call["h2"]) // period))
        out["lik"], out["PP"] = float(call["lik"]), float(call["pp"])
        out["CI"] = "{}-{}|{}-{}".format(*(int(x) for x in call["ci"]))
        out["P_h1"], out["P_h2"] = sparsify_marginal(res.P_h1, lazy=lazy), sparsify_marginal(res.P_h2, lazy=lazy)
        if getattr(res, "joint_units", None) is not None:    # (the batch's joint entries, already in units and normalised)
            d = SparseDist(*res.joint_units)
            out["P_h1h2"] = d if lazy else d.as_dict()
        elif getattr(res, "joint", None) is not None:
            out["P_h1h2"] = sparsify_joint_triples(res.joint[0], res.joint[1], period, lazy=lazy)
        elif res.grid is not None:
            out["P_h1h2"] = sparsify_joint(res.grid, period)
    out["label"] = calc_label(tred, out["alleles"])
    return out


def pair_summary(global_lens, target_lens):
    """PEDP, PEG, PET, P_PEG, P_PET of the JSON from the two pair-length lists."""
    g, t = list(global_lens), list(target_lens)
    return {"PEDP": len(t), "PEG": mean_std(g), "PET": mean_std(t), "P_PEG": histogram(g), "P_PET": histogram(t)}


_HIST_BINS = 40
_HIST_LEFT = ["%d:" % (SPAN // _HIST_BINS * j) for j in range(_HIST_BINS)]


def _pool_strings(pool, first, count):
    """mean_std and histogram strings of many slices pool[first[k] : first[k] + count[k]] at once (the slices are
    consecutive and in order, as the scan's pools are).  Same values as mean_std / histogram per slice: the sums of
    these small integers are exact in float64, the bins of np.histogram over (0, SPAN) are x // 25 for integers."""
    g = len(first)
    count = np.asarray(count, np.int64)
    from . import bamio
    native = bamio.pair_stats(pool, first, count) if g else None
    if native is not None:                  # the same numbers from one native pass over the pool
        mean, std, hist = native
        ms = ["%.0f+/-%.0fbp" % (m, sd) if c else "" for m, sd, c in zip(mean.tolist(), std.tolist(), count.tolist())]
        hs = [",".join([a + str(b) for a, b in zip(_HIST_LEFT, row)]) if c else "" for row, c in zip(hist.tolist(), count.tolist())]
        return ms, hs
    lo, hi = (int(first[0]), int(first[-1] + count[-1])) if g else (0, 0)
    x = np.asarray(pool[lo:hi], np.int64)
    unit = np.repeat(np.arange(g), count)
    n = np.maximum(count, 1).astype(np.float64)
    xf = x.astype(np.float64)
    mean = np.bincount(unit, xf, g) / n
    dev = xf - mean[unit]
    std = np.sqrt(np.bincount(unit, dev * dev, g) / n)
    ok = (x >= 0) & (x <= SPAN)
    width = SPAN // _HIST_BINS
    hist = np.bincount(unit[ok] * _HIST_BINS + np.minimum(x[ok] // width, _HIST_BINS - 1),
                       minlength=g * _HIST_BINS).reshape(g, _HIST_BINS).tolist()
    ms = ["%.0f+/-%.0fbp" % (m, sd) if c else "" for m, sd, c in zip(mean.tolist(), std.tolist(), count.tolist())]
    hs = [",".join([a + str(b) for a, b in zip(_HIST_LEFT, row)]) if c else "" for row, c in zip(hist, count.tolist())]
    return ms, hs


def pair_summaries(scan):
    """pair_summary for every locus of a bam_parser.SampleScan, computed over the scan's pools in one go."""
    u = scan.unit
    peg, p_peg = _pool_strings(scan.global_lens, u["global_first"], u["n_global"])
    pet, p_pet = _pool_strings(scan.target_lens, u["target_first"], u["n_target"])
    return [{"PEDP": int(n), "PEG": a, "PET": b, "P_PEG": c, "P_PET": d}
            for n, a, b, c, d in zip(u["n_target"].tolist(), peg, pet, p_peg, p_pet)]


class IntegratedCaller:
    """Single-locus view with the reference's constructor and result attributes (models.py:101-147, 394-415):
    IntegratedCaller(bamParser, maxinsert=, fullsearch=).call() fills alleles, label, CI, PP, P_h1, P_h2, P_h1h2;
    PEDP / PEG / PET / P_PEG / P_PET are there from construction.  The grid runs in libtredgpu.so."""

    def __init__(self, bamParser, score=1.0, gc=.68, maxinsert=300, fullsearch=False, pe=None):
        self.bamParser, self.tred = bamParser, bamParser.tred
        self.maxinsert, self.fullsearch = maxinsert, fullsearch
        self.pe = pe if pe is not None else PEextractor(bamParser)
        for k, v in pair_summary(self.pe.global_lens, self.pe.target_lens).items():
            setattr(self, k, v)
        self.P_h1 = self.P_h2 = self.P_h1h2 = ""

    def unit(self, reads=()):
        from .engine import Unit
        bp = self.bamParser
        return Unit(self.tred, bp.READLEN, reads, bp.depth, bp.ploidy, self.pe.global_lens, self.pe.target_lens,
                    maxinsert=self.maxinsert, fullsearch=self.fullsearch, clip=bp.clip)

    def from_result(self, res):
        for k, v in format_call(self.tred, res).items():
            setattr(self, k, v)

    def call(self, engine=None, **kwargs):
        """Grid from bamParser.counts / .rept (filled by bamParser.parse())."""
        from .engine import Engine
        engine = engine or Engine()
        bp = self.bamParser
        hist = lambda c: {int(k): int(v) for k, v in c.items()}
        self.from_result(engine.grid_from_counts(self.unit(), hist(bp.counts["FULL"]), hist(bp.counts["PREF"]),
                                                 int(bp.rept)))
