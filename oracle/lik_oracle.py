"""lik_oracle.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Plain numpy restatement of the reference's likelihood model, function by function:

  StepModel / NoiseModel      /root/reference/tredparse/models.py:42-84   (constants from
                              tredparse_amd/data/model.json, parsed from the reference's data files
                              by tools/make_site_table.py)
  pdf_spanning                models.py:149-168
  pdf_partial                 models.py:170-180
  get_alpha                   models.py:182-190
  evaluate_spanning/_partial  models.py:192-207, safe_log :418-423
  evaluate_rept               models.py:209-221  (scipy.stats.poisson.pmf, scipy 1.15.3 here:
                              exp(xlogy(k, mu) - gammaln(k + 1) - mu))
  evaluate                    models.py:223-302
  sparsify / calc_CI / calc_PP / calc_label / call   models.py:304-415
  PEMaxLikModel               models.py:426-473  (scipy.stats.gaussian_kde, Scott bandwidth)

It keeps the reference's dense 1000-vectors and Python loops on purpose -- it is the checker, not
the thing measured.  Third-party arithmetic (scipy poisson / gaussian_kde, numpy log/exp) is called
exactly as the reference calls it; the reference pins no versions (requirements.txt), so this
container's scipy 1.15.3 / numpy 2.2.6 define parity.
Parity pin: tests/test_oracle_lik.py checks this file against tests/golden/grid_*.json, which
tools/gen_golden.py produced by running the reference's own models.py (through tools/refshim.py).
"""
import json
import os
from collections import defaultdict
from math import exp

import numpy as np
from scipy.stats import gaussian_kde, poisson

SPAN = 1000                      # bam_parser.py:29
FLANKMATCH = 9                   # bam_parser.py:30
MAX_PERIOD = 6                   # models.py:33
SMALL_VALUE = exp(-10)           # models.py:34
REALLY_SMALL_VALUE = exp(-100)   # models.py:35
MIN_SPANNING_PAIRS = 5           # models.py:39

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tredparse_amd", "data", "model.json")


def load_model(path=_DATA):
    with open(path) as fp:
        m = json.load(fp)
    step = {int(k): np.array(v, float) for k, v in m["step_size_by_period"].items()}
    for i in range(MAX_PERIOD, 3 * MAX_PERIOD):  # models.py:59-60
        step[i] = step[MAX_PERIOD]
    return step, list(m["stutter_weights"])


def safe_log(pdf):  # models.py:418-423
    pdf[pdf < SMALL_VALUE] = SMALL_VALUE
    return np.log(pdf)


class PEModel:  # models.py:426-473
    def __init__(self, global_lens, target_lens, ref, minpe):
        self.MINPE = minpe
        kde = gaussian_kde(global_lens)
        pdf = kde.evaluate(np.arange(SPAN))
        self.pdf = pdf / pdf.sum()
        self.target_lens = target_lens
        self.ref = ref
        self.db = {}

    def roll(self, h):
        if h in self.db:
            return self.db[h]
        shift = self.ref - h
        p = np.roll(self.pdf, shift)
        if shift > 0:
            p[:shift] = SMALL_VALUE
        elif shift < 0:
            p[shift:] = SMALL_VALUE
        p[:self.MINPE] = SMALL_VALUE
        self.db[h] = p
        return p

    def evaluate(self, h1, h2):
        alpha = .5
        mm = safe_log(alpha * self.roll(h1) + (1 - alpha) * self.roll(h2))
        return sum(mm[tl] for tl in self.target_lens)


class Caller:
    """IntegratedCaller (models.py:101-415) on plain inputs.

    full / partial: dicts {repeat units: count} (counts["FULL"], counts["PREF"] incl. POST)
    """

    def __init__(self, period, readlen, ploidy, depth, full, partial, n_rept, global_lens, target_lens,
                 ref_len, minpe, maxinsert=300, fullsearch=False, score=1.0, gc=.68, model=None):
        self.step, self.weights = model or load_model()
        self.period, self.readlen, self.ploidy = period, readlen, ploidy
        self.t1 = readlen - FLANKMATCH
        self.t2 = readlen - 2 * FLANKMATCH
        self.t3 = readlen - 3 * FLANKMATCH
        self.max_partial = self.t2
        self.score, self.gc = score, gc
        self.full, self.partial, self.rept = dict(full), dict(partial), n_rept
        self.half_depth = depth / 2
        self.maxinsert, self.fullsearch = maxinsert, fullsearch
        self.pemodel = PEModel(list(global_lens), list(target_lens), ref_len, minpe) \
            if (len(global_lens) >= 100 and len(target_lens) >= MIN_SPANNING_PAIRS) else None
        self.spanning_db, self.partial_db = {}, {}

    def predict(self, x):  # models.py:79-84
        z = self.weights[0]
        for b, xx in zip(self.weights[1:], x):
            z += b * xx
        return 1.0 / (1 + exp(-1 * z))

    def pdf_spanning(self, h):  # models.py:149-168
        if h in self.spanning_db:
            return self.spanning_db[h]
        a = np.zeros(SPAN)
        stutter_prob = self.predict((self.period, h // self.period, self.gc, self.score))
        p = self.step[self.period] * stutter_prob
        lp = len(p)
        dev = lp // 2
        p[dev] = 1 - stutter_prob
        start, end = h - dev, h + dev + 1
        if start < 0:
            start = 0
        if end > SPAN:
            end = SPAN
        a[start:end] = p[lp - end + start:lp]
        self.spanning_db[h] = a
        return a

    def pdf_partial(self, h):  # models.py:170-180
        if h in self.partial_db:
            return self.partial_db[h]
        if h > self.max_partial:
            h = self.max_partial
        a = np.zeros(SPAN)
        c = 1. / (h + 1)
        a[:h] = c
        a += c * self.pdf_spanning(h)
        self.partial_db[h] = a
        return a

    def get_alpha(self, h1, h2, mode=0):  # models.py:182-190
        if mode == 0:
            s1, s2 = max(0, self.t2 - h1), max(0, self.t2 - h2)
        else:
            s1, s2 = min(h1, self.t1), min(h2, self.t1)
        return s1 * 1. / (s1 + s2) if (s1 + s2) else .5

    def evaluate_spanning(self, obs, h1, h2):  # models.py:192-198
        alpha = self.get_alpha(h1, h2, mode=0)
        ls = safe_log(alpha * self.pdf_spanning(h1) + (1 - alpha) * self.pdf_spanning(h2))
        return sum(ls[k] * c for k, c in obs.items())

    def evaluate_partial(self, obs, h1, h2):  # models.py:200-207
        alpha = self.get_alpha(h1, h2, mode=1)
        lp = safe_log(alpha * self.pdf_partial(h1) + (1 - alpha) * self.pdf_partial(h2))
        return sum(lp[k] * c for k, c in obs.items())

    def evaluate_rept(self, n, h1, h2):  # models.py:209-221
        d1 = max(h1 - self.readlen, 1)
        d2 = max(h2 - self.readlen, 1)
        mu = (d1 + d2) * self.half_depth / self.readlen
        return np.log(max(poisson.pmf(n, mu), REALLY_SMALL_VALUE))

    def evaluate(self):  # models.py:223-302 (+ call :394-404 for the bp conversion)
        period = self.period
        # ascending key order (the reference iterates dicts in hash/insertion order; the sums differ
        # by O(1e-13) at most -- SURVEY.md 9.1)
        obs_spanning = dict(sorted((k * period, v) for k, v in self.full.items()))
        obs_partial = dict(sorted((k * period, v) for k, v in self.partial.items()))
        n_obs_rept = self.rept
        max_full = max(obs_spanning.keys()) if obs_spanning else 0
        max_partial = max(obs_partial.keys()) if obs_partial else 0
        reads_above_full = sum(c for k, c in obs_partial.items() if k > max_full + period)
        run_pe = max_partial >= self.t3 and reads_above_full > 1 and (self.pemodel is not None)
        possible = set(obs_spanning.keys())
        if obs_partial:
            if max_partial > self.max_partial:
                self.max_partial = max_partial
            possible.add(max_partial)
        res = {"run_pe": bool(run_pe)}
        if not possible:
            res.update(status=1, mls=[])
            return res
        base_range = sorted(possible)
        extended_range = base_range + list(range(max_partial + period, period * self.maxinsert + 1, period))
        if self.fullsearch:
            h1range = h2range = list(range(period, period * self.maxinsert + 1, period))
        else:
            h1range = base_range if max_full else extended_range
            h2range = extended_range if (n_obs_rept or run_pe) else base_range
        mls = []
        for h1 in h1range:
            h2_range = [h1] if self.ploidy == 1 else h2range
            for h2 in h2_range:
                if h1 > h2:
                    continue
                ml1 = self.evaluate_spanning(obs_spanning, h1, h2) if obs_spanning else 0
                ml2 = self.evaluate_partial(obs_partial, h1, h2) if obs_partial else 0
                ml3 = self.evaluate_rept(n_obs_rept, h1, h2)
                ml4 = self.pemodel.evaluate(h1, h2) if run_pe else 0
                mls.append((h1, h2, float(ml1), float(ml2), float(ml3), float(ml4)))
        tot = [(m[2] + m[3] + m[4] + m[5], (m[0], m[1])) for m in mls]
        P_h1, P_h2, P_h1h2 = defaultdict(float), defaultdict(float), {}
        max_ml = max(tot)[0]
        for ml, (h1, h2) in tot:
            mlexp = exp(ml - max_ml)
            P_h1[h1] += mlexp
            P_h2[h2] += mlexp
            P_h1h2[(h1, h2)] = mlexp
        h1_lo, h1_hi = calc_CI(P_h1)
        h2_lo, h2_hi = calc_CI(P_h2)
        lik, alleles = max(tot, key=lambda x: (x[0], -x[1][0]))
        res.update(status=0, mls=mls, alleles=alleles, lik=lik,
                   CI=(h1_lo // period, h1_hi // period, h2_lo // period, h2_hi // period),
                   P_h1=dict(P_h1), P_h2=dict(P_h2), P_h1h2=P_h1h2, tot=tot)
        return res


def calc_CI(P):  # models.py:319-340
    cum_sum = 0
    alpha, beta = .025, .975
    lo, hi = 0, 0
    in_range = False
    total_prob = sum(P.values())
    k = 0
    for k, v in sorted(P.items()):
        cum_sum += v
        if (not in_range) and cum_sum > alpha * total_prob:
            in_range = True
            lo = k
        if cum_sum > beta * total_prob:
            break
    hi = k
    return lo, hi


def calc_PP(tot, lik, period, cutoff_risk, is_expansion, is_recessive):  # models.py:342-368
    if is_expansion:
        if not is_recessive:
            path = [x[0] for x in tot if max(x[1]) // period >= cutoff_risk]
        else:
            path = [x[0] for x in tot if min(x[1]) // period >= cutoff_risk]
    else:
        if not is_recessive:
            path = [x[0] for x in tot if min(x[1]) // period <= cutoff_risk]
        else:
            path = [x[0] for x in tot if max(x[1]) // period <= cutoff_risk]
    path = np.array(path)
    all_liks = np.array([x[0] for x in tot])
    return min(1, np.exp(path - lik).sum() / np.exp(all_liks - lik).sum())


def calc_label(alleles, cutoff_prerisk, cutoff_risk, is_expansion, is_recessive):  # models.py:370-392
    a, b = sorted(alleles)
    label = "ok" if a != -1 else "missing"
    if is_expansion:
        crit = a if is_recessive else b
        if cutoff_prerisk <= crit < cutoff_risk:
            label = "prerisk"
        elif crit >= cutoff_risk:
            label = "risk"
    else:
        crit = b if is_recessive else a
        if cutoff_prerisk <= crit < cutoff_risk:
            label = "prerisk"
        elif 0 < crit <= cutoff_risk:
            label = "risk"
    return label


def sparsify(P, period):  # models.py:304-317
    Z = {}
    total = sum(v for v in P.values())
    for k, v in P.items():
        if v < SMALL_VALUE:
            continue
        kk = [x // period for x in (k if isinstance(k, (list, tuple)) else [k])]
        Z[",".join(str(x) for x in kk)] = v / total
    return Z


def kde_pdf(global_lens):
    """models.py:428-435."""
    pdf = gaussian_kde(list(global_lens)).evaluate(np.arange(SPAN))
    return pdf / pdf.sum()
