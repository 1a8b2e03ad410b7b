#!/usr/bin/env python3
"""Randomised parity campaign for tredgpu_likelihood_grid on adversarial inputs: histograms, repeat-only counts
and pair-length lists drawn directly (not through reads) -- many distinct FULL sizes, PREF beyond FULL, empty
evidence, zero-variance pair lengths (singular KDE), too few pairs for the paired-end model, one allele, full
search -- for every locus of the table (periods 2..12), against the numpy oracle.  Compared per case: status,
pair enumeration and the four log-likelihood terms of every pair (1e-6), arg-max, CI, lik, PP, run_pe.
Not part of the test suite; prints one JSON line.   usage: python tools/fuzz_hist.py [cases] [seed]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def draw_case(rng, loci):
    locus = loci[int(rng.integers(len(loci)))]
    period = len(locus["repeat"])
    readlen = int(rng.choice([100, 150, 250]))
    mu = -(-readlen // period)
    c = {"locus_rec": locus, "readlen": readlen, "ploidy": int(rng.choice([1, 2], p=[0.2, 0.8])),
         "depth": float(rng.choice([3, 12, 30, 80])), "maxinsert": int(rng.choice([30, 100, 300])),
         "fullsearch": bool(rng.random() < 0.1)}
    nf = int(rng.choice([0, 0, 1, 2, 3, 6, 12]))
    c["full"] = {int(k): int(rng.integers(1, 25)) for k in rng.choice(np.arange(1, mu + 1), min(nf, mu), replace=False)}
    npf = int(rng.choice([0, 1, 3, 8, 20]))
    c["partial"] = {int(k): int(rng.integers(1, 8)) for k in rng.choice(np.arange(1, mu + 1), min(npf, mu), replace=False)}
    c["rept"] = int(rng.choice([0, 0, 1, 5, 40, 300]))
    ng = int(rng.choice([0, 60, 150, 2000]))
    gl = np.clip(np.rint(rng.normal(350, rng.choice([1e-9, 20, 80]), ng)), 1, 999).astype(int)
    c["global_lens"] = [int(x) for x in gl]
    nt = int(rng.choice([0, 3, 6, 20, 60, 140]))
    c["target_lens"] = [int(x) for x in np.clip(np.rint(rng.normal(330, 120, nt)), 0, 999)]
    if nt and rng.random() < 0.08:     # python indexing: negative lengths count from the end, >= 1000 raise IndexError
        c["target_lens"][int(rng.integers(nt))] = int(rng.choice([-1, -40, 1000, 1500]))
    chrom, span = locus["repeat_location"].split(":")
    start, end = (int(x) for x in span.split("-"))
    c["ref_len"], c["minpe"] = end - start + 1, end - start + 20
    return c


def campaign(cases_n=300, seed=1):
    """Runs the campaign and returns its summary (tests/test_fuzz_gpu.py runs a fixed-seed slice of it)."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from oracle import lik_oracle as lo
    from tredparse_amd import _lib, synth
    loci = synth.load_loci()
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    step, w = lo.load_model()
    ctx.set_model(np.array([step[p] for p in range(1, 7)]), np.array(w))
    cases = [draw_case(rng, loci) for _ in range(cases_n)]
    # oracle first (gives the dump capacities)
    want = []
    for c in cases:
        l = c["locus_rec"]
        # (zero-variance pair lengths: gaussian_kde raises LinAlgError -- SURVEY 8a, a17 -- unless the rounding of its
        #  weighted mean leaves a variance of ~1e-28, in which case it builds a one-hot pdf; the kernels follow scipy in
        #  both, csrc/grid.hip kde_of_equal_lengths: the oracle below simply calls scipy)
        try:
            res = lo.Caller(len(l["repeat"]), c["readlen"], c["ploidy"], c["depth"], c["full"], c["partial"], c["rept"],
                            c["global_lens"], c["target_lens"], c["ref_len"], c["minpe"], maxinsert=c["maxinsert"],
                            fullsearch=c["fullsearch"]).evaluate()
        except np.linalg.LinAlgError:
            res = {"raised": "LinAlgError"}
        except IndexError:
            res = {"raised": "IndexError"}
        want.append(res)
    hs = 128
    n = len(cases)
    units = np.zeros(n, _lib.UNIT_DTYPE)
    full = np.zeros((n, hs), np.int32); pref = np.zeros((n, hs), np.int32); rept = np.zeros((n, hs), np.int32)
    gl, tl = [], []
    for i, c in enumerate(cases):
        u = synth.unit_params_for(c["locus_rec"], c["readlen"], c["depth"], len(c["global_lens"]), len(c["target_lens"]),
                                  len(gl), len(tl), ploidy=c["ploidy"], maxinsert=c["maxinsert"], fullsearch=c["fullsearch"])
        units[i] = u
        for k, v in c["full"].items(): full[i, k] = v
        for k, v in c["partial"].items(): pref[i, k] = v
        rept[i, 0] = c["rept"]
        gl += c["global_lens"]; tl += c["target_lens"]
    gl = np.asarray(gl or [0], np.int32); tl = np.asarray(tl or [0], np.int32)
    goff = np.zeros(n + 1, np.int64)
    goff[1:] = np.cumsum([max(len(w.get("mls", [])), 1) for w in want])
    dump = np.zeros((int(goff[-1]), 6), np.float64)
    calls = np.zeros(n, _lib.CALL_DTYPE)
    t0 = time.time()
    ctx.likelihood_grid(_lib.MEM_HOST, units, n, hs, full, pref, rept, gl, len(gl), tl, len(tl), calls, goff, dump, None, 0)
    # the sparse joint distribution (P_h1h2) through its own entry point
    cap = np.array([max(len(w.get("P_h1h2", {})), 1) for w in want], np.int64)
    joff = np.zeros(n + 1, np.int64); joff[1:] = np.cumsum(cap)
    trip = np.zeros((int(joff[-1]), 3), np.float64); jn = np.zeros(n, np.int32); jt = np.zeros(n, np.float64)
    calls_j = np.zeros(n, _lib.CALL_DTYPE)
    ctx.likelihood_grid_joint(_lib.MEM_HOST, units, n, hs, full, pref, rept, gl, len(gl), tl, len(tl), calls_j, None, 0,
                              joff, trip, jn, jt)
    bad = singular = empty = near_ties = 0
    worst = 0.0
    for i, (c, w) in enumerate(zip(cases, want)):
        call, l = calls[i], c["locus_rec"]
        if w.get("raised"):
            singular += 1
            ok = call["status"] == (-2 if w["raised"] == "LinAlgError" else -3)
        elif w["status"] == 1:
            empty += 1
            ok = call["status"] == 1
        else:
            got, exp = dump[goff[i]:goff[i] + call["n_pairs"]], np.asarray(w["mls"], np.float64)
            ok = call["status"] == 0 and call["n_pairs"] == len(exp) and np.array_equal(got[:, :2], exp[:, :2])
            if ok:
                d = float(np.abs(got[:, 2:] - exp[:, 2:]).max())
                worst = max(worst, d)
                ppv = lo.calc_PP(w["tot"], w["lik"], len(l["repeat"]), l["cutoff_risk"], l["mutation_nature"] == "increase",
                                 l["inheritance"][-1] == "R")
                same_call = (call["h1"], call["h2"]) == tuple(w["alleles"])
                if not same_call and d <= 1e-6 and len(set(c["global_lens"])) == 1:
                    # equal pair lengths that scipy lets through give a one-hot paired-end pdf: the paired-end term then takes
                    # a handful of values (sums of log .5, log 1 and log e^-10 in the order of the spanning pairs) and whole
                    # families of pairs tie to the last bits -- in the reference too, where the rounding of its left-to-right
                    # sum decides.  A different pair whose likelihood is the reference's maximum to 1e-9 is such a tie
                    exp_tot = exp[:, 2:].sum(axis=1)
                    at = np.nonzero((exp[:, 0] == call["h1"]) & (exp[:, 1] == call["h2"]))[0]
                    if len(at) and abs(exp_tot[at[0]] - exp_tot.max()) <= 1e-9:
                        near_ties += 1
                        same_call = True
                ok = (d <= 1e-6 and same_call and tuple(call["ci"]) == tuple(w["CI"])
                      and abs(call["lik"] - w["lik"]) <= 1e-6 and abs(call["pp"] - ppv) <= 1e-9
                      and bool(call["run_pe"]) == w["run_pe"])
                if ok:   # joint: same kept pairs, same normalised values
                    P = w["P_h1h2"]
                    tot = sum(P.values())
                    keep = {k: v / tot for k, v in P.items() if v >= lo.SMALL_VALUE}
                    got_j = {(int(a), int(b)): v / jt[i] for a, b, v in trip[joff[i]:joff[i] + min(jn[i], cap[i])]}
                    ok = (calls_j[i].tobytes() == call.tobytes() and jn[i] == len(keep) and set(got_j) == set(keep)
                          and all(abs(got_j[k] - keep[k]) <= 1e-9 for k in keep))
        if not ok:
            bad += 1
            print("MISMATCH case", i, l["name"], {k: c[k] for k in ("readlen", "ploidy", "depth", "maxinsert", "fullsearch",
                  "full", "partial", "rept")}, len(c["global_lens"]), len(c["target_lens"]), call, w.get("alleles"),
                  w.get("CI"), file=sys.stderr)
    ctx.close()
    return {"cases": n, "mismatches": bad, "cases_the_reference_raises_on": singular, "no_evidence_cases": empty,
            "pairs_compared": int(sum(len(w.get("mls", [])) for w in want)), "max_abs_diff_ml_terms": worst,
            "equal_pair_length_ties_resolved_differently": near_ties,
            "seed": seed}


def main():
    res = campaign(*[int(x) for x in sys.argv[1:3]])
    from tredparse_amd import _lib
    res["library"] = _lib.version()
    print(json.dumps(res))
    return 1 if res["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
