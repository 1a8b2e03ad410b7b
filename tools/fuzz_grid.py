#!/usr/bin/env python3
"""Randomised parity campaign for the likelihood grid: whole fused path (SW -> histograms -> grid) on random
synthetic batches (all loci, ploidy 1 and 2, coverage 5..100, maxinsert 60..400, expansions, few / many spanning
pairs) against the numpy oracle (oracle/lik_oracle.py, pinned to the reference's models.py by tests/golden/grid.*)
unit by unit: (h1, h2), CI, number of pairs, lik and PP.  Not part of the test suite (the oracle needs seconds per
big grid); prints one JSON line.

usage: python tools/fuzz_grid.py [rounds] [seed] [max_pairs_for_oracle]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def campaign(rounds=6, seed=1, max_pairs=20000):
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
    checked = skipped = bad = 0
    worst = 0.0
    t0 = time.time()
    for k in range(rounds):
        p = synth.SynthParams(coverage=float(rng.choice([5, 15, 30, 60, 100])), readlen=150,
                              min_units=int(rng.integers(1, 10)), max_units=int(rng.integers(20, 70)),
                              expanded_max=int(rng.choice([0, 120, 200])), expanded_frac=0.4)
        sel = [loci[i] for i in rng.permutation(len(loci))[:int(rng.integers(3, 10))]]
        maxinsert = int(rng.choice([60, 150, 300, 400]))
        b = synth.build_batch(int(rng.integers(1 << 30)), sel, int(rng.integers(1, 4)), p, maxinsert=maxinsert, workers=8)
        units = b.units.copy()
        units["ploidy"] = rng.choice([1, 2], len(units), p=[0.25, 0.75])
        units["fullsearch"] = rng.random(len(units)) < 0.15
        ctx.set_ladders(b.ladders)
        n, g, hs = b.n_reads, b.n_units, b.hist_stride
        tag = np.zeros(n, np.uint8); h = np.zeros(n, np.int16); sc = np.zeros(n, np.int16)
        full = np.zeros((g, hs), np.int32); pref = np.zeros((g, hs), np.int32); rept = np.zeros((g, hs), np.int32)
        calls = np.zeros(g, _lib.CALL_DTYPE)
        ctx.genotype_batch(_lib.MEM_HOST, b.packed, b.read_off, b.read_len, n, b.unit_read_off, b.unit_ladder, units, g,
                           _lib.default_sw_params(max_read_len=150), None, b.global_lens, len(b.global_lens),
                           b.target_lens, len(b.target_lens), tag, h, sc, hs, full, pref, rept, calls)
        for u in range(g):
            up, c = units[u], calls[u]
            if c["n_pairs"] > max_pairs:
                skipped += 1
                continue
            locus = sel[b.unit_ladder[u]]
            f = {i: int(v) for i, v in enumerate(full[u]) if v}
            pp = {i: int(v) for i, v in enumerate(pref[u]) if v}
            res = lo.Caller(int(up["period"]), 150, int(up["ploidy"]), 2 * float(up["half_depth"]), f, pp, int(rept[u].sum()),
                            b.global_lens[up["pe_off"]:up["pe_off"] + up["n_global"]],
                            b.target_lens[up["tl_off"]:up["tl_off"] + up["n_target"]], int(up["ref_len"]),
                            int(up["minpe"]), maxinsert=int(up["maxinsert"]), fullsearch=bool(up["fullsearch"])).evaluate()
            checked += 1
            if res["status"] == 1:
                ok = c["status"] == 1
            else:
                ppv = lo.calc_PP(res["tot"], res["lik"], int(up["period"]), locus["cutoff_risk"],
                                 locus["mutation_nature"] == "increase", locus["inheritance"][-1] == "R")
                worst = max(worst, abs(c["lik"] - res["lik"]), abs(c["pp"] - ppv))
                ok = (c["status"] == 0 and (c["h1"], c["h2"]) == tuple(res["alleles"]) and tuple(c["ci"]) == tuple(res["CI"])
                      and c["n_pairs"] == len(res["mls"]) and abs(c["lik"] - res["lik"]) <= 1e-6 and abs(c["pp"] - ppv) <= 1e-9
                      and bool(c["run_pe"]) == res["run_pe"])
            if not ok:
                bad += 1
                print("MISMATCH round", k, "unit", u, locus["name"], "ploidy", int(up["ploidy"]), c, res.get("alleles"),
                      res.get("CI"), res.get("lik"), file=sys.stderr)
    ctx.close()
    return {"units_checked": checked, "units_skipped_big": skipped, "mismatches": bad, "rounds": rounds,
            "seed": seed, "max_abs_diff_lik_or_pp": worst, "seconds": round(time.time() - t0, 1)}


def main():
    a = [int(x) for x in sys.argv[1:4]]
    res = campaign(*a)
    from tredparse_amd import _lib
    res["library"] = _lib.version()
    print(json.dumps(res))
    return 1 if res["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
