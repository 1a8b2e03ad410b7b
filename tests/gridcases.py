"""Shared helpers for the likelihood-grid tests: golden cases -> oracle / C-ABI inputs."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_cases():
    cases = json.load(open(os.path.join(GOLD, "grid.json")))["cases"]
    arrays = np.load(os.path.join(GOLD, "grid.npz"))
    loci = {l["name"]: l for l in json.load(open(os.path.join(ROOT, "tredparse_amd", "data", "treds.json")))["loci"]}
    for i, c in enumerate(cases):
        c["mls"] = arrays["mls_{}".format(i)] if "mls_{}".format(i) in arrays else None
        c["kde"] = arrays["kde_{}".format(i)] if "kde_{}".format(i) in arrays else None
        c["locus_rec"] = loci[c["locus"]]
    return cases


def oracle_caller(case):
    from oracle import lik_oracle as lo
    l = case["locus_rec"]
    return lo.Caller(len(l["repeat"]), case["readlen"], case["ploidy"], case["depth"],
                     {int(k): v for k, v in case["full"].items()}, {int(k): v for k, v in case["partial"].items()},
                     case["rept"], case["global_lens"], case["target_lens"], case["ref_len"], case["minpe"],
                     maxinsert=case["maxinsert"], fullsearch=case["fullsearch"])
