"""CPU: the numpy likelihood oracle against golden vectors produced by the reference's models.py."""
import numpy as np
import pytest

from oracle import lik_oracle as lo
from tests.gridcases import load_cases, oracle_caller

CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_grid_matches_reference(case):
    exp = case["expected"]
    l = case["locus_rec"]
    period = len(l["repeat"])
    if exp["raised"]:
        with pytest.raises(Exception) as ei:
            oracle_caller(case).evaluate()
        assert type(ei.value).__name__ == exp["raised"]
        return
    res = oracle_caller(case).evaluate()
    if exp["alleles"] == [-1, -1]:
        assert res["status"] == 1
        return
    mls = np.asarray(res["mls"], float)
    want = case["mls"]
    assert mls.shape == want.shape == (exp["n_pairs"], 6)
    assert np.array_equal(mls[:, :2], want[:, :2])                    # same pairs, same order
    assert np.abs(mls[:, 2:] - want[:, 2:]).max() <= 1e-9             # per-term log-likelihoods
    assert sorted(x // period for x in res["alleles"]) == exp["alleles"]
    assert "{}-{}|{}-{}".format(*res["CI"]) == exp["CI"]
    pp = lo.calc_PP(res["tot"], res["lik"], period, l["cutoff_risk"], l["mutation_nature"] == "increase",
                    l["inheritance"][-1] == "R")
    assert abs(pp - exp["PP"]) <= 1e-12
    assert lo.calc_label(exp["alleles"], l["cutoff_prerisk"], l["cutoff_risk"], l["mutation_nature"] == "increase",
                         l["inheritance"][-1] == "R") == exp["label"]
    for name in ("P_h1", "P_h2", "P_h1h2"):
        got = lo.sparsify(res[name], period)
        assert set(got) == set(exp[name]), name
        for k in got:
            assert abs(got[k] - exp[name][k]) <= 1e-12
    if case["kde"] is not None:
        assert np.abs(lo.kde_pdf(case["global_lens"]) - case["kde"]).max() <= 1e-15
