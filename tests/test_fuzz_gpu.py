"""GPU: fixed-seed slices of the randomised parity campaigns (tools/fuzz_*.py) so that every run of the suite
re-draws them: the long campaigns logged under profiles/ are the same code with more rounds.

  fuzz_parity     pruned production kernel path vs the reference's own compiled ssw.c (oracle/_ref) read by read:
                  (tag, h, score); plus the per-template dump field by field on adversarial units
  fuzz_selfcheck  production launch vs arg-max over the unpruned per-template dump (GPU only, many reads)
  fuzz_hist       adversarial grid inputs vs the numpy oracle: status, enumeration, four terms per pair, CI, PP
  fuzz_grid       SW -> histograms -> grid on random synthetic batches vs the numpy oracle
"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libssw.so")


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref (the compiled reference ssw.c) is not built")
def test_fuzz_parity_slice():
    import fuzz_parity
    res = fuzz_parity.campaign(rounds=80, seed=20270301)
    assert res["reads"] > 100000 and res["template_pairs"] > 200000
    assert res["mismatches"] == 0 and res["pair_mismatches"] == 0, res


def test_fuzz_selfcheck_slice():
    import fuzz_selfcheck
    res = fuzz_selfcheck.campaign(rounds=12, seed=20270302)
    assert res["reads"] > 100000
    assert res["mismatches"] == 0, res


def test_fuzz_hist_slice():
    import fuzz_hist
    res = fuzz_hist.campaign(cases_n=250, seed=20270303)
    assert res["pairs_compared"] > 100000 and res["cases_the_reference_raises_on"] > 0
    assert res["mismatches"] == 0 and res["max_abs_diff_ml_terms"] <= 1e-6, res


def test_fuzz_grid_slice():
    import fuzz_grid
    res = fuzz_grid.campaign(rounds=12, seed=20270304, max_pairs=6000)
    assert res["units_checked"] > 50
    assert res["mismatches"] == 0 and res["max_abs_diff_lik_or_pp"] <= 1e-6, res


def test_fuzz_inflate_slice():
    import fuzz_inflate
    res = fuzz_inflate.campaign(rounds=6, seed=20270305)
    assert res["streams"] == 1800 and res["intact"] == 1440 and res["damaged_refused"] > 100
    assert res["mismatches"] == 0, res
