"""The one reference-published value the build disagrees with: README.md:76-86 prints DM1 = 5|62 for tests/t002.bam,
v0.7.8's own code gives 5|66.  tools/pin_pysam.py (build container, the reference's IntegratedCaller through
tools/refshim.py) swept every input that comes from pysam -- the local depth, the paired-end record set, the unmapped
reads of the window fetch -- and recorded the result in tests/golden/pin_pysam.json:

  * README's FR / PR / RR strings are reproduced exactly (with the unmapped reads: without them RR loses two reads and
    the call is 63-64), so the read evidence is the same;
  * allele 2 = 62 needs a depth of 68-72 (86-92 without the paired-end term); the file offers 43.6 (columns inside the
    window), 47.1 (htslib's default pileup: all columns of the overlapping reads) and at most 50.1 (no record filtered);
  * no reading of PEextractor's record set moves the call below 65.

So 62 is not reachable from v0.7.8's code over this file: the README line predates the code, and 66 stands.  What the
sweep does pin is the model's dependence on the depth -- replayed here against the oracle (CPU) and the kernels (GPU)."""
import json
import os

import numpy as np
import pytest

from tests.gridcases import GOLD, ROOT

REC = json.load(open(os.path.join(GOLD, "pin_pysam.json")))
LOCUS = [l for l in json.load(open(os.path.join(ROOT, "tredparse_amd", "data", "treds.json")))["loci"] if l["name"] == "DM1"][0]
SWEEP = [(d, a) for row in REC["depth_sweep"] if row["paired_end_term"] for d, a in row["calls"]]


def test_the_sweep_says_62_is_out_of_reach():
    assert REC["readme"] == [5, 62] and REC["baseline"]["alleles"] == [5, 66]
    assert REC["depth_truncated"] < REC["depth_all_columns"] < REC["depth_no_filter"] < 51
    depth_hits = [h for h in REC["hits_62"] if h["sweep"] == "depth"]
    assert depth_hits and len(depth_hits) == len(REC["hits_62"])         # nothing but a different depth gives 62 ...
    assert min(h["depth"] for h in depth_hits) >= 68 > REC["depth_no_filter"] and REC["reachable_62"] == []   # ... and no pileup has it
    assert all(v["alleles"][1] >= 65 for v in REC["pe_variants"])
    assert all(v["alleles"][1] in (63, 64) and sum(v["rept"].values()) == 9 for v in REC["no_unmapped_in_fetch"])
    assert REC["unit"]["rept"] == 11                                    # README's RR 49|3;50|8: the unmapped reads are in


def test_oracle_replays_the_depth_sweep():
    from oracle import lik_oracle as lo
    u = REC["unit"]
    for depth, alleles in SWEEP:
        res = lo.Caller(u["period"], u["readlen"], u["ploidy"], depth, {int(k): v for k, v in u["full"].items()},
                        {int(k): v for k, v in u["pref"].items()}, u["rept"], u["global_lens"], u["target_lens"], u["ref_len"],
                        u["minpe"]).evaluate()
        assert res["status"] == 0 and sorted(a // u["period"] for a in res["alleles"]) == alleles, depth


@pytest.mark.gpu
def test_kernels_replay_the_depth_sweep(ctx):
    from oracle import lik_oracle as lo
    from tredparse_amd import _lib, synth
    step, w = lo.load_model()
    ctx.set_model(np.array([step[p] for p in range(1, 7)]), np.array(w))
    u, n, hs = REC["unit"], len(SWEEP), 128
    units = np.zeros(n, _lib.UNIT_DTYPE)
    full, pref, rept = (np.zeros((n, hs), np.int32) for _ in range(3))
    gl, tl = np.asarray(u["global_lens"], np.int32), np.asarray(u["target_lens"], np.int32)
    for i, (depth, _) in enumerate(SWEEP):
        p = synth.unit_params_for(LOCUS, u["readlen"], depth, len(gl), len(tl), 0, 0, ploidy=u["ploidy"], maxinsert=300, fullsearch=False)
        p["ref_len"], p["minpe"] = u["ref_len"], u["minpe"]
        units[i] = p
        for k, v in u["full"].items():
            full[i, int(k)] = v
        for k, v in u["pref"].items():
            pref[i, int(k)] = v
        rept[i, 0] = u["rept"]
    calls = np.zeros(n, _lib.CALL_DTYPE)
    ctx.likelihood_grid(_lib.MEM_HOST, units, n, hs, full, pref, rept, gl, len(gl), tl, len(tl), calls, None, None, None, 0)
    for (depth, alleles), c in zip(SWEEP, calls):
        assert c["status"] == 0 and sorted([c["h1"] // u["period"], c["h2"] // u["period"]]) == alleles, depth


# ---- tools/pin_pysam_live.py: the one-command check for a maintainer who has pysam ---------------
def _live():
    import importlib.util
    spec = importlib.util.spec_from_file_location("pin_pysam_live", os.path.join(ROOT, "tools", "pin_pysam_live.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_live_pysam_check_skips_cleanly_without_pysam(tmp_path, capsys):
    live = _live()
    if live.import_pysam() is not None:
        pytest.skip("pysam is installed here: run tools/pin_pysam_live.py itself")
    out = str(tmp_path / "rec.json")
    assert live.main(["--json", out]) == 0
    assert "skipped" in capsys.readouterr().out and json.load(open(out))["skipped"] is True
    assert live.main(["--require"]) == 2


def test_live_pysam_check_compares_what_it_says(monkeypatch, capsys):
    """The comparison itself, with a stand-in pysam module built on this repository's pure-Python reader: identical as it
    is, and a fetch that drops the placed-unmapped reads / a pileup that truncates to the window are both found out."""
    import types
    from tredparse_amd import bamio
    live = _live()

    class Col(object):
        def __init__(self, pos, n):
            self.reference_pos, self.n = pos, n

    def make(drop_unmapped=False, truncate=False):
        class AlignmentFile(bamio.PyAlignmentFile):
            def fetch(self, *a, **k):
                for r in bamio.PyAlignmentFile.fetch(self, *a, **k):
                    if not (drop_unmapped and a and r.is_unmapped):
                        yield r

            def pileup(self, chrom, start, end):
                cols = live.column_sums(bamio.PyAlignmentFile.fetch(self, chrom, start, end), start, end)
                return [Col(c, n) for c, n in sorted(cols.items()) if not truncate or start <= c < end]
        mod = types.ModuleType("pysam")
        mod.AlignmentFile, mod.__version__ = AlignmentFile, "stand-in"
        return mod

    bam = os.path.join(GOLD, "bam", "t002.bam")
    monkeypatch.setattr(live, "import_pysam", lambda: make())
    assert live.main([bam]) == 0 and "identical" in capsys.readouterr().out
    monkeypatch.setattr(live, "import_pysam", lambda: make(drop_unmapped=True))
    assert live.main([bam]) == 1 and '"kind": "fetch/window"' in capsys.readouterr().out
    monkeypatch.setattr(live, "import_pysam", lambda: make(truncate=True))
    assert live.main([bam]) == 1 and '"kind": "pileup/sum"' in capsys.readouterr().out
