"""GPU: run() under every non-default option against the REFERENCE's run() under the same options
(tests/golden/run_flags.json, tools/gen_golden.py `flags`: tredparse/tred.py:180-278 through tools/refshim.py).

--useclippedreads, --noalts, --norepeatpairs, --haploid, --fullsearch and --maxinsert as they reach run()
(tred.py:501-512) and repo.set_ploidy (tred.py:492; meta.py set_ploidy).  Three of them change nothing on the
reference's two mini-BAMs (no ALT-region reads, no pairs of repeat-only records), so most cases run on
tests/golden/bam/synf.bam -- one synthetic sample with expanded alleles, mates mismapped into ALT regions, unmapped
mates inside the tract and secondary copies of repeat-only reads -- where every flag moves the call, the read counts
or the intervals.  All keys of the result are compared: ints and strings exactly, floats to 1e-9."""
import gzip
import json
import os
import shutil

import pytest

from tredparse_amd import tred as tredmod
from tredparse_amd.engine import Engine
from tredparse_amd.meta import TREDsRepo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "run_flags.json")))["cases"]


def _close(a, b, tol=1e-9):
    if isinstance(a, float) or isinstance(b, float):
        return abs(float(a) - float(b)) <= tol * max(1.0, abs(float(b)))
    return a == b


@pytest.fixture(scope="module")
def engine():
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def bams(tmp_path_factory):
    """sample -> BAM path; synf.bam's index is stored gzipped and unpacked next to a copy of the BAM."""
    d = tmp_path_factory.mktemp("flagbams")
    shutil.copy(os.path.join(GOLD, "bam", "synf.bam"), str(d / "synf.bam"))
    with gzip.open(os.path.join(GOLD, "bam", "synf.bam.bai.gz"), "rb") as src, open(str(d / "synf.bam.bai"), "wb") as dst:
        shutil.copyfileobj(src, dst)
    return {"synf": str(d / "synf.bam"), "t001": os.path.join(GOLD, "bam", "t001.bam"),
            "t002": os.path.join(GOLD, "bam", "t002.bam")}


def _compare(got, want, where):
    assert set(got) == set(want), where
    for k in sorted(want):
        w, g = want[k], got[k]
        if k.endswith(".details"):
            assert g == w, (where, k)
        elif isinstance(w, dict):
            assert set(g) == set(w), (where, k)
            for kk in w:
                assert _close(g[kk], w[kk]), (where, k, kk, g[kk], w[kk])
        else:
            assert _close(g, w), (where, k, g, w)


@pytest.mark.parametrize("case", CASES, ids=["{}-{}".format(c["name"], c["sample"]) for c in CASES])
def test_run_under_flags_matches_reference(engine, bams, case):
    repo = TREDsRepo(ref="hg38", sites=os.path.join(GOLD, "no_sites"))
    repo.set_ploidy(case["haploid"])
    arg = (case["sample"], bams[case["sample"]], repo, list(case["loci"]), case["maxinsert"], case["fullsearch"],
           case["clip"], case["alts"], case["repeatpairs"], "INFO")
    got = tredmod.run(arg, engine=engine)["tredCalls"]
    _compare(got, case["tredCalls"], (case["name"], case["sample"]))


def test_the_flags_do_change_the_synthetic_sample():
    """Guard against a golden that pins nothing: on synf every option moves some entry away from the default run."""
    by = {c["name"]: c["tredCalls"] for c in CASES if c["sample"] == "synf"}
    for name in by:
        if name != "default":
            assert any(by[name][k] != by["default"][k] for k in by["default"] if not k.endswith(".details")), name


def test_cli_flags_reach_run(engine, bams, tmp_path, monkeypatch):
    """The same through main(): --haploid / --norepeatpairs / --noalts / --maxinsert on the command line
    (tred.py:64-113, 501-512) give the JSON the reference's run() gives for those options."""
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr("tredparse_amd.engine.Engine", lambda *a, **k: engine)
    want = {c["name"]: c for c in CASES if c["sample"] == "synf"}
    for name, flags in (("noalts_norepeatpairs", ["--noalts", "--norepeatpairs"]),
                        ("haploid", ["--haploid", "chrX", "--haploid", "chr19"]),
                        ("fullsearch_maxinsert60", ["--fullsearch", "--maxinsert", "60"]),
                        ("clip", ["--useclippedreads"])):
        work = tmp_path / name
        argv = [bams["synf"], "--workdir", str(work)] + flags
        for t in want[name]["loci"]:
            argv += ["--tred", t]
        tredmod.main(argv, quiet=True)
        got = json.load(open(work / "synf.json"))["tredCalls"]
        _compare(got, want[name]["tredCalls"], name)


# (seed, loci or None = the 30 distinct ones, SynthParams keywords, alt_rate) of tools/gen_golden.py SYN_SAMPLES
SYN_SAMPLES = {
    "synall": (None, dict(coverage=12.0, expanded_max=120, expanded_frac=0.25), 0.3),
    "syn100": (["HD", "DM1", "SCA10", "DM2", "ULD", "FXS", "SCA36", "FRDA", "OPMD", "SCA3"],
               dict(coverage=20.0, readlen=100, ins_mean=300.0, ins_sd=50.0, expanded_max=90, expanded_frac=0.3), 0.3),
    "syn250": (["HD", "DM1", "SCA10", "DM2", "ULD", "FXS", "SCA36", "FRDA", "OPMD", "SCA3"],
               dict(coverage=10.0, readlen=250, ins_mean=550.0, ins_sd=80.0, max_units=75, expanded_max=150, expanded_frac=0.3), 0.3),
    "syn100x": (["HD", "DM1", "SCA1", "FXS"],
                dict(coverage=100.0, min_units=42, max_units=60, expanded_max=200, expanded_frac=0.8), 0.4),
    # whole-genome-shaped: background reads over the alternative regions' index windows and the chrY depth windows
    "synwgs": (["HD", "DM1", "FXS", "SCA10"], dict(coverage=10.0, expanded_max=120, expanded_frac=0.5, wgs_like=True), 0.4),
}


@pytest.mark.parametrize("name", sorted(SYN_SAMPLES))
def test_synthetic_samples_match_reference(engine, tmp_path, name):
    """The reference's run() (default flags) on synthetic samples with reads at EVERY listed locus
    (tests/golden/run_synall.json): all 30 distinct loci at 150 bp -- BASELINE configs[1] asks for all TRED loci, the
    reference's mini-BAMs cover one each --, ten loci at 100 bp and at 250 bp (READLEN taken from the file, other
    ladders and kernel instantiations), four loci at 100x with alleles up to 200 repeats (configs[4]).  The samples are
    regenerated here from their seeds; the record table's digest must equal the one the golden was made from.  Every
    key of the result is compared (`details` as id / tag / h)."""
    import hashlib
    import numpy as np
    from tredparse_amd import synth, synth_bam
    gold = json.load(open(os.path.join(GOLD, "run_synall.json")))["samples"][name]
    names, kw, alt_rate = SYN_SAMPLES[name]
    loci = synth_bam.bench_loci() if names is None else [l for l in synth.load_loci() if l["name"] in names]
    assert [l["name"] for l in loci] == gold["loci"]
    kw = dict(kw)
    wgs_like = kw.pop("wgs_like", False)
    recs, h_true = synth_bam.simulate_sample(gold["seed"], loci, synth.SynthParams(**kw), alt_rate=alt_rate, wgs_like=wgs_like)
    h = hashlib.sha256()
    for k in recs.FIELDS:
        h.update(np.ascontiguousarray(getattr(recs, k)).tobytes())
    assert h.hexdigest() == gold["records_sha256"], "the synthetic generator no longer reproduces the golden's sample"
    bam = str(tmp_path / (name + ".bam"))
    synth_bam.write_bam(bam, recs, sample=name, level=1)
    repo = TREDsRepo(ref="hg38", sites=os.path.join(GOLD, "no_sites"))
    got = tredmod.run((name, bam, repo, gold["loci"], 300, False, False, True, True, "INFO"), engine=engine)["tredCalls"]
    want = gold["tredCalls"]
    for k in list(got):
        if k.endswith(".details"):
            got[k] = [[d["id"], d["tag"], int(d["h"])] for d in got[k]]
    _compare(got, want, name)
    assert got["readLen"] == kw.get("readlen", 150)
    assert all(want[n + ".1"] > 0 for n in gold["loci"])            # evidence and a call at every locus
