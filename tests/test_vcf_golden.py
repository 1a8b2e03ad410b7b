"""CPU: to_vcf against the text the REFERENCE's to_vcf writes (tests/golden/vcf_t001_t002.json: tredparse/tred.py
:281-293, :316-374 run through tools/refshim.py on the reference's own run() results of tests/t001.bam and
tests/t002.bam, all 32 loci; plus hg19_nochr coordinates with a three-locus call list).  Every line is compared byte
for byte except the two that carry the day and the install path (##fileDate, ##source), whose form is checked."""
import gzip
import json
import os
import re

import pytest

from tredparse_amd import tred as tredmod
from tredparse_amd.meta import TREDsRepo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case,sample,ref,treds", [
    ("t001", "t001", "hg38", None), ("t002", "t002", "hg38", None),
    ("t001_hg19_nochr_3loci", "t001", "hg19_nochr", ["HD", "DM1", "SCA1"])])
def test_vcf_text_equals_the_references(case, sample, ref, treds, tmp_path, monkeypatch):
    want = json.load(open(os.path.join(GOLD, "vcf_t001_t002.json")))["vcf"][case]
    calls = json.load(open(os.path.join(GOLD, "run_t001_t002.json")))["samples"][sample]
    repo = TREDsRepo(ref=ref, sites=os.path.join(GOLD, "no_sites"))
    key = sample if treds is None else "t001_hg19"
    results = {"samplekey": key, "bam": "tests/{}.bam".format(sample), "tredCalls": calls}
    monkeypatch.chdir(tmp_path)
    tredmod.to_vcf(results, ref, repo, treds=treds or list(repo.names))
    got = gzip.open(key + ".tred.vcf.gz", "rt").read().splitlines()
    assert len(got) == len(want)
    for g, w in zip(got, want):
        if w.startswith("##fileDate="):
            assert re.match(r"^##fileDate=\d{8}$", g)
        elif w.startswith("##source="):
            assert g.startswith("##source=") and g.endswith(" tests/{}.bam".format(sample))
        else:
            assert g == w
    records = [l for l in got if not l.startswith("#")]
    assert len(records) == (32 if treds is None else 3)
    assert records == sorted(records, key=lambda l: (l.split("\t")[0], int(l.split("\t")[1])))
