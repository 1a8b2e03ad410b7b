"""CPU: the reporter against the REFERENCE's own reporter (tests/golden/report.json: tredparse/tredreport.py main run
through tools/refshim.py on eight per-sample JSONs -- the reference's two run() results and edited copies reaching a
male at an X-linked locus, a pre-risk sample, carriers of both mutation natures, a case below --minPP, a long
evidence string, the AR exemption of the details file).  Every file the reference writes is compared byte for byte:
<tsv>, <tsv>.cases.txt, <tsv>.details.txt, <tsv>.report.txt (tredreport.py:36-141, 198-302)."""
import json
import os

import pytest

from tredparse_amd import tredreport

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "report.json")


@pytest.fixture(scope="module")
def golden():
    with open(GOLD) as fp:
        return json.load(fp)


@pytest.mark.parametrize("run", ["default", "columns_minpp", "two_reference_samples"])
def test_reporter_files_equal_the_references(golden, run, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    for key, calls in golden["inputs"].items():
        with open(key + ".json", "w") as fp:
            json.dump({"samplekey": key, "bam": key + ".bam", "tredCalls": calls}, fp)
    case = [r for r in golden["runs"] if r["name"] == run][0]
    tredreport.main(case["files"] + case["options"])
    tsv = case["options"][case["options"].index("--tsv") + 1]
    for suffix, want in case["outputs"].items():
        with open(tsv + suffix[len("tsv"):]) as fp:
            got = fp.read()
        assert got == want, (run, suffix)


def test_float_column_layout():
    """The float layout of the case tables (6 decimals, common trailing zeros dropped, scientific once an entry
    would print as 0), checked on the values of the golden and on the switch to scientific notation."""
    assert tredreport._float_cells([0.9999999999900089]) == ["1.0"]
    assert tredreport._float_cells([1.0, 0.42, 0.97]) == ["1.00", "0.42", "0.97"]
    assert tredreport._float_cells([0.5, 1e-9]) == ["5.000000e-01", "1.000000e-09"]
    assert tredreport._float_cells([0.123456789, 0.5]) == ["0.123457", "0.500000"]
