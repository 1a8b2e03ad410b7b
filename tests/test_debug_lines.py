"""CPU: the text of the --log DEBUG lines (tred.debug_lines; the reference's bam_parser.py:177-178 and models.py:270-272).
The numbers behind them are compared with a capture of the reference's own loggers on the GPU (tests/test_e2e_gpu.py);
here: the layout -- Python 2's str(float), the literal 0 of a term the reference does not evaluate, the tag names."""
import json
import logging
import os

import numpy as np

from tredparse_amd import tred

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_python2_float_layout():
    assert tred._py2_str(-61.0) == "-61.0" and tred._py2_str(0.0) == "0.0" and tred._py2_str(1e-05) == "1e-05"
    assert tred._py2_str(-0.020661398520546232) == "-0.0206613985205"       # 12 significant digits
    assert tred._py2_str(-171.44913740846184) == "-171.449137408" and tred._py2_str(1e+16) == "1e+16"
    assert tred._py2_str(float("inf")) == "inf"


def test_lines_of_a_unit(caplog):
    want = json.load(open(os.path.join(GOLD, "debug_t001_HD.json")))

    class Locus(object):
        repeat = "CAG"

    class Scan(object):
        loci = [Locus()]
        seqs = [r[2] for r in want["reads"][:4]] + ["ACGT"]

        def reads_of(self, k):
            return 0, len(self.seqs)

        def sequence(self, i):
            return self.seqs[i]

    class Res(object):
        tags = np.array([2, 2, 1, 3, 0], np.uint8)          # PREF PREF FULL POST and an untagged read
        hs = np.array([7, 7, 15, 20, 0], np.int16)
        grid = np.array([[45.0, 45.0, -0.5, -123.25, -0.125, -75.0], [45.0, 123.0, -1.0, -98.0, -0.125, -72.5]])
        call = {"run_pe": 1}

    with caplog.at_level(logging.DEBUG):
        tred.debug_lines(Scan(), 0, Res())
    reads = [r.getMessage() for r in caplog.records if r.name == "BamParser"]
    pairs = [r.getMessage() for r in caplog.records if r.name == "IntegratedCaller"]
    assert reads == ["PREF: h=  7, seq=" + Scan.seqs[0], "PREF: h=  7, seq=" + Scan.seqs[1], "FULL: h= 15, seq=" + Scan.seqs[2],
                     "POST: h= 20, seq=" + Scan.seqs[3]]
    assert pairs == ["*** (15, 15) -0.5 -123.25 -0.125 -75.0 -198.875", "*** (15, 41) -1.0 -98.0 -0.125 -72.5 -171.625"]
    # a unit without spanning reads and without the paired-end term prints those terms as the reference's literal 0
    Res.tags = np.array([2, 3], np.uint8)
    Res.hs = np.array([7, 9], np.int16)
    Res.call = {"run_pe": 0}
    Scan.seqs = Scan.seqs[:2]
    caplog.clear()
    with caplog.at_level(logging.DEBUG):
        tred.debug_lines(Scan(), 0, Res())
    assert [r.getMessage() for r in caplog.records if r.name == "IntegratedCaller"][0] == "*** (15, 15) 0 -123.25 -0.125 0 -198.875"
