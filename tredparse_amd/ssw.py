"""`ssw.Aligner` with the reference's signature (src/ssw_wrap.py:110-143, 177-227) on top of the GPU
kernel: one alignment = one read against a plain reference registered as a max_units = 0 ladder.
Meant for tests and spot checks -- the product path batches whole ladders (bam_parser / engine)."""
import numpy as np

from . import _lib


class PyAlignRes(object):
    def __init__(self, rec, query_seq, ref_seq):
        self.score, self.ref_begin, self.ref_end, self.query_begin, self.query_end = (int(x) for x in rec[:5])
        self.score2 = None
        self.ref_seq, self.query_seq = ref_seq, query_seq


class Aligner(object):
    def __init__(self, ref_seq="", match=2, mismatch=2, gap_open=3, gap_extend=1, report_secondary=False,
                 report_cigar=False, ctx=None):
        self.ref_seq = ref_seq
        self.match, self.mismatch, self.gap_open, self.gap_extend = match, mismatch, gap_open, gap_extend
        self.ctx = ctx or _lib.Context(0)

    def align(self, query_seq, min_score=0, min_len=0):
        ctx = self.ctx
        ctx.set_ladders([(self.ref_seq, "A", "", 0)])
        packed, woff, rlen = _lib.pack_reads([query_seq])
        tag = np.zeros(1, np.uint8); h = np.zeros(1, np.int16); sc = np.zeros(1, np.int16)
        dump = np.zeros((1, 1, 6), np.int16)
        p = _lib.SwParams(self.match, self.mismatch, self.gap_open, self.gap_extend, 9, 0, 0, 0)
        ctx.sw_classify(_lib.MEM_HOST, packed, woff, rlen, 1, np.array([0, 1], np.int32), np.zeros(1, np.int32), 1, p,
                        tag, h, sc, dump, 1)
        rec = dump[0, 0]
        score, match_len = int(rec[0]), int(rec[4]) - int(rec[3]) + 1
        if score >= min_score and match_len >= min_len:   # ssw_wrap.py:214-220
            return PyAlignRes(rec, query_seq, self.ref_seq)
        return None
