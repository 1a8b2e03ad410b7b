"""`ssw.Aligner` with the reference's signature (src/ssw_wrap.py:110-143, 177-227) on top of the GPU kernel: one
alignment = one read against a plain reference registered as a max_units = 0 ladder.

For tests, spot checks and code that still thinks one (reference, read) pair at a time -- the product path batches
whole template ladders (bam_parser / engine) and never comes through here.  All Aligners of a process share ONE
private GPU context (created on first use, never the engine's: registering a one-template ladder replaces a
context's ladder table), the reference is re-registered only when it differs from the one registered last, and
`align_many` takes any number of reads per call.
"""
import numpy as np

from . import _lib

_shared = {"ctx": None, "registered": None}


def _context():
    if _shared["ctx"] is None:
        _shared["ctx"] = _lib.Context(0)
    return _shared["ctx"]


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
        self._own = ctx            # a caller-supplied context is used as is (and its ladders replaced)

    def _ready(self):
        ctx = self._own or _context()
        key = (id(ctx), self.ref_seq)
        if self._own is not None or _shared["registered"] != key:
            ctx.set_ladders([(self.ref_seq, "A", "", 0)])
            if self._own is None:
                _shared["registered"] = key
        return ctx

    def align_many(self, queries, min_score=0, min_len=0):
        """[PyAlignRes or None] for every query, one kernel launch."""
        queries = list(queries)
        n = len(queries)
        if n == 0:
            return []
        ctx = self._ready()
        packed, woff, rlen = _lib.pack_reads(queries)
        tag, h, sc = np.zeros(n, np.uint8), np.zeros(n, np.int16), np.zeros(n, np.int16)
        dump = np.zeros((n, 1, 6), np.int16)
        p = _lib.SwParams(self.match, self.mismatch, self.gap_open, self.gap_extend, 9, 0, 0, 0)
        ctx.sw_classify(_lib.MEM_HOST, packed, woff, rlen, n, np.array([0, n], np.int32), np.zeros(1, np.int32), 1, p,
                        tag, h, sc, dump, 1)
        out = []
        for q, rec in zip(queries, dump[:, 0]):
            keep = int(rec[0]) >= min_score and int(rec[4]) - int(rec[3]) + 1 >= min_len    # ssw_wrap.py:214-220
            out.append(PyAlignRes(rec, q, self.ref_seq) if keep else None)
        return out

    def align(self, query_seq, min_score=0, min_len=0):
        return self.align_many([query_seq], min_score, min_len)[0]
