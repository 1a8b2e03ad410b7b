"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (tredparse_amd) never does.  See sw_oracle.c / ladder_model.c / ref_driver.c for
what each entry restates (reference file:line cited there).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TAGS = ("", "FULL", "PREF", "POST", "REPT", "HANG")

_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(HERE, "liboracle.so")
    srcs = [os.path.join(HERE, f) for f in ("sw_oracle.c", "ladder_model.c")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    ref_missing = os.path.exists("/root/reference/src/ssw.c") and not os.path.exists(
        os.path.join(HERE, "_ref", "libref_driver.so"))
    if stale or ref_missing:
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return so


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_sw_pairs.argtypes = [_i8p, _i64p, _i8p, _i64p, _i32p, _i32p, C.c_int64,
                                         C.c_int, C.c_int, C.c_int, C.c_int, _i32p, C.c_int]
        _lib.oracle_classify_batch.argtypes = [_i8p, _i64p, _i32p, _i32p, C.c_int64, _i8p, _i64p, _i64p,
                                               _i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                               _i32p, C.c_int]
        _lib.oracle_build_ladder.argtypes = [_i8p, C.c_int, _i8p, C.c_int, _i8p, C.c_int, C.c_int,
                                             _i8p, _i64p]
        _lib.oracle_ladder_size.restype = C.c_int64
        _lib.oracle_ladder_size.argtypes = [C.c_int] * 4
        _lib.ladder_model_strand.argtypes = [_i8p, C.c_int, _i8p, C.c_int, _i8p, C.c_int, _i8p, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p]
    return _lib


def have_ref():
    return os.path.exists(os.path.join(HERE, "_ref", "libref_driver.so"))


def ref():
    """The reference's own ssw.c, compiled in place (oracle/_ref)."""
    global _ref
    if _ref is None:
        build()
        _ref = C.CDLL(os.path.join(HERE, "_ref", "libref_driver.so"))
        _ref.ref_sw_pairs.argtypes = lib().oracle_sw_pairs.argtypes
        _ref.ref_classify_batch.argtypes = lib().oracle_classify_batch.argtypes
    return _ref


_CODE = np.full(256, 4, np.int8)
for _c, _v in zip("ACGTacgt", (0, 1, 2, 3, 0, 1, 2, 3)):
    _CODE[ord(_c)] = _v


def encode(seq):
    """ssw_wrap.py:61,229-244 -- ACGT (either case) -> 0..3, anything else -> 4."""
    return _CODE[np.frombuffer(seq.encode("latin-1"), np.uint8)]


def csr(seqs):
    off = np.zeros(len(seqs) + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    codes = np.concatenate([encode(s) for s in seqs]) if seqs else np.zeros(0, np.int8)
    return np.ascontiguousarray(codes, np.int8), off


def rc(s):
    """bam_parser.py:448-450."""
    return s.translate(str.maketrans("ATCGatcgNnXx", "TAGCtagcNnXx"))[::-1]


def build_ladder(prefix, repeat, suffix, max_units):
    """bam_parser.py:84-100 -> list of (units, template string) in db order."""
    out = []
    for u in range(1, max_units + 1):
        t = prefix + repeat * u + suffix
        out.append((u, t))
        out.append((u, rc(t)))
    return out


def _pairs(fn, reads, refs, pair_read, pair_ref, scoring, threads):
    rc_, ro = csr(reads)
    tc, to = csr(refs)
    pr = np.ascontiguousarray(pair_read, np.int32)
    pt = np.ascontiguousarray(pair_ref, np.int32)
    out = np.zeros((len(pr), 5), np.int32)
    m, x, go, ge = scoring
    fn(rc_, ro, tc, to, pr, pt, len(pr), m, x, go, ge, out.reshape(-1), threads)
    return out


def sw_pairs(reads, refs, pair_read, pair_ref, scoring=(1, 5, 7, 2), threads=0):
    """[(score, ref_begin, ref_end, read_begin, read_end)] from the C restatement."""
    return _pairs(lib().oracle_sw_pairs, reads, refs, pair_read, pair_ref, scoring, threads)


REF_CRASHED = -9998   # oracle/ref_driver.c: the reference's ssw_align faulted on this pair (its CIGAR pass, ssw.c:549-633)


def _in_fresh_process(func, *args):
    """func(*args) of this module in a newly started interpreter (spawn): a process in which the reference has faulted
    once has run off its heap buffers and is not trusted with further reference results."""
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(1) as pool:
        return pool.apply(func, args)


def _ref_sw_pairs_once(reads, refs, pair_read, pair_ref, scoring, threads):
    return _pairs(ref().ref_sw_pairs, reads, refs, pair_read, pair_ref, scoring, threads)


def ref_sw_pairs(reads, refs, pair_read, pair_ref, scoring=(1, 5, 7, 2), threads=0):
    """Same, computed by the reference's compiled ssw.c (ssw_wrap.py:177-227 call pattern).  A pair the reference
    faults on comes back as five REF_CRASHED -- and every OTHER pair of that call is then computed again in a fresh
    process without the faulting ones: the fault is a heap overrun inside the reference (ref_driver.c), after which
    the results of the same process, other threads' included, are not evidence."""
    out = _ref_sw_pairs_once(reads, refs, pair_read, pair_ref, scoring, threads)
    crashed = out[:, 0] == REF_CRASHED
    while crashed.any() and not crashed.all():
        keep = np.nonzero(~crashed)[0]
        pr, pt = np.asarray(pair_read, np.int32)[keep], np.asarray(pair_ref, np.int32)[keep]
        again = _in_fresh_process(_ref_sw_pairs_once, list(reads), list(refs), pr, pt, tuple(scoring), threads)
        out[keep] = again
        more = np.zeros_like(crashed)
        more[keep] = again[:, 0] == REF_CRASHED
        if not more.any():
            break
        crashed |= more
    return out


class LocusSet:
    """Template ladders of several loci, flattened for the batch classifiers."""

    def __init__(self, loci):
        """loci: list of (prefix, repeat, suffix, max_units)."""
        tmpls, lad_off, periods = [], [0], []
        for prefix, repeat, suffix, max_units in loci:
            tmpls.extend(t for _, t in build_ladder(prefix, repeat, suffix, max_units))
            lad_off.append(len(tmpls))
            periods.append(len(repeat))
        self.codes, self.tmpl_off = csr(tmpls)
        # one trailing entry so every locus slice has its end offset
        self.ladder_tmpl_off = np.asarray(lad_off[:-1], np.int64)
        self.periods = np.asarray(periods, np.int32)
        self.max_units = np.asarray([l[3] for l in loci], np.int32)
        self.templates = tmpls
        self.lad_off = lad_off
        self.loci = [tuple(l) for l in loci]


def _classify(fn, reads, read_locus, locus_set, clip, scoring, threads):
    rc_, ro = csr(reads)
    rl = np.ascontiguousarray(read_locus, np.int32)
    mu = np.ascontiguousarray(locus_set.max_units[rl], np.int32)
    out = np.zeros((len(reads), 3), np.int32)
    m, x, go, ge = scoring
    fn(rc_, ro, rl, mu, len(reads), locus_set.codes, locus_set.tmpl_off, locus_set.ladder_tmpl_off,
       locus_set.periods, int(clip), m, x, go, ge, out.reshape(-1), threads)
    return out


def classify(reads, read_locus, locus_set, clip=False, scoring=(1, 5, 7, 2), threads=0):
    """Per read (tag, h, score) -- bam_parser.py:123-182 via the C restatement."""
    return _classify(lib().oracle_classify_batch, reads, read_locus, locus_set, clip, scoring, threads)


def _ref_classify_once(reads, read_locus, loci, clip, scoring, threads):
    return _classify(ref().ref_classify_batch, reads, read_locus, LocusSet(loci), clip, scoring, threads)


def ref_classify(reads, read_locus, locus_set, clip=False, scoring=(1, 5, 7, 2), threads=0):
    """Same, every alignment computed by the reference's compiled ssw.c (tag -1: the reference faulted on the read; the
    other reads of the call are then classified again in a fresh process, see ref_sw_pairs)."""
    out = _classify(ref().ref_classify_batch, reads, read_locus, locus_set, clip, scoring, threads)
    crashed = out[:, 0] == -1
    loci = getattr(locus_set, "loci", None)
    while loci is not None and crashed.any() and not crashed.all():
        keep = np.nonzero(~crashed)[0]
        again = _in_fresh_process(_ref_classify_once, [reads[i] for i in keep], np.asarray(read_locus, np.int32)[keep], loci,
                                  bool(clip), tuple(scoring), threads)
        out[keep] = again
        more = np.zeros_like(crashed)
        more[keep] = again[:, 0] == -1
        if not more.any():
            break
        crashed |= more
    return out


def ladder_model(read, prefix, repeat, suffix, max_units, scoring=(1, 5, 7, 2)):
    """CPU model of the kernel algorithm: (2*max_units, 5) results in db order (fwd, rc per u)."""
    r = encode(read)
    out = np.zeros((2 * max_units, 5), np.int32)
    m, x, go, ge = scoring
    for strand, (a, rep, b) in enumerate(((prefix, repeat, suffix), (rc(suffix), rc(repeat), rc(prefix)))):
        o = np.zeros((max_units, 5), np.int32)
        rcode = lib().ladder_model_strand(np.ascontiguousarray(r), len(r), encode(a), len(a), encode(rep),
                                          len(rep), encode(b), len(b), max_units, m, x, go, ge,
                                          o.reshape(-1))
        if rcode != 0:
            raise ValueError("ladder model limits exceeded")
        out[strand::2] = o
    return out
