"""Host front end: from a BAM file to the packed arrays the GPU batch consumes.

What the reference does read by read in Python through pysam (tredparse/bam_parser.py: BamParser.parse :184-257,
PEextractor :316-369, BamReadLen :372-391, BamDepth :394-429) happens here in ONE native call per sample
(libtredbam.so ``tredbam_scan``, include/tredbam.h): window depth, read selection incl. unmapped mates and the
alternative-locus mate rescue, paired-end lengths -- for every requested locus, with the selected reads already in
libtredgpu's 2-bit layout.  No per-record Python object exists on this path; a `SampleScan` is a handful of numpy
arrays and can be produced by worker threads (the native call releases the GIL).

The per-locus classes of the reference -- ``BamParser(inputParams)`` with ``counts / details / rept``,
``PEextractor(bp)`` with ``global_lens / target_lens / MINPE / ref``, ``BamDepth``, ``BamReadLen`` -- are kept as
thin single-locus views of the same scan for callers and tests that address one locus at a time.
"""
import logging
import os
from collections import Counter
from collections.abc import Sequence

import numpy as np

from . import _lib, bamio

SPAN = 1000                 # half width of the fetch window around the tract, and the pair-length cap
FLANKMATCH = 9              # bases of flank an alignment must reach to count as anchored
DNAPE_ELONGATE = 10 * SPAN  # pairs are collected this far on either side of the tract
MAX_READ_LEN = 480          # longest read the SW kernel holds (include/tredgpu.h)
MAX_TEMPLATE_LEN = 511      # longest template (prefix + repeat * max_units + suffix)
_Y_SKIP = frozenset((1, 4, 6, 7, 10, 11, 13, 16, 18, 19))   # rows of the chrY table that still attract reads
_PKG = os.path.dirname(os.path.abspath(__file__))
_SEQ4 = np.frombuffer(b"=ACMGRSVTWYHKDBN", np.uint8)
_log = logging.getLogger("tredparse_amd.bam")

_RC = str.maketrans("ACGTNXacgtnx", "TGCANXtgcanx")


def rc(seq):
    """Reverse complement (N and X stay)."""
    return seq.translate(_RC)[::-1]


def open_bam(path):
    """The native reader of one BAM (its own file handle, block cache and index: one per thread)."""
    if path.endswith(".cram"):
        raise ValueError("CRAM input needs htslib; this front end reads BAM only")
    if bamio._native() is None:
        raise RuntimeError("tredparse_amd/libtredbam.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    return bamio.NativeAlignmentFile(path)


def read_alignment(path):
    """pysam.AlignmentFile-like reader (bamio) -- for code that still walks records."""
    if path.endswith(".cram"):
        raise ValueError("CRAM input needs htslib; this front end reads BAM only")
    return bamio.AlignmentFile(path, "rb")


class SampleScan(object):
    """Everything read from one BAM for a list of loci.

    per locus (index k, same order as `names`): unit[k] (bamio.SCAN_UNIT_DTYPE), depth[k], ploidy[k]
    per read: packed words at word_off, read_len, seq4 at seq4_off, name at name_off, name_id
    pools: global_lens, target_lens (unit[k] holds first / count)
    """
    __slots__ = ("path", "names", "loci", "readlen", "gender", "ydepth", "unit", "depth", "ploidy", "dropped",
                 "packed", "word_off", "read_len", "seq4", "seq4_off", "name_blob", "name_off", "name_id",
                 "global_lens", "target_lens", "opened", "_text", "device")

    def reads_of(self, k):
        u = self.unit[k]
        return int(u["read_first"]), int(u["read_first"]) + int(u["n_reads"])

    def sequence(self, i):
        """Read i as the string the BAM record decodes to.  The whole 4-bit pool is decoded once, on first use
        (every record starts on a byte, so read i is text[2 * seq4_off[i] :][:read_len[i]])."""
        text = getattr(self, "_text", None)
        if text is None:
            nib = np.empty(2 * len(self.seq4), np.uint8)
            nib[0::2], nib[1::2] = self.seq4 >> 4, self.seq4 & 15
            text = self._text = _SEQ4[nib].tobytes().decode("ascii")
        a = 2 * int(self.seq4_off[i])
        return text[a:a + int(self.read_len[i])]

    def name(self, i):
        return self.name_blob[int(self.name_off[i]):int(self.name_off[i + 1])].decode()

    def pair_lengths(self, k):
        u = self.unit[k]
        g0, t0 = int(u["global_first"]), int(u["target_first"])
        return self.global_lens[g0:g0 + int(u["n_global"])], self.target_lens[t0:t0 + int(u["n_target"])]


_y_region_cache = {}


def y_regions(build):
    """(contig, start, end) of the first five usable single-copy chrY regions of the build's table."""
    if build not in _y_region_cache:
        table = os.path.join(_PKG, "data", "chrY.{}.unique_ccn.tsv".format(build.split("_")[0]))
        out = []
        with open(table) as fp:
            for i, line in enumerate(fp):
                if i in _Y_SKIP:
                    continue
                if len(out) == 5:
                    break
                contig, lo, hi = line.split()[:3]
                out.append((contig, int(lo), int(hi)))
        _y_region_cache[build] = out
    return _y_region_cache[build]


def _y_depth(f, build):
    """Median pileup depth of the first five usable single-copy chrY regions (sex inference)."""
    depths = [f.pileup_depth_sum(contig, lo, hi) / float(hi - lo + 1) for contig, lo, hi in y_regions(build)]
    return float(np.median(depths))


_site_cache = {}


def _site_arrays(repo, names, loci, f):
    """The tredbam_site / tredbam_region arrays of a locus list for a file's contig table -- the same for every BAM
    of a cohort, so built once per (locus table, locus list, contig names)."""
    key = (id(repo), repo.ref, tuple(names), tuple(f.references))
    hit = _site_cache.get(key)
    if hit is None:
        sites = np.zeros(len(loci), bamio.SITE_DTYPE)
        regions = []
        strip = "nochr" in repo.ref       # the ALT table names contigs chrN in every build
        for k, t in enumerate(loci):
            mine = [(f.tid(c[3:] if strip else c), a, b) for c, a, b in t.alt]
            sites[k] = (f.tid(t.chr), t.repeat_start, t.repeat_end, len(regions), len(mine))
            regions += mine
        hit = (sites, np.array(regions, bamio.REGION_DTYPE) if regions else np.zeros(0, bamio.REGION_DTYPE))
        if len(_site_cache) > 64:
            _site_cache.clear()
        _site_cache[key] = hit
    return hit


def walk_need(coffset, host, results, alt_need=None):
    """Which planned blocks of a sample a scan with walked pair lengths reads (bamio plan_blocks / WALK_RESULT_DTYPE):
    those flagged `host` (alternative loci, extra regions) and, per walked site, the blocks between the virtual offsets
    of its window's first record and the end of its last one.  alt_need: the alternative loci were walked elsewhere too
    -- of their blocks only those flagged there (they hold the records that count) are read."""
    if alt_need is None:
        need = (np.asarray(host) != 0).astype(np.uint8)
    else:
        need = (((np.asarray(host) & 2) != 0) | (np.asarray(alt_need) != 0)).astype(np.uint8)
    ok = (results["status"] == 0) & (results["n_window"] > 0)
    lo = np.searchsorted(coffset, (results["win_vbeg"][ok] >> np.uint64(16)).astype(np.int64), side="left")
    hi = np.searchsorted(coffset, ((results["win_vend"][ok] - np.uint64(1)) >> np.uint64(16)).astype(np.int64), side="right")
    for a, b in zip(lo, hi):
        need[a:b] = 1
    return need


def scan_sample(path, repo, names, clip=False, alts=True, readlen=None, want_sex=None, handle=None, pe=None, alt=None):
    """Read one sample: sex and read length, then depth / reads / pair lengths of every locus in `names`.
    Never raises for a bad file: `opened` is False and nothing else is filled (the reference returns a result
    with only inferredGender / depthY for such a sample).  pe: (results, global pool, target pool) of the pair walks
    done where the blocks were inflated, alt: those of the walks over the alternative loci (bamio scan)."""
    s = SampleScan()
    s.path, s.names, s.loci = path, list(names), [repo[n] for n in names]
    s.gender, s.ydepth, s.readlen, s.opened = "Unknown", -1, 150, False
    try:
        f = handle or open_bam(path)
    except (IOError, ValueError) as e:
        _log.error("Cannot retrieve file `%s` (%s)", path, e)
        return s
    s.opened = True
    if want_sex is None:
        want_sex = any(t.is_xlinked for t in s.loci)
    if want_sex:
        try:
            s.ydepth = _y_depth(f, repo.ref)
            s.gender = "Male" if s.ydepth > 1 else "Female"
        except Exception:          # no chrY, no index ...: sex stays unknown
            pass
    if readlen is not None:
        s.readlen = int(readlen)
    else:
        try:
            s.readlen = f.max_read_len(101)
        except Exception:
            pass
    sites, regions = _site_arrays(repo, s.names, s.loci, f)
    s.unit, pools = f.scan(sites, regions, s.readlen, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN,
                           use_alts=alts and not clip, **({"pe": pe, "alt": alt} if pe is not None else {}))
    s.packed, s.word_off, s.read_len = pools["packed"], pools["word_off"], pools["read_len"]
    s.seq4, s.seq4_off = pools["seq4"], pools["seq4_off"]
    s.name_blob, s.name_off, s.name_id = pools["names"], pools["name_off"], pools["name_id"]
    s.global_lens, s.target_lens = pools["global_lens"], pools["target_lens"]
    window = np.array([t.repeat_end + SPAN - max(0, t.repeat_start - SPAN) + 1 for t in s.loci], np.float64)
    s.depth = np.where(s.unit["depth_status"] == 0, s.unit["depth_sum"] / window, 30.0)
    s.ploidy = np.array([1 if (s.gender == "Male" and t.is_xlinked) else t.ploidy for t in s.loci], np.int32)
    admit(s)
    if handle is None:
        f.close()
    return s


def admit(s):
    """Fill s.dropped = {locus index: reason} with the units that cannot go to the GPU batch: an unreadable file, a
    pair-length extraction the reference would have died in, a read beyond the SW kernel's length limit, a template
    ladder beyond its column limit.  Each costs only its own unit (the reference, too, loses just the failing
    locus, tred.py:245-249); everything else of the sample is genotyped."""
    s.dropped = {}
    for k, (t, u) in enumerate(zip(s.loci, s.unit)):
        if u["depth_status"] != 0:
            _log.error("Exception on `%s` %s (depth query failed). Set depth=30", s.path, t.name)
        if u["status"] & bamio.UNIT_NO_FETCH:
            _log.error("No reads extracted for region %s:%d-%d", t.chr, max(0, t.repeat_start - SPAN), t.repeat_end + SPAN)
        a, b = s.reads_of(k)
        longest = int(s.read_len[a:b].max()) if b > a else 0
        why = None
        if u["status"] & bamio.UNIT_FAILED:
            why = "the BAM could not be read"
        elif u["status"] & bamio.UNIT_NO_SEQ:
            why = "a selected read without a sequence (SEQ '*': len(None) in the reference)"
        elif u["pe_status"] != 0:
            why = "a paired read without an alignment end (pair-length extraction)"
        elif longest > MAX_READ_LEN:
            why = "a {} bp read: the SW kernel holds reads up to {} bp".format(longest, MAX_READ_LEN)
        elif len(t.prefix) + t.period * -(-s.readlen // t.period) + len(t.suffix) > MAX_TEMPLATE_LEN:
            why = "template ladder longer than {} columns".format(MAX_TEMPLATE_LEN)
        if why:
            _log.error("Exception on `%s` %s (%s)", s.path, t.name, why)
            s.dropped[k] = why
    return s.dropped


# ---- what becomes of the per-read results ---------------------------------------------------------------------
class Details(Sequence):
    """A locus' `details` -- [{"tag", "h", "id", "seq"}] for every read with a usable tag, in BAM order -- held as
    index arrays into the scan's pools.  Behaves like that list (len, indexing, iteration, == against a list build
    the dicts on first use); json_text() is the list as tred.to_json prints it, written natively from the pools
    without a Python object per read (the driver's formatting time was mostly these thousands of small strings)."""
    __slots__ = ("scan", "reads", "tags", "hs", "_items")

    def __init__(self, scan, reads, tags, hs):
        self.scan, self.reads, self.tags, self.hs, self._items = scan, reads, tags, hs, None

    def items(self):
        if self._items is None:
            names, s = _lib.TAG_NAMES, self.scan
            self._items = [{"tag": names[t], "h": h, "id": s.name(i), "seq": s.sequence(i)}
                           for t, h, i in zip(self.tags.tolist(), self.hs.tolist(), self.reads.tolist())]
        return self._items

    def __len__(self):
        return len(self.reads)

    def __getitem__(self, i):
        return self.items()[i]

    def __iter__(self):
        return iter(self.items())

    def __eq__(self, other):
        if isinstance(other, Details):
            other = other.items()
        return self.items() == other if isinstance(other, list) else NotImplemented

    def __ne__(self, other):
        eq = self.__eq__(other)
        return eq if eq is NotImplemented else not eq

    __hash__ = None

    def __repr__(self):
        return repr(self.items())

    def json_text(self):
        """The list at nesting depth 2 of json.dumps(..., sort_keys=True, indent=4), or None (generic encoder)."""
        s = self.scan
        return bamio.details_json(s.seq4, s.seq4_off, s.read_len, s.name_blob, s.name_off, self.reads, self.tags, self.hs)


def tally(scan, k, tags, hs, repeatpairs=True, lazy=False):
    """The reference's bookkeeping for one locus (bam_parser.py:174-182, 248-287) from the kernel's per-read
    (tag, h): returns (counts, details, rept).  counts["PREF"] and counts["POST"] are ONE Counter (reads anchored
    on either flank are pooled), counts["HANG"] counts every read that aligned somewhere; details lists the reads
    with a usable tag in BAM order; with repeatpairs off every read whose name carries two or more REPT records is
    removed before counting.  lazy: details as a Details view of the pools instead of the list of dicts."""
    a, b = scan.reads_of(k)
    tags, hs = np.asarray(tags), np.asarray(hs)
    hit = np.nonzero(tags != _lib.TAG_NONE)[0]
    keep = hit[tags[hit] != _lib.TAG_HANG]
    if not repeatpairs and len(keep):
        ids = scan.name_id[a:b][keep]
        rept_ids = ids[tags[keep] == _lib.TAG_REPT]
        twice = np.nonzero(np.bincount(rept_ids, minlength=int(ids.max()) + 1) > 1)[0]
        keep = keep[~np.isin(ids, twice)]
    kt, kh = tags[keep], hs[keep]
    flank = Counter(kh[(kt == _lib.TAG_PREF) | (kt == _lib.TAG_POST)].tolist())
    counts = {"FULL": Counter(kh[kt == _lib.TAG_FULL].tolist()), "PREF": flank, "POST": flank,
              "REPT": Counter(kh[kt == _lib.TAG_REPT].tolist()), "HANG": Counter(hs[hit].tolist())}
    details = Details(scan, keep.astype(np.int64) + a, kt.astype(np.uint8), kh.astype(np.int32))
    if not lazy:
        details = details.items()
    return counts, details, sum(counts["REPT"].values())


# ---- single-locus views with the reference's class names ------------------------------------------------------
class InputParams(object):
    """Parameters of one sample x locus run (the reference's utils.InputParams): positional core, free-form extras
    (`maxinsert`, `fullsearch`, `log`) in .kwargs."""

    def __init__(self, bam, READLEN, repo, tredName, gender="Unknown", depth=30, clip=False, alts=True,
                 repeatpairs=False, **kwargs):
        self.bam, self.READLEN, self.tredName, self.gender, self.depth = bam, READLEN, tredName, gender, depth
        self.clip, self.alts, self.repeatpairs = clip, alts, repeatpairs
        self.tred, self.ref, self.repo, self.kwargs = repo.get(tredName), repo.ref, repo, kwargs

    def getLogLevel(self, default="INFO"):
        return getattr(logging, str(self.kwargs.get("log", default)).upper(), logging.INFO)


class BamParser(object):
    """One locus of one BAM.  parse() = scan + one GPU classification + tally; collect() only scans and returns the
    selected (name, sequence) pairs in the order the reference would align them."""

    def __init__(self, inputParams):
        p = self.inputParams = inputParams
        t = self.tred = p.tred
        self.bam, self.gender, self.depth, self.READLEN = p.bam, p.gender, p.depth, p.READLEN
        self.clip, self.alts, self.repeatpairs, self.ref = p.clip, p.alts, p.repeatpairs, p.ref
        self.logger = logging.getLogger("BamParser")
        self.logger.setLevel(p.getLogLevel())
        self.chr, self.repeat, self.alt = t.chr, t.repeat, t.alt
        self.startRepeat, self.endRepeat = t.repeat_start, t.repeat_end
        self.referenceLen = t.repeat_end - t.repeat_start + 1
        self.fullPrefix, self.fullSuffix = t.prefix, t.suffix
        self.period = self.repeatSize = len(t.repeat)
        self.max_units = -(-self.READLEN // self.period)
        self.ploidy = 1 if (self.gender == "Male" and t.is_xlinked) else t.ploidy
        self.scan = None
        self.details, self.rept = [], 0
        flank = Counter()
        self.counts = {"FULL": Counter(), "PREF": flank, "POST": flank, "REPT": Counter(), "HANG": Counter()}

    def _scan(self):
        if self.scan is None:
            self.scan = scan_sample(self.bam, self.inputParams.repo, [self.tred.name], clip=self.clip, alts=self.alts,
                                    readlen=self.READLEN, want_sex=False)
            if not self.scan.opened:
                raise IOError("cannot read `{}`".format(self.bam))
            if 0 in self.scan.dropped:
                raise ValueError(self.scan.dropped[0])
        return self.scan

    def collect(self):
        s = self._scan()
        a, b = s.reads_of(0)
        self.reads = [(s.name(i), s.sequence(i)) for i in range(a, b)]
        return self.reads

    def finish(self, tags, hs):
        self.counts, self.details, self.rept = tally(self._scan(), 0, tags, hs,
                                                     repeatpairs=self.repeatpairs or self.clip)

    def parse(self, engine=None):
        from .engine import Engine, PackedUnits
        s = self._scan()
        engine = engine or Engine()
        tags, hs, _ = engine.classify_packed(PackedUnits.from_scans([(s, [0])], clip=self.clip))
        self.finish(tags, hs)


class PEextractor(object):
    """Pair lengths around one locus: global_lens (pairs not spanning the tract), target_lens (spanning pairs)."""

    def __init__(self, bp):
        s = bp._scan()
        g, t = s.pair_lengths(0)
        self.global_lens, self.target_lens = g.tolist(), t.tolist()
        self.ref = bp.referenceLen
        self.MINPE = bp.endRepeat - bp.startRepeat + 2 * FLANKMATCH + 2


class BamReadLen(object):
    def __init__(self, bamfile, logger=None):
        self.bamfile = bamfile

    @property
    def readlen(self):
        f = open_bam(self.bamfile)
        try:
            return f.max_read_len(101)
        finally:
            f.close()


class BamDepth(object):
    def __init__(self, bamfile, ref, logger=None):
        self.bamfile, self.ref = bamfile, ref

    def region_depth(self, chr, start, end, verbose=False):
        f = open_bam(self.bamfile)
        try:
            return f.pileup_depth_sum(chr, start, end) / float(end - start + 1)
        finally:
            f.close()

    def get_Y_depth(self, N=5):
        f = open_bam(self.bamfile)
        try:
            return _y_depth(f, self.ref)
        finally:
            f.close()
