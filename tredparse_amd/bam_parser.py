"""Host BAM front end mirroring tredparse/bam_parser.py -- same class names, attributes and read-selection
semantics, with the per-read Smith-Waterman loop (bam_parser.py:123-182) replaced by one batched call
into libtredgpu.so.

  BamParser.collect()   the fetch / filter part of parse() (bam_parser.py:184-243): which reads go to SW
  BamParser.finish()    what _parseReadSW + tally_counts + rept do with the per-read (tag, h) results
                        (bam_parser.py:174-182, 248-257, 259-287)
  BamParser.parse()     = collect + Engine.classify + finish, for single-unit use as in the reference
  PEextractor           bam_parser.py:316-369      BamReadLen :372-391      BamDepth :394-429
"""
import logging
import math
import os
from collections import defaultdict

import numpy as np

from . import _lib, bamio

SPAN = 1000
FLANKMATCH = 9
DNAPE_ELONGATE = SPAN * 10  # How far do we look beyond the target for paired-end
_complement = str.maketrans('ATCGatcgNnXx', 'TAGCtagcNnXx')
HERE = os.path.dirname(os.path.abspath(__file__))


def rc(s):
    return s.translate(_complement)[::-1]


_open_files = {}   # path -> open AlignmentFile: one sample's 30-odd loci open its BAM about a hundred times


class _SharedFile(object):
    """What read_alignment hands out: the cached file with a close() that leaves it open."""

    def __init__(self, f):
        self._f = f
        self.references, self.lengths = f.references, f.lengths
        self.fetch, self.pileup_depth_sum, self.getrname = f.fetch, f.pileup_depth_sum, f.getrname
        self.get_reference_name = f.getrname
        if hasattr(f, "pe_lengths"):
            self.pe_lengths, self.check_region, self.fetch_reads = f.pe_lengths, f.check_region, f.fetch_reads

    def close(self):
        pass


def read_alignment(samfile):
    if samfile.endswith(".cram"):
        raise ValueError("CRAM input needs htslib; this front end reads BAM only")
    try:
        key = (samfile, os.path.getmtime(samfile))
    except OSError:
        key = None
    f = _open_files.get(key) if key else None
    if f is None:
        f = bamio.AlignmentFile(samfile, "rb")
        if key:
            if len(_open_files) >= 8:
                _open_files.pop(next(iter(_open_files))).close()
            _open_files[key] = f
    return _SharedFile(f) if key else f


def test_fetch(samfile, chr, start, end, logger):
    try:
        if hasattr(samfile, "check_region"):    # native file layer: same ValueErrors without reading the region
            samfile.check_region(chr, start, end)
        else:
            next(iter(samfile.fetch(chr, start, end)), None)
        return True
    except ValueError:
        logger.error("No reads extracted for region {}:{}-{}".format(chr, start, end))
        return False


class BamParser:
    '''
    Find TRED repeats from aligned reads bam file
    :inputParams: InputParams object
    '''
    def __init__(self, inputParams):
        self.inputParams = inputParams
        self.logger = logging.getLogger('BamParser')
        self.logger.setLevel(inputParams.getLogLevel())
        self.bam = inputParams.bam
        self.gender = inputParams.gender
        self.depth = inputParams.depth
        self.READLEN = inputParams.READLEN
        self.clip = inputParams.clip
        self.alts = inputParams.alts
        self.repeatpairs = inputParams.repeatpairs
        self.ref = inputParams.ref
        self.tred = inputParams.tred
        self.repeatSize = len(self.tred.repeat)
        self.chr = self.tred.chr

        # X-linked TRED (bam_parser.py:57-61)
        if self.gender == 'Male' and self.tred.is_xlinked:
            self.ploidy = 1
        else:
            self.ploidy = self.tred.ploidy

        self.repeat = self.tred.repeat
        self.alt = self.tred.alt
        self.startRepeat, self.endRepeat = self.tred.repeat_start, self.tred.repeat_end
        self.referenceLen = self.tred.repeat_end - self.tred.repeat_start + 1
        self.fullPrefix, self.fullSuffix = self.tred.prefix, self.tred.suffix
        self.period = len(self.repeat)
        self.max_units = int(math.ceil(self.READLEN * 1. / self.period))

        counts = {}
        counts["PREF"] = counts["POST"] = defaultdict(int)   # one shared dict, as in the reference (:77)
        for tag in ("FULL", "REPT", "HANG"):
            counts[tag] = defaultdict(int)
        self.counts = counts
        self.details = []
        self.reads = []   # (query_name, query_sequence) in the order the reference would align them
        self.rept = 0

    # ---- read selection (bam_parser.py:184-243) -------------------------------------------------------
    def collect(self, pad=SPAN):
        WINDOW_START = max(0, self.startRepeat - pad)
        WINDOW_END = self.endRepeat + pad
        READ_START = max(0, self.startRepeat - self.READLEN)
        READ_END = self.endRepeat + self.READLEN
        samfile = read_alignment(self.bam)
        chr, start, end = self.chr, WINDOW_START, WINDOW_END
        self.reads = []
        if test_fetch(samfile, chr, start, end, self.logger):
            if hasattr(samfile, "fetch_reads"):    # native file layer: position filter applied before wrapping
                window = samfile.fetch_reads(chr, start, end, READ_START, READ_END)
            else:
                window = samfile.fetch(chr, start, end)
            for read in window:
                if not read.is_unmapped:
                    if read.reference_start < READ_START:
                        continue
                    if read.reference_start > READ_END:
                        continue
                self.reads.append((read.query_name, read.query_sequence))
            if self.alts:
                for c, s, e in self.alt:
                    if self.clip:
                        continue
                    try:
                        if "nochr" in self.ref:
                            c = c[3:]
                        for read in samfile.fetch(c, s, e):
                            rid = read.next_reference_id
                            if rid == -1:
                                continue
                            rname = samfile.getrname(rid)
                            rstart = read.next_reference_start
                            if rname != chr:
                                continue
                            if rstart < WINDOW_START:
                                continue
                            if rstart > WINDOW_END:
                                continue
                            self.reads.append((read.query_name, read.query_sequence))
                    except Exception as ex:
                        self.logger.debug("Fetch failed for region {}:{}-{} ({})".format(c, s, e, ex))
                        continue
        samfile.close()
        return self.reads

    # ---- what the reference does with each read's best (score, units, tag) (:174-182) + tally (:248-268) ----
    def finish(self, tags, hs):
        for (rid, seq), t, h in zip(self.reads, tags, hs):
            t, h = int(t), int(h)
            if t == _lib.TAG_NONE:
                continue
            self.counts["HANG"][h] += 1
            if t == _lib.TAG_HANG:
                continue
            self.details.append({'tag': _lib.TAG_NAMES[t], 'h': h, 'id': rid, 'seq': seq})
        if not (self.repeatpairs or self.clip):
            self.remove_pairs_of_rept()
        self.tally_counts()
        self.rept = sum(self.counts["REPT"].values()) if self.counts["REPT"] else 0

    def parse(self, pad=SPAN, engine=None):
        from .engine import Engine, Unit
        self.collect(pad)
        engine = engine or Engine()
        unit = Unit(self.tred, self.READLEN, [s for _, s in self.reads], self.depth, self.ploidy, [], [],
                    clip=self.clip)
        tags, hs, _, _, _ = engine.classify([unit])
        self.finish(tags, hs)

    def tally_counts(self):
        for x in self.details:
            self.counts[x["tag"]][x["h"]] += 1

    def remove_pairs_of_rept(self):
        rept_counts = defaultdict(int)
        for read in self.details:
            if read["tag"] == "REPT":
                rept_counts[read["id"]] += 1
        remove_ids = set(rid for rid, count in rept_counts.items() if count > 1)
        self.details = [x for x in self.details if x["id"] not in remove_ids]


class BamParserResults:
    '''Encapsulates all results: counts from BamParser and calls from the caller (bam_parser.py:290-313)'''
    def __init__(self, inputParams, bamParser, caller):
        self.inputParams = inputParams
        self.tred = bamParser.tred
        self.counts = bamParser.counts
        self.details = bamParser.details
        self.FDP = sum(bamParser.counts["FULL"].values())
        self.PDP = sum(bamParser.counts["PREF"].values())
        self.RDP = bamParser.rept
        for k in ("PEDP", "PEG", "PET", "CI", "PP", "label", "alleles", "P_h1", "P_h2", "P_h1h2", "P_PEG", "P_PET"):
            setattr(self, k, getattr(caller, k))


class PEextractor:
    """Infer distance paired-end reads spanning a certain region (bam_parser.py:316-369)."""
    def __init__(self, bp):
        samfile = read_alignment(bp.bam)
        chr, start, end = bp.chr, bp.startRepeat, bp.endRepeat
        self.ref = bp.referenceLen
        pstart = max(start - DNAPE_ELONGATE, 0)
        pend = end + DNAPE_ELONGATE
        self.global_lens, self.target_lens = [], []
        tstart = start - FLANKMATCH
        tend = end + FLANKMATCH
        self.MINPE = end - start + 2 * FLANKMATCH + 2
        if hasattr(samfile, "pe_lengths"):   # native file layer: the whole selection in one call
            if test_fetch(samfile, chr, pstart, pend, bp.logger):
                self.global_lens, self.target_lens = samfile.pe_lengths(chr, pstart, pend, tstart, tend, SPAN)
            samfile.close()
            return
        cache = {}
        if test_fetch(samfile, chr, pstart, pend, bp.logger):
            cache = defaultdict(list)
            for x in samfile.fetch(chr, pstart, pend):
                if not x.is_paired:
                    continue
                if x.is_unmapped:
                    continue
                if x.is_duplicate:
                    continue
                cache[x.query_name].append(x)
        for name, reads in cache.items():
            if len(reads) < 2:
                continue
            a, b = reads[:2]
            if not ((not a.is_reverse) and b.is_reverse):  # Mapped in +, - orientation
                continue
            tlen = self.get_target_length(a, b)
            if tlen >= SPAN:
                continue
            if a.reference_start < tstart and b.reference_end > tend:
                self.target_lens.append(tlen)
            else:
                self.global_lens.append(tlen)
        samfile.close()

    def get_target_length(self, a, b):
        start, end = a.reference_start, b.reference_end
        if a.query_alignment_start > 0:  # has clips
            start -= a.query_alignment_start
        if b.query_alignment_end < b.query_length:  # has clips
            end += b.query_length - b.query_alignment_end
        return end - start


class BamReadLen:
    """Returns the read length in BAM file (bam_parser.py:372-391)."""
    def __init__(self, bamfile, logger):
        self.bamfile = bamfile
        self.logger = logger

    @property
    def readlen(self, firstN=100):
        sam = read_alignment(self.bamfile)
        rls = []
        for read in sam.fetch():
            rls.append(read.query_length)
            if len(rls) > firstN:
                break
        sam.close()
        return max(rls)


class BamDepth:
    """Average depth of a region, for the repeat model and for sex inference (bam_parser.py:394-429)."""
    def __init__(self, bamfile, ref, logger):
        self.bamfile = bamfile
        self.logger = logger
        self.ref = ref

    def region_depth(self, chr, start, end, verbose=False):
        sam = read_alignment(self.bamfile)
        try:
            total = sam.pileup_depth_sum(chr, start, end)
        finally:
            sam.close()
        return total * 1. / (end - start + 1)

    def get_Y_depth(self, N=5):
        UNIQY = os.path.join(HERE, "data", "chrY.{}.unique_ccn.tsv".format(self.ref.split('_')[0]))
        depths = []
        with open(UNIQY) as fp:
            for i, row in enumerate(fp):
                if i in (1, 4, 6, 7, 10, 11, 13, 16, 18, 19):   # regions that still attract reads (:419)
                    continue
                if len(depths) >= N:
                    break
                c, start, end = row.split()[:3]
                depths.append(self.region_depth(c, int(start), int(end)))
        return np.median(depths)
