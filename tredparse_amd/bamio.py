"""Minimal BAM/BGZF/BAI reader (pure Python + zlib) -- the host I/O front end's file layer.

The reference reads BAMs through pysam/htslib (bam_parser.py:22,432-436), which is neither vendored
nor installable here; this module provides the handful of htslib behaviours the reference relies on:

  AlignmentFile(path).fetch()                   all records in file order        (bam_parser.py:384)
  AlignmentFile(path).fetch(chrom, start, end)  records overlapping [start, end) (bam_parser.py:206,226,333),
                                                found through the .bai (bins + linear index) like htslib;
                                                placed-unmapped reads (mate-anchored) are returned at their
                                                mate's position, as htslib does
  .pileup_depth_sum(chrom, start, end)          sum of per-column read counts of pileup(chrom, start, end)
                                                (bam_parser.py:404-407): every reference position covered by a
                                                read that overlaps the region counts, also outside the region,
                                                skipping unmapped / secondary / QC-fail / duplicate reads
                                                (htslib's default pileup mask)
  .getrname(tid), .references, .lengths
  Read: query_name, query_sequence, query_length, flag properties, reference_start, reference_end,
        next_reference_id, next_reference_start, query_alignment_start, query_alignment_end, cigartuples

CRAM is not supported (the reference dispatches .cram to htslib, bam_parser.py:435).
"""
import os
import struct
import zlib

_SEQ = "=ACMGRSVTWYHKDBN"
_CIGAR_CONSUMES_REF = (True, False, True, True, False, False, False, True, True)   # MIDNSHP=X
_CIGAR_CONSUMES_QUERY = (True, True, False, False, True, False, False, True, True)

FUNMAP, FPAIRED, FREVERSE, FSECONDARY, FQCFAIL, FDUP = 0x4, 0x1, 0x10, 0x100, 0x200, 0x400


class Read(object):
    __slots__ = ("tid", "pos", "mapq", "flag", "next_tid", "next_pos", "tlen", "query_name", "cigartuples",
                 "_seq_raw", "l_seq", "_end")

    # -- pysam-style accessors used by the reference (bam_parser.py:130-131,207-212,228-232,334-369,385)
    @property
    def is_unmapped(self): return bool(self.flag & FUNMAP)
    @property
    def is_paired(self): return bool(self.flag & FPAIRED)
    @property
    def is_reverse(self): return bool(self.flag & FREVERSE)
    @property
    def is_duplicate(self): return bool(self.flag & FDUP)
    @property
    def is_secondary(self): return bool(self.flag & FSECONDARY)
    @property
    def is_qcfail(self): return bool(self.flag & FQCFAIL)
    @property
    def reference_start(self): return self.pos
    @property
    def next_reference_id(self): return self.next_tid
    @property
    def next_reference_start(self): return self.next_pos
    @property
    def query_length(self): return self.l_seq

    @property
    def reference_end(self):
        """One past the last aligned reference base; None without an alignment (pysam semantics)."""
        if self._end is None:
            if self.is_unmapped or not self.cigartuples:
                self._end = -1
            else:
                self._end = self.pos + sum(n for op, n in self.cigartuples if _CIGAR_CONSUMES_REF[op])
        return None if self._end < 0 else self._end

    @property
    def query_sequence(self):
        raw, n = self._seq_raw, self.l_seq
        out = []
        for i in range(n):
            b = raw[i >> 1]
            out.append(_SEQ[(b >> 4) if not (i & 1) else (b & 15)])
        return "".join(out)

    @property
    def query_alignment_start(self):
        s = 0
        for op, n in self.cigartuples or ():
            if op == 4: s += n       # soft clip
            elif op == 5: continue   # hard clip
            else: break
        return s

    @property
    def query_alignment_end(self):
        e = self.l_seq
        for op, n in reversed(self.cigartuples or ()):
            if op == 4: e -= n
            elif op == 5: continue
            else: break
        return e


class _Bgzf(object):
    """Random access to a BGZF file through virtual offsets (coffset << 16 | uoffset)."""

    def __init__(self, path):
        self.fp = open(path, "rb")
        self.block_coffset = -1
        self.block = b""
        self.block_clen = 0
        self.upos = 0

    def close(self):
        self.fp.close()

    def _load(self, coffset):
        self.fp.seek(coffset)
        hdr = self.fp.read(18)
        if len(hdr) < 18:
            self.block, self.block_coffset, self.block_clen = b"", coffset, 0
            return False
        if hdr[:4] != b"\x1f\x8b\x08\x04":
            raise IOError("not a BGZF block at {}".format(coffset))
        xlen = struct.unpack_from("<H", hdr, 10)[0]
        extra = hdr[12:] + self.fp.read(xlen - 6)
        bsize, p = None, 0
        while p + 4 <= len(extra):
            si1, si2, slen = extra[p], extra[p + 1], struct.unpack_from("<H", extra, p + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", extra, p + 4)[0]
            p += 4 + slen
        if bsize is None:
            raise IOError("BGZF block without BC field")
        clen = bsize + 1
        data = self.fp.read(clen - 12 - xlen)
        self.block = zlib.decompress(data[:-8], -15)
        self.block_coffset, self.block_clen = coffset, clen
        return True

    def seek(self, voffset):
        coffset, uoffset = voffset >> 16, voffset & 0xFFFF
        if coffset != self.block_coffset:
            self._load(coffset)
        self.upos = uoffset

    def tell(self):
        return (self.block_coffset << 16) | self.upos

    def read(self, n):
        out = []
        while n > 0:
            if self.upos >= len(self.block):
                nxt = self.block_coffset + self.block_clen
                if not self._load(nxt) and not self.block:
                    break
                self.upos = 0
                if not self.block:
                    continue
            chunk = self.block[self.upos:self.upos + n]
            out.append(chunk)
            self.upos += len(chunk)
            n -= len(chunk)
        return b"".join(out)


def _reg2bins(beg, end):
    """htslib reg2bins for the standard 5-level scheme (SAM spec 5.3)."""
    end -= 1
    bins = [0]
    for shift, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        bins.extend(range(base + (beg >> shift), base + (end >> shift) + 1))
    return bins


class AlignmentFile(object):
    def __init__(self, path, mode="rb"):
        if path.endswith(".cram"):
            raise ValueError("CRAM is not supported by this front end")
        if not os.path.exists(path):
            raise IOError("file `{}` not found".format(path))
        self.path = path
        self.bg = _Bgzf(path)
        self.bg.seek(0)
        if self.bg.read(4) != b"BAM\x01":
            raise ValueError("not a BAM file: {}".format(path))
        l_text = struct.unpack("<i", self.bg.read(4))[0]
        self.text = self.bg.read(l_text)
        n_ref = struct.unpack("<i", self.bg.read(4))[0]
        self.references, self.lengths = [], []
        for _ in range(n_ref):
            l_name = struct.unpack("<i", self.bg.read(4))[0]
            self.references.append(self.bg.read(l_name)[:-1].decode())
            self.lengths.append(struct.unpack("<i", self.bg.read(4))[0])
        self._tid = {n: i for i, n in enumerate(self.references)}
        self._first = self.bg.tell()
        self._index = None

    def close(self):
        self.bg.close()

    def getrname(self, tid):
        return self.references[tid]

    get_reference_name = getrname

    # ---- records -------------------------------------------------------------------------------
    def _next(self):
        head = self.bg.read(4)
        if len(head) < 4:
            return None
        size = struct.unpack("<i", head)[0]
        buf = self.bg.read(size)
        if len(buf) < size:
            return None
        tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, ntid, npos, tlen = struct.unpack_from("<iiBBHHHiiii", buf, 0)
        r = Read()
        r.tid, r.pos, r.mapq, r.flag, r.next_tid, r.next_pos, r.tlen, r.l_seq = tid, pos, mapq, flag, ntid, npos, tlen, l_seq
        p = 32
        r.query_name = buf[p:p + l_name - 1].decode()
        p += l_name
        cig = struct.unpack_from("<{}I".format(n_cig), buf, p) if n_cig else ()
        r.cigartuples = [(c & 15, c >> 4) for c in cig]
        p += 4 * n_cig
        r._seq_raw = buf[p:p + (l_seq + 1) // 2]
        r._end = None
        return r

    def _load_index(self):
        if self._index is not None:
            return
        for cand in (self.path + ".bai", os.path.splitext(self.path)[0] + ".bai"):
            if os.path.exists(cand):
                break
        else:
            raise ValueError("no .bai index next to {}".format(self.path))
        data = open(cand, "rb").read()
        if data[:4] != b"BAI\x01":
            raise ValueError("bad BAI magic")
        n_ref = struct.unpack_from("<i", data, 4)[0]
        p = 8
        index = []
        for _ in range(n_ref):
            n_bin = struct.unpack_from("<i", data, p)[0]; p += 4
            bins = {}
            for _ in range(n_bin):
                b, n_chunk = struct.unpack_from("<Ii", data, p); p += 8
                chunks = struct.unpack_from("<{}Q".format(2 * n_chunk), data, p); p += 16 * n_chunk
                bins[b] = [(chunks[2 * k], chunks[2 * k + 1]) for k in range(n_chunk)]
            n_intv = struct.unpack_from("<i", data, p)[0]; p += 4
            lin = struct.unpack_from("<{}Q".format(n_intv), data, p); p += 8 * n_intv
            index.append((bins, lin))
        self._index = index

    def fetch(self, chrom=None, start=None, end=None):
        """Records in file order; with a region, those overlapping [start, end) (0-based, half-open)."""
        if chrom is None:
            self.bg.seek(self._first)
            while True:
                r = self._next()
                if r is None:
                    return
                yield r
        if chrom not in self._tid:
            raise ValueError("invalid contig `{}`".format(chrom))
        tid = self._tid[chrom]
        start = max(0, int(start) if start is not None else 0)
        end = int(end) if end is not None else self.lengths[tid]
        if start > end:
            raise ValueError("invalid coordinates: start > end")
        self._load_index()
        bins, lin = self._index[tid]
        min_off = lin[min(start >> 14, len(lin) - 1)] if lin else 0
        chunks = []
        for b in _reg2bins(start, max(end, start + 1)):
            for cb, ce in bins.get(b, ()):
                if ce > min_off:
                    chunks.append((max(cb, min_off), ce))
        chunks.sort()
        merged = []
        for cb, ce in chunks:
            if merged and cb <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], ce))
            else:
                merged.append((cb, ce))
        for cb, ce in merged:
            self.bg.seek(cb)
            while self.bg.tell() < ce:
                r = self._next()
                if r is None:
                    break
                if r.tid != tid or r.pos >= end:
                    if r.tid > tid or (r.tid == tid and r.pos >= end):
                        break
                    continue
                rend = r.reference_end
                if rend is None or rend <= r.pos:
                    rend = r.pos + 1           # unmapped-but-placed / zero-length: one base (htslib bam_endpos)
                if rend > start:
                    yield r

    def pileup_depth_sum(self, chrom, start, end):
        total = 0
        for r in self.fetch(chrom, start, end):
            if r.flag & (FUNMAP | FSECONDARY | FQCFAIL | FDUP):
                continue
            rend = r.reference_end
            if rend is not None:
                total += rend - r.pos
        return total
