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

Two implementations of the same layer: `AlignmentFile` is the native one when tredparse_amd/libtredbam.so
(include/tredbam.h, csrc/bamread.cpp) is built -- a region's records arrive in one call and are only wrapped
here -- and the pure-Python `PyAlignmentFile` otherwise (TREDBAM_PURE_PYTHON=1 forces it);
tests/test_host_frontend.py checks them against each other record for record.
"""
import ctypes as C
import os
import struct
import zlib

_SEQ = "=ACMGRSVTWYHKDBN"
_CIGAR_CONSUMES_REF = (True, False, True, True, False, False, False, True, True)   # MIDNSHP=X
_CIGAR_CONSUMES_QUERY = (True, True, False, False, True, False, False, True, True)

FUNMAP, FPAIRED, FREVERSE, FSECONDARY, FQCFAIL, FDUP = 0x4, 0x1, 0x10, 0x100, 0x200, 0x400


class Read(object):
    __slots__ = ("tid", "pos", "mapq", "flag", "next_tid", "next_pos", "tlen", "l_seq", "_end")

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


class PyRead(Read):
    """A record parsed by the pure-Python layer (packed 4-bit sequence kept raw until asked for)."""
    __slots__ = ("query_name", "cigartuples", "_seq_raw")

    @property
    def query_sequence(self):
        raw, n = self._seq_raw, self.l_seq
        if n == 0:
            return None                                  # (SEQ '*': pysam's AlignedSegment.query_sequence is None)
        out = []
        for i in range(n):
            b = raw[i >> 1]
            out.append(_SEQ[(b >> 4) if not (i & 1) else (b & 15)])
        return "".join(out)


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
        if self.block_clen > 0 and self.upos >= len(self.block):      # at a block's end = at the next block's start
            return (self.block_coffset + self.block_clen) << 16      # (htslib's bgzf_tell; also 64 KiB blocks)
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


class PyAlignmentFile(object):
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
        r = PyRead()
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


# ---- native file layer (libtredbam.so) ---------------------------------------------------------------------
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtredbam.so")
_lib = None


def _native():
    """The ctypes handle of libtredbam.so, or None when it is not built / switched off."""
    global _lib
    if _lib is None:
        if os.environ.get("TREDBAM_PURE_PYTHON") or not os.path.exists(_LIB_PATH):
            _lib = False
        else:
            lib = C.CDLL(_LIB_PATH)
            lib.tredbam_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
            lib.tredbam_open.restype = C.c_int
            lib.tredbam_close.argtypes = [C.c_void_p]
            lib.tredbam_close.restype = None
            lib.tredbam_last_error.argtypes = [C.c_void_p]
            lib.tredbam_last_error.restype = C.c_char_p
            lib.tredbam_n_ref.argtypes = [C.c_void_p]
            lib.tredbam_n_ref.restype = C.c_int32
            lib.tredbam_ref_name.argtypes = [C.c_void_p, C.c_int32]
            lib.tredbam_ref_name.restype = C.c_char_p
            lib.tredbam_ref_len.argtypes = [C.c_void_p, C.c_int32]
            lib.tredbam_ref_len.restype = C.c_int64
            lib.tredbam_tid.argtypes = [C.c_void_p, C.c_char_p]
            lib.tredbam_tid.restype = C.c_int32
            lib.tredbam_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
            lib.tredbam_fetch.restype = C.c_int64
            lib.tredbam_fetch_reads.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                                C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
            lib.tredbam_fetch_reads.restype = C.c_int64
            lib.tredbam_pileup_depth_sum.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
            lib.tredbam_pileup_depth_sum.restype = C.c_int
            lib.tredbam_pe_lengths.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                               C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                                               C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
            lib.tredbam_pe_lengths.restype = C.c_int
            lib.tredbam_max_read_len.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]
            lib.tredbam_max_read_len.restype = C.c_int
            lib.tredbam_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(ScanOpts), C.c_void_p]
            lib.tredbam_scan.restype = C.c_int
            lib.tredbam_scan_pools.argtypes = [C.c_void_p, C.POINTER(Pools)]
            lib.tredbam_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(ScanOpts), C.c_void_p, C.c_int32,
                                         C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
            lib.tredbam_plan.restype = C.c_int64
            lib.tredbam_plan_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
            lib.tredbam_plan_fill.restype = C.c_int
            lib.tredbam_preload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            lib.tredbam_preload.restype = C.c_int
            lib.tredbam_preload_crc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            lib.tredbam_preload_crc.restype = C.c_int
            lib.tredbam_preload_clear.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
            lib.tredbam_preload_clear.restype = None
            lib.tredbam_scan_pools.restype = C.c_int
            lib.tredbam_plan_walks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(ScanOpts), C.c_void_p, C.c_void_p,
                                               C.c_int64]
            lib.tredbam_plan_walks.restype = C.c_int64
            lib.tredbam_plan_region_walks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
            lib.tredbam_plan_region_walks.restype = C.c_int64
            lib.tredbam_plan_blocks.argtypes = [C.c_void_p] * 5
            lib.tredbam_plan_blocks.restype = C.c_int64
            lib.tredbam_scan_pe.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(ScanOpts), C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
            lib.tredbam_scan_pe.restype = C.c_int
            lib.tredbam_plan_alt_walks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(ScanOpts),
                                                   C.c_void_p, C.c_void_p, C.c_int64]
            lib.tredbam_plan_alt_walks.restype = C.c_int64
            lib.tredbam_scan_walked.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(ScanOpts), C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            lib.tredbam_scan_walked.restype = C.c_int
            lib.tredbam_pe_pool_sizes.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
            lib.tredbam_pe_pool_sizes.restype = C.c_int
            lib.tredbam_details_json.argtypes = [C.c_void_p] * 8 + [C.c_int64, C.c_void_p, C.c_int64]
            lib.tredbam_details_json.restype = C.c_int64
            lib.tredbam_sparse_json.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64]
            lib.tredbam_sparse_json.restype = C.c_int64
            lib.tredbam_pair_stats.argtypes = [C.c_void_p] * 3 + [C.c_int64] + [C.c_void_p] * 3
            lib.tredbam_pair_stats.restype = C.c_int
            lib.tredbam_sparse_json_many.argtypes = [C.c_void_p] * 5 + [C.c_int64, C.c_int32, C.c_void_p, C.c_int64,
                                                                          C.c_void_p, C.c_void_p]
            lib.tredbam_sparse_json_many.restype = C.c_int64
            lib.tredbam_details_json_many.argtypes = [C.c_void_p] * 9 + [C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                                                          C.c_void_p]
            lib.tredbam_details_json_many.restype = C.c_int64
            lib.tredbam_emit_sample_files.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
            lib.tredbam_emit_sample_files.restype = C.c_int
            lib.tredbam_emit_last_error.argtypes = []
            lib.tredbam_emit_last_error.restype = C.c_char_p
            lib.tredbam_pairwise_sum.argtypes = [C.c_void_p, C.c_int64]
            lib.tredbam_pairwise_sum.restype = C.c_double
            _lib = lib
    return _lib or None


_REC = struct.Struct("<10iHBB")   # tredbam_rec (include/tredbam.h)

# ---- whole-sample scan (tredbam_scan, include/tredbam.h) ----------------------------------------------------
import numpy as np   # noqa: E402  (only the scan path needs it)

SITE_DTYPE = np.dtype([("tid", "<i4"), ("repeat_start", "<i4"), ("repeat_end", "<i4"), ("alt_first", "<i4"),
                       ("n_alt", "<i4")])
REGION_DTYPE = np.dtype([("tid", "<i4"), ("start", "<i4"), ("end", "<i4")])
SCAN_UNIT_DTYPE = np.dtype([("status", "<i4"), ("n_reads", "<i4"), ("read_first", "<i8"), ("depth_sum", "<i8"),
                            ("depth_status", "<i4"), ("pe_status", "<i4"), ("n_global", "<i4"), ("n_target", "<i4"),
                            ("global_first", "<i8"), ("target_first", "<i8")])
UNIT_NO_FETCH, UNIT_FAILED, UNIT_NO_SEQ = 1, 2, 4
# the pair walks handed to the device (include/tredbam.h; the same layouts as tredgpu.h's tredgpu_walk_*)
WALK_TASK_DTYPE = np.dtype([(k, "<i4") for k in ("tid", "start", "end", "tstart", "tend", "span", "chunk_first", "n_chunks",
                                                 "block_first", "block_end", "win_lo", "win_hi")])
WALK_CHUNK_DTYPE = np.dtype([("begin_block", "<i4"), ("begin_upos", "<i4"), ("end_voffset", "<u8")])
WALK_RESULT_DTYPE = np.dtype([("status", "<i4"), ("n_global", "<i4"), ("n_target", "<i4"), ("n_window", "<i4"),
                              ("global_first", "<i8"), ("target_first", "<i8"), ("win_vbeg", "<u8"), ("win_vend", "<u8")])


def sparse_json(a, b, values, depth):
    """tredbam_sparse_json: the text of {"a" | "a,b": value} as the driver's JSON prints it at nesting `depth`, or None
    (library absent / a case for the generic encoder)."""
    lib = _native()
    if lib is None:
        return None
    n = len(values)
    a = np.ascontiguousarray(a, np.int32)
    b = None if b is None else np.ascontiguousarray(b, np.int32)
    values = np.ascontiguousarray(values, np.float64)
    cap = 64 + n * (4 * (depth + 1) + 80)
    buf = C.create_string_buffer(cap)
    got = lib.tredbam_sparse_json(a.ctypes.data, None if b is None else b.ctypes.data, values.ctypes.data, n, depth, buf, cap)
    if got == -1:
        return None
    if got < 0:
        raise RuntimeError("tredbam_sparse_json failed ({})".format(got))
    return buf.raw[:got].decode("ascii")


def details_json(seq4, seq4_off, read_len, names, name_off, reads, tags, hs):
    """tredbam_details_json over numpy pools (names: bytes); the text, or None when the generic encoder has to do it
    (library absent, a name with bytes json.dumps escapes other than the quote and the backslash)."""
    lib = _native()
    if lib is None:
        return None
    n = len(reads)
    reads = np.ascontiguousarray(reads, np.int64)
    tags = np.ascontiguousarray(tags, np.uint8)
    hs = np.ascontiguousarray(hs, np.int32)
    name_len = int((name_off[reads + 1] - name_off[reads]).sum()) if n else 0
    cap = 64 + 200 * n + 2 * name_len + (int(read_len[reads].sum()) if n else 0)
    buf = C.create_string_buffer(cap)
    got = lib.tredbam_details_json(seq4.ctypes.data, seq4_off.ctypes.data, read_len.ctypes.data, names,
                                   name_off.ctypes.data, reads.ctypes.data, tags.ctypes.data, hs.ctypes.data, n, buf, cap)
    if got == -1:
        return None
    if got < 0:
        raise RuntimeError("tredbam_details_json failed ({})".format(got))
    return buf.raw[:got].decode("ascii")


def pair_stats(pool, first, count):
    """tredbam_pair_stats: (mean[g], sd[g], hist[g, 40]) of the slices pool[first[k] : first[k] + count[k]], or None
    (library absent)."""
    lib = _native()
    if lib is None:
        return None
    g = len(first)
    pool = np.ascontiguousarray(pool, np.int32)
    first = np.ascontiguousarray(first, np.int64)
    count = np.ascontiguousarray(count, np.int32)
    mean, sd, hist = np.empty(g), np.empty(g), np.empty((g, 40), np.int32)
    rc = lib.tredbam_pair_stats(pool.ctypes.data if len(pool) else None, first.ctypes.data, count.ctypes.data, g,
                                mean.ctypes.data, sd.ctypes.data, hist.ctypes.data)
    if rc != 0:
        raise RuntimeError("tredbam_pair_stats failed ({})".format(rc))
    return mean, sd, hist


def _texts(buf, got, out_off, status):
    """[text or None] of the items of a *_many call."""
    raw = buf[:got].tobytes().decode("ascii")
    o = out_off.tolist()
    return [raw[o[k]:o[k + 1]] if st == 0 else None for k, st in enumerate(status.tolist())]


def sparse_json_many(dists, depth):
    """tredbam_sparse_json_many: the texts of many distributions [(a, b or None, values)] in one native call (a sample
    has 90 of them); an entry is None where the generic encoder has to print that one.  None: library absent."""
    lib = _native()
    if lib is None:
        return None
    if not dists:
        return []
    sizes = np.fromiter((len(d[2]) for d in dists), np.int64, len(dists))
    off = np.zeros(len(dists) + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    total = int(off[-1])
    a = np.empty(total, np.int32)
    b = np.zeros(total, np.int32)
    v = np.empty(total, np.float64)
    two = np.zeros(len(dists), np.uint8)
    for k, (da, db, dv) in enumerate(dists):
        lo, hi = int(off[k]), int(off[k + 1])
        a[lo:hi] = da
        v[lo:hi] = dv
        if db is not None:
            b[lo:hi] = db
            two[k] = 1
    cap = 64 * len(dists) + total * (4 * (depth + 1) + 80)
    buf = np.empty(cap, np.uint8)
    out_off = np.empty(len(dists) + 1, np.int64)
    status = np.empty(len(dists), np.int8)
    got = lib.tredbam_sparse_json_many(a.ctypes.data, b.ctypes.data, v.ctypes.data, off.ctypes.data, two.ctypes.data,
                                       len(dists), depth, buf.ctypes.data, cap, out_off.ctypes.data, status.ctypes.data)
    if got < 0:
        raise RuntimeError("tredbam_sparse_json_many failed ({})".format(got))
    return _texts(buf, got, out_off, status)


def details_json_many(seq4, seq4_off, read_len, names, name_off, lists):
    """tredbam_details_json_many: the texts of many `details` lists [(reads, tags, hs)] over ONE scan's pools in one
    native call; an entry is None where the generic encoder has to print that list.  None: library absent."""
    lib = _native()
    if lib is None:
        return None
    if not lists:
        return []
    sizes = np.fromiter((len(x[0]) for x in lists), np.int64, len(lists))
    off = np.zeros(len(lists) + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    cat = lambda i, dt: (np.ascontiguousarray(np.concatenate([x[i] for x in lists]), dt) if int(off[-1])
                         else np.zeros(1, dt))
    reads, tags, hs = cat(0, np.int64), cat(1, np.uint8), cat(2, np.int32)
    n = int(off[-1])
    name_len = int((name_off[reads[:n] + 1] - name_off[reads[:n]]).sum()) if n else 0
    cap = 64 * len(lists) + 200 * n + 2 * name_len + (int(read_len[reads[:n]].sum()) if n else 0)
    buf = np.empty(cap, np.uint8)
    out_off = np.empty(len(lists) + 1, np.int64)
    status = np.empty(len(lists), np.int8)
    got = lib.tredbam_details_json_many(seq4.ctypes.data, seq4_off.ctypes.data, read_len.ctypes.data, names,
                                        name_off.ctypes.data, reads.ctypes.data, tags.ctypes.data, hs.ctypes.data,
                                        off.ctypes.data, len(lists), buf.ctypes.data, cap, out_off.ctypes.data,
                                        status.ctypes.data)
    if got < 0:
        raise RuntimeError("tredbam_details_json_many failed ({})".format(got))
    return _texts(buf, got, out_off, status)


# ---- a sample's output files written natively (tredbam_emit_sample_files, include/tredbam.h) -------------------------
class EmitLocus(C.Structure):
    _fields_ = [("name", C.c_char_p), ("motif", C.c_char_p), ("chrom", C.c_char_p), ("info", C.c_char_p)] + \
               [(k, C.c_int32) for k in ("pos", "ref_copy", "period", "cutoff_prerisk", "cutoff_risk", "is_expansion",
                                         "is_recessive", "in_vcf")]


class EmitBatch(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("tag", "h", "unit_read_off", "calls", "marg")] + [("marg_len", C.c_int64)] + \
               [(k, C.c_void_p) for k in ("joint_a", "joint_b", "joint_v", "joint_lo", "joint_n")] + \
               [("repeatpairs", C.c_int32), ("pad", C.c_int32)]


class EmitSample(C.Structure):
    _fields_ = [("samplekey", C.c_char_p), ("bam", C.c_char_p), ("gender", C.c_char_p), ("ydepth", C.c_double),
                ("opened", C.c_int32), ("readlen", C.c_int32)] + \
               [(k, C.c_void_p) for k in ("seq4", "seq4_off", "read_len", "names", "name_off", "name_id", "global_lens",
                                          "target_lens", "unit", "depth", "unit_index")]


class EmitOpts(C.Structure):
    _fields_ = [("ref", C.c_char_p), ("source", C.c_char_p), ("filedate", C.c_char_p), ("vcf_meta", C.c_char_p),
                ("write_json", C.c_int32), ("write_vcf", C.c_int32), ("gzip_level", C.c_int32), ("pad", C.c_int32)]


def emit_locus_table(repo, names):
    """The tredbam_emit_locus array of a locus list (kept alive by the returned object's `.keep`)."""
    arr = (EmitLocus * max(1, len(names)))()
    keep = []
    for k, n in enumerate(names):
        t = repo[n]
        chrom, pos, ref_copy, motif, info = repo.get_info(n)
        texts = [x.encode("utf-8") for x in (t.name, motif, chrom, info)]
        keep.append(texts)
        arr[k] = EmitLocus(texts[0], texts[1], texts[2], texts[3], int(pos), int(ref_copy), int(t.period), int(t.cutoff_prerisk),
                           int(t.cutoff_risk), int(bool(t.is_expansion)), int(bool(t.is_recessive)), 1)
    arr.keep = keep
    return arr


class ScanOpts(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("readlen", "pad", "flank", "pe_reach", "span", "use_alts", "want_depth",
                                         "want_pe")]


class Pools(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_words", C.c_int64), ("n_global", C.c_int64), ("n_target", C.c_int64),
                ("packed", C.c_void_p), ("word_off", C.c_void_p), ("read_len", C.c_void_p), ("seq4", C.c_void_p),
                ("seq4_off", C.c_void_p), ("names", C.c_void_p), ("name_off", C.c_void_p), ("name_id", C.c_void_p),
                ("global_lens", C.c_void_p), ("target_lens", C.c_void_p)]


assert SITE_DTYPE.itemsize == 20 and REGION_DTYPE.itemsize == 12 and SCAN_UNIT_DTYPE.itemsize == 56
ALT_RESULT_DTYPE = np.dtype([("status", "<i4"), ("n", "<i4"), ("vbeg", "<u8", (6,))])
assert ALT_RESULT_DTYPE.itemsize == 56
assert WALK_TASK_DTYPE.itemsize == 48 and WALK_CHUNK_DTYPE.itemsize == 16 and WALK_RESULT_DTYPE.itemsize == 48


def _copy(ptr, count, dtype):
    """numpy copy of `count` items at C address ptr (the pools are only valid until the handle's next scan)."""
    dtype = np.dtype(dtype)
    if not count:
        return np.zeros(0, dtype)
    return np.frombuffer(C.string_at(ptr, int(count) * dtype.itemsize), dtype).copy()


class NativeRead(Read):
    """A record of a libtredbam buffer; name, CIGAR and sequence are decoded on first use (most records of a
    region are dismissed on flags and positions alone)."""
    __slots__ = ("_d", "_q", "_n_cig", "_l_name", "_name", "_cig")

    @property
    def query_name(self):
        if self._name is None:
            self._name = self._d[self._q:self._q + self._l_name - 1].decode()
        return self._name

    @property
    def cigartuples(self):
        if self._cig is None:
            q = self._q + ((self._l_name + 3) & ~3)
            self._cig = [(c & 15, c >> 4) for c in struct.unpack_from("<{}I".format(self._n_cig), self._d, q)] \
                if self._n_cig else []
        return self._cig

    @property
    def query_sequence(self):
        if self.l_seq == 0:
            return None                                  # (as pysam)
        q = self._q + ((self._l_name + 3) & ~3) + 4 * self._n_cig
        return self._d[q:q + self.l_seq].decode()


class NativeAlignmentFile(object):
    """Same interface as PyAlignmentFile on top of libtredbam.so."""

    def __init__(self, path, mode="rb"):
        if path.endswith(".cram"):
            raise ValueError("CRAM is not supported by this front end")
        if not os.path.exists(path):
            raise IOError("file `{}` not found".format(path))
        self.path = path
        self._lib = _native()
        h = C.c_void_p()
        if self._lib.tredbam_open(path.encode(), C.byref(h)) != 0:
            raise ValueError(self._lib.tredbam_last_error(None).decode())
        self._h = h
        n = self._lib.tredbam_n_ref(h)
        self.references = [self._lib.tredbam_ref_name(h, t).decode() for t in range(n)]
        self.lengths = [int(self._lib.tredbam_ref_len(h, t)) for t in range(n)]
        self._tid = {nm: i for i, nm in enumerate(self.references)}

    def close(self):
        if self._h:
            self._lib.tredbam_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def getrname(self, tid):
        return self.references[tid]

    get_reference_name = getrname

    def _err(self):
        return self._lib.tredbam_last_error(self._h).decode()

    def _records(self, tid, start, end, limit=0, pos_range=None):
        buf, nbytes = C.c_void_p(), C.c_int64()
        if pos_range is None:
            n = self._lib.tredbam_fetch(self._h, tid, start, end, limit, C.byref(buf), C.byref(nbytes))
        else:
            n = self._lib.tredbam_fetch_reads(self._h, tid, start, end, pos_range[0], pos_range[1], C.byref(buf),
                                              C.byref(nbytes))
        if n < 0:
            raise ValueError(self._err())
        data = C.string_at(buf, nbytes.value) if n else b""
        out, p = [], 0
        unpack = _REC.unpack_from
        for _ in range(n):
            size, rtid, pos, rend, ntid, npos, tlen, l_seq, n_cig, l_name, flag, mapq, _pad = unpack(data, p)
            r = NativeRead()
            r.tid, r.pos, r.mapq, r.flag, r.next_tid, r.next_pos, r.tlen, r.l_seq = rtid, pos, mapq, flag, ntid, npos, tlen, l_seq
            r._d, r._q, r._n_cig, r._l_name, r._name, r._cig, r._end = data, p + _REC.size, n_cig, l_name, None, None, rend
            out.append(r)
            p += size
        return out

    def fetch(self, chrom=None, start=None, end=None):
        if chrom is None:
            # file order; read in growing slabs so that callers that stop early (bam_parser.py:380-391) stay cheap
            got, limit = 0, 256
            while True:
                recs = self._records(-1, 0, 0, limit)
                for r in recs[got:]:
                    yield r
                if len(recs) < limit:
                    return
                got, limit = len(recs), limit * 8
        if chrom not in self._tid:
            raise ValueError("invalid contig `{}`".format(chrom))
        tid = self._tid[chrom]
        start = max(0, int(start) if start is not None else 0)
        end = int(end) if end is not None else self.lengths[tid]
        if start > end:
            raise ValueError("invalid coordinates: start > end")
        for r in self._records(tid, start, end):
            yield r

    def pileup_depth_sum(self, chrom, start, end):
        if chrom not in self._tid:
            raise ValueError("invalid contig `{}`".format(chrom))
        start = max(0, int(start))
        total = C.c_int64()
        if self._lib.tredbam_pileup_depth_sum(self._h, self._tid[chrom], start, int(end), C.byref(total)) != 0:
            raise ValueError(self._err())
        return int(total.value)


    def fetch_reads(self, chrom, start, end, pos_lo, pos_hi):
        """fetch(chrom, start, end) restricted to unmapped records and those with pos_lo <= pos <= pos_hi
        (the read selection of bam_parser.py:206-214, filtered before anything is wrapped)."""
        self.check_region(chrom, start, end)
        return self._records(self._tid[chrom], max(0, int(start)), int(end), pos_range=(int(pos_lo), int(pos_hi)))

    def check_region(self, chrom, start, end):
        """The ValueErrors fetch(chrom, start, end) would raise, without reading anything."""
        if chrom not in self._tid:
            raise ValueError("invalid contig `{}`".format(chrom))
        start = max(0, int(start) if start is not None else 0)
        end = int(end) if end is not None else self.lengths[self._tid[chrom]]
        if start > end:
            raise ValueError("invalid coordinates: start > end")
        if not self._has_index():
            raise ValueError("no .bai index next to {}".format(self.path))

    def _has_index(self):
        return any(os.path.exists(c) for c in (self.path + ".bai", os.path.splitext(self.path)[0] + ".bai"))

    def max_read_len(self, first_n=101):
        """Largest query length among the first records of the file (the sample's READLEN)."""
        out = C.c_int32()
        if self._lib.tredbam_max_read_len(self._h, int(first_n), C.byref(out)) != 0:
            raise ValueError(self._err())
        return int(out.value)

    def tid(self, chrom):
        return self._tid.get(chrom, -1)

    def scan(self, sites, alts, readlen, pad=1000, flank=9, pe_reach=10000, span=1000, use_alts=True,
             want_depth=True, want_pe=True, pe=None, alt=None):
        """tredbam_scan: `sites` (SITE_DTYPE) and `alts` (REGION_DTYPE) -> (units SCAN_UNIT_DTYPE, dict of pool
        arrays).  One native call; the GIL is released while it runs.  pe = (results WALK_RESULT_DTYPE per site, global
        pool, target pool): pair lengths computed where the blocks were inflated (tredbam_scan_pe); alt (with pe): the
        results ALT_RESULT_DTYPE of the walks over the alternative loci, one per entry of `alts` (tredbam_scan_walked)."""
        sites = np.ascontiguousarray(sites, SITE_DTYPE)
        alts = np.ascontiguousarray(alts if len(alts) else np.zeros(1, REGION_DTYPE), REGION_DTYPE)
        units = np.zeros(len(sites), SCAN_UNIT_DTYPE)
        o = ScanOpts(int(readlen), int(pad), int(flank), int(pe_reach), int(span), int(bool(use_alts)),
                     int(bool(want_depth)), int(bool(want_pe)))
        if pe is not None:
            res = np.ascontiguousarray(pe[0], WALK_RESULT_DTYPE)
            gp = np.ascontiguousarray(pe[1] if len(pe[1]) else np.zeros(1, np.int32), np.int32)
            tp = np.ascontiguousarray(pe[2] if len(pe[2]) else np.zeros(1, np.int32), np.int32)
            if len(res) != len(sites):
                raise ValueError("one walk result per site")
            ok = res["status"] == 0
            if ok.any() and ((res["global_first"][ok] < 0).any() or (res["target_first"][ok] < 0).any()
                             or (res["global_first"][ok] + res["n_global"][ok]).max() > len(pe[1])
                             or (res["target_first"][ok] + res["n_target"][ok]).max() > len(pe[2])):
                raise ValueError("walk results point outside their pools")
            self._lib.tredbam_pe_pool_sizes(self._h, len(pe[1]), len(pe[2]))      # (the library checks every slice again)
            if alt is not None:
                ar = np.ascontiguousarray(alt, ALT_RESULT_DTYPE)
                if len(ar) < int((sites["alt_first"] + sites["n_alt"]).max() if len(sites) else 0):
                    raise ValueError("one alternative-locus result per region")
                if len(ar) == 0:
                    ar = np.zeros(1, ALT_RESULT_DTYPE)
                rc = self._lib.tredbam_scan_walked(self._h, sites.ctypes.data, len(sites), alts.ctypes.data, C.byref(o),
                                                   res.ctypes.data, gp.ctypes.data, tp.ctypes.data, ar.ctypes.data, units.ctypes.data)
            else:
                rc = self._lib.tredbam_scan_pe(self._h, sites.ctypes.data, len(sites), alts.ctypes.data, C.byref(o),
                                               res.ctypes.data, gp.ctypes.data, tp.ctypes.data, units.ctypes.data)
        else:
            rc = self._lib.tredbam_scan(self._h, sites.ctypes.data, len(sites), alts.ctypes.data, C.byref(o),
                                        units.ctypes.data)
        if rc != 0:
            raise ValueError(self._err())
        p = Pools()
        self._lib.tredbam_scan_pools(self._h, C.byref(p))
        n = p.n_reads
        pools = {"packed": _copy(p.packed, p.n_words, "<u4"), "word_off": _copy(p.word_off, n + 1, "<i8"),
                 "read_len": _copy(p.read_len, n, "<i4"), "seq4": _copy(p.seq4, _last(p.seq4_off, n), "u1"),
                 "seq4_off": _copy(p.seq4_off, n + 1, "<i8"), "names": C.string_at(p.names, _last(p.name_off, n)),
                 "name_off": _copy(p.name_off, n + 1, "<i8"), "name_id": _copy(p.name_id, n, "<i4"),
                 "global_lens": _copy(p.global_lens, p.n_global, "<i4"),
                 "target_lens": _copy(p.target_lens, p.n_target, "<i4")}
        return units, pools

    # ---- blocks inflated elsewhere (the GPU's batch decoder, _lib.Inflater) ----
    def plan(self, sites, alts, readlen, pad=1000, flank=9, pe_reach=10000, span=1000, use_alts=True, want_depth=True,
             want_pe=True, extra=()):
        """tredbam_plan: the BGZF blocks scan(...) with the same arguments will read -> (n_blocks, bytes their payloads
        take in a staging buffer, bytes they inflate to).  extra: (contig, start, end) of other queries to cover."""
        sites = np.ascontiguousarray(sites, SITE_DTYPE)
        alts = np.ascontiguousarray(alts if len(alts) else np.zeros(1, REGION_DTYPE), REGION_DTYPE)
        o = ScanOpts(int(readlen), int(pad), int(flank), int(pe_reach), int(span), int(bool(use_alts)),
                     int(bool(want_depth)), int(bool(want_pe)))
        cb, ob = C.c_int64(), C.c_int64()
        ex = np.zeros(max(len(extra), 1), REGION_DTYPE)
        for k, (contig, lo, hi) in enumerate(extra):
            ex[k] = (self._tid.get(contig, -1), lo, hi)
        n = self._lib.tredbam_plan(self._h, sites.ctypes.data, len(sites), alts.ctypes.data, C.byref(o), ex.ctypes.data, len(extra),
                                   C.byref(cb), C.byref(ob))
        if n < 0:
            raise ValueError(self._err())
        return int(n), cb.value, ob.value

    def plan_walks(self, sites, readlen, pad=1000, flank=9, pe_reach=10000, span=1000):
        """tredbam_plan_walks (after plan() with the same arguments): (tasks WALK_TASK_DTYPE per site, chunks
        WALK_CHUNK_DTYPE) of the pair-length walks, for tredgpu's inflate_walk."""
        sites = np.ascontiguousarray(sites, SITE_DTYPE)
        o = ScanOpts(int(readlen), int(pad), int(flank), int(pe_reach), int(span), 0, 1, 1)
        tasks = np.zeros(len(sites), WALK_TASK_DTYPE)
        cap = 8 * len(sites) + 16
        while True:
            chunks = np.zeros(cap, WALK_CHUNK_DTYPE)
            n = self._lib.tredbam_plan_walks(self._h, sites.ctypes.data, len(sites), C.byref(o), tasks.ctypes.data,
                                             chunks.ctypes.data, cap)
            if n == -3:
                cap *= 4
                continue
            if n < 0:
                raise ValueError(self._err())
            return tasks, chunks[:n]

    def plan_alt_walks(self, sites, alts, readlen, pad=1000, flank=9, pe_reach=10000, span=1000, use_alts=True):
        """tredbam_plan_alt_walks (after plan()): (tasks WALK_TASK_DTYPE, one per entry of `alts`; chunks) of the walks over
        the alternative loci."""
        sites = np.ascontiguousarray(sites, SITE_DTYPE)
        n_alts = len(alts)
        alts = np.ascontiguousarray(alts if n_alts else np.zeros(1, REGION_DTYPE), REGION_DTYPE)
        o = ScanOpts(int(readlen), int(pad), int(flank), int(pe_reach), int(span), int(bool(use_alts)), 1, 1)
        tasks = np.zeros(max(n_alts, 1), WALK_TASK_DTYPE)
        cap = 4 * n_alts + 16
        while True:
            chunks = np.zeros(cap, WALK_CHUNK_DTYPE)
            n = self._lib.tredbam_plan_alt_walks(self._h, sites.ctypes.data, len(sites), alts.ctypes.data, n_alts, C.byref(o),
                                                 tasks.ctypes.data, chunks.ctypes.data, cap)
            if n == -3:
                cap *= 4
                continue
            if n < 0:
                raise ValueError(self._err())
            return tasks[:n_alts], chunks[:n]

    def plan_region_walks(self, regions):
        """tredbam_plan_region_walks (after plan() with the regions among its `extra`): (tasks WALK_TASK_DTYPE, chunks) of plain
        region walks -- regions: [(contig, start, end)]; a task's window is its region and it forms no pairs (span 0)."""
        rg = np.zeros(max(len(regions), 1), REGION_DTYPE)
        for k, (contig, lo, hi) in enumerate(regions):
            rg[k] = (self._tid.get(contig, -1), lo, hi)
        tasks = np.zeros(max(len(regions), 1), WALK_TASK_DTYPE)
        cap = 8 * len(regions) + 16
        while True:
            chunks = np.zeros(cap, WALK_CHUNK_DTYPE)
            n = self._lib.tredbam_plan_region_walks(self._h, rg.ctypes.data, len(regions), tasks.ctypes.data, chunks.ctypes.data, cap)
            if n == -3:
                cap *= 4
                continue
            if n < 0:
                raise ValueError(self._err())
            return tasks[:len(regions)], chunks[:n]

    def plan_blocks(self):
        """tredbam_plan_blocks: (compressed offset, compressed length, trailer CRC-32, read-by-the-scan-itself flag) of
        the planned blocks, in the plan's (file) order."""
        n = self._lib.tredbam_plan_blocks(self._h, None, None, None, None)
        coff, clen = np.zeros(n, np.int64), np.zeros(n, np.int32)
        crc, host = np.zeros(n, np.uint32), np.zeros(n, np.uint8)
        if n:
            self._lib.tredbam_plan_blocks(self._h, coff.ctypes.data, clen.ctypes.data, crc.ctypes.data, host.ctypes.data)
        return coff, clen, crc, host

    def plan_fill(self, comp_addr, comp_base, out_base, comp_off, out_off):
        """Copies the planned payloads into the staging buffer at comp_addr (from byte comp_base on) and writes this
        sample's n + 1 entries of the two offset arrays (int64 numpy views of the inflater's)."""
        if self._lib.tredbam_plan_fill(self._h, comp_addr, comp_base, out_base, comp_off.ctypes.data, out_off.ctypes.data) != 0:
            raise ValueError(self._err())

    def preload(self, out_addr, out_off, status, crc=None):
        """Hands the inflated blocks of the plan in (pointers only: the staging buffer must outlive the scan).  crc: the
        decoder's CRC-32 of every block it wrote -- blocks whose checksum equals their trailer's count as verified,
        the others are left to the scan."""
        status = np.ascontiguousarray(status, np.int32)
        if crc is not None:
            crc = np.ascontiguousarray(crc, np.uint32)
            n = self._lib.tredbam_preload_crc(self._h, out_addr, out_off.ctypes.data, status.ctypes.data, crc.ctypes.data)
        else:
            n = self._lib.tredbam_preload(self._h, out_addr, out_off.ctypes.data, status.ctypes.data)
        if n < 0:
            raise ValueError(self._err())
        return n

    def preload_clear(self):
        """Forgets the preloaded blocks; (block loads served from them, block loads inflated here) since preload."""
        h, m = C.c_int64(), C.c_int64()
        self._lib.tredbam_preload_clear(self._h, C.byref(h), C.byref(m))
        return h.value, m.value

    def pe_lengths(self, chrom, start, end, tstart, tend, span):
        """(global_lens, target_lens) of PEextractor (bam_parser.py:316-369) for the window [start, end)."""
        if chrom not in self._tid:
            raise ValueError("invalid contig `{}`".format(chrom))
        tid, start = self._tid[chrom], max(0, int(start))
        cap_g, cap_t = 8192, 1024
        while True:
            g, t = (C.c_int32 * cap_g)(), (C.c_int32 * cap_t)()
            ng, nt = C.c_int64(), C.c_int64()
            rc = self._lib.tredbam_pe_lengths(self._h, tid, start, int(end), int(tstart), int(tend), int(span),
                                              g, cap_g, C.byref(ng), t, cap_t, C.byref(nt))
            if rc == -9:
                raise TypeError(self._err())     # `None - int` in the reference's get_target_length
            if rc != 0:
                raise ValueError(self._err())
            if ng.value <= cap_g and nt.value <= cap_t:
                return list(g[:ng.value]), list(t[:nt.value])
            cap_g, cap_t = max(cap_g, ng.value), max(cap_t, nt.value)


def _last(off_ptr, n):
    """off[n] of an int64 offset array at C address off_ptr."""
    return int(C.cast(off_ptr, C.POINTER(C.c_int64))[n]) if off_ptr else 0


def AlignmentFile(path, mode="rb"):
    """pysam.AlignmentFile stand-in: native when libtredbam.so is there, pure Python otherwise."""
    return NativeAlignmentFile(path, mode) if _native() is not None else PyAlignmentFile(path, mode)
