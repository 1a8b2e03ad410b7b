"""Synthetic 30x paired-end BAM files around the TRED loci (BASELINE configs 3-5: "1k synthetic 30x 150 bp BAMs").

There is no reference counterpart: tredparse only ever reads real BAMs.  A sample is simulated at the level a BAM
shows it -- sorted alignment records with flags, mate fields and CIGARs -- so that the whole product path (native
BGZF/BAI reader, read selection, pair-length extraction, depth) is exercised, not only the kernels:

  * per locus two haplotypes  flank + prefix + repeat * h + suffix + flank  (flank 10.5 kb: the paired-end window of
    bam_parser.py:328-333 is +-10 kb), fragments with insert ~ N(350, 80^2), both mates 150 bp;
  * a mate lying inside the tract has no anchor: it is stored unmapped at its mate's position (flag 0x4, the mate
    carries 0x8) -- the reads parse() keeps regardless of position (bam_parser.py:207-214) -- or, for a share of
    them, "mismapped" into one of the locus' alternative regions with its mate field pointing home (the reads the
    ALT rescue of :217-243 brings back);
  * reads crossing a tract whose length differs from the reference carry an insertion or deletion in their CIGAR;
    a few percent of the plain reads are soft-clipped, a percent of the pairs are duplicates, some reads QC-fail,
    some have a secondary copy -- the flags depth (:404-411) and PEextractor (:334-340) look at;
  * per-base substitutions, rare N.

`write_bam` emits BGZF blocks (records never straddle a block) and the .bai (bins + 16 kb linear index) per the SAM
specification.  `expected_scan` derives, in numpy and from the record table alone, what the front end has to find
for a locus (selected reads in order, depth, pair lengths): the independent statement tests compare the native
scan with, and the source of the packed batch "for the same seed".
"""
import os
import struct
import zlib

import numpy as np

from .synth import SynthParams, _COMP, encode, load_loci

CONTIGS = ["chr{}".format(i) for i in range(1, 23)] + ["chrX", "chrY", "chr22_KI270733v1_random", "chrUn_GL000220v1"]
CONTIG_LEN = 250000000
REACH = 10500                  # simulated flank on either side of the tract
FUNMAP, FMUNMAP, FREV, FMREV, FR1, FR2, FSEC, FQC, FDUP = 0x4, 0x8, 0x10, 0x20, 0x40, 0x80, 0x100, 0x200, 0x400
_NIB = np.array([1, 2, 4, 8, 15], np.uint8)         # A C G T N as BAM 4-bit codes
OP_M, OP_I, OP_D, OP_S = 0, 1, 2, 4


class Records(object):
    """Struct of arrays, one entry per alignment record, in file order."""
    FIELDS = ("tid", "pos", "flag", "mtid", "mpos", "tlen", "frag", "locus", "n_cig", "cig", "codes")

    def __init__(self, **kw):
        for k in self.FIELDS:
            setattr(self, k, kw[k])

    def __len__(self):
        return len(self.tid)

    def take(self, idx):
        return Records(**{k: getattr(self, k)[idx] for k in self.FIELDS})

    @staticmethod
    def concat(parts):
        return Records(**{k: np.concatenate([getattr(p, k) for p in parts]) for k in Records.FIELDS})

    @property
    def ref_end(self):
        """pysam reference_end: pos + reference-consuming CIGAR lengths; -1 for unmapped reads."""
        op, ln = self.cig & 15, self.cig >> 4
        span = np.where(np.isin(op, (OP_M, OP_D, 3, 7, 8)), ln, 0).sum(axis=1)          # M D N = X consume the reference
        return np.where((self.flag & FUNMAP) != 0, -1, self.pos + span).astype(np.int64)

    def names(self, sample):
        return ["{}.{:02d}.{:07d}".format(sample, l, f) for l, f in zip(self.locus.tolist(), self.frag.tolist())]


def _locus_records(rng, li, locus, h_pair, p, tid_of, alt_rate):
    """All records of one locus of one sample (unsorted)."""
    L = p.readlen
    prefix, suffix = encode(locus["prefix"]), encode(locus["suffix"])
    period = len(locus["repeat"])
    chrom, span = locus["repeat_location"].split(":")
    start, end = (int(x) for x in span.split("-"))
    ref_len = end - start + 1
    tid = tid_of[chrom]
    alts = []
    for a in locus.get("alts", "").split("|"):
        if a:
            c, se = a.split(":")
            s, e = (int(x) for x in se.split("-"))
            if c in tid_of and e - s > 60:
                alts.append((tid_of[c], s, e))
    rep = encode(locus["repeat"].replace("N", "ACGT"[int(rng.integers(0, 4))]))
    rs = REACH + len(prefix)                           # tract start in haplotype coordinates
    out = []
    frag_base = 0
    for hap in range(2):
        hlen = int(h_pair[hap]) * period
        tot = 2 * REACH + len(prefix) + len(suffix) + hlen
        nf = int(rng.poisson((p.coverage / 2.0) * tot / (2.0 * L)))
        ins = np.clip(np.rint(rng.normal(p.ins_mean, p.ins_sd, nf)), L, 999).astype(np.int64)
        fs = (rng.random(nf) * (tot - ins)).astype(np.int64)
        re_ = rs + hlen - 1
        frag = frag_base + np.arange(nf)
        frag_base += nf

        def to_ref(x):
            """haplotype coordinate -> reference coordinate (positions inside the tract clamp to its reference copy)."""
            inside = np.clip(x - rs, 0, ref_len - 1)
            return np.where(x < rs, start + (x - rs), np.where(x <= re_, start + inside, end + 1 + (x - re_ - 1)))

        mates = []
        for m in range(2):
            s = fs if m == 0 else fs + ins - L
            e = s + L - 1
            lost = (s >= rs - 10) & (e <= re_ + 10)          # no usable flank: the aligner cannot place it
            pos = to_ref(s)
            rend = to_ref(e) + 1
            rspan = rend - pos
            # CIGAR: plain, or an indel where the tract length differs from the reference's
            cig = np.zeros((nf, 3), np.uint32)
            n_cig = np.ones(nf, np.int32)
            cig[:, 0] = (L << 4) | OP_M
            short, long_ = rspan < L, rspan > L
            left = np.clip(np.minimum(rs - s + 3, rspan - 1), 1, None)      # bases before the indel
            left = np.minimum(left, np.minimum(rspan, L) - 1)
            for sel, op, gap in ((short, OP_I, L - rspan), (long_, OP_D, rspan - L)):
                k = np.nonzero(sel & ~lost)[0]
                cig[k, 0] = (left[k].astype(np.uint32) << 4) | OP_M
                cig[k, 1] = (gap[k].astype(np.uint32) << 4) | op
                cig[k, 2] = ((np.minimum(rspan, L)[k] - left[k]).astype(np.uint32) << 4) | OP_M
                n_cig[k] = 3
            plain = (rspan == L) & ~lost
            clip = plain & (rng.random(nf) < 0.03)
            k = np.nonzero(clip)[0]
            amount = rng.integers(5, 31, len(k))
            lead = rng.random(len(k)) < 0.5
            pos = pos.copy()
            pos[k[lead]] += amount[lead]
            cig[k[lead], 0] = (amount[lead].astype(np.uint32) << 4) | OP_S
            cig[k[lead], 1] = ((L - amount[lead]).astype(np.uint32) << 4) | OP_M
            cig[k[~lead], 0] = ((L - amount[~lead]).astype(np.uint32) << 4) | OP_M
            cig[k[~lead], 1] = (amount[~lead].astype(np.uint32) << 4) | OP_S
            n_cig[k] = 2
            # bases from the haplotype: random flanks, prefix / repeat / suffix by position
            rel = (s[:, None] - REACH + np.arange(L, dtype=np.int64)[None, :]).astype(np.int32)
            bases = rng.integers(0, 4, rel.shape, dtype=np.uint8)
            rrel = rel - len(prefix)
            srel = rrel - hlen
            np.copyto(bases, prefix[np.clip(rel, 0, len(prefix) - 1)], where=(rel >= 0) & (rel < len(prefix)))
            np.copyto(bases, rep[np.remainder(rrel, period)], where=(rrel >= 0) & (srel < 0))
            np.copyto(bases, suffix[np.clip(srel, 0, len(suffix) - 1)], where=(srel >= 0) & (srel < len(suffix)))
            r = rng.integers(0, 65536, bases.shape, dtype=np.uint16)
            t_sub, t_n = int(p.sub * 65536), int((p.sub + p.nrate) * 65536)
            sub = r < t_sub
            bases[sub] = (bases[sub] + rng.integers(1, 4, int(sub.sum()), dtype=np.uint8)) % 4
            bases[(r >= t_sub) & (r < t_n)] = 4
            mates.append(dict(lost=lost, pos=pos, cig=cig, n_cig=n_cig, bases=bases))
        a, b = mates
        both_lost = a["lost"] & b["lost"]                     # would sit in the unmapped tail of the file: never fetched
        dup = rng.random(nf) < 0.01
        qc = rng.random((2, nf)) < 0.005
        to_alt = (rng.random(nf) < alt_rate) if alts else np.zeros(nf, bool)
        alt_pick = rng.integers(0, max(len(alts), 1), nf)
        for m, (me, other) in enumerate(((a, b), (b, a))):
            keep = ~both_lost
            flag = np.full(nf, 0x1 | (FR1 if m == 0 else FR2), np.int32)
            flag |= np.where(m == 1, FREV, FMREV)
            flag |= np.where(dup, FDUP, 0) | np.where(qc[m], FQC, 0)
            flag |= np.where(other["lost"] & ~to_alt, FMUNMAP, 0)
            rtid = np.full(nf, tid, np.int32)
            pos = me["pos"].copy()
            mtid = np.full(nf, tid, np.int32)
            mpos = other["pos"].copy()
            cig, n_cig, codes = me["cig"].copy(), me["n_cig"].copy(), me["bases"].copy()
            lost = me["lost"]
            # mates of lost reads that went to an alternative locus point there
            if alts:
                at = np.array([x[0] for x in alts], np.int32)[alt_pick]
                a_lo = np.array([x[1] for x in alts], np.int64)[alt_pick]
                a_hi = np.array([x[2] for x in alts], np.int64)[alt_pick]
                alt_pos = a_lo - L + 20 + ((a_hi - a_lo + L - 40) * ((frag * 2654435761 % 1000) / 1000.0)).astype(np.int64)
                go = lost & to_alt
                rtid[go], pos[go] = at[go], alt_pos[go]
                cig[go] = 0
                cig[go, 0] = (L << 4) | OP_M
                n_cig[go] = 1
                og = other["lost"] & to_alt
                mtid[og], mpos[og] = at[og], alt_pos[og]
            un = lost & ~to_alt
            flag[un] |= FUNMAP
            flag[un] &= ~FREV
            pos[un] = other["pos"][un]                       # an unmapped read is stored where its mate lies
            n_cig[un] = 0
            cig[un] = 0
            flip = un & (rng.random(nf) < 0.5)                # unmapped reads come in sequencing orientation
            codes[flip] = _COMP[codes[flip][:, ::-1]]
            tl = np.where(lost | other["lost"], 0, (1 if m == 0 else -1) * ins)
            idx = np.nonzero(keep)[0]
            out.append(Records(tid=rtid[idx], pos=pos[idx], flag=flag[idx], mtid=mtid[idx], mpos=mpos[idx],
                               tlen=tl[idx].astype(np.int32), frag=frag[idx],
                               locus=np.full(len(idx), li, np.int32), n_cig=n_cig[idx], cig=cig[idx], codes=codes[idx]))
        # a few secondary copies of mapped first mates, right behind the original
        sec = np.nonzero(~a["lost"] & ~both_lost & (rng.random(nf) < 0.005))[0]
        if len(sec):
            r0 = out[-2]
            pick = np.nonzero(np.isin(r0.frag, frag[sec]))[0]
            extra = r0.take(pick)
            extra.flag = extra.flag | FSEC
            out.append(extra)
    recs = Records.concat(out)
    recs.frag = recs.frag.astype(np.int64)
    return recs


BACKGROUND_LOCUS = 99          # the `locus` field (and name part) of whole-genome background reads


def background_windows(loci):
    """What a whole-genome BAM holds and the locus-only samples lack, as far as this path ever reads it: the 16 kb index
    windows the walks over the loci's alternative regions go through (a region query starts parsing at the linear index'
    offset of the window that holds its start: htslib, and csrc/bamread.cpp region_chunks, read everything from there to the
    region's end) and, when a locus is X-linked, the five chrY windows of the sex inference (bam_parser.y_regions).
    [(contig index, start, end)] -- merged, and without what lies within REACH + 2 kb of a locus (those stretches are the
    loci's own, simulated with their haplotypes)."""
    from .bam_parser import y_regions
    tid_of = {c: i for i, c in enumerate(CONTIGS)}
    spans = []
    for locus in loci:
        for a in locus.get("alts", "").split("|"):
            if a:
                c, se = a.split(":")
                lo, hi = (int(x) for x in se.split("-"))
                if c in tid_of:
                    spans.append((tid_of[c], (lo >> 14) << 14, hi))
    if any(str(l.get("inheritance", "")).startswith("X") for l in loci):
        spans += [(tid_of[c], (lo >> 14) << 14, hi) for c, lo, hi in y_regions("hg38") if c in tid_of]
    own = []
    for locus in loci:
        c, se = locus["repeat_location"].split(":")
        lo, hi = (int(x) for x in se.split("-"))
        own.append((tid_of[c], lo - REACH - 2000, hi + REACH + 2000))
    merged = []
    for t, lo, hi in sorted(spans):
        if merged and merged[-1][0] == t and lo <= merged[-1][2]:
            merged[-1][2] = max(merged[-1][2], hi)
        else:
            merged.append([t, lo, hi])
    out = []
    for t, lo, hi in merged:                   # cut the loci's own stretches out
        pieces = [(lo, hi)]
        for ot, olo, ohi in own:
            if ot == t:
                pieces = [q for a, b in pieces for q in ((a, min(b, olo)), (max(a, ohi), b)) if q[1] - q[0] > 0]
        out += [(t, a, b) for a, b in pieces]
    return out


def background_records(rng, windows, p):
    """Plain read pairs at p.coverage over `windows` (background_windows): random bases, one M operation, proper pairs
    whose mates lie next to them -- none points at a locus, so none is ever selected; the walks only have to get past them."""
    L = p.readlen
    parts = []
    frag0 = 0
    for t, lo, hi in windows:
        nf = int(rng.poisson(p.coverage * (hi - lo) / (2.0 * L)))
        if nf == 0:
            continue
        ins = np.clip(np.rint(rng.normal(p.ins_mean, p.ins_sd, nf)), L, 999).astype(np.int64)
        fs = lo + (rng.random(nf) * max(1, hi - lo)).astype(np.int64)
        frag = frag0 + np.arange(nf, dtype=np.int64)
        frag0 += nf
        for mate in range(2):
            pos = fs if mate == 0 else fs + ins - L
            other = fs + ins - L if mate == 0 else fs
            flag = np.full(nf, 0x1 | 0x2 | (FR1 | FMREV if mate == 0 else FR2 | FREV), np.int64)
            cig = np.zeros((nf, 3), np.int64)
            cig[:, 0] = (L << 4) | OP_M
            parts.append(Records(tid=np.full(nf, t, np.int64), pos=pos, flag=flag, mtid=np.full(nf, t, np.int64), mpos=other,
                                 tlen=np.where(mate == 0, ins, -ins), frag=frag, locus=np.full(nf, BACKGROUND_LOCUS, np.int64),
                                 n_cig=np.ones(nf, np.int64), cig=cig, codes=rng.integers(0, 4, (nf, L)).astype(np.uint8)))
    return parts


def simulate_sample(seed, loci, p=None, h_pairs=None, alt_rate=0.3, wgs_like=False):
    """Records of one sample over `loci` (entries of data/treds.json), sorted as a coordinate-sorted BAM is.
    wgs_like: also background reads over everything else a whole-genome BAM makes this path read through
    (background_windows).  Returns (Records, h_true int[n_loci, 2])."""
    p = p or SynthParams()
    rng = np.random.default_rng(seed)
    tid_of = {c: i for i, c in enumerate(CONTIGS)}
    if h_pairs is None:
        h_pairs = rng.integers(p.min_units, p.max_units + 1, (len(loci), 2))
        if p.expanded_max > 0:
            big = rng.random(len(loci)) < p.expanded_frac
            h_pairs[big, 1] = rng.integers(p.max_units, p.expanded_max + 1, int(big.sum()))
    h_pairs = np.sort(np.asarray(h_pairs, np.int64), axis=1)
    parts = [_locus_records(rng, li, locus, h_pairs[li], p, tid_of, alt_rate) for li, locus in enumerate(loci)]
    if wgs_like:
        ref = parts[0]
        for b in background_records(rng, background_windows(loci), p):
            for k in Records.FIELDS:           # (same dtypes and CIGAR width as the loci's records)
                v = getattr(b, k)
                setattr(b, k, v.astype(getattr(ref, k).dtype) if k != "cig" else v[:, :ref.cig.shape[1]].astype(ref.cig.dtype))
            parts.append(b)
    recs = Records.concat(parts)
    order = np.lexsort((np.arange(len(recs)), recs.pos, recs.tid))      # stable: mates / copies keep their order
    return recs.take(order), h_pairs.astype(np.int32)


# ---- BAM + BAI -------------------------------------------------------------------------------------------------
def _reg2bin(beg, end):
    """SAM specification 5.3 (vectorised): smallest bin containing [beg, end)."""
    end = end - 1
    out = np.zeros(len(beg), np.int64)
    done = np.zeros(len(beg), bool)
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        hit = ~done & ((beg >> shift) == (end >> shift))
        out[hit] = base + (beg[hit] >> shift)
        done |= hit
    return out


def _bgzf_block(data, level):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    head = struct.pack("<4BI2BH2BHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, len(body) + 25)
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


_EOF_BLOCK = _bgzf_block(b"", 6)


def _record_sizes(recs, sample, no_seq=None, aux=b"", lengths=None):
    """(bytes of every record without its 4-byte block_size, name length incl. NUL, bytes of a packed sequence).  no_seq: a mask
    of records written WITHOUT their sequence (SEQ and QUAL '*', l_seq 0: what some aligners leave of secondary alignments)."""
    n = len(recs)
    L = recs.codes.shape[1]
    name_len = len("{}.{:02d}.{:07d}".format(sample, 0, 0)) + 1 if n else 1
    if n:                                   # (names are fixed-width as long as the counters stay inside their fields)
        name_len = max(name_len, len("{}.{:02d}.{:07d}".format(sample, int(recs.locus.max()), int(recs.frag.max()))) + 1)
    seq_len = (L + 1) // 2
    Ls = np.full(n, L, np.int64) if lengths is None else np.asarray(lengths, np.int64)
    body = (Ls + 1) // 2 + Ls
    if no_seq is not None:
        body = np.where(no_seq, 0, body)
    return 32 + name_len + 4 * recs.n_cig.astype(np.int64) + body + len(aux), name_len, seq_len


def _encode_records(recs, sample, name_len, bins, decoy_mask=None, no_seq=None, aux=b"", lengths=None):
    """The BAM bytes of `recs` (block_size word + record, one after the other) and the records' byte offsets in them."""
    n = len(recs)
    L = recs.codes.shape[1]
    names = recs.names(sample)
    name_mat = np.zeros((n, name_len), np.uint8)
    if n:
        width = len(names[0])
        if all(len(x) == width for x in (names[0], names[-1])) and width == name_len - 1:
            name_mat[:, :width] = np.frombuffer("".join(names).encode(), np.uint8).reshape(n, width)
        else:
            for i, x in enumerate(names):
                name_mat[i, :len(x)] = np.frombuffer(x.encode(), np.uint8)
    nib = _NIB[recs.codes]
    Ls = np.full(n, L, np.int64) if lengths is None else np.asarray(lengths, np.int64)
    if lengths is not None:
        nib = np.where(np.arange(L)[None, :] < Ls[:, None], nib, 0).astype(np.uint8)      # (nothing behind a shortened read's last base)
    if L % 2:
        nib = np.concatenate([nib, np.zeros((n, 1), np.uint8)], axis=1)
    seq = (nib[:, 0::2] << 4) | nib[:, 1::2]
    seqw = (Ls + 1) // 2
    with_seq = np.ones(n, bool) if no_seq is None else ~np.asarray(no_seq, bool)
    size = 32 + name_len + 4 * recs.n_cig + np.where(with_seq, seqw + Ls, 0) + len(aux)   # without the 4-byte block_size
    off = np.zeros(n + 1, np.int64)
    np.cumsum(size + 4, out=off[1:])
    flat = np.zeros(int(off[-1]), np.uint8)
    fixed = np.zeros(n, np.dtype([("bs", "<i4"), ("tid", "<i4"), ("pos", "<i4"), ("l_name", "u1"), ("mapq", "u1"),
                                  ("bin", "<u2"), ("n_cig", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("mtid", "<i4"),
                                  ("mpos", "<i4"), ("tlen", "<i4")]))
    fixed["bs"], fixed["tid"], fixed["pos"], fixed["l_name"] = size, recs.tid, recs.pos, name_len
    fixed["mapq"] = np.where((recs.flag & FUNMAP) != 0, 0, 60)
    fixed["bin"], fixed["n_cig"], fixed["flag"], fixed["l_seq"] = bins, recs.n_cig, recs.flag, np.where(with_seq, Ls, 0)
    fixed["mtid"], fixed["mpos"], fixed["tlen"] = recs.mtid, recs.mpos, recs.tlen

    def scatter(col0, mat):
        idx = (off[:-1] + col0)[:, None] + np.arange(mat.shape[1])[None, :]
        flat[idx] = mat

    scatter(0, fixed.view(np.uint8).reshape(n, 36))
    scatter(36, name_mat)
    cig_bytes = np.ascontiguousarray(recs.cig.astype("<u4")).view(np.uint8).reshape(n, 12)
    for k in (1, 2, 3):
        sel = np.nonzero(recs.n_cig == k)[0]
        if len(sel):
            idx = (off[sel] + 36 + name_len)[:, None] + np.arange(4 * k)[None, :]
            flat[idx] = cig_bytes[sel, :4 * k]
    seq_at = 36 + name_len + 4 * recs.n_cig
    ws = np.nonzero(with_seq)[0]
    if lengths is None:
        idx = (off[ws] + seq_at[ws])[:, None] + np.arange(seq.shape[1])[None, :]
        flat[idx] = seq[ws]
        idx = (off[ws] + seq_at[ws] + seq.shape[1])[:, None] + np.arange(L)[None, :]
        flat[idx] = 0xff                                                    # no base qualities
    else:                                                                   # every record its own width
        idx = (off[ws] + seq_at[ws])[:, None] + np.arange(seq.shape[1])[None, :]
        keep = np.arange(seq.shape[1])[None, :] < seqw[ws][:, None]
        flat[idx[keep]] = seq[ws][keep]
        idx = (off[ws] + seq_at[ws] + seqw[ws])[:, None] + np.arange(L)[None, :]
        keep = np.arange(L)[None, :] < Ls[ws][:, None]
        flat[idx[keep]] = 0xff
    if len(aux) and n:                                                     # the same optional fields behind every record
        idx = (off[1:] - len(aux))[:, None] + np.arange(len(aux))[None, :]
        flat[idx] = np.frombuffer(aux, np.uint8)
    if decoy_mask is not None and L >= 48 and n:
        decoy_mask = decoy_mask & with_seq & (Ls == L)
        fake = np.zeros(1, fixed.dtype)
        fake["bs"], fake["pos"], fake["l_name"], fake["mtid"], fake["mpos"] = 40, 5, 2, -1, -1
        sel = np.nonzero(decoy_mask)[0]
        rows = np.repeat(fake, len(sel))
        rows["tid"] = recs.tid[sel]
        body = np.concatenate([rows.view(np.uint8).reshape(len(sel), 36), np.tile(np.frombuffer(b"a\0", np.uint8), (len(sel), 1))], axis=1)
        at = (off[sel] + seq_at[sel] + seq.shape[1] + 4)[:, None] + np.arange(38)[None, :]
        flat[at] = body
    return flat, off


def trim_records(recs, lengths):
    """`recs` with every record cut to lengths[i] bases at the end of its CIGAR (reads trimmed before alignment): the last
    operation gives up the bases; a record whose last operation is too short for that keeps its length.  Returns (records --
    the CIGAR matrix is a copy, the other arrays are shared --, the lengths really applied)."""
    L = recs.codes.shape[1]
    want = np.minimum(np.asarray(lengths, np.int64), L)
    cig = recs.cig.copy()
    last = np.maximum(recs.n_cig.astype(np.int64) - 1, 0)
    rows = np.arange(len(recs))
    op_len = (cig[rows, last] >> 4).astype(np.int64)
    cut = L - want
    ok = (recs.n_cig > 0) & (cut > 0) & (op_len > cut)
    cig[rows[ok], last[ok]] = (((op_len[ok] - cut[ok]) << 4) | (cig[rows[ok], last[ok]] & 15)).astype(cig.dtype)
    out = Records(**{k: (cig if k == "cig" else getattr(recs, k)) for k in Records.FIELDS})
    return out, np.where(ok, want, L)


WRITE_SLICE = 150000          # records encoded at a time (the index matrices of the scatter are 8 bytes per record byte)


def write_bam(path, recs, sample="s", level=1, block=0xff00, split_records=False, decoys=0.0, decoy_seed=1, no_seq=None, aux=b"",
              lengths=None):
    """Write `recs` (sorted) as <path> and <path>.bai.  Returns the number of uncompressed bytes.  split_records: cut the
    record stream into blocks of `block` bytes wherever that falls (records then straddle blocks, as in files written by
    samtools) instead of at record boundaries.  decoys: that share of the reads gets base qualities that read as the
    head of a BAM record of the read's contig (tests of the device walk's guessed record starts: DESIGN 4.5).
    Whole-genome-shaped samples (two million records) are encoded WRITE_SLICE records at a time; the bytes are the same.
    no_seq: a mask of records written without sequence and qualities (l_seq 0).  aux: bytes of optional fields (tag, type,
    value ...) appended to every record, as aligners leave them (NM, MD, AS, RG ...).  lengths: every record's own sequence
    length (<= the codes' width; trim_records makes the CIGARs agree): reads trimmed before alignment."""
    n = len(recs)
    rend = recs.ref_end
    end_for_bin = np.where(rend > recs.pos, rend, recs.pos + 1)
    bins = _reg2bin(recs.pos.astype(np.int64), end_for_bin)
    no_seq = None if no_seq is None else np.asarray(no_seq, bool)
    lengths = None if lengths is None else np.asarray(lengths, np.int64)
    size, name_len, _ = _record_sizes(recs, sample, no_seq, aux, lengths)
    decoy_mask = (np.random.default_rng(decoy_seed).random(n) < decoys) if (decoys > 0 and n) else None
    off = np.zeros(n + 1, np.int64)
    np.cumsum(size + 4, out=off[1:])
    header = b"BAM\x01"
    text = "@HD\tVN:1.5\tSO:coordinate\n" + "".join("@SQ\tSN:{}\tLN:{}\n".format(c, CONTIG_LEN) for c in CONTIGS)
    header += struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(CONTIGS))
    for c in CONTIGS:
        header += struct.pack("<i", len(c) + 1) + c.encode() + b"\x00" + struct.pack("<i", CONTIG_LEN)
    # blocks: the header alone, then whole records
    if split_records:
        flat, off2 = _encode_records(recs, sample, name_len, bins, decoy_mask, no_seq, aux, lengths)
        assert np.array_equal(off, off2)
        blob = flat.tobytes()
        starts = list(range(0, len(blob), block)) or [0]
        voff = np.zeros(n + 1, np.int64)
        with open(path, "wb") as fp:
            fp.write(_bgzf_block(header, level))
            co = []
            for a in starts:
                co.append(fp.tell())
                fp.write(_bgzf_block(blob[a:a + block], level))
            co.append(fp.tell())
            fp.write(_EOF_BLOCK)
        co = np.array(co, np.int64)
        which = np.minimum(off // block, len(starts) - 1)                  # a position at a block's end is the next block's start
        which = np.where(off >= len(blob), len(starts), which)
        inside = np.where(which < len(starts), off - which * block, 0)
        v = (co[which] << 16) | inside
        _write_bai(path + ".bai", recs.tid, recs.pos.astype(np.int64), end_for_bin, bins, v[:-1], v[1:])
        return len(blob)
    cuts = [0]
    while cuts[-1] < n:
        k = int(np.searchsorted(off, off[cuts[-1]] + block, side="right")) - 1
        cuts.append(max(k, cuts[-1] + 1))
    voff = np.zeros(n + 1, np.int64)
    with open(path, "wb") as fp:
        fp.write(_bgzf_block(header, level))
        bi = 0
        while bi < len(cuts) - 1:
            # the blocks that start within the next WRITE_SLICE records (at least one), encoded together
            bj = bi + 1
            while bj < len(cuts) - 1 and cuts[bj + 1] - cuts[bi] <= WRITE_SLICE:
                bj += 1
            ra, rb = cuts[bi], cuts[bj]
            flat, off2 = _encode_records(recs.take(slice(ra, rb)), sample, name_len, bins[ra:rb],
                                         None if decoy_mask is None else decoy_mask[ra:rb], None if no_seq is None else no_seq[ra:rb], aux,
                                         None if lengths is None else lengths[ra:rb])
            blob = flat.tobytes()
            for a, b in zip(cuts[bi:bj], cuts[bi + 1:bj + 1]):
                co = fp.tell()
                voff[a:b] = (co << 16) | (off[a:b] - off[a])
                fp.write(_bgzf_block(blob[off[a] - off[ra]:off[b] - off[ra]], level))
            bi = bj
        voff[n] = fp.tell() << 16
        fp.write(_EOF_BLOCK)
    # a record's end offset: the next record's start, or the next block's start for the last record of a block
    vend = voff[1:].copy()
    for b in cuts[1:-1]:
        vend[b - 1] = voff[b]
    _write_bai(path + ".bai", recs.tid, recs.pos.astype(np.int64), end_for_bin, bins, voff[:-1], vend)
    return int(off[-1])


def _write_bai(path, tid, pos, end, bins, vbeg, vend):
    out = [b"BAI\x01", struct.pack("<i", len(CONTIGS))]
    for t in range(len(CONTIGS)):
        sel = np.nonzero(tid == t)[0]
        if not len(sel):
            out.append(struct.pack("<ii", 0, 0))
            continue
        b, vb, ve = bins[sel], vbeg[sel], vend[sel]
        chunks = {}
        for i in range(len(sel)):                     # records are in file order: extend the bin's last chunk
            lst = chunks.setdefault(int(b[i]), [])
            if lst and lst[-1][1] == int(vb[i]):
                lst[-1][1] = int(ve[i])
            else:
                lst.append([int(vb[i]), int(ve[i])])
        out.append(struct.pack("<i", len(chunks)))
        for k in sorted(chunks):
            out.append(struct.pack("<Ii", k, len(chunks[k])))
            for c in chunks[k]:
                out.append(struct.pack("<QQ", c[0], c[1]))
        n_win = int((end[sel].max() - 1) >> 14) + 1
        lin = np.zeros(n_win, np.int64)
        first = np.full(n_win, np.iinfo(np.int64).max, np.int64)
        for w0, w1, v in zip((pos[sel] >> 14).tolist(), ((end[sel] - 1) >> 14).tolist(), vb.tolist()):
            for w in range(w0, w1 + 1):
                if v < first[w]:
                    first[w] = v
        last = 0
        for w in range(n_win):                        # windows without a read inherit the previous offset (samtools)
            if first[w] != np.iinfo(np.int64).max:
                last = int(first[w])
            lin[w] = last
        out.append(struct.pack("<i", n_win) + lin.astype("<u8").tobytes())
    with open(path, "wb") as fp:
        fp.write(b"".join(out))


# ---- what the front end has to find (numpy, from the record table alone) ------------------------------------------
def expected_scan(recs, locus, readlen, alts=True, sample="s"):
    """For one locus: (indices of the selected reads in the order the reference aligns them, depth, global_lens,
    target_lens) by the rules of bam_parser.py:196-243, :316-369, :404-411."""
    tid_of = {c: i for i, c in enumerate(CONTIGS)}
    chrom, span = locus["repeat_location"].split(":")
    start, end = (int(x) for x in span.split("-"))
    tid = tid_of[chrom]
    rend = recs.ref_end
    epos = np.where(rend > recs.pos, rend, recs.pos + 1)
    unmapped = (recs.flag & FUNMAP) != 0

    def overlapping(t, lo, hi):
        return np.nonzero((recs.tid == t) & (recs.pos < hi) & (epos > lo))[0]

    lo, hi = max(0, start - 1000), end + 1000
    win = overlapping(tid, lo, hi)
    sel = win[unmapped[win] | ((recs.pos[win] >= max(0, start - readlen)) & (recs.pos[win] <= end + readlen))]
    picked = [sel]
    if alts:
        for a in locus.get("alts", "").split("|"):
            if not a:
                continue
            c, se = a.split(":")
            s, e = (int(x) for x in se.split("-"))
            if c not in tid_of:
                continue
            r = overlapping(tid_of[c], s, e)
            picked.append(r[(recs.mtid[r] == tid) & (recs.mpos[r] >= lo) & (recs.mpos[r] <= hi)])
    reads = np.concatenate(picked)
    counted = win[(recs.flag[win] & (FUNMAP | FSEC | FQC | FDUP)) == 0]
    depth = float((rend[counted] - recs.pos[counted]).sum()) / float(hi - lo + 1)
    pe = overlapping(tid, max(start - 10000, 0), end + 10000)
    pe = pe[((recs.flag[pe] & 0x1) != 0) & ((recs.flag[pe] & (FUNMAP | FDUP)) == 0)]
    key = recs.locus[pe].astype(np.int64) << 40 | recs.frag[pe]
    gl, tl, seen = [], [], {}
    for i, k in zip(pe.tolist(), key.tolist()):
        seen.setdefault(k, []).append(i)
    op, ln = recs.cig & 15, recs.cig >> 4
    for pair in seen.values():
        if len(pair) < 2:
            continue
        a, b = pair[:2]
        if (recs.flag[a] & FREV) or not (recs.flag[b] & FREV):
            continue
        lead = int(ln[a, 0]) if op[a, 0] == OP_S else 0
        last = int(recs.n_cig[b]) - 1
        trail = int(ln[b, last]) if last >= 0 and op[b, last] == OP_S else 0
        t = (int(rend[b]) + trail) - (int(recs.pos[a]) - lead)
        if t >= 1000:
            continue
        (tl if recs.pos[a] < start - 9 and rend[b] > end + 9 else gl).append(t)
    return reads, depth, gl, tl


def bench_loci():
    """The 30 loci with distinct coordinates (FXTAS/FXS and SBMA/AR share a region)."""
    return [l for l in load_loci() if l["name"] not in ("FXTAS", "AR")]


def make_bams(outdir, n_samples, seed=20260101, loci=None, p=None, workers=1, prefix="syn", wgs_like=False):
    """n_samples synthetic BAMs (<outdir>/<prefix>NNNN.bam + .bai); returns [(samplekey, path, h_true)].  wgs_like: with the
    background a whole-genome BAM adds (simulate_sample): ~15 x the records, ~2 GB of memory per worker while a file is made."""
    loci = loci or bench_loci()
    os.makedirs(outdir, exist_ok=True)
    tasks = [(seed + i, "{}{:04d}".format(prefix, i), outdir, loci, p, wgs_like) for i in range(n_samples)]
    if workers > 1 and n_samples > 1:
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=min(workers, n_samples)) as ex:
            return list(ex.map(_make_one, tasks))
    return [_make_one(t) for t in tasks]


def _make_one(task):
    seed, key, outdir, loci, p, wgs_like = task
    recs, h_true = simulate_sample(seed, loci, p, wgs_like=wgs_like)
    path = os.path.join(outdir, key + ".bam")
    write_bam(path, recs, sample=key)
    return key, path, h_true
