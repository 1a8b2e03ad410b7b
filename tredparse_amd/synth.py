"""Synthetic sample x locus units.

There is no reference counterpart: the reference only ever reads real BAMs.  The generator
produces, per unit, what tredparse's host front end (bam_parser.py:184-257 BamParser.parse read
selection, :316-369 PEextractor, :404-411 region_depth) would hand to the hot path for a diploid
sample sequenced at a given coverage with 150 bp paired reads:

  * reads: fragments are sampled from two haplotypes  flank + prefix + repeat*h + suffix + flank,
    insert ~ N(350, 80^2); both mates are emitted in reference orientation (as a BAM stores mapped
    reads).  A read is kept when it would be fetched and pass the position filter
    (start within [repeat_start - READLEN, repeat_end + READLEN], bam_parser.py:196-213); reads lying
    entirely inside the repeat are emitted as "unmapped with an anchored mate" in random orientation.
  * per-base errors: substitutions, rare indels, rare N.
  * depth (with the pileup inflation the reference's region_depth has), global / spanning pair lengths.

Everything is vectorised with numpy over all units of one locus so that 30k-unit batches build in
seconds; all randomness comes from one numpy Generator (seeded by the caller).
"""
import json
import os

import numpy as np

from . import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
_COMP = np.array([3, 2, 1, 0, 4], np.uint8)
_CODE = np.full(256, 4, np.uint8)
for _c, _v in zip("ACGTacgt", (0, 1, 2, 3, 0, 1, 2, 3)):
    _CODE[ord(_c)] = _v
_LETTERS = np.frombuffer(b"ACGTN", np.uint8)


def encode(s):
    return _CODE[np.frombuffer(s.encode("latin-1"), np.uint8)]


def decode(codes):
    return _LETTERS[np.asarray(codes, np.uint8)].tobytes().decode()


def load_loci():
    with open(os.path.join(HERE, "data", "treds.json")) as fp:
        return json.load(fp)["loci"]


class SynthParams:
    def __init__(self, coverage=30.0, readlen=150, ins_mean=350.0, ins_sd=80.0, sub=0.01, indel=0.001,
                 nrate=0.005, flank=1200, min_units=5, max_units=60, expanded_max=0, expanded_frac=0.0):
        self.coverage = coverage
        self.readlen = readlen
        self.ins_mean = ins_mean
        self.ins_sd = ins_sd
        self.sub = sub
        self.indel = indel
        self.nrate = nrate
        self.flank = flank
        self.min_units = min_units
        self.max_units = max_units
        self.expanded_max = expanded_max      # config 5: one allele up to this many units ...
        self.expanded_frac = expanded_frac    # ... in this fraction of the units


class LocusBatch:
    """All units of one locus: flat read matrix + per-unit metadata."""

    def __init__(self, name, reads, unit_read_off, h_true, depth, global_lens, gl_off, target_lens, tl_off):
        self.name = name
        self.reads = reads                  # uint8 [n_reads, L] base codes 0..4
        self.unit_read_off = unit_read_off  # int64 [n_units+1]
        self.h_true = h_true                # int32 [n_units, 2]
        self.depth = depth                  # float64 [n_units]
        self.global_lens = global_lens      # int32 pool
        self.gl_off = gl_off                # int64 [n_units+1]
        self.target_lens = target_lens
        self.tl_off = tl_off


def _mutate(rng, reads, p):
    """In-place substitutions / N; indels applied per affected read (rare)."""
    n, L = reads.shape
    r = rng.integers(0, 65536, reads.shape, dtype=np.uint16)
    t_sub, t_n = int(p.sub * 65536), int((p.sub + p.nrate) * 65536)
    sub = r < t_sub
    reads[sub] = (reads[sub] + rng.integers(1, 4, int(sub.sum()), dtype=np.uint8)) % 4
    isn = (r >= t_sub) & (r < t_n)
    reads[isn] = 4
    if p.indel > 0:
        hit = np.nonzero(rng.random(n) < 1 - (1 - 2 * p.indel) ** L)[0]
        pos = rng.integers(1, L - 1, len(hit))
        ins = rng.random(len(hit)) < 0.5
        newb = rng.integers(0, 4, len(hit), dtype=np.uint8)
        for i, q, is_ins, b in zip(hit, pos, ins, newb):
            row = reads[i]
            if is_ins:
                row[q + 1:] = row[q:-1].copy()
                row[q] = b
            else:
                row[q:-1] = row[q + 1:].copy()
                row[-1] = b
    return reads


def simulate_locus(rng, locus, n_units, p, h_pairs=None):
    """Simulate n_units diploid samples at one locus.  h_pairs: optional int array [n_units, 2]."""
    L = p.readlen
    prefix, suffix = encode(locus["prefix"]), encode(locus["suffix"])
    period = len(locus["repeat"])
    chrom, span = locus["repeat_location"].split(":")
    start, end = (int(x) for x in span.split("-"))
    ref_len = end - start + 1
    if h_pairs is None:
        h_pairs = rng.integers(p.min_units, p.max_units + 1, (n_units, 2))
        if p.expanded_max > 0:
            big = rng.random(n_units) < p.expanded_frac
            h_pairs[big, 1] = rng.integers(p.max_units, p.expanded_max + 1, int(big.sum()))
    h_pairs = np.sort(np.asarray(h_pairs, np.int64), axis=1)

    # fragments per haplotype: coverage/2 over the local haplotype
    all_reads, all_unit, all_order = [], [], []
    tl_lists = [[] for _ in range(n_units)]
    for hap in range(2):
        hlen = h_pairs[:, hap] * period                 # repeat tract length per unit
        tot = 2 * p.flank + len(prefix) + len(suffix) + hlen
        nfrag = rng.poisson((p.coverage / 2.0) * tot / (2.0 * L))
        unit = np.repeat(np.arange(n_units), nfrag)
        nf = len(unit)
        ins = np.clip(np.rint(rng.normal(p.ins_mean, p.ins_sd, nf)), L, 999).astype(np.int64)
        fstart = (rng.random(nf) * (tot[unit] - ins)).astype(np.int64)
        rs = p.flank + len(prefix)                      # repeat start in haplotype coordinates
        re_ = rs + hlen[unit] - 1                       # repeat end (inclusive)
        # spanning pairs (bam_parser.py:343-359): read1 starts left of start-9, read2 ends right of end+9
        a_start, b_end = fstart, fstart + ins - 1
        spanning = (a_start < rs - 9) & (b_end > re_ + 9) & (a_start + L - 1 < re_) & (b_end - L + 1 > rs)
        tlen_ref = ins - (hlen[unit] - ref_len)
        ok = spanning & (tlen_ref < 1000) & (tlen_ref > 0)
        for g, t in zip(unit[ok], tlen_ref[ok]):
            tl_lists[g].append(int(t))
        for mate in range(2):
            s = fstart if mate == 0 else fstart + ins - L
            e = s + L - 1
            inside = (s >= rs - 10) & (e <= re_ + 10)   # no usable flank: unmapped, mate-anchored
            near = (s >= rs - L) & (s <= re_ + L)       # position filter of parse()
            keep = near | inside
            idx = np.nonzero(keep)[0]
            if len(idx) == 0:
                continue
            # materialise the read bases from the haplotype (flanks are random per unit and position)
            rel = (s[idx, None] - p.flank + np.arange(L, dtype=np.int32)[None, :]).astype(np.int32)
            hl = hlen[unit[idx], None].astype(np.int32)
            bases = rng.integers(0, 4, rel.shape, dtype=np.uint8)     # flank bases (i.i.d.)
            rep = encode(locus["repeat"].replace("N", "ACGT"[int(rng.integers(0, 4))]))
            # one lookup table: [prefix | repeat * many | suffix] indexed by position class
            rrel = rel - len(prefix)
            srel = rrel - hl
            np.copyto(bases, prefix[np.clip(rel, 0, len(prefix) - 1)], where=(rel >= 0) & (rel < len(prefix)))
            np.copyto(bases, rep[np.remainder(rrel, period)], where=(rrel >= 0) & (srel < 0))
            np.copyto(bases, suffix[np.clip(srel, 0, len(suffix) - 1)], where=(srel >= 0) & (srel < len(suffix)))
            # unmapped reads come in sequencing orientation: reverse-complement half of them
            flip = inside[idx] & (rng.random(len(idx)) < 0.5)
            bases[flip] = _COMP[bases[flip][:, ::-1]]
            all_reads.append(bases)
            all_unit.append(unit[idx])
            all_order.append(s[idx])
    reads = np.concatenate(all_reads) if all_reads else np.zeros((0, L), np.uint8)
    unit = np.concatenate(all_unit) if all_unit else np.zeros(0, np.int64)
    order = np.concatenate(all_order) if all_order else np.zeros(0, np.int64)
    # BAM order within a unit: by position
    perm = np.lexsort((order, unit))
    reads, unit = reads[perm], unit[perm]
    reads = _mutate(rng, np.ascontiguousarray(reads), p)
    unit_read_off = np.zeros(n_units + 1, np.int64)
    np.cumsum(np.bincount(unit, minlength=n_units), out=unit_read_off[1:])

    # region_depth's pileup counts columns outside the window too (bam_parser.py:404-411)
    window = ref_len + 2000
    depth = p.coverage * (1.0 + 2.0 * (L - 1) / window) * rng.normal(1.0, 0.03, n_units)
    # global pairs within +-10 kb (bam_parser.py:328-359)
    n_gl = rng.poisson(20000.0 * p.coverage / (2.0 * L), n_units)
    gl_off = np.zeros(n_units + 1, np.int64)
    np.cumsum(n_gl, out=gl_off[1:])
    gl = np.clip(np.rint(rng.normal(p.ins_mean, p.ins_sd, int(gl_off[-1]))), L, 999).astype(np.int32)
    tl_off = np.zeros(n_units + 1, np.int64)
    np.cumsum([len(t) for t in tl_lists], out=tl_off[1:])
    tl = np.asarray([t for ts in tl_lists for t in ts], np.int32)
    return LocusBatch(locus["name"], reads, unit_read_off, h_pairs.astype(np.int32), depth, gl, gl_off, tl, tl_off)


def unit_params_for(locus, readlen, depth, n_global, n_target, pe_off, tl_off, ploidy=2, maxinsert=300,
                    fullsearch=False):
    """Fill one tredgpu_unit_params record from a locus table entry (meta.py:103-129)."""
    u = np.zeros((), _lib.UNIT_DTYPE)
    chrom, span = locus["repeat_location"].split(":")
    start, end = (int(x) for x in span.split("-"))
    u["period"] = len(locus["repeat"])
    u["readlen"] = readlen
    u["ploidy"] = ploidy
    u["maxinsert"] = maxinsert
    u["fullsearch"] = int(fullsearch)
    u["ref_len"] = end - start + 1                     # bam_parser.py:68
    u["minpe"] = end - start + 2 * 9 + 2               # bam_parser.py:361
    u["cutoff_risk"] = locus["cutoff_risk"]
    u["is_expansion"] = int(locus["mutation_nature"] == "increase")
    u["is_recessive"] = int(locus["inheritance"][-1] == "R")
    u["pe_off"], u["n_global"], u["tl_off"], u["n_target"] = pe_off, n_global, tl_off, n_target
    u["half_depth"] = depth / 2
    return u


class Batch:
    """A packed multi-locus batch in the C-ABI's layout (host numpy arrays)."""

    def __init__(self, loci, readlen):
        self.loci = loci
        self.readlen = readlen
        self.ladders = [(l["prefix"], l["repeat"], l["suffix"], -(-readlen // len(l["repeat"]))) for l in loci]
        self.hist_stride = max(l[3] for l in self.ladders) + 2


def _sim_task(args):
    seed, li, locus, n_samples, p = args
    lb = simulate_locus(np.random.default_rng([seed, li]), locus, n_samples, p)
    lb.packed, lb.read_off, lb.read_len = _lib.pack_codes(lb.reads)  # pack in the worker
    return lb


def build_batch(rng, loci, n_samples, p, maxinsert=300, fullsearch=False, workers=0):
    """n_samples x len(loci) units; unit index = locus_index * n_samples + sample.
    rng: numpy Generator or an int seed; workers > 1 simulates loci in a process pool (call it
    before the process touches the GPU)."""
    b = Batch(loci, p.readlen)
    seed = int(rng.integers(0, 2 ** 31)) if hasattr(rng, "integers") else int(rng)
    tasks = [(seed, li, locus, n_samples, p) for li, locus in enumerate(loci)]
    if workers and workers > 1 and len(tasks) > 1:
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=min(workers, len(tasks))) as ex:
            sims = list(ex.map(_sim_task, tasks))
    else:
        sims = [_sim_task(t) for t in tasks]
    parts = []
    read_base, gl_base, tl_base = 0, 0, 0
    gls, tls, uoffs, ulad, htrue, unit_blocks = [], [], [np.zeros(1, np.int64)], [], [], []
    for li, (locus, lb) in enumerate(zip(loci, sims)):
        parts.append(lb.reads)
        uoffs.append(lb.unit_read_off[1:] + read_base)
        read_base += len(lb.reads)
        ulad.append(np.full(n_samples, li, np.int32))
        htrue.append(lb.h_true)
        blk = np.zeros(n_samples, _lib.UNIT_DTYPE)
        blk[:] = unit_params_for(locus, p.readlen, 0.0, 0, 0, 0, 0, maxinsert=maxinsert, fullsearch=fullsearch)
        blk["half_depth"] = lb.depth / 2
        blk["pe_off"] = lb.gl_off[:-1] + gl_base
        blk["n_global"] = np.diff(lb.gl_off)
        blk["tl_off"] = lb.tl_off[:-1] + tl_base
        blk["n_target"] = np.diff(lb.tl_off)
        unit_blocks.append(blk)
        gls.append(lb.global_lens)
        tls.append(lb.target_lens)
        gl_base += len(lb.global_lens)
        tl_base += len(lb.target_lens)
    b.codes = np.concatenate(parts)
    b.n_reads = len(b.codes)
    b.packed = np.concatenate([lb.packed for lb in sims])
    woffs, wbase = [np.zeros(1, np.int64)], 0
    for lb in sims:
        woffs.append(lb.read_off[1:] + wbase)
        wbase += int(lb.read_off[-1])
    b.read_off = np.concatenate(woffs)
    b.read_len = np.concatenate([lb.read_len for lb in sims])
    b.unit_read_off = np.concatenate(uoffs).astype(np.int32)
    b.unit_ladder = np.concatenate(ulad)
    b.n_units = len(b.unit_ladder)
    b.units = np.concatenate(unit_blocks)
    b.global_lens = np.concatenate(gls).astype(np.int32) if gls else np.zeros(0, np.int32)
    b.target_lens = np.concatenate(tls).astype(np.int32) if tls else np.zeros(0, np.int32)
    b.h_true = np.concatenate(htrue)
    return b
