#!/usr/bin/env python3
"""Randomised campaign for the read selection on the device (include/tredgpu.h section 5: select_kernel,
pack_selected_kernel, tredgpu_genotype_selected) against the host's scan (tredbam_scan, bamread.cpp scan_impl -- the
restatement of BamParser.parse / BamDepth / PEextractor, tredparse/bam_parser.py:184-257, 316-369, 404-411).  GPU box.

Every round simulates a few samples (random coverage, read length, loci subset -- X-linked ones now and then, which brings
the chrY depth windows in), perturbs the records' flags (duplicates, secondary, QC-fail, unpaired, strand-flipped), writes
each as a BAM whose BGZF blocks are cut at a random size WITHOUT regard to record boundaries (200 bytes .. 64 KiB), picks
the options at random (alternative loci on / off, --useclippedreads, --fullsearch with a small --maxinsert) and sends the
samples through the product's feeder with the selection on (feeder._InflateFeeder(select=True) -> tred.genotype_scans).
One sample in seven carries a record without a sequence (SEQ '*'): where the selection takes it, the sample must come back to the
host's scan and lose that locus, as the reference does.  Compared per sample with scan_sample + the host-packed genotyping call: sex and chrY depth, per locus depth sum, read count
and pair-length slices; the selected reads' lengths, 4-bit sequences and names in order; tags, repeat counts, scores, calls,
marginals and joint entries bit for bit.  Prints one JSON line.

usage: python tools/fuzz_select.py [rounds = 20] [seed = 1]
"""
import json
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def compare(a, s, pieces, engine):
    """[what differs] between a device-selected sample (scan s, its genotyped pieces) and the host's scan of the same task."""
    from tredparse_amd import tred as t
    from tredparse_amd.bam_parser import scan_sample
    from tredparse_amd.engine import PackedUnits
    o = t._options(a)
    h = scan_sample(o["bam"], o["repo"], o["names"], clip=o["clip"], alts=o["alts"])
    bad = []
    if (s.gender, s.ydepth, s.readlen, s.opened, s.dropped) != (h.gender, h.ydepth, h.readlen, h.opened, h.dropped):
        bad.append("sample")
    for key in ("n_reads", "read_first", "depth_sum", "depth_status", "pe_status", "n_global", "n_target", "status"):
        if not (s.unit[key] == h.unit[key]).all():
            bad.append("unit." + key)
    if not (np.array_equal(s.depth, h.depth) and np.array_equal(s.ploidy, h.ploidy)):
        bad.append("depth")
    for k in range(len(s.names)):
        if not all(np.array_equal(x, y) for x, y in zip(s.pair_lengths(k), h.pair_lengths(k))):
            bad.append("pairs")
            break
    if bad:
        return bad, 0, 0
    if not (np.array_equal(s.read_len, h.read_len) and np.array_equal(s.seq4_off, h.seq4_off) and np.array_equal(s.seq4, h.seq4)):
        bad.append("sequences")
    if not (np.array_equal(s.name_off, h.name_off) and s.name_blob == h.name_blob):
        bad.append("names")
    ks = [k for k in range(len(h.names)) if k not in h.dropped]          # (a locus the reference would lose is in no batch)
    if not ks:
        return bad + ([] if not pieces else ["units"]), 0, 0
    hb = engine.genotype_packed(PackedUnits.from_scans([(h, ks)], maxinsert=o["maxinsert"], fullsearch=o["fullsearch"], clip=o["clip"]))
    (br, i0, dks), = pieces
    if dks != ks:
        return bad + ["units"], 0, 0
    lo, hi = int(br.batch.unit_read_off[i0]), int(br.batch.unit_read_off[i0 + len(ks)])
    if dks != ks or hi - lo != hb.batch.n_reads:
        return bad + ["units"], 0, 0
    if not (np.array_equal(br.tag[lo:hi], hb.tag) and np.array_equal(br.h[lo:hi], hb.h) and np.array_equal(br.score[lo:hi], hb.score)):
        bad.append("tags")
    for key in (k for k in br.calls.dtype.names if k != "pad"):
        x, y = br.calls[key][i0:i0 + len(ks)], hb.calls[key]
        if not (np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)):
            bad.append("calls." + key)
    m = min(br.marg.shape[2], hb.marg.shape[2])
    if not np.array_equal(br.marg[i0:i0 + len(ks), :, :m], hb.marg[:, :, :m]):
        bad.append("marginals")
    for j in range(len(ks)):
        (ta, tot_a), (tb, tot_b) = br.joint[i0 + j], hb.joint[j]
        if tot_a != tot_b or sorted(map(tuple, ta.tolist())) != sorted(map(tuple, tb.tolist())):
            bad.append("joint")
            break
    return bad, hi - lo, len(ks)


def device_scans(args, engine, batch):
    """The feeder with the selection on, a chunk at a time: [(arg, scan, pieces)] in task order."""
    from tredparse_amd import tred as t
    from tredparse_amd.feeder import _InflateFeeder
    chunks = [args[i:i + batch] for i in range(0, len(args), batch)]
    ex = ThreadPoolExecutor(max_workers=2)
    feeder = _InflateFeeder(chunks, ex, 0, walk=True, select=True)
    out = []
    try:
        for _ in chunks:
            chunk, futs = feeder.next()
            scans = [f.result() for f in futs]
            picks, parts = t.genotype_scans(engine, chunk, scans)
            out += [(a, s, parts.get(si, [])) for si, (a, s) in enumerate(zip(chunk, scans))]
    finally:
        feeder.close()
        ex.shutdown()
    return out


# optional fields as an aligner leaves them behind every record (every third round): the last one an array whose bytes read as a
# record's head (block_size 40, a contig id, a position) to whoever mistakes them for one
AUX = b"NMC\x02MDZ75A74\x00ASC\x91RGZgroup1\x00XSC\x13" + b"ZBBC" + (40).to_bytes(4, "little") + bytes([40, 0, 0, 0, 3, 0, 0, 0, 5, 0, 0, 0, 2, 0, 73, 18] + [0] * 24)


def main():
    from tredparse_amd import _lib, synth, synth_bam, tred as t
    from tredparse_amd.engine import Engine
    from tredparse_amd.meta import TREDsRepo
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    repo = TREDsRepo()
    all_loci = synth.load_loci()
    root = tempfile.mkdtemp(prefix="tred_fuzzsel_")
    engine = Engine(0)
    out = {"rounds": rounds, "samples": 0, "on_device": 0, "declined": 0, "units": 0, "reads": 0, "mismatching_samples": 0, "what": {},
           "block_sizes": [], "read_lengths": [], "x_linked_samples": 0, "samples_with_a_record_without_sequence": 0, "samples_with_a_mapped_record_without_cigar": 0, "samples_with_trimmed_reads": 0,
           "of_which_lost_a_locus": 0}
    for k in t.TIMING:
        t.TIMING[k] = 0
    t0 = time.time()
    for rnd in range(rounds):
        loci = [all_loci[i] for i in sorted(rng.choice(len(all_loci), size=int(rng.integers(2, 8)), replace=False))]
        names = [l["name"] for l in loci]
        args = []
        for k in range(int(rng.integers(2, 6))):
            cov = float(rng.choice([3, 10, 30, 60]))
            readlen = int(rng.choice([100, 150, 150, 250]))
            recs, _ = synth_bam.simulate_sample(int(rng.integers(1 << 30)), loci,
                                                synth.SynthParams(coverage=cov, readlen=readlen, expanded_max=120, expanded_frac=0.3))
            n = len(recs.flag)
            recs.flag[rng.random(n) < 0.03] |= 0x400                       # duplicates
            recs.flag[rng.random(n) < 0.01] |= 0x100                       # secondary
            recs.flag[rng.random(n) < 0.01] |= 0x200                       # QC fail
            recs.flag[rng.random(n) < 0.02] &= ~0x1                        # unpaired
            recs.flag[rng.random(n) < 0.05] ^= 0x10                        # strand flipped
            # CIGAR forms the simulator does not write: a leading hard clip (no query bases: query_alignment_start skips it), '='
            # in place of M (consumes the reference all the same)
            hc = np.nonzero((rng.random(n) < 0.03) & (recs.n_cig <= 2) & (recs.n_cig >= 1))[0]
            for i in hc.tolist():
                m = int(recs.n_cig[i])
                recs.cig[i, 1:m + 1] = recs.cig[i, :m].copy()
                recs.cig[i, 0] = (5 << 4) | 5
                recs.n_cig[i] = m + 1
            eq = np.nonzero(rng.random(n) < 0.03)[0]
            for i in eq.tolist():
                for j in range(int(recs.n_cig[i])):
                    if (int(recs.cig[i, j]) & 15) == 0:
                        recs.cig[i, j] = (int(recs.cig[i, j]) & ~15) | 7
                        break
            if rng.random() < 0.15 and n:                                 # one mapped record without a CIGAR ('*': pysam's
                i = int(rng.integers(n))                                   # reference_end is None, PEextractor dies on such a mate)
                if not (recs.flag[i] & 0x4):
                    recs.n_cig[i] = 0
                    recs.cig[i] = 0
                    out["samples_with_a_mapped_record_without_cigar"] += 1
            block = int(rng.choice([200, 333, 1000, 4096, 20000, 0xff00]))
            path = os.path.join(root, "r{}_{}.bam".format(rnd, k))
            no_seq = None
            if rng.random() < 0.15 and n:                                 # one record without its sequence (SEQ '*'), anywhere
                no_seq = np.zeros(n, bool)
                no_seq[int(rng.integers(n))] = True
                out["samples_with_a_record_without_sequence"] += 1
            lengths = None
            if rnd % 4 == 1 and n:                                        # reads trimmed before alignment: two in five shortened
                recs, lengths = synth_bam.trim_records(recs, np.where(rng.random(n) < 0.4, rng.integers(readlen // 3, readlen + 1, n), readlen))
                out["samples_with_trimmed_reads"] += 1
            synth_bam.write_bam(path, recs, sample="f{}_{}".format(rnd, k), block=block, split_records=True,
                                decoys=0.5 if rnd % 5 == 4 else 0.0, decoy_seed=rnd, no_seq=no_seq,
                                aux=AUX if rnd % 3 == 2 else b"", lengths=lengths)
            mode = int(rng.integers(0, 5))            # plain, plain, --noalts, --useclippedreads, --fullsearch --maxinsert 60
            args.append(("f{}_{}".format(rnd, k), path, repo, names, 60 if mode == 4 else 300, mode == 4, mode == 3, mode != 2, True, "ERROR"))
            out["block_sizes"].append(block)
            out["read_lengths"].append(readlen)
            out["x_linked_samples"] += int(any(repo[nm].is_xlinked for nm in names))
        for a, s, pieces in device_scans(args, engine, batch=int(rng.integers(1, 5))):
            out["samples"] += 1
            if getattr(s, "device", None) is None:
                out["declined"] += 1                     # (or taken back: a record without a sequence among the selected)
                out["of_which_lost_a_locus"] += int(bool(s.dropped))
            else:
                out["on_device"] += 1
            bad, reads, units = compare(a, s, pieces, engine)
            out["reads"] += reads
            out["units"] += units
            if bad:
                out["mismatching_samples"] += 1
                for b in bad:
                    out["what"][b] = out["what"].get(b, 0) + 1
        for a in args:
            os.remove(a[1]); os.remove(a[1] + ".bai")
    t.release_inflaters()
    out["walk_blocks_fetched"] = int(t.TIMING["walk_blocks_fetched"])
    out["seconds"] = round(time.time() - t0, 1)
    out["block_sizes"] = sorted(set(out["block_sizes"]))
    out["read_lengths"] = sorted(set(out["read_lengths"]))
    out["library"] = _lib.version()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
