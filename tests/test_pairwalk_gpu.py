"""The pair walk on the device (csrc/walk.hip pair_walk_kernel, tredgpu_inflate_walk) against the host's file layer:
per locus the two pair-length lists in PEextractor's order (tredparse/bam_parser.py:316-369, restated by
bamread.cpp's PairTable and checked against the reference's own numbers in test_host_frontend.py), the virtual offsets
between which the locus' window lies against a walk through the package's pure-Python BAM layer, and the whole scan with
the device's results handed in (tredbam_scan_pe) against the plain scan -- reading only the blocks tredgpu_inflater_fetch
brought back."""
import os

import numpy as np
import pytest

from tredparse_amd import _lib, bamio, synth, synth_bam
from tredparse_amd.bam_parser import DNAPE_ELONGATE, FLANKMATCH, SPAN, _site_arrays, walk_need
from tredparse_amd.meta import TREDsRepo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def inf():
    f = _lib.Inflater(0)
    yield f
    f.close()


@pytest.fixture(scope="module")
def synthetic(tmp_path_factory):
    root = tmp_path_factory.mktemp("walk")
    loci = [l for l in synth.load_loci() if l["name"] in ("HD", "DM1", "SCA1", "FRDA", "AR", "FXS", "SCA17")]
    made = synth_bam.make_bams(str(root), 2, seed=91, loci=loci, p=synth.SynthParams(coverage=30, expanded_max=120, expanded_frac=0.3))
    out = [(path, TREDsRepo(), [l["name"] for l in loci]) for _, path, _ in made]
    # the same kind of sample in blocks cut without regard to records (every record straddles 300-byte blocks; samtools'
    # files look like the 20 000-byte one), some reads marked duplicate / unpaired / strand-flipped
    recs, _ = synth_bam.simulate_sample(92, loci[:4], synth.SynthParams(coverage=20, expanded_max=120, expanded_frac=0.3))
    rng = np.random.default_rng(92)
    recs.flag[rng.random(len(recs.flag)) < 0.03] |= 0x400
    recs.flag[rng.random(len(recs.flag)) < 0.02] &= ~0x1
    recs.flag[rng.random(len(recs.flag)) < 0.05] ^= 0x10
    for block in (300, 20000):
        path = os.path.join(str(root), "cut{}.bam".format(block))
        synth_bam.write_bam(path, recs, sample="cut", block=block, split_records=True)
        out.append((path, TREDsRepo(), [l["name"] for l in loci[:4]]))
    return out


def _window_span(path, chrom, p_lo, p_hi, w_lo, w_hi):
    """(n, vbeg, vend) of the records of [w_lo, w_hi) within the walk over [p_lo, p_hi), through the pure-Python layer
    (bamio.PyAlignmentFile.fetch's loop with the position taken before and after every record)."""
    f = bamio.PyAlignmentFile(path)
    if chrom not in f._tid:
        return 0, 0, 0
    tid = f._tid[chrom]
    f._load_index()
    bins, lin = f._index[tid]
    start, end = max(0, p_lo), p_hi
    min_off = lin[min(start >> 14, len(lin) - 1)] if lin else 0
    chunks = sorted((max(cb, min_off), ce) for b in bamio._reg2bins(start, max(end, start + 1)) for cb, ce in bins.get(b, ()) if ce > min_off)
    merged = []
    for cb, ce in chunks:
        if merged and cb <= merged[-1][1]:
            merged[-1] = (merged[-1][0], max(merged[-1][1], ce))
        else:
            merged.append((cb, ce))
    n = vbeg = vend = 0
    for cb, ce in merged:
        f.bg.seek(cb)
        while f.bg.tell() < ce:
            at = f.bg.tell()
            r = f._next()
            if r is None:
                break
            if r.tid != tid or r.pos >= end:
                if r.tid > tid or (r.tid == tid and r.pos >= end):
                    break
                continue
            rend = r.reference_end
            if rend is None or rend <= r.pos:
                rend = r.pos + 1
            if rend > start and r.pos < w_hi and rend > max(0, w_lo):
                if n == 0:
                    vbeg = at
                n += 1
                vend = f.bg.tell()
    f.close()
    return n, vbeg, vend


def _lay_out(inf, handles, plans):
    """The planned blocks of several files in one inflater, and the walk tables of the whole call."""
    n_all = sum(p[0] for p in plans)
    comp, out, coff, ooff = inf.reserve(sum(p[1] for p in plans), sum(p[2] for p in plans), n_all)
    at = cb = ob = 0
    firsts = []
    for f, (n, cbytes, obytes) in zip(handles, plans):
        f.plan_fill(inf.comp_addr, cb, ob, coff[at:at + n + 1], ooff[at:at + n + 1])
        firsts.append(at)
        at, cb, ob = at + n, cb + cbytes, ob + obytes
    return n_all, comp, coff, ooff, firsts


def _walk_inputs(handles, sites_of, readlens, firsts):
    blk = [f.plan_blocks() for f in handles]
    tasks, chunks, c0 = [], [], 0
    for f, sites, rl, b0, (coff, _, _, _) in zip(handles, sites_of, readlens, firsts, blk):
        t, c = f.plan_walks(sites, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
        t, c = t.copy(), c.copy()
        t["chunk_first"] += c0
        t["block_first"] += b0
        t["block_end"] += b0
        c["begin_block"][c["begin_block"] >= 0] += b0
        tasks.append(t)
        chunks.append(c)
        c0 += len(c)
    cat = lambda k: np.concatenate([b[k] for b in blk])      # noqa: E731
    return cat(0), cat(1), cat(2), [b[3] for b in blk], [b[0] for b in blk], np.concatenate(tasks), np.concatenate(chunks), [len(t) for t in tasks]


def _alt_inputs(handles, sites_of, regions_of, readlens, firsts):
    tasks, chunks, c0 = [], [], 0
    for f, sites, regions, rl, b0 in zip(handles, sites_of, regions_of, readlens, firsts):
        t, c = f.plan_alt_walks(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
        t, c = t.copy(), c.copy()
        t["chunk_first"] += c0
        t["block_first"] += b0
        t["block_end"] += b0
        c["begin_block"][c["begin_block"] >= 0] += b0
        tasks.append(t)
        chunks.append(c)
        c0 += len(c)
    return np.concatenate(tasks), np.concatenate(chunks), [len(t) for t in tasks]


def _zlib_blocks(handles, plans):
    """The planned blocks of the files, inflated by zlib, in the layout of the call."""
    import struct
    import zlib
    for f, (n, _, _) in zip(handles, plans):
        coff, clen, _, _ = f.plan_blocks()
        with open(f.path if hasattr(f, "path") else f.filename, "rb") as fp:
            for c, l in zip(coff, clen):
                fp.seek(int(c))
                raw = fp.read(int(l))
                xlen = struct.unpack_from("<H", raw, 10)[0]
                yield zlib.decompressobj(-15).decompress(raw[12 + xlen:-8])


def _cases(synthetic):
    repo = TREDsRepo(ref="hg38", sites=os.path.join(GOLD, "no_sites"))
    out = [(os.path.join(GOLD, "bam", s + ".bam"), repo, sorted(repo.names)) for s in ("t001", "t002")]
    return out + list(synthetic)


def test_walked_pair_lengths_offsets_and_scan(inf, synthetic):
    cases = _cases(synthetic)
    handles = [bamio.AlignmentFile(p) for p, _, _ in cases]
    sites_of, regions_of, readlens, plans = [], [], [], []
    for f, (path, repo, names) in zip(handles, cases):
        loci = [repo[n] for n in names]
        sites, regions = _site_arrays(repo, names, loci, f)
        rl = f.max_read_len(101)
        plans.append(f.plan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN))
        sites_of.append(sites); regions_of.append(regions); readlens.append(rl)
    n_all, comp, _, ooff, firsts = _lay_out(inf, handles, plans)
    bcoff, bclen, bcrc, host_of, coff_of, tasks, chunks, n_tasks = _walk_inputs(handles, sites_of, readlens, firsts)
    alt_tasks, alt_chunks, n_alt = _alt_inputs(handles, sites_of, regions_of, readlens, firsts)
    status, crc, res, gp, tp, ares, alt_need = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks, alt_tasks=alt_tasks, alt_chunks=alt_chunks)
    assert (status == 0).all() and (crc == bcrc).all()
    assert inf.walk_ms() > 0
    # the alternative loci: every region against the plain Python walk of the same task over zlib's bytes
    from .walk_model import _walk
    ooff_all = ooff[:n_all + 1]
    out_bytes = b"".join(_zlib_blocks(handles, plans))
    walkable = alt_tasks["n_chunks"] >= 0
    assert walkable.sum() > 100 and (ares["status"][walkable] == 0).all() and (ares["status"][~walkable] == 1).all()
    total_hits = 0
    for t in np.flatnonzero(walkable):
        hits = _walk(alt_tasks[t], alt_chunks, out_bytes, ooff_all, bcoff, bclen, lambda k: True, alt=True)
        assert [h[0] for h in hits] == [int(v) for v in ares["vbeg"][t][:ares["n"][t]]], t
        for _, kb, ka in hits:
            assert alt_need[kb:ka + 1].all()
        total_hits += len(hits)
    assert total_hits > 20 and alt_need.sum() <= 3 * total_hits
    t0 = a0 = 0
    total_pairs = total_window = 0
    for f, (path, repo, names), sites, regions, rl, first, host, coff, nt, na in zip(handles, cases, sites_of, regions_of, readlens, firsts,
                                                                                      host_of, coff_of, n_tasks, n_alt):
        r = res[t0:t0 + nt]
        t0 += nt
        ar = ares[a0:a0 + na]
        a0 += na
        plain_f = bamio.AlignmentFile(path)
        units, pools = plain_f.scan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
        for k, name in enumerate(names):
            t = repo[name]
            if sites["tid"][k] < 0:
                assert r["status"][k] == 1                  # not walkable: the scan's business
                continue
            assert r["status"][k] == 0, (path, name, r[k])
            g = gp[r["global_first"][k]:r["global_first"][k] + r["n_global"][k]]
            tt = tp[r["target_first"][k]:r["target_first"][k] + r["n_target"][k]]
            eg, et = plain_f.pe_lengths(t.chr, t.repeat_start - DNAPE_ELONGATE, t.repeat_end + DNAPE_ELONGATE,
                                        t.repeat_start - FLANKMATCH, t.repeat_end + FLANKMATCH, SPAN)
            assert list(g) == eg and list(tt) == et, (path, name)
            total_pairs += len(eg) + len(et)
            n, vbeg, vend = _window_span(path, t.chr, t.repeat_start - DNAPE_ELONGATE, t.repeat_end + DNAPE_ELONGATE,
                                         t.repeat_start - SPAN, t.repeat_end + SPAN)
            assert (int(r["n_window"][k]), int(r["win_vbeg"][k]), int(r["win_vend"][k])) == (n, vbeg, vend), (path, name)
            total_window += n
        # the scan with these results, over the fetched blocks only
        need = walk_need(coff, host, r, alt_need[first:first + len(coff)])
        assert need.sum() < len(need) or len(need) < 8
        assert need.sum() <= (np.asarray(host) != 0).sum() + 4 * len(names) + 8      # (the alternative loci's blocks stay behind)
        full = np.zeros(n_all, np.uint8)
        full[first:first + len(need)] = need
        inf.fetch(full)
        n_here = len(need)
        f.preload(inf.out_addr, ooff[first:first + n_here + 1], np.where(need != 0, status[first:first + n_here], 1).astype(np.int32),
                  crc[first:first + n_here])
        u2, p2 = f.scan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN, pe=(r, gp, tp), alt=ar)
        hits, misses = f.preload_clear()
        for key in units.dtype.names:
            assert (units[key] == u2[key]).all(), (path, key)
        for key in pools:
            assert (pools[key] == p2[key]) if isinstance(pools[key], bytes) else np.array_equal(pools[key], p2[key]), (path, key)
        assert misses == 0 and hits > 0, (path, hits, misses)     # every block the scan read had been fetched
        plain_f.close()
    assert total_pairs > 2000 and total_window > 500
    for f in handles:
        f.close()


def test_a_block_the_decoder_cannot_vouch_for_ends_the_task_not_the_sample(inf, synthetic):
    """A payload damaged in the staging buffer (the file is fine): the block's status or CRC is off, the walks that touch
    it end with status 2, every other region is walked, and the scan -- computing those sites itself -- gives the plain
    scan's result."""
    path, repo, names = synthetic[0]
    f = bamio.AlignmentFile(path)
    loci = [repo[n] for n in names]
    sites, regions = _site_arrays(repo, names, loci, f)
    rl = f.max_read_len(101)
    plan = f.plan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    n_all, comp, comp_off, ooff, firsts = _lay_out(inf, [f], [plan])
    bcoff, bclen, bcrc, host_of, coff_of, tasks, chunks, _ = _walk_inputs([f], [sites], [rl], firsts)
    # damage the middle of the payload of a block in the first task's first chunk
    victim = int(chunks["begin_block"][tasks["chunk_first"][0]]) + 1
    mid = int(comp_off[victim] + (comp_off[victim + 1] - comp_off[victim]) // 2)
    comp[mid:mid + 8] ^= 0x5A
    status, crc, res, gp, tp = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks)
    assert status[victim] != 0 or crc[victim] != bcrc[victim]
    assert res["status"][0] == 2 and (res["status"][1:] == 0).sum() >= len(res) - 2
    need = walk_need(coff_of[0], host_of[0], res)
    inf.fetch(need)
    ok_status = np.where((need != 0) & (crc == bcrc), status, 1).astype(np.int32)
    f.preload(inf.out_addr, ooff[:n_all + 1], ok_status, crc)
    u2, p2 = f.scan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN, pe=(res, gp, tp))
    f.preload_clear()
    g = bamio.AlignmentFile(path)
    units, pools = g.scan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    for key in units.dtype.names:
        assert (units[key] == u2[key]).all(), key
    for key in pools:
        assert (pools[key] == p2[key]) if isinstance(pools[key], bytes) else np.array_equal(pools[key], p2[key]), key
    f.close(); g.close()


def test_a_short_random_campaign(capsys, monkeypatch):
    """tools/fuzz_walk.py, three rounds: random loci / coverage / flags, blocks cut at random sizes without regard to
    records -- no region and no scan may differ (the long campaigns are in profiles/r04_fuzz_walk_*.json)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import fuzz_walk
    monkeypatch.setattr(sys, "argv", ["fuzz_walk.py", "3", "20261004"])
    fuzz_walk.main()
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["samples"] >= 3 and out["regions_walked"] > 0 and out["alt_regions"] > 0
    assert out["mismatching_regions"] == 0 and out["mismatching_scans"] == 0


def _odd_bam(path, rng, block, no_end=False, decoys=False, bad_record=False):
    """A BAM around the HD locus whose records have heads of every size: names of 1-250 characters, CIGARs of 1-220
    operations (soft / hard clips at the ends, M I D N = X inside), secondary / supplementary copies under the same name,
    duplicates, unmapped mates -- in blocks of `block` bytes cut without regard to records."""
    import struct
    t = TREDsRepo()["HD"]
    tid = synth_bam.CONTIGS.index(t.chr)
    recs = []
    for i in range(700):
        name = ("q%d_" % i + "x" * int(rng.integers(0, 245)))[:250]
        pos = int(t.repeat_start + rng.integers(-11000, 11000))
        n_mates = int(rng.choice([1, 2, 2, 2, 3]))
        if no_end and i == 350:
            pos, n_mates = t.repeat_start - 200, 2
        for mate in range(n_mates):
            n_ops = int(rng.choice([1, 2, 3, 5, 40, 130, 220]))
            ops = []
            if rng.random() < 0.3:
                ops.append((5, int(rng.integers(1, 30))))
            if rng.random() < 0.5:
                ops.append((4, int(rng.integers(1, 40))))
            for _ in range(n_ops):
                ops.append((int(rng.choice([0, 0, 0, 1, 2, 3, 7, 8])), int(rng.integers(1, 12))))
            if rng.random() < 0.5:
                ops.append((4, int(rng.integers(1, 40))))
            if rng.random() < 0.2:
                ops.append((5, int(rng.integers(1, 30))))
            l_seq = sum(n for op, n in ops if op in (0, 1, 4, 7, 8))
            flag = 0x1 | (0x10 if (mate == 1) != (rng.random() < 0.1) else 0) | (0x40 if mate == 0 else 0x80)
            if rng.random() < 0.04:
                flag |= 0x400
            if rng.random() < 0.03:
                flag |= 0x4
                ops = []
            if mate == 2:
                flag |= 0x800
            if no_end and i == 350 and mate == 1:
                flag, ops = (flag | 0x10) & ~0x404, []       # a mapped reverse mate without a CIGAR: no alignment end
            if no_end and i == 350 and mate == 0:
                flag &= ~0x414
            p = pos + (int(rng.integers(150, 700)) if mate else 0)
            ref = sum(n for op, n in ops if op in (0, 2, 3, 7, 8))
            mtid, mpos = (tid, int(t.repeat_start + rng.integers(-900, 900))) if rng.random() < 0.5 else (-1, -1)
            recs.append((tid, p, name, flag, ops, l_seq, ref, mtid, mpos))
    # mates rescued from two of the locus' alternative loci: three in the first, nine (more than a result holds) in the second
    for (contig, lo, hi), count in zip([a for a in t.alt if a[0] in synth_bam.CONTIGS][:2], (3, 9)):
        for k in range(count + 4):
            in_window = k < count
            recs.append((synth_bam.CONTIGS.index(contig), int(lo + rng.integers(0, hi - lo - 60)), "alt%d_%d" % (lo, k), 0x1 | 0x40,
                         [(0, 80)], 80, 80, tid if in_window else tid + 1, int(t.repeat_start + rng.integers(-800, 800))))
    recs.sort(key=lambda r: (r[0], r[1]))
    blob, offs, ends = b"", [], []
    for rtid, p, name, flag, ops, l_seq, ref, mtid, mpos in recs:
        end = p + ref if (ref and not flag & 4) else p + 1
        body = struct.pack("<iiBBHHHiiii", rtid, p, len(name) + 1, 0 if flag & 4 else 60, int(synth_bam._reg2bin(np.array([p]), np.array([end]))[0]),
                           len(ops), flag, l_seq, mtid, mpos, 0)
        body += name.encode() + b"\0" + b"".join(struct.pack("<I", n << 4 | op) for op, n in ops)
        qual = b"\xff" * l_seq
        if decoys and l_seq >= 48:
            # base qualities that read as the head of a record of the region's contig: a length word that covers the fixed
            # fields, the contig, a position, a two-byte name that ends in NUL, no CIGAR, no bases -- what follows them is not
            fake = struct.pack("<iiiBBHHHiiii", 40, rtid, 5, 2, 0, 0, 0, 0, 0, -1, -1, 0) + b"a\0"
            qual = b"\xff" * 4 + fake + b"\xff" * (l_seq - 4 - len(fake))
        body += bytes((l_seq + 1) // 2) + qual
        if bad_record and len(offs) == len(recs) // 2:
            # a record whose fixed fields promise more bases than the record holds (its length word is right: the chain
            # goes on behind it)
            body = body[:16] + struct.pack("<i", 1 << 20) + body[20:]
        offs.append(len(blob))
        blob += struct.pack("<i", len(body)) + body
        ends.append(end)
    offs.append(len(blob))
    text = "@HD\tVN:1.5\tSO:coordinate\n" + "".join("@SQ\tSN:{}\tLN:{}\n".format(c, synth_bam.CONTIG_LEN) for c in synth_bam.CONTIGS)
    header = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(synth_bam.CONTIGS))
    for c in synth_bam.CONTIGS:
        header += struct.pack("<i", len(c) + 1) + c.encode() + b"\x00" + struct.pack("<i", synth_bam.CONTIG_LEN)
    starts = list(range(0, len(blob), block))
    with open(path, "wb") as fp:
        fp.write(synth_bam._bgzf_block(header, 1))
        co = []
        for a in starts:
            co.append(fp.tell())
            fp.write(synth_bam._bgzf_block(blob[a:a + block], 1))
        co.append(fp.tell())
        fp.write(synth_bam._EOF_BLOCK)
    off = np.array(offs, np.int64)
    co = np.array(co, np.int64)
    which = np.where(off >= len(blob), len(starts), np.minimum(off // block, len(starts) - 1))
    v = (co[which] << 16) | np.where(which < len(starts), off - which * block, 0)
    pos = np.array([r[1] for r in recs], np.int64)
    end = np.array(ends, np.int64)
    synth_bam._write_bai(path + ".bai", np.array([r[0] for r in recs]), pos, end, synth_bam._reg2bin(pos, end), v[:-1], v[1:])
    return len(recs)


@pytest.mark.parametrize("block", [500, 6000, 0xff00])
def test_heads_longer_than_the_window_and_names_seen_three_times(inf, tmp_path, block):
    """Records whose name and CIGAR do not fit the part of the LDS window a record is promised (512 bytes), CIGARs of
    200 operations with clips at both ends, third records under a name, unmapped and duplicate reads: lists, offsets and the
    scan as the host computes them."""
    path = str(tmp_path / "odd.bam")
    assert _odd_bam(path, np.random.default_rng(block), block) > 1000
    repo, names = TREDsRepo(), ["HD"]
    f = bamio.AlignmentFile(path)
    sites, regions = _site_arrays(repo, names, [repo["HD"]], f)
    plan = f.plan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    n_all, comp, _, ooff, firsts = _lay_out(inf, [f], [plan])
    bcoff, bclen, bcrc, host_of, coff_of, tasks, chunks, _ = _walk_inputs([f], [sites], [150], firsts)
    alt_tasks, alt_chunks, _ = _alt_inputs([f], [sites], [regions], [150], firsts)
    status, crc, res, gp, tp, ares, alt_need = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks, alt_tasks=alt_tasks, alt_chunks=alt_chunks)
    assert (status == 0).all() and res["status"][0] == 0, res
    # the two alternative loci that hold rescued mates: three records found in the first, the second has more than a
    # result holds and is left to the scan (status 6); every other region is empty
    walkable = alt_tasks["n_chunks"] >= 0
    assert sorted(ares["status"][walkable].tolist())[-2:] == [0, 6] and (ares["status"][walkable] != 0).sum() == 1
    assert sorted(ares["n"][ares["status"] == 0].tolist())[-2:] == [0, 3]
    t = repo["HD"]
    g = bamio.AlignmentFile(path)
    eg, et = g.pe_lengths(t.chr, t.repeat_start - DNAPE_ELONGATE, t.repeat_end + DNAPE_ELONGATE, t.repeat_start - FLANKMATCH,
                          t.repeat_end + FLANKMATCH, SPAN)
    assert list(gp[res["global_first"][0]:][:res["n_global"][0]]) == eg and list(tp[res["target_first"][0]:][:res["n_target"][0]]) == et
    assert len(eg) + len(et) > 20
    n, vbeg, vend = _window_span(path, t.chr, t.repeat_start - DNAPE_ELONGATE, t.repeat_end + DNAPE_ELONGATE, t.repeat_start - SPAN, t.repeat_end + SPAN)
    assert (int(res["n_window"][0]), int(res["win_vbeg"][0]), int(res["win_vend"][0])) == (n, vbeg, vend) and n > 30
    need = walk_need(coff_of[0], host_of[0], res, alt_need)
    inf.fetch(need)
    f.preload(inf.out_addr, ooff[:n_all + 1], np.where(need != 0, status, 1).astype(np.int32), crc)
    u2, p2 = f.scan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN, pe=(res, gp, tp), alt=ares)
    f.preload_clear()
    units, pools = g.scan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    for key in units.dtype.names:
        assert (units[key] == u2[key]).all(), key
    for key in pools:
        assert (pools[key] == p2[key]) if isinstance(pools[key], bytes) else np.array_equal(pools[key], p2[key]), key
    f.close(); g.close()


def test_record_starts_guessed_wrong_are_found_out(inf, tmp_path):
    """walk_chain_par_kernel guesses where records start from what the bytes look like, and proves every guess by the
    chain that arrives there.  Here most reads carry base qualities that look like a record's head: lanes that start
    inside such a read guess wrong, their neighbours step over the guess, and the region goes to the serial chain --
    lists, offsets and scan as the host computes them, and the library says that the serial chain was used."""
    path = str(tmp_path / "decoy.bam")
    assert _odd_bam(path, np.random.default_rng(11), 6000, decoys=True) > 1000
    repo, names = TREDsRepo(), ["HD"]
    f = bamio.AlignmentFile(path)
    sites, regions = _site_arrays(repo, names, [repo["HD"]], f)
    plan = f.plan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    n_all, comp, _, ooff, firsts = _lay_out(inf, [f], [plan])
    bcoff, bclen, bcrc, host_of, coff_of, tasks, chunks, _ = _walk_inputs([f], [sites], [150], firsts)
    alt_tasks, alt_chunks, _ = _alt_inputs([f], [sites], [regions], [150], firsts)
    status, crc, res, gp, tp, ares, alt_need = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks, alt_tasks=alt_tasks, alt_chunks=alt_chunks)
    assert (status == 0).all() and res["status"][0] == 0, res
    assert inf.walk_serial_regions() == 1                   # (the one region of the call)
    t = repo["HD"]
    g = bamio.AlignmentFile(path)
    eg, et = g.pe_lengths(t.chr, t.repeat_start - DNAPE_ELONGATE, t.repeat_end + DNAPE_ELONGATE, t.repeat_start - FLANKMATCH,
                          t.repeat_end + FLANKMATCH, SPAN)
    assert list(gp[res["global_first"][0]:][:res["n_global"][0]]) == eg and list(tp[res["target_first"][0]:][:res["n_target"][0]]) == et
    n, vbeg, vend = _window_span(path, t.chr, t.repeat_start - DNAPE_ELONGATE, t.repeat_end + DNAPE_ELONGATE, t.repeat_start - SPAN, t.repeat_end + SPAN)
    assert (int(res["n_window"][0]), int(res["win_vbeg"][0]), int(res["win_vend"][0])) == (n, vbeg, vend) and n > 30
    need = walk_need(coff_of[0], host_of[0], res, alt_need)
    inf.fetch(need)
    f.preload(inf.out_addr, ooff[:n_all + 1], np.where(need != 0, status, 1).astype(np.int32), crc)
    u2, p2 = f.scan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN, pe=(res, gp, tp), alt=ares)
    f.preload_clear()
    units, pools = g.scan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    for key in units.dtype.names:
        assert (units[key] == u2[key]).all(), key
    for key in pools:
        assert (pools[key] == p2[key]) if isinstance(pools[key], bytes) else np.array_equal(pools[key], p2[key]), key
    f.close(); g.close()


def test_the_serial_chain_and_the_lanes_list_the_same_records(inf, synthetic, monkeypatch):
    """TREDGPU_WALK_SERIAL=1 sends every region through walk_chain_kernel (one record after the other, the way every
    region went before the lanes shared the chain): the results of a call are the same numbers either way, and without
    the switch no region of these files needs the serial chain."""
    cases = _cases(synthetic)
    handles = [bamio.AlignmentFile(p) for p, _, _ in cases]
    sites_of, regions_of, readlens, plans = [], [], [], []
    for f, (path, repo, names) in zip(handles, cases):
        sites, regions = _site_arrays(repo, names, [repo[n] for n in names], f)
        rl = f.max_read_len(101)
        plans.append(f.plan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN))
        sites_of.append(sites); regions_of.append(regions); readlens.append(rl)
    n_all, comp, _, ooff, firsts = _lay_out(inf, handles, plans)
    bcoff, bclen, bcrc, host_of, coff_of, tasks, chunks, _ = _walk_inputs(handles, sites_of, readlens, firsts)
    alt_tasks, alt_chunks, _ = _alt_inputs(handles, sites_of, regions_of, readlens, firsts)
    a = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks, alt_tasks=alt_tasks, alt_chunks=alt_chunks)
    assert inf.walk_serial_regions() <= int((a[2]["status"] != 0).sum())     # (only what the lanes hand back: regions the plan does not hold)
    monkeypatch.setenv("TREDGPU_WALK_SERIAL", "1")
    b = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks, alt_tasks=alt_tasks, alt_chunks=alt_chunks)
    assert inf.walk_serial_regions() == int((tasks["n_chunks"] >= 0).sum())
    res_a, res_b = a[2], b[2]
    walked = res_a["status"] == 0
    assert walked.sum() > 20 and np.array_equal(res_a["status"], res_b["status"])
    for key in ("n_global", "n_target", "n_window", "win_vbeg", "win_vend"):
        assert np.array_equal(res_a[key][walked], res_b[key][walked]), key
    for k in np.flatnonzero(walked):                        # (the pools are filled in the order the regions finish)
        for pool, first, count in ((3, "global_first", "n_global"), (4, "target_first", "n_target")):
            assert np.array_equal(a[pool][res_a[first][k]:][:res_a[count][k]], b[pool][res_b[first][k]:][:res_b[count][k]])
    assert np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6])     # the alternative loci's results and wanted blocks
    for f in handles:
        f.close()


def test_a_fetched_block_that_is_not_what_the_decoder_wrote_is_found_out(inf, synthetic):
    """The decoder's CRC vouches for the bytes on the device; the scan reads their copy in host memory.  One block in
    sixteen (by its place in the file) is checked again on the host at first use: damage exactly those in the fetched
    copy -- the scan drops them, inflates them itself and gives the plain scan's numbers."""
    import ctypes
    path, repo, names = synthetic[0]
    f, g = bamio.AlignmentFile(path), bamio.AlignmentFile(path)
    sites, regions = _site_arrays(repo, names, [repo[n] for n in names], f)
    rl = f.max_read_len(101)
    plan = f.plan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    n_all, comp, _, ooff, firsts = _lay_out(inf, [f], [plan])
    bcoff, bclen, bcrc, host_of, coff_of, tasks, chunks, _ = _walk_inputs([f], [sites], [rl], firsts)
    status, crc, res, gp, tp = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks)
    need = walk_need(coff_of[0], host_of[0], res)
    inf.fetch(need)
    sampled = ((((np.asarray(coff_of[0]).astype(np.uint64) >> np.uint64(4)) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(60)) == 0) & (need != 0)
    assert sampled.sum() >= 1, "no block of the sample falls into the checked sixteenth: pick another seed"
    host = np.ctypeslib.as_array((ctypes.c_uint8 * int(ooff[n_all])).from_address(inf.out_addr))
    for k in np.flatnonzero(sampled):
        host[ooff[k] + 40] ^= 0x55
    f.preload(inf.out_addr, ooff[:n_all + 1], np.where(need != 0, status, 1).astype(np.int32), crc)
    u2, p2 = f.scan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN, pe=(res, gp, tp))
    hits, misses = f.preload_clear()
    units, pools = g.scan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    assert misses >= 1 and hits > 0
    for key in units.dtype.names:
        assert (units[key] == u2[key]).all(), key
    for key in pools:
        assert (pools[key] == p2[key]) if isinstance(pools[key], bytes) else np.array_equal(pools[key], p2[key]), key
    f.close(); g.close()


def test_regions_at_100x_use_the_large_name_table(inf, tmp_path):
    """A +-10 kb region at 100x holds ~13 000 records in ~65 blocks and ~6 500 names: the launch takes the 8 192-name table
    (128 KB of LDS per workgroup) -- lists and window as the host computes them, nothing declined."""
    loci = [l for l in synth.load_loci() if l["name"] in ("HD", "DM1")]
    made = synth_bam.make_bams(str(tmp_path), 1, seed=17, loci=loci, p=synth.SynthParams(coverage=100, expanded_max=120, expanded_frac=0.3))
    path, repo, names = made[0][1], TREDsRepo(), [l["name"] for l in loci]
    f, g = bamio.AlignmentFile(path), bamio.AlignmentFile(path)
    sites, regions = _site_arrays(repo, names, [repo[n] for n in names], f)
    plan = f.plan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    n_all, comp, _, ooff, firsts = _lay_out(inf, [f], [plan])
    bcoff, bclen, bcrc, host_of, coff_of, tasks, chunks, _ = _walk_inputs([f], [sites], [150], firsts)
    status, crc, res, gp, tp = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks, pool_pairs=_lib.walk_pool_pairs(tasks, ooff))
    assert (status == 0).all() and (res["status"] == 0).all(), res["status"]
    assert inf.walk_serial_regions() == 0
    for k, name in enumerate(names):
        t = repo[name]
        eg, et = g.pe_lengths(t.chr, t.repeat_start - DNAPE_ELONGATE, t.repeat_end + DNAPE_ELONGATE, t.repeat_start - FLANKMATCH,
                              t.repeat_end + FLANKMATCH, SPAN)
        assert list(gp[res["global_first"][k]:][:res["n_global"][k]]) == eg and list(tp[res["target_first"][k]:][:res["n_target"][k]]) == et
        assert len(eg) + len(et) > 5000
        n, vbeg, vend = _window_span(path, t.chr, t.repeat_start - DNAPE_ELONGATE, t.repeat_end + DNAPE_ELONGATE, t.repeat_start - SPAN, t.repeat_end + SPAN)
        assert (int(res["n_window"][k]), int(res["win_vbeg"][k]), int(res["win_vend"][k])) == (n, vbeg, vend)
    f.close(); g.close()


def test_a_pair_without_alignment_end_is_the_hosts_to_report(inf, tmp_path):
    """The reference dies in get_target_length when the second read of a +/- pair has no alignment end (`None - int`); the
    kernel ends that region with status 5, the scan walks it itself and reports what the plain scan reports (pe_status
    -9: the locus is dropped)."""
    path = str(tmp_path / "noend.bam")
    _odd_bam(path, np.random.default_rng(7), 6000, no_end=True)
    repo, names = TREDsRepo(), ["HD"]
    f = bamio.AlignmentFile(path)
    sites, regions = _site_arrays(repo, names, [repo["HD"]], f)
    plan = f.plan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    n_all, comp, _, ooff, firsts = _lay_out(inf, [f], [plan])
    bcoff, bclen, bcrc, host_of, coff_of, tasks, chunks, _ = _walk_inputs([f], [sites], [150], firsts)
    status, crc, res, gp, tp = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks)
    assert res["status"][0] == 5
    need = walk_need(coff_of[0], host_of[0], res)
    inf.fetch(need)
    f.preload(inf.out_addr, ooff[:n_all + 1], np.where(need != 0, status, 1).astype(np.int32), crc)
    u2, p2 = f.scan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN, pe=(res, gp, tp))
    f.preload_clear()
    g = bamio.AlignmentFile(path)
    units, pools = g.scan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    assert units["pe_status"][0] == -9 and u2["pe_status"][0] == -9
    for key in units.dtype.names:
        assert (units[key] == u2[key]).all(), key
    for key in pools:
        assert (pools[key] == p2[key]) if isinstance(pools[key], bytes) else np.array_equal(pools[key], p2[key]), key
    f.close(); g.close()


def test_a_record_that_makes_no_sense_is_the_hosts_to_report(inf, tmp_path):
    """One record in the middle of the region says it has a million bases in 300 bytes: the kernel ends that region with
    status 3, the scan walks it itself and says what the plain scan says about the file."""
    path = str(tmp_path / "bad.bam")
    _odd_bam(path, np.random.default_rng(5), 6000, bad_record=True)
    repo, names = TREDsRepo(), ["HD"]
    f = bamio.AlignmentFile(path)
    sites, regions = _site_arrays(repo, names, [repo["HD"]], f)
    plan = f.plan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    n_all, comp, _, ooff, firsts = _lay_out(inf, [f], [plan])
    bcoff, bclen, bcrc, host_of, coff_of, tasks, chunks, _ = _walk_inputs([f], [sites], [150], firsts)
    status, crc, res, gp, tp = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks)
    assert (status == 0).all() and res["status"][0] == 3, res
    need = walk_need(coff_of[0], host_of[0], res)
    inf.fetch(need)
    f.preload(inf.out_addr, ooff[:n_all + 1], np.where(need != 0, status, 1).astype(np.int32), crc)
    g = bamio.AlignmentFile(path)
    try:
        want = g.scan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
    except Exception as e:
        want = e
    try:
        got = f.scan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN, pe=(res, gp, tp))
    except Exception as e:
        got = e
    f.preload_clear()
    if isinstance(want, Exception):
        assert type(got) is type(want) and str(got) == str(want)
    else:
        (units, pools), (u2, p2) = want, got
        for key in units.dtype.names:
            assert (units[key] == u2[key]).all(), key
        for key in pools:
            assert (pools[key] == p2[key]) if isinstance(pools[key], bytes) else np.array_equal(pools[key], p2[key]), key
    f.close(); g.close()


def test_dense_fetch_hands_over_the_same_blocks(synthetic):
    """An inflater without a host copy of the whole output (tredgpu_inflater_host_out(0): 45 MB less pinned memory per
    sample): tredgpu_inflater_fetch_dense packs the wanted blocks (and short gaps) one after the other -- every block it
    says it copied is zlib's byte for byte, every wanted block is among them, and a scan preloaded from the dense buffer
    equals the plain scan.  The calls that need the host copy refuse."""
    import ctypes
    inf = _lib.Inflater(0, host_out=False)
    cases = _cases(synthetic)
    handles = [bamio.AlignmentFile(p) for p, _, _ in cases]
    sites_of, regions_of, readlens, plans = [], [], [], []
    for f, (path, repo, names) in zip(handles, cases):
        loci = [repo[n] for n in names]
        sites, regions = _site_arrays(repo, names, loci, f)
        rl = f.max_read_len(101)
        plans.append(f.plan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN))
        sites_of.append(sites); regions_of.append(regions); readlens.append(rl)
    n_all, comp, _, ooff, firsts = _lay_out(inf, handles, plans)
    assert inf.out_addr == 0
    bcoff, bclen, bcrc, host_of, coff_of, tasks, chunks, n_tasks = _walk_inputs(handles, sites_of, readlens, firsts)
    alt_tasks, alt_chunks, n_alt = _alt_inputs(handles, sites_of, regions_of, readlens, firsts)
    status, crc, res, gp, tp, ares, alt_need = inf.run_walk(n_all, bcoff, bclen, bcrc, tasks, chunks, alt_tasks=alt_tasks, alt_chunks=alt_chunks,
                                                            pool_pairs=_lib.walk_pool_pairs(tasks, ooff))
    assert (status == 0).all() and (crc == bcrc).all() and (res["status"][tasks["n_chunks"] >= 0] == 0).all()
    with pytest.raises(_lib.TredGpuError):
        inf.fetch(np.ones(n_all, np.uint8))
    with pytest.raises(_lib.TredGpuError):
        inf.run(n_all)
    need = np.zeros(n_all, np.uint8)
    t0 = 0
    for first, host, coff, nt in zip(firsts, host_of, coff_of, n_tasks):
        need[first:first + len(coff)] = walk_need(coff, host, res[t0:t0 + nt], alt_need[first:first + len(coff)])
        t0 += nt
    rng = np.random.default_rng(5)
    need[rng.integers(0, n_all, 40)] = 1                              # (and some the walk did not ask for)
    addr, off = inf.fetch_dense(need)
    size = np.diff(off)
    want = np.diff(ooff[:n_all + 1])
    # a block is there in full or not at all: what was not copied is EMPTY (dense_off[k + 1] == dense_off[k]; the runs lie
    # one behind the other without padding -- ADVICE r5: a run's alignment used to make the block before it 1..15 bytes long)
    assert ((size == want) | (size == 0)).all() and (size[need != 0] == want[need != 0]).all()
    gaps = np.nonzero((need == 0) & (size == 0))[0]
    assert len(gaps) > 0 and (off[gaps + 1] == off[gaps]).all()
    assert off[-1] < 0.6 * ooff[n_all]                                      # (most bytes stay behind)
    dense = np.ctypeslib.as_array(ctypes.cast(addr, ctypes.POINTER(ctypes.c_uint8)), shape=(int(off[-1]),))
    for k, data in enumerate(_zlib_blocks(handles, plans)):
        if size[k] == want[k] and want[k]:
            assert dense[off[k]:off[k + 1]].tobytes() == data, k
    t0 = a0 = 0
    for f, (path, repo, names), sites, regions, rl, first, nt, na, coff in zip(handles, cases, sites_of, regions_of, readlens, firsts, n_tasks,
                                                                              n_alt, coff_of):
        r, ar = res[t0:t0 + nt], ares[a0:a0 + na]
        t0, a0 = t0 + nt, a0 + na
        n_here = len(coff)
        f.preload(addr, off[first:first + n_here + 1], np.where(need[first:first + n_here] != 0, status[first:first + n_here], 1).astype(np.int32),
                  crc[first:first + n_here])
        u2, p2 = f.scan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN, pe=(r, gp, tp), alt=ar)
        hits, misses = f.preload_clear()
        plain_f = bamio.AlignmentFile(path)
        units, pools = plain_f.scan(sites, regions, rl, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN)
        plain_f.close()
        for key in units.dtype.names:
            assert (units[key] == u2[key]).all(), (path, key)
        for key in pools:
            assert (pools[key] == p2[key]) if isinstance(pools[key], bytes) else np.array_equal(pools[key], p2[key]), (path, key)
        assert misses == 0 and hits > 0, (path, hits, misses)
        f.close()
    inf.close()
