"""GPU parity: HIP template-ladder SW + tagging vs the CPU oracle (bit-exact, integer work).

Every call goes through the C ABI (libtredgpu.so).  The oracle (oracle/sw_oracle.c) is the plain
restatement of ssw.c / bam_parser.py pinned against the compiled reference in test_oracle_sw.py.
"""
import numpy as np
import pytest

from oracle import pyoracle as po
from tredparse_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _run_gpu(ctx, ladders, reads, unit_read_off, unit_ladder, dump=True, clip=False):
    ctx.set_ladders(ladders)
    packed, woff, rlen = _lib.pack_reads(reads)
    n = len(reads)
    nt = max(2 * l[3] for l in ladders)
    tag = np.zeros(n, np.uint8)
    h = np.zeros(n, np.int16)
    score = np.zeros(n, np.int16)
    dumpa = np.zeros((n, nt, 6), np.int16) if dump else None
    uro = np.asarray(unit_read_off, np.int32)
    ul = np.asarray(unit_ladder, np.int32)
    ctx.sw_classify(_lib.MEM_HOST, packed, woff, rlen, n, uro, ul, len(ul), _lib.default_sw_params(clip=clip),
                    tag, h, score, dumpa, nt if dump else 0)
    return tag, h, score, dumpa


def _oracle(ladders, reads, unit_read_off, unit_ladder, clip=False):
    ls = po.LocusSet(ladders)
    read_locus = np.zeros(len(reads), np.int32)
    for g, lad in enumerate(unit_ladder):
        read_locus[unit_read_off[g]:unit_read_off[g + 1]] = lad
    cls = po.classify(reads, read_locus, ls, clip=clip, threads=8)
    return cls, ls, read_locus


@pytest.mark.parametrize("readlen,names", [(150, ["HD", "DM1", "SCA10", "ULD", "OPMD", "ALS"]),
                                            (100, ["HD", "DM2", "BPES"]), (250, ["SCA36", "FRDA"]),
                                            (36, ["HD"])])
def test_ladder_dump_matches_oracle(ctx, loci, readlen, names):
    rng = np.random.default_rng(7 + readlen)
    p = synth.SynthParams(coverage=12, readlen=readlen, sub=0.02, indel=0.004, nrate=0.01,
                          min_units=1, max_units=max(3, readlen // 3 + 8))
    sel = [l for l in loci if l["name"] in names]
    ladders, reads, uro, ul = [], [], [0], []
    for li, locus in enumerate(sel):
        lb = synth.simulate_locus(rng, locus, 3, p)
        ladders.append((locus["prefix"], locus["repeat"], locus["suffix"], -(-readlen // len(locus["repeat"]))))
        for g in range(3):
            rr = [synth.decode(r) for r in lb.reads[lb.unit_read_off[g]:lb.unit_read_off[g + 1]]]
            rr += ["N" * readlen, synth.decode(rng.integers(0, 4, readlen).astype(np.uint8))]
            reads += rr
            uro.append(len(reads))
            ul.append(li)
    tag, h, score, dump = _run_gpu(ctx, ladders, reads, uro, ul)
    cls, ls, read_locus = _oracle(ladders, reads, uro, ul)
    # field-by-field per template against the restated ssw_align
    n_bad = 0
    for r, read in enumerate(reads):
        g = read_locus[r]
        t0, t1 = ls.lad_off[g], ls.lad_off[g + 1]
        refs = ls.templates[t0:t1]
        want = po.sw_pairs([read], refs, [0] * len(refs), list(range(len(refs))))
        got = dump[r, :len(refs), :5].astype(np.int32)
        if not np.array_equal(got, want):
            n_bad += 1
            if n_bad < 5:
                k = np.nonzero((got != want).any(axis=1))[0][0]
                print("read", r, "template", k, "gpu", got[k], "oracle", want[k])
    assert n_bad == 0
    assert np.array_equal(tag, cls[:, 0].astype(np.uint8))
    assert np.array_equal(h, cls[:, 1].astype(np.int16))
    assert np.array_equal(score, cls[:, 2].astype(np.int16))
    assert (tag != 0).sum() > 0


def test_ragged_units_and_lengths(ctx, loci):
    """Empty units, units of 1..9 reads (partial quads), mixed read lengths in one launch."""
    rng = np.random.default_rng(3)
    hd = [l for l in loci if l["name"] == "HD"][0]
    ladders = [(hd["prefix"], hd["repeat"], hd["suffix"], 50)]
    p = synth.SynthParams(coverage=40, readlen=150)
    lb = synth.simulate_locus(rng, hd, 1, p, h_pairs=[[17, 42]])
    pool = [synth.decode(r) for r in lb.reads]
    reads, uro, ul = [], [0], []
    for n in (0, 1, 2, 3, 4, 5, 9, 0, 7):
        for k in range(n):
            s = pool[int(rng.integers(0, len(pool)))]
            cut = int(rng.integers(0, 60))
            reads.append(s[cut:] if rng.random() < 0.5 else s[:150 - cut])
        uro.append(len(reads))
        ul.append(0)
    tag, h, score, _ = _run_gpu(ctx, ladders, reads, uro, ul, dump=False)
    cls, _, _ = _oracle(ladders, reads, uro, ul)
    assert np.array_equal(tag, cls[:, 0].astype(np.uint8))
    assert np.array_equal(h, cls[:, 1].astype(np.int16))
    assert np.array_equal(score, cls[:, 2].astype(np.int16))
    # clip mode: REPT cut-off from the read's own length (bam_parser.py:154-155)
    tag2, h2, score2, _ = _run_gpu(ctx, ladders, reads, uro, ul, dump=False, clip=True)
    cls2, _, _ = _oracle(ladders, reads, uro, ul, clip=True)
    assert np.array_equal(tag2, cls2[:, 0].astype(np.uint8))
    assert np.array_equal(h2, cls2[:, 1].astype(np.int16))


def test_plain_reference_alignment(ctx):
    """max_units = 0 ladders: one arbitrary reference per unit (the Aligner.align use case)."""
    rng = np.random.default_rng(11)
    refs = [synth.decode(rng.integers(0, 4, n).astype(np.uint8)) for n in (40, 120, 300, 511)]
    reads, uro, ul = [], [0], []
    for i, ref in enumerate(refs):
        for k in range(5):
            a = int(rng.integers(0, max(1, len(ref) - 30)))
            frag = ref[a:a + 150]
            frag = frag[:len(frag) // 2] + "ACGTNACG" + frag[len(frag) // 2:]
            reads.append(frag[:150] if len(frag) > 30 else frag + "ACGT" * 10)
        uro.append(len(reads))
        ul.append(i)
    ladders = [(r, "A", "", 0) for r in refs]
    ctx.set_ladders(ladders)
    packed, woff, rlen = _lib.pack_reads(reads)
    n = len(reads)
    tag = np.zeros(n, np.uint8); h = np.zeros(n, np.int16); sc = np.zeros(n, np.int16)
    dump = np.zeros((n, 1, 6), np.int16)
    ctx.sw_classify(_lib.MEM_HOST, packed, woff, rlen, n, np.asarray(uro, np.int32), np.asarray(ul, np.int32),
                    len(ul), _lib.default_sw_params(), tag, h, sc, dump, 1)
    pr = list(range(n))
    pt = [u for u in range(len(refs)) for _ in range(5)]
    want = po.sw_pairs(reads, refs, pr, pt)
    assert np.array_equal(dump[:, 0, :5].astype(np.int32), want)


def test_aligner_mirror_matches_reference_goldens(ctx):
    """tredparse_amd.ssw.Aligner (the reference's ssw.Aligner signature, src/ssw_wrap.py:110-143,177-227) on
    pairs taken from the reference-generated golden file."""
    import os
    from tredparse_amd.ssw import Aligner
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sw_pairs.npz"))
    reads, refs = [str(x) for x in z["reads"]], [str(x) for x in z["refs"]]
    pr, pt, want = z["pair_read"], z["pair_ref"], z["result"].astype(np.int32)
    rng = np.random.default_rng(0)
    for k in rng.choice(len(pr), 60, replace=False):
        ref = refs[pt[k]]
        if len(ref) > 511:
            continue
        al = Aligner(ref_seq=ref, match=1, mismatch=5, gap_open=7, gap_extend=2, ctx=ctx)
        res = al.align(reads[pr[k]], min_score=0, min_len=0)
        got = (res.score, res.ref_begin, res.ref_end, res.query_begin, res.query_end)
        assert got == tuple(int(x) for x in want[k]), k
        # the min_score / min_len filter of Aligner.align (ssw_wrap.py:214-220)
        assert al.align(reads[pr[k]], min_score=int(want[k][0]) + 1, min_len=0) is None
    # without ctx=: all Aligners share one private context (not one GPU context each), the reference is registered
    # once per change, and align_many batches the reads of one reference into one launch
    by_ref = {}
    for k in rng.choice(len(pr), 400, replace=False):
        if len(refs[pt[k]]) <= 511:
            by_ref.setdefault(int(pt[k]), []).append(int(k))
    from tredparse_amd import ssw
    for t, ks in list(by_ref.items())[:12]:
        al = Aligner(ref_seq=refs[t], match=1, mismatch=5, gap_open=7, gap_extend=2)
        got = al.align_many([reads[pr[k]] for k in ks])
        for k, res in zip(ks, got):
            assert (res.score, res.ref_begin, res.ref_end, res.query_begin, res.query_end) == tuple(int(x) for x in want[k])
        assert al.align(reads[pr[ks[0]]]).score == int(want[ks[0]][0])
    assert ssw._shared["ctx"] is not None and ssw._shared["ctx"] is not ctx


def _checker():
    """The reference's own compiled ssw.c when oracle/_ref travelled with the checkout (it is git-ignored: a clean
    clone on a box without /root/reference has none), else the C restatement, which tests/test_oracle_sw.py pins to
    that same reference pair by pair."""
    return po.ref_classify if po.have_ref() else po.classify


@pytest.mark.parametrize("readlen", [36, 50, 64])
def test_short_reads_through_the_pruned_path(ctx, loci, readlen):
    """Reads of up to 64 bp use 4 rows per lane: a 6-mer window then spans three lanes, so the in-kernel 6-mer
    bound must stay off there (it once capped scores too low and dropped the last templates of the ladder:
    found by tools/fuzz_parity.py).  Non-dump path, every read against the reference's own ssw.c."""
    sel = [l for l in loci if l["name"] in ("HD", "SCA3", "DM1", "FRDA", "SCA10")]
    p = synth.SynthParams(coverage=40, readlen=readlen, min_units=1, max_units=30, sub=0.02, nrate=0.01)
    b = synth.build_batch(900 + readlen, sel, 3, p, workers=4)
    ctx.set_ladders(b.ladders)
    n = b.n_reads
    tag = np.zeros(n, np.uint8); h = np.zeros(n, np.int16); sc = np.zeros(n, np.int16)
    ctx.sw_classify(_lib.MEM_HOST, b.packed, b.read_off, b.read_len, n, b.unit_read_off, b.unit_ladder, b.n_units,
                    _lib.default_sw_params(max_read_len=readlen), tag, h, sc)
    reads = [synth.decode(r) for r in b.codes]
    cls = _checker()(reads, np.repeat(b.unit_ladder, np.diff(b.unit_read_off)), po.LocusSet(b.ladders), threads=0)
    bad = np.nonzero((tag != cls[:, 0]) | (h != cls[:, 1]) | (sc != cls[:, 2]))[0]
    assert len(bad) == 0, (len(bad), bad[:5], tag[bad[:5]], h[bad[:5]], cls[bad[:5]])
    assert (tag == 4).sum() > 20      # REPT reads: the case that exposed it


def test_quads_across_units_any_unit_order(ctx, loci):
    """Quads are packed by (ladder, class) across units: a read's result must not depend on which units surround
    it.  The same reads are classified with the units in locus order and interleaved at random (with empty units
    in between); per read the outputs must be identical, and equal to the reference's."""
    sel = [l for l in loci if l["name"] in ("HD", "DM1", "SCA10", "OPMD", "FRDA", "ULD", "BPES")]
    p = synth.SynthParams(coverage=25, readlen=150, min_units=3, max_units=55)
    b = synth.build_batch(4242, sel, 5, p, workers=4)
    reads = [synth.decode(r) for r in b.codes]
    uro = np.asarray(b.unit_read_off, np.int64)
    ctx.set_ladders(b.ladders)

    def run(order, empties):
        rr, off, lad, src = [], [0], [], []
        for k, g in enumerate(order):
            if k in empties:
                off.append(len(rr)); lad.append(int(b.unit_ladder[g]))
            rr += reads[uro[g]:uro[g + 1]]
            src += list(range(uro[g], uro[g + 1]))
            off.append(len(rr)); lad.append(int(b.unit_ladder[g]))
        packed, woff, rlen = _lib.pack_reads(rr)
        n = len(rr)
        tag = np.zeros(n, np.uint8); h = np.zeros(n, np.int16); sc = np.zeros(n, np.int16)
        ctx.sw_classify(_lib.MEM_HOST, packed, woff, rlen, n, np.asarray(off, np.int32), np.asarray(lad, np.int32),
                        len(lad), _lib.default_sw_params(max_read_len=150), tag, h, sc)
        out = np.zeros((len(reads), 3), np.int32)
        out[np.asarray(src)] = np.stack([tag, h, sc], 1)
        return out

    rng = np.random.default_rng(5)
    a = run(list(range(b.n_units)), set())
    c = run(list(rng.permutation(b.n_units)), set(rng.integers(0, b.n_units, 6).tolist()))
    assert np.array_equal(a, c)
    cls = _checker()(reads, np.repeat(b.unit_ladder, np.diff(uro)), po.LocusSet(b.ladders), threads=0)
    assert np.array_equal(a, cls[:, :3].astype(np.int32))
    assert len(set(a[:, 0])) >= 5      # all tags occur
