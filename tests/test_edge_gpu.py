"""GPU: edge cases of the C ABI -- empty and ragged inputs, degenerate reads, maximum sizes, other scorings,
--norepeatpairs, error reporting (SURVEY section 5 failure semantics)."""
import ctypes as C

import numpy as np
import pytest

from oracle import lik_oracle as lo
from oracle import pyoracle as po
from tredparse_amd import _lib, synth

pytestmark = pytest.mark.gpu
HD = ("GAGTCCCTCAAGTCCTTC", "CAG", "CAACAGCCGCCACCGCCG", 50)


def _classify(ctx, ladders, reads, uro, ul, params=None, dump=False):
    ctx.set_ladders(ladders)
    packed, woff, rlen = _lib.pack_reads(reads)
    n = len(reads)
    tag = np.zeros(max(n, 1), np.uint8); h = np.zeros(max(n, 1), np.int16); sc = np.zeros(max(n, 1), np.int16)
    nt = max(2 * l[3] for l in ladders) if dump else 0
    d = np.zeros((max(n, 1), max(nt, 1), 6), np.int16) if dump else None
    ctx.sw_classify(_lib.MEM_HOST, packed, woff, rlen, n, np.asarray(uro, np.int32), np.asarray(ul, np.int32), len(ul),
                    params or _lib.default_sw_params(), tag, h, sc, d, nt)
    return tag[:n], h[:n], sc[:n], d


def test_empty_and_zero_read_units(ctx):
    step, w = lo.load_model()
    ctx.set_model(np.array([step[p] for p in range(1, 7)]), np.array(w))
    tag, h, sc, _ = _classify(ctx, [HD], [], [0], [])
    assert len(tag) == 0
    # three units, the middle one without reads; fused path gives "no evidence" for it
    lb = synth.simulate_locus(np.random.default_rng(2), [l for l in synth.load_loci() if l["name"] == "HD"][0], 2,
                              synth.SynthParams(coverage=20), h_pairs=[[15, 41], [17, 19]])
    reads = [synth.decode(r) for r in lb.reads]
    n0 = int(lb.unit_read_off[1])
    uro = np.array([0, n0, n0, len(reads)], np.int32)
    packed, woff, rlen = _lib.pack_reads(reads)
    units = np.zeros(3, _lib.UNIT_DTYPE)
    locus = [l for l in synth.load_loci() if l["name"] == "HD"][0]
    for i in range(3):
        units[i] = synth.unit_params_for(locus, 150, 30.0, 0, 0, 0, 0)
    n, hs = len(reads), 52
    tag = np.zeros(n, np.uint8); h = np.zeros(n, np.int16); sc = np.zeros(n, np.int16)
    full = np.zeros((3, hs), np.int32); pref = np.zeros((3, hs), np.int32); rept = np.zeros((3, hs), np.int32)
    calls = np.zeros(3, _lib.CALL_DTYPE)
    ctx.set_ladders([HD])
    ctx.genotype_batch(_lib.MEM_HOST, packed, woff, rlen, n, uro, np.zeros(3, np.int32), units, 3,
                       _lib.default_sw_params(), None, np.zeros(1, np.int32), 0, np.zeros(1, np.int32), 0, tag, h, sc, hs,
                       full, pref, rept, calls)
    assert calls["status"].tolist() == [0, 1, 0]
    assert (calls[0]["h1"] // 3, calls[2]["h1"] // 3) == (15, 17)
    assert full[1].sum() == 0 and pref[1].sum() == 0


def test_degenerate_reads_and_length_limits(ctx):
    rng = np.random.default_rng(8)
    tmpl = HD[0] + "CAG" * 30 + HD[2]
    reads = ["A", "ACGTA", "N" * 150, "", tmpl[:160], tmpl[5:155], "CAG" * 50, "acgt" * 30 + "nnn", tmpl[:64], tmpl[3:115]]
    tag, h, sc, dump = _classify(ctx, [HD], reads, [0, len(reads)], [0], dump=True)
    ls = po.LocusSet([HD])
    cls = po.classify(reads, np.zeros(len(reads), np.int32), ls)
    assert np.array_equal(tag, cls[:, 0].astype(np.uint8)) and np.array_equal(h, cls[:, 1].astype(np.int16))
    for r, read in enumerate(reads):
        want = po.sw_pairs([read], ls.templates, [0] * 100, list(range(100)))
        assert np.array_equal(dump[r, :, :5].astype(np.int32), want), r
    # a read longer than the declared bound is flagged, never truncated
    p = _lib.default_sw_params(max_read_len=150)
    long_reads = [tmpl[:150], (tmpl * 2)[:200]]
    t2, _, _, _ = _classify(ctx, [HD], long_reads, [0, 2], [0], params=p)
    assert t2[1] == _lib.TAG_INVALID and t2[0] != _lib.TAG_INVALID
    # 250 bp reads take the 16-rows-per-lane instantiation
    mu = -(-250 // 3)
    lad = (HD[0], "CAG", HD[2], mu)
    r250 = [(HD[0] + "CAG" * 70 + HD[2] + "ACGT" * 30)[:250], po.rc((("TTGCA" * 10) + HD[0] + "CAG" * 40 + HD[2] + "GATTACA" * 20)[:250])]
    t3, h3, s3, _ = _classify(ctx, [lad], r250, [0, 2], [0])
    c3 = po.classify(r250, np.zeros(2, np.int32), po.LocusSet([lad]))
    assert np.array_equal(t3, c3[:, 0]) and np.array_equal(h3, c3[:, 1]) and np.array_equal(s3, c3[:, 2])


@pytest.mark.parametrize("scoring", [(2, 2, 3, 1), (1, 4, 6, 1), (1, 5, 7, 2)])
def test_other_scorings_dump(ctx, scoring):
    rng = np.random.default_rng(sum(scoring))
    lad = ("GGCAGCCGCGGGCGGCGG", "CTG", "GGGCTTCAGCGACATGGT", 34)
    hap = "".join("ACGT"[i] for i in rng.integers(0, 4, 200)) + lad[0] + "CTG" * 22 + lad[2] + "".join("ACGT"[i] for i in rng.integers(0, 4, 200))
    reads = [hap[s:s + 100] for s in range(120, 300, 9)]
    reads = [r if i % 3 else po.rc(r) for i, r in enumerate(reads)]
    p = _lib.SwParams(scoring[0], scoring[1], scoring[2], scoring[3], 9, 0, 0, 0)
    tag, h, sc, dump = _classify(ctx, [lad], reads, [0, len(reads)], [0], params=p, dump=True)
    ls = po.LocusSet([lad])
    for r, read in enumerate(reads):
        want = po.sw_pairs([read], ls.templates, [0] * 68, list(range(68)), scoring=scoring)
        assert np.array_equal(dump[r, :, :5].astype(np.int32), want), (scoring, r)
    cls = po.classify(reads, np.zeros(len(reads), np.int32), ls, scoring=scoring)
    t2, h2, s2, _ = _classify(ctx, [lad], reads, [0, len(reads)], [0], params=p)     # pruned path
    assert np.array_equal(t2, cls[:, 0]) and np.array_equal(h2, cls[:, 1]) and np.array_equal(s2, cls[:, 2])


def test_norepeatpairs_removes_rept_mates(ctx):
    """remove_pairs_of_rept (bam_parser.py:270-287) on the device histograms."""
    tag = np.array([4, 4, 4, 1, 4, 2, 4, 4], np.uint8)      # REPT REPT REPT FULL REPT PREF | REPT REPT
    h = np.array([50, 49, 50, 20, 50, 12, 50, 50], np.int16)
    pair = np.array([7, 7, 8, 8, 9, 9, 3, 4], np.int32)      # reads 0,1 are mates and both REPT -> dropped
    uro = np.array([0, 6, 8], np.int32)
    hs = 52
    full = np.zeros((2, hs), np.int32); pref = np.zeros((2, hs), np.int32); rept = np.zeros((2, hs), np.int32)
    ctx.tally(_lib.MEM_HOST, tag, h, 8, uro, 2, pair, hs, full, pref, rept)
    assert rept[0].sum() == 2 and rept[0, 50] == 2 and full[0, 20] == 1 and pref[0, 12] == 1
    assert rept[1].sum() == 2
    ctx.tally(_lib.MEM_HOST, tag, h, 8, uro, 2, None, hs, full, pref, rept)
    assert rept[0].sum() == 4


class _FakeScan(object):
    """The handful of SampleScan members bam_parser.tally reads, over plain arrays."""

    def __init__(self, name_id):
        self.name_id = np.asarray(name_id, np.int32)

    def reads_of(self, k):
        return 0, len(self.name_id)

    def name(self, i):
        return "q{}".format(self.name_id[i])

    def sequence(self, i):
        return "ACGT"


@pytest.mark.parametrize("n_reads", [3, 700, 2500])     # 2500: beyond the LDS table, the pairwise fallback
def test_norepeatpairs_device_histograms_equal_host_counts(ctx, n_reads):
    """--norepeatpairs: the device histograms (what the grid genotypes from) and the host's counts (what the JSON
    prints as FR / PR / RR / FDP ...) agree, also when a name carries three records -- two REPT and a FULL one
    (supplementary alignment, ALT refetch): the reference removes every record of that name."""
    from tredparse_amd.bam_parser import tally
    rng = np.random.default_rng(n_reads)
    if n_reads == 3:
        tag = np.array([4, 1, 4], np.uint8)
        h = np.array([50, 20, 50], np.int16)
        pair = np.array([5, 5, 5], np.int32)
    else:
        tag = rng.choice(np.array([0, 1, 2, 3, 4, 4, 4, 5], np.uint8), n_reads)
        h = rng.integers(1, 51, n_reads).astype(np.int16)
        pair = rng.integers(0, n_reads // 2, n_reads).astype(np.int32)     # names with 1, 2, 3, ... records
    hs = 52
    full = np.zeros((1, hs), np.int32); pref = np.zeros((1, hs), np.int32); rept = np.zeros((1, hs), np.int32)
    ctx.tally(_lib.MEM_HOST, tag, h, n_reads, np.array([0, n_reads], np.int32), 1, pair, hs, full, pref, rept)
    counts, details, n_rept = tally(_FakeScan(pair), 0, tag, h, repeatpairs=False)
    for dev, key in ((full, "FULL"), (pref, "PREF"), (rept, "REPT")):
        assert {int(k): int(v) for k, v in enumerate(dev[0]) if v} == dict(counts[key]), key
    assert int(rept.sum()) == n_rept
    if n_reads == 3:
        assert full.sum() == 0 and rept.sum() == 0 and details == []
    else:
        assert 0 < n_rept < int((tag == 4).sum())          # some pairs removed, some REPT reads kept


def test_scoring_bound_of_the_packed_values(ctx):
    """(rows + 511) * gap_extend + max_read_len * match must stay below 2^13: accepted under the bound
    (and the alignment is right: checked against the oracle), refused just over it."""
    lad = HD
    tmpl = HD[0] + "CAG" * 20 + HD[2]
    reads = [tmpl[3:93], tmpl[10:60] + "T" + tmpl[60:99], "CAG" * 30]
    ok = _lib.SwParams(4, 9, 10, 10, 9, 0, 100, 0)           # (100 + 511) * 10 + 100 * 4 = 6510
    cls = po.classify(reads, np.zeros(len(reads), np.int32), po.LocusSet([lad]), scoring=(4, 9, 10, 10))
    t, hh, sc, _ = _classify(ctx, [lad], reads, [0, len(reads)], [0], params=ok)
    assert np.array_equal(t, cls[:, 0]) and np.array_equal(hh, cls[:, 1]) and np.array_equal(sc, cls[:, 2])
    edge = _lib.SwParams(1, 5, 12, 12, 9, 0, 100, 0)         # 7 rows x 16 lanes: (112 + 511) * 12 + 100 = 7576
    _classify(ctx, [lad], reads, [0, len(reads)], [0], params=edge)
    over1 = _lib.SwParams(1, 5, 13, 13, 9, 0, 100, 0)        # (112 + 511) * 13 + 100 = 8199
    with pytest.raises(_lib.TredGpuError, match="packed DP values"):
        _classify(ctx, [lad], reads, [0, len(reads)], [0], params=over1)
    over = _lib.SwParams(1, 5, 11, 11, 9, 0, 0, 0)           # no max_read_len: 256 assumed -> 767 * 11 + 256 = 8693
    with pytest.raises(_lib.TredGpuError, match="packed DP values"):
        _classify(ctx, [lad], reads, [0, len(reads)], [0], params=over)
    over2 = _lib.SwParams(8, 9, 9, 9, 9, 0, 0, 0)            # 767 * 9 + 2048 = 8951
    with pytest.raises(_lib.TredGpuError, match="packed DP values"):
        _classify(ctx, [lad], reads, [0, len(reads)], [0], params=over2)
    # no max_read_len named and a 400 bp read in HOST memory: the call resolves the 512-row instantiation, and the bound is
    # checked for THAT one ((512 + 511) * 5 + 400 * 8 = 8315), not for the 320 bp assumed before the reads were seen
    # ((320 + 511) * 5 + 320 * 8 = 6715, which passes) -- ADVICE r5
    long_tmpl = HD[0] + "CAG" * 130 + HD[2]
    tall = _lib.SwParams(8, 9, 5, 5, 9, 0, 0, 0)
    _classify(ctx, [lad], reads, [0, len(reads)], [0], params=tall)                  # 90-99 bp reads: fine
    with pytest.raises(_lib.TredGpuError, match="packed DP values") as e:
        _classify(ctx, [lad], reads + [long_tmpl[:400]], [0, len(reads) + 1], [0], params=tall)
    assert "(-2)" in str(e.value)


def test_error_reporting(ctx):
    lib = ctx.lib
    p = _lib.default_sw_params()
    one = np.zeros(4, np.int32)
    rc = lib.tredgpu_sw_classify(ctx.h, 0, None, None, None, 5, None, None, 1, C.byref(p), None, None, None, None, 0)
    assert rc < 0 and b"NULL" in lib.tredgpu_last_error(ctx.h)
    bad = _lib.SwParams(1, 5, 2, 7, 9, 0, 0, 0)            # gap_extend > gap_open
    rc = lib.tredgpu_sw_classify(ctx.h, 0, one.ctypes.data, one.ctypes.data, one.ctypes.data, 0, one.ctypes.data,
                                 one.ctypes.data, 0, C.byref(bad), one.ctypes.data, one.ctypes.data, one.ctypes.data, None, 0)
    assert rc < 0 and b"scoring" in lib.tredgpu_last_error(ctx.h)
    with pytest.raises(_lib.TredGpuError):
        ctx.set_ladders([("ACGT" * 5, "CAG", "ACGT" * 5, 200)])      # template longer than 511
    ctx.set_ladders([HD])
    packed, woff, rlen = _lib.pack_reads(["ACGT" * 30])
    with pytest.raises(_lib.TredGpuError):                            # ladder index out of range
        ctx.sw_classify(_lib.MEM_HOST, packed, woff, rlen, 1, np.array([0, 1], np.int32), np.array([3], np.int32), 1, p,
                        np.zeros(1, np.uint8), np.zeros(1, np.int16), np.zeros(1, np.int16))


def test_300bp_reads_take_the_20_rows_per_lane_instantiation(ctx):
    """2 x 300 bp runs (the reference has no length limit, ssw.c:780-871 / bam_parser.py:73): reads up to 320 bp are
    held by sw_cont_kernel<20, 2>.  Per-template records against the oracle field by field, tags of a synthetic 300 bp
    batch over loci of every period against the oracle, and reads beyond the limit (480 bp) refused, not truncated."""
    rng = np.random.default_rng(300)
    mu = -(-300 // 3)
    lad = (HD[0], "CAG", HD[2], mu)
    flank = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    reads = [(flank(40) + HD[0] + "CAG" * 60 + HD[2] + flank(100))[:300],            # spanning
             po.rc((flank(10) + HD[0] + "CAG" * 90 + HD[2] + flank(50))[:300]),       # prefix read, other strand
             ("CAG" * 110)[:300], ("AGC" * 110)[1:299], "N" * 300,                     # inside the repeat; all N
             (flank(150) + "CAG" * 20 + HD[2] + flank(100))[:300],                    # suffix read
             ("CAG" * 99)[:297] + "TTT", flank(300), (HD[0] + "CAG" * 100)[:310]]      # 310 bp: still inside the limit
    tag, h, sc, dump = _classify(ctx, [lad], reads, [0, len(reads)], [0], dump=True)
    ls = po.LocusSet([lad])
    cls = po.classify(reads, np.zeros(len(reads), np.int32), ls)
    assert np.array_equal(tag, cls[:, 0]) and np.array_equal(h, cls[:, 1]) and np.array_equal(sc, cls[:, 2])
    assert set(tag.tolist()) >= {_lib.TAG_FULL, _lib.TAG_REPT, _lib.TAG_NONE}
    nt = 2 * mu
    for r, read in enumerate(reads):
        want = po.sw_pairs([read], ls.templates, [0] * nt, list(range(nt)))
        assert np.array_equal(dump[r, :, :5].astype(np.int32), want), r
    # a synthetic 300 bp batch over periods 3, 4, 5, 6 and 12 through the production (pruned) path
    loci = [l for l in synth.load_loci() if l["name"] in ("HD", "DM2", "SCA10", "SCA36", "ULD", "OPMD")]
    b = synth.build_batch(301, loci, 2, synth.SynthParams(coverage=20, readlen=300, max_units=90))
    ctx.set_ladders(b.ladders)
    n = b.n_reads
    t2 = np.zeros(n, np.uint8); h2 = np.zeros(n, np.int16); s2 = np.zeros(n, np.int16)
    ctx.sw_classify(_lib.MEM_HOST, b.packed, b.read_off, b.read_len, n, b.unit_read_off, b.unit_ladder, b.n_units,
                    _lib.default_sw_params(max_read_len=300), t2, h2, s2)
    rs = [synth.decode(x) for x in b.codes]
    c2 = po.classify(rs, np.repeat(b.unit_ladder, np.diff(b.unit_read_off)), po.LocusSet(b.ladders), threads=0)
    assert np.array_equal(t2, c2[:, 0]) and np.array_equal(h2, c2[:, 1]) and np.array_equal(s2, c2[:, 2])
    assert (t2 == _lib.TAG_FULL).sum() > 20 and (t2 == _lib.TAG_REPT).sum() > 0
    # beyond 480 bp: refused by the call (host memory: the lengths are looked at), never truncated
    with pytest.raises(_lib.TredGpuError, match="TREDGPU_MAX_READ_LEN"):
        _classify(ctx, [lad], [reads[0], (reads[0] * 2)[:490]], [0, 2], [0])


def test_reads_of_321_to_470_bp_take_the_32_rows_per_lane_instantiation(ctx):
    """Merged pairs / long runs: reads beyond 320 bp are held by sw_cont_kernel<32, 1> (16 lanes x 32 rows: the nine row
    bits of the packed DP values; ladders stay below 512 columns).  Per-template records against the oracle field by
    field, and a synthetic 400 bp batch over loci of several periods through the production path."""
    rng = np.random.default_rng(400)
    flank = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    for readlen in (330, 400, 470):
        mu = -(-readlen // 3)
        lad = (HD[0], "CAG", HD[2], mu)
        reads = [(flank(60) + HD[0] + "CAG" * 70 + HD[2] + flank(400))[:readlen],          # spanning
                 po.rc((flank(10) + HD[0] + "CAG" * 150 + HD[2] + flank(50))[:readlen]),   # prefix read, other strand
                 ("CAG" * 160)[:readlen], ("AGC" * 160)[1:readlen - 1], "N" * readlen,      # inside the repeat; all N
                 (flank(200) + "CAG" * 20 + HD[2] + flank(400))[:readlen],                 # suffix read
                 flank(readlen), (HD[0] + "CAG" * 160)[:321]]                              # the shortest read of the instantiation
        tag, h, sc, dump = _classify(ctx, [lad], reads, [0, len(reads)], [0], dump=True)
        ls = po.LocusSet([lad])
        cls = po.classify(reads, np.zeros(len(reads), np.int32), ls)
        assert np.array_equal(tag, cls[:, 0]) and np.array_equal(h, cls[:, 1]) and np.array_equal(sc, cls[:, 2]), readlen
        assert set(tag.tolist()) >= {_lib.TAG_FULL, _lib.TAG_REPT, _lib.TAG_NONE}
        nt = 2 * mu
        for r, read in enumerate(reads):
            want = po.sw_pairs([read], ls.templates, [0] * nt, list(range(nt)))
            assert np.array_equal(dump[r, :, :5].astype(np.int32), want), (readlen, r)
    loci = [l for l in synth.load_loci() if l["name"] in ("HD", "DM2", "SCA10", "SCA36", "ULD")]
    b = synth.build_batch(401, loci, 2, synth.SynthParams(coverage=20, readlen=400, max_units=120))
    ctx.set_ladders(b.ladders)
    n = b.n_reads
    t2 = np.zeros(n, np.uint8); h2 = np.zeros(n, np.int16); s2 = np.zeros(n, np.int16)
    ctx.sw_classify(_lib.MEM_HOST, b.packed, b.read_off, b.read_len, n, b.unit_read_off, b.unit_ladder, b.n_units,
                    _lib.default_sw_params(max_read_len=400), t2, h2, s2)
    rs = [synth.decode(x) for x in b.codes]
    c2 = po.classify(rs, np.repeat(b.unit_ladder, np.diff(b.unit_read_off)), po.LocusSet(b.ladders), threads=0)
    assert np.array_equal(t2, c2[:, 0]) and np.array_equal(h2, c2[:, 1]) and np.array_equal(s2, c2[:, 2])
    assert (t2 == _lib.TAG_FULL).sum() > 20 and (t2 == _lib.TAG_REPT).sum() > 0


@pytest.mark.parametrize("repeatpairs", [True, False])
def test_fused_host_call_equals_the_three_separate_calls(tmp_path, repeatpairs):
    """tredgpu_genotype_batch_joint (SW -> tally -> grid with marginals and sparse joint composed on the device, one wait:
    what engine.genotype_packed calls for every product batch) against tredgpu_sw_classify + tredgpu_tally +
    tredgpu_likelihood_grid_joint one after the other: the same tags, calls, marginals and joint entries bit for bit."""
    from tredparse_amd import bam_parser, synth_bam
    from tredparse_amd.engine import Engine, PackedUnits
    from tredparse_amd.meta import TREDsRepo
    made = synth_bam.make_bams(str(tmp_path), 3, seed=41, workers=1)
    repo = TREDsRepo("hg38", sites=str(tmp_path / "no_sites"))
    names = [l["name"] for l in synth_bam.bench_loci()]
    scans = [bam_parser.scan_sample(path, repo, names) for _, path, _ in made]
    b = PackedUnits.from_scans([(s, list(range(len(names)))) for s in scans], repeatpairs=repeatpairs)
    eng = Engine(0)
    one, three = eng.genotype_packed(b), eng._genotype_packed_stepwise(b)
    assert b.n_units == 90 and b.n_reads > 5000
    for key in ("tag", "h", "score", "rept", "marg"):
        assert np.array_equal(getattr(one, key), getattr(three, key)), key
    assert one.calls.tobytes() == three.calls.tobytes() and (one.calls["status"] == 0).sum() > 60
    a1, b1, v1, lo1, n1 = one.joint_units
    a3, b3, v3, lo3, n3 = three.joint_units
    assert np.array_equal(n1, n3) and n1.sum() > 500
    for u in range(b.n_units):         # (the entries of a unit arrive in any order: compare them sorted)
        k1 = sorted(zip(a1[lo1[u]:lo1[u] + n1[u]].tolist(), b1[lo1[u]:lo1[u] + n1[u]].tolist(), v1[lo1[u]:lo1[u] + n1[u]].tolist()))
        k3 = sorted(zip(a3[lo3[u]:lo3[u] + n3[u]].tolist(), b3[lo3[u]:lo3[u] + n3[u]].tolist(), v3[lo3[u]:lo3[u] + n3[u]].tolist()))
        assert k1 == k3, u
    eng.close()
