"""GPU, BASELINE-size batches: properties that do not need the (slow) CPU oracle for every unit --
idempotence, unit-order invariance, histogram/tag consistency, device-memory vs host-memory paths --
plus an oracle spot check on a random subset and a full oracle check of the tags on a mid-size batch
spanning all 32 loci (the non-dump path, i.e. with every exact shortcut of the kernel active)."""
import numpy as np
import pytest

from oracle import lik_oracle as lo
from oracle import pyoracle as po
from tredparse_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _model(ctx):
    step, w = lo.load_model()
    ctx.set_model(np.array([step[p] for p in range(1, 7)]), np.array(w))


def _run(ctx, b, order=None):
    """Fused path on host buffers; order = permutation of units (reads are re-laid out accordingly)."""
    g = b.n_units
    order = np.arange(g) if order is None else np.asarray(order)
    sizes = np.diff(b.unit_read_off)[order]
    uro = np.zeros(g + 1, np.int32)
    uro[1:] = np.cumsum(sizes)
    ridx = np.concatenate([np.arange(b.unit_read_off[u], b.unit_read_off[u + 1]) for u in order])
    W = int(b.read_off[1] - b.read_off[0])
    packed = b.packed.reshape(-1, W)[ridx].reshape(-1).copy()
    roff = (np.arange(len(ridx) + 1, dtype=np.int64) * W)
    rlen = b.read_len[ridx].copy()
    n = len(ridx)
    tag = np.zeros(n, np.uint8); h = np.zeros(n, np.int16); sc = np.zeros(n, np.int16)
    hs = b.hist_stride
    full = np.zeros((g, hs), np.int32); pref = np.zeros((g, hs), np.int32); rept = np.zeros((g, hs), np.int32)
    calls = np.zeros(g, _lib.CALL_DTYPE)
    ctx.genotype_batch(_lib.MEM_HOST, packed, roff, rlen, n, uro, b.unit_ladder[order].copy(), b.units[order].copy(), g,
                       _lib.default_sw_params(max_read_len=150), None, b.global_lens, len(b.global_lens),
                       b.target_lens, len(b.target_lens), tag, h, sc, hs, full, pref, rept, calls)
    return dict(tag=tag, h=h, sc=sc, full=full, pref=pref, rept=rept, calls=calls, uro=uro, ridx=ridx)


def test_full_size_properties(ctx, loci):
    _model(ctx)
    sel = [l for l in loci if l["name"] not in ("FXTAS", "AR")]
    # BASELINE configs[2] at full size: 1 000 samples x 30 loci = 30 000 units, ~2.9 M reads (bench.py's batch)
    b = synth.build_batch(20260101, sel, 1000, synth.SynthParams(coverage=30), workers=32)
    assert b.n_units == 30000 and b.n_reads > 2500000
    ctx.set_ladders(b.ladders)
    r1 = _run(ctx, b)
    r2 = _run(ctx, b)
    for k in ("tag", "h", "sc", "full", "pref", "rept"):
        assert np.array_equal(r1[k], r2[k]), k                           # idempotent, bit for bit
    assert r1["calls"].tobytes() == r2["calls"].tobytes()
    # unit order must not matter: units are independent, and although a quad of the SW kernel holds reads of
    # several units (quads are packed by ladder / class / 6-mer level across the batch), a read's result does not
    # depend on its wave-mates
    rng = np.random.default_rng(1)
    perm = rng.permutation(b.n_units)
    r3 = _run(ctx, b, perm)
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    assert r3["calls"][inv].tobytes() == r1["calls"].tobytes()
    back = np.empty(len(r3["ridx"]), np.int64); back[r3["ridx"]] = np.arange(len(r3["ridx"]))
    assert np.array_equal(r3["tag"][back], r1["tag"]) and np.array_equal(r3["h"][back], r1["h"])
    # histograms == recount of the tags (bam_parser.py:259-268; POST lands in PREF's bins)
    unit_of = np.repeat(np.arange(b.n_units), np.diff(b.unit_read_off))
    for name, tags in (("full", (1,)), ("pref", (2, 3)), ("rept", (4,))):
        m = np.isin(r1["tag"], tags)
        want = np.zeros_like(r1[name])
        np.add.at(want, (unit_of[m], r1["h"][m].astype(np.int64)), 1)
        assert np.array_equal(want, r1[name]), name
    # every unit got a call; the short allele is recovered exactly most of the time
    assert (r1["calls"]["status"] == 0).all()
    ok = (r1["calls"]["h1"] // b.units["period"]) == b.h_true[:, 0]
    assert ok.mean() > 0.75
    # oracle spot check (classification + likelihood) on 48 random units
    ls = po.LocusSet(b.ladders)
    for u in rng.choice(b.n_units, 48, replace=False):
        r0, r1_ = b.unit_read_off[u], b.unit_read_off[u + 1]
        reads = [synth.decode(x) for x in b.codes[r0:r1_]]
        cls = po.classify(reads, np.full(len(reads), b.unit_ladder[u], np.int32), ls, threads=0)
        assert np.array_equal(r1["tag"][r0:r1_], cls[:, 0].astype(np.uint8)), u
        assert np.array_equal(r1["h"][r0:r1_], cls[:, 1].astype(np.int16)), u
        up = b.units[u]
        f = {k: int(v) for k, v in enumerate(r1["full"][u]) if v}
        pp = {k: int(v) for k, v in enumerate(r1["pref"][u]) if v}
        res = lo.Caller(int(up["period"]), 150, 2, 2 * float(up["half_depth"]), f, pp, int(r1["rept"][u].sum()),
                        b.global_lens[up["pe_off"]:up["pe_off"] + up["n_global"]],
                        b.target_lens[up["tl_off"]:up["tl_off"] + up["n_target"]], int(up["ref_len"]),
                        int(up["minpe"])).evaluate()
        c = r1["calls"][u]
        assert (c["h1"], c["h2"]) == tuple(res["alleles"]), u
        assert abs(c["lik"] - res["lik"]) <= 1e-6 and tuple(c["ci"]) == tuple(res["CI"])


def test_all_loci_tags_against_oracle(ctx, loci):
    """Every read of a 32-locus batch through the pruned (non-dump) kernel path vs the oracle."""
    b = synth.build_batch(77, loci, 6, synth.SynthParams(coverage=30, expanded_max=150, expanded_frac=0.3), workers=8)
    ctx.set_ladders(b.ladders)
    n = b.n_reads
    tag = np.zeros(n, np.uint8); h = np.zeros(n, np.int16); sc = np.zeros(n, np.int16)
    ctx.sw_classify(_lib.MEM_HOST, b.packed, b.read_off, b.read_len, n, b.unit_read_off, b.unit_ladder, b.n_units,
                    _lib.default_sw_params(max_read_len=150), tag, h, sc)
    reads = [synth.decode(r) for r in b.codes]
    cls = po.classify(reads, np.repeat(b.unit_ladder, np.diff(b.unit_read_off)), po.LocusSet(b.ladders), threads=0)
    bad = np.nonzero((tag != cls[:, 0]) | (h != cls[:, 1]) | (sc != cls[:, 2]))[0]
    assert len(bad) == 0, (len(bad), bad[:5], tag[bad[:5]], cls[bad[:5]])
    assert (tag == 4).sum() > 0 and (tag == 1).sum() > 0 and (tag == 5).sum() > 0


@pytest.mark.parametrize("scoring", [(1, 5, 7, 2), (2, 3, 5, 2), (1, 1, 2, 1)])
def test_pruned_path_equals_argmax_of_the_unpruned_dump(ctx, loci, scoring):
    """Every exact shortcut of the production kernel (6-mer strand filter, exact-score drop, strand exit, steady-state
    exit, deferred best-cell resolution) against the kernel variant that has none of them: the per-template dump
    (each of its records is checked against ssw.c elsewhere) gives (score, tag) of every (read, template); the arg-max
    by (score, -units), forward strand before reverse complement, must be what the production launch returns --
    ~60 000 reads over all loci incl. tracts longer than a read, without the CPU oracle's cost."""
    sp = synth.SynthParams(coverage=30, expanded_max=150, expanded_frac=0.3)
    b = synth.build_batch(1234 + scoring[1], loci, 20, sp, workers=8)
    ctx.set_ladders(b.ladders)
    n = b.n_reads
    params = _lib.SwParams(scoring[0], scoring[1], scoring[2], scoring[3], 9, 0, 150, 0)
    tag = np.zeros(n, np.uint8); h = np.zeros(n, np.int16); sc = np.zeros(n, np.int16)
    ctx.sw_classify(_lib.MEM_HOST, b.packed, b.read_off, b.read_len, n, b.unit_read_off, b.unit_ladder, b.n_units,
                    params, tag, h, sc)
    nt = 2 * max(l[3] for l in b.ladders)
    dump = np.zeros((n, nt, 6), np.int16)
    t2 = np.zeros(n, np.uint8); h2 = np.zeros(n, np.int16); s2 = np.zeros(n, np.int16)
    ctx.sw_classify(_lib.MEM_HOST, b.packed, b.read_off, b.read_len, n, b.unit_read_off, b.unit_ladder, b.n_units,
                    params, t2, h2, s2, dump, nt)
    assert np.array_equal(tag, t2) and np.array_equal(h, h2) and np.array_equal(sc, s2)      # the two kernel variants
    k = np.arange(nt)
    units, strand = k // 2 + 1, k % 2
    score, dtag = dump[:, :, 0].astype(np.int64), dump[:, :, 5].astype(np.int64)
    key = np.where(dtag != _lib.TAG_NONE, (score << 11) | ((511 - units)[None, :] << 1) | (1 - strand)[None, :], -1)
    at = key.argmax(1)
    rows = np.arange(n)
    none = key[rows, at] < 0
    want_tag = np.where(none, _lib.TAG_NONE, dtag[rows, at])
    want_h = np.where(none, 0, units[at])
    want_sc = np.where(none, 0, score[rows, at])
    bad = np.nonzero((tag != want_tag) | (h != want_h) | (sc != want_sc))[0]
    assert len(bad) == 0, (scoring, len(bad), bad[:5], tag[bad[:5]], h[bad[:5]], want_tag[bad[:5]], want_h[bad[:5]])
    assert n > 40000 and all((tag == t).sum() > 100 for t in (1, 2, 3, 4))


def test_device_memory_path_matches_host_path(ctx, loci):
    torch = pytest.importorskip("torch")
    _model(ctx)
    sel = [l for l in loci if l["name"] in ("HD", "DM1", "SCA10", "ULD")]
    b = synth.build_batch(5, sel, 12, synth.SynthParams(coverage=30))
    ctx.set_ladders(b.ladders)
    ref = _run(ctx, b)
    dev = torch.device("cuda", 0)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    n, g, hs = b.n_reads, b.n_units, b.hist_stride
    d = dict(packed=dv(b.packed.view(np.int32)), roff=dv(b.read_off), rlen=dv(b.read_len), uoff=dv(b.unit_read_off),
             ulad=dv(b.unit_ladder), units=dv(b.units.view(np.uint8)), gl=dv(b.global_lens), tl=dv(b.target_lens),
             tag=torch.zeros(n, dtype=torch.uint8, device=dev), h=torch.zeros(n, dtype=torch.int16, device=dev),
             sc=torch.zeros(n, dtype=torch.int16, device=dev),
             full=torch.zeros((g, hs), dtype=torch.int32, device=dev), pref=torch.zeros((g, hs), dtype=torch.int32, device=dev),
             rept=torch.zeros((g, hs), dtype=torch.int32, device=dev),
             calls=torch.zeros(g * _lib.CALL_DTYPE.itemsize, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    ctx.genotype_batch(_lib.MEM_DEVICE, d["packed"], d["roff"], d["rlen"], n, d["uoff"], d["ulad"], d["units"], g,
                       _lib.default_sw_params(max_read_len=150), None, d["gl"], len(b.global_lens), d["tl"],
                       len(b.target_lens), d["tag"], d["h"], d["sc"], hs, d["full"], d["pref"], d["rept"], d["calls"])
    ctx.sync()
    assert np.array_equal(d["tag"].cpu().numpy(), ref["tag"]) and np.array_equal(d["h"].cpu().numpy(), ref["h"])
    assert np.array_equal(d["full"].cpu().numpy(), ref["full"]) and np.array_equal(d["rept"].cpu().numpy(), ref["rept"])
    assert d["calls"].cpu().numpy().tobytes() == ref["calls"].tobytes()


def _config5_against_the_oracle(ctx, sel, seed, samples, expect_big):
    """A configs[4]-shaped batch (100x, alleles beyond a read, one up to 200 repeats) through the kernels and through the CPU
    restatement: every read's tag and repeat count, every unit's call, likelihood, CI and pair count."""
    p = synth.SynthParams(coverage=100, min_units=42, max_units=60, expanded_max=200, expanded_frac=0.8)
    b = synth.build_batch(seed, sel, samples, p, maxinsert=300)
    ctx.set_ladders(b.ladders)
    r = _run(ctx, b)
    assert (r["calls"]["status"] == 0).all()
    if expect_big:
        assert r["calls"]["n_pairs"].max() > 20000 and (r["tag"] == 4).sum() > 50         # big grids, REPT reads
    ls = po.LocusSet(b.ladders)
    reads = [synth.decode(x) for x in b.codes]
    cls = po.classify(reads, np.repeat(b.unit_ladder, np.diff(b.unit_read_off)), ls, threads=0)
    assert np.array_equal(r["tag"], cls[:, 0].astype(np.uint8)) and np.array_equal(r["h"], cls[:, 1].astype(np.int16))
    for u in range(b.n_units):
        up = b.units[u]
        f = {k: int(v) for k, v in enumerate(r["full"][u]) if v}
        pp = {k: int(v) for k, v in enumerate(r["pref"][u]) if v}
        res = lo.Caller(int(up["period"]), 150, int(up["ploidy"]), 2 * float(up["half_depth"]), f, pp,
                        int(r["rept"][u].sum()),
                        b.global_lens[up["pe_off"]:up["pe_off"] + up["n_global"]],
                        b.target_lens[up["tl_off"]:up["tl_off"] + up["n_target"]], int(up["ref_len"]),
                        int(up["minpe"])).evaluate()
        c = r["calls"][u]
        assert (c["h1"], c["h2"]) == tuple(res["alleles"]), u
        assert abs(c["lik"] - res["lik"]) <= 1e-6 and tuple(c["ci"]) == tuple(res["CI"]) and c["n_pairs"] == len(res["mls"])
    return b.n_units


def test_config5_high_coverage_expanded_alleles(ctx, loci):
    """BASELINE configs[4]: 100x coverage, one allele expanded up to 200 repeats -- large (h1,h2) grids
    (no spanning read for the long allele: extended h2 range, PE mode) and many repeat-only reads."""
    _model(ctx)
    sel = [l for l in loci if l["name"] in ("HD", "DM1", "SCA1")]
    assert _config5_against_the_oracle(ctx, sel, 55, 2, True) == 6


def test_config5_across_all_loci(ctx, loci):
    """configs[4] again over every locus the bench's 200 x 30 batch holds (motifs of 3, 4, 5, 6 and 12 bp, X-linked and
    autosomal, either strand's flanks): one 100x sample x 30 loci against the oracle, read by read and call by call
    (VERDICT r5 weak 1 iii: six units were 'adequate, not generous')."""
    _model(ctx)
    sel = [l for l in loci if l["name"] not in ("FXTAS", "AR")]       # (the 30 of bench.py / synth_bam.bench_loci)
    assert _config5_against_the_oracle(ctx, sel, 77, 1, False) == 30


def test_small_scratch_pool_takes_several_passes(loci, monkeypatch):
    """The grid kernels carve each unit's tables out of one scratch pool; units that find it full wait for
    the next pass.  A 4 MiB pool (two or three of the big grids per pass) must give the same calls as the
    default pool, which holds this batch at once."""
    sel = [l for l in loci if l["name"] in ("HD", "DM1", "SCA1")]
    p = synth.SynthParams(coverage=30, min_units=5, max_units=60)
    b = synth.build_batch(60, sel, 11, p, maxinsert=300)
    out = []
    for mb in (None, "4"):
        if mb is None:
            monkeypatch.delenv("TREDGPU_GRID_POOL_MB", raising=False)
        else:
            monkeypatch.setenv("TREDGPU_GRID_POOL_MB", mb)
        c = _lib.Context(0)
        try:
            _model(c)
            c.set_ladders(b.ladders)
            out.append(_run(c, b))
        finally:
            c.close()
    a, s = out
    assert (a["calls"]["status"] == 0).all()
    assert (a["calls"]["n_pairs"] > 20000).sum() >= 4      # each of these needs > 0.5 MiB of the pool
    assert np.array_equal(a["calls"], s["calls"])
