"""GPU parity: HIP likelihood grid / KDE / fused genotyping path vs the reference goldens and the
CPU oracle.  Tolerances: per-term log-likelihoods 1e-6 absolute (BASELINE.json north_star; observed
~1e-12), (h1,h2) calls, CI and grid enumeration bit-identical."""
import numpy as np
import pytest

from oracle import lik_oracle as lo
from oracle import pyoracle as po
from tests.gridcases import load_cases, oracle_caller
from tredparse_amd import _lib, synth

pytestmark = pytest.mark.gpu
CASES = load_cases()
ML_TOL = 1e-6


def _set_model(ctx):
    step, w = lo.load_model()
    ctx.set_model(np.array([step[p] for p in range(1, 7)]), np.array(w))


def _case_inputs(cases, hist_stride):
    n = len(cases)
    units = np.zeros(n, _lib.UNIT_DTYPE)
    full = np.zeros((n, hist_stride), np.int32)
    pref = np.zeros((n, hist_stride), np.int32)
    rept = np.zeros((n, hist_stride), np.int32)
    gl, tl = [], []
    for i, c in enumerate(cases):
        u = synth.unit_params_for(c["locus_rec"], c["readlen"], c["depth"], len(c["global_lens"]),
                                  len(c["target_lens"]), len(gl), len(tl), ploidy=c["ploidy"],
                                  maxinsert=c["maxinsert"], fullsearch=c["fullsearch"])
        u["ref_len"], u["minpe"] = c["ref_len"], c["minpe"]
        units[i] = u
        for k, v in c["full"].items():
            full[i, int(k)] = v
        for k, v in c["partial"].items():
            pref[i, int(k)] = v
        rept[i, 0] = c["rept"]
        gl += c["global_lens"]
        tl += c["target_lens"]
    return units, full, pref, rept, np.asarray(gl, np.int32), np.asarray(tl, np.int32)


@pytest.mark.parametrize("hs", [128, 400])   # 400: histograms too wide for the kernel's LDS staging (other code path)
def test_grid_matches_reference_goldens(ctx, hs):
    _set_model(ctx)
    units, full, pref, rept, gl, tl = _case_inputs(CASES, hs)
    n = len(CASES)
    cap = np.array([max(c["expected"].get("n_pairs", 0), 1) for c in CASES], np.int64)
    goff = np.zeros(n + 1, np.int64)
    goff[1:] = np.cumsum(cap)
    dump = np.zeros((int(goff[-1]), 6), np.float64)
    ms = 512
    marg = np.zeros((n, 2, ms), np.float64)
    calls = np.zeros(n, _lib.CALL_DTYPE)
    ctx.likelihood_grid(_lib.MEM_HOST, units, n, hs, full, pref, rept, gl, len(gl), tl, len(tl), calls, goff, dump,
                        marg, ms)
    for i, c in enumerate(CASES):
        exp, call = c["expected"], calls[i]
        period = len(c["locus_rec"]["repeat"])
        if exp["raised"]:
            assert call["status"] == -2, c["name"]           # singular KDE -> LinAlgError in the reference
            continue
        if exp["alleles"] == [-1, -1]:
            assert call["status"] == 1, c["name"]
            continue
        assert call["status"] == 0, (c["name"], call)
        assert call["n_pairs"] == exp["n_pairs"], c["name"]
        got = dump[goff[i]:goff[i] + call["n_pairs"]]
        want = c["mls"]
        assert np.array_equal(got[:, :2], want[:, :2]), c["name"]            # enumeration order (models.py:260-264)
        assert np.abs(got[:, 2:] - want[:, 2:]).max() <= ML_TOL, c["name"]
        assert sorted([call["h1"] // period, call["h2"] // period]) == exp["alleles"], c["name"]
        assert "{}-{}|{}-{}".format(*call["ci"]) == exp["CI"], c["name"]
        assert abs(call["pp"] - exp["PP"]) <= 1e-9, c["name"]
        tot = want[:, 2:].sum(axis=1)
        assert abs(call["lik"] - tot.max()) <= ML_TOL
        # marginals (un-normalised) -> the reference's sparsified P_h1 / P_h2 (models.py:304-317)
        for which, name in enumerate(("P_h1", "P_h2")):
            m = marg[i, which]
            z = {str(k): m[k] / m.sum() for k in np.nonzero(m >= lo.SMALL_VALUE)[0]}
            assert set(z) == set(exp[name]), (c["name"], name)
            for k in z:
                assert abs(z[k] - exp[name][k]) <= 1e-9


def test_kde_matches_reference(ctx):
    _set_model(ctx)
    cs = [c for c in CASES if c["kde"] is not None]
    units, _, _, _, gl, _ = _case_inputs(cs, 128)
    pdf = np.zeros((len(cs), 1000))
    st = np.zeros(len(cs), np.int32)
    ctx.pe_kde(_lib.MEM_HOST, units, len(cs), gl, len(gl), pdf, st)
    assert (st == 0).all()
    for i, c in enumerate(cs):
        assert np.abs(pdf[i] - c["kde"]).max() <= 1e-12
        assert abs(pdf[i].sum() - 1) < 1e-12


def test_kde_of_equal_pair_lengths_follows_scipy(ctx):
    """Zero-variance pair lengths (models.py:428-435 hands them to scipy): gaussian_kde raises LinAlgError when the
    rounding of its weighted mean gives back the common value exactly, and otherwise builds a pdf that is 1 at that
    value and 0 elsewhere -- which of the two is a pure function of (value, count).  The kernel reproduces numpy's
    pairwise reduction for it (csrc/grid.hip kde_of_equal_lengths): same outcome as scipy here for every drawn case,
    through both entry points (the KDE alone, and a unit of the grid whose model exists but is not used)."""
    import warnings
    from scipy.stats import gaussian_kde
    _set_model(ctx)
    rng = np.random.default_rng(5)
    cases = [(int(n), int(c)) for n, c in zip(rng.integers(100, 3000, 160), rng.integers(1, 1000, 160))]
    cases += [(100, 22), (100, 1), (128, 999), (129, 500), (256, 350), (2727, 390), (1000, 0)]
    locus = CASES[0]["locus_rec"]
    units = np.zeros(len(cases), _lib.UNIT_DTYPE)
    gl, want = [], []
    for i, (n, c) in enumerate(cases):
        units[i] = synth.unit_params_for(locus, 150, 30.0, n, 5, len(gl), 0)
        gl += [c] * n
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                pdf = gaussian_kde([c] * n).evaluate(np.arange(1000))
                want.append(pdf / pdf.sum())
            except np.linalg.LinAlgError:
                want.append(None)
    assert any(w is None for w in want) and any(w is not None for w in want)
    gl = np.asarray(gl, np.int32)
    pdf = np.zeros((len(cases), 1000))
    st = np.zeros(len(cases), np.int32)
    ctx.pe_kde(_lib.MEM_HOST, units, len(cases), gl, len(gl), pdf, st)
    for i, w in enumerate(want):
        assert st[i] == (-2 if w is None else 0), cases[i]
        if w is not None:
            assert np.array_equal(pdf[i], w), cases[i]
    # the grid: a unit with five spanning pairs and equal pair lengths, no partial read near the read length -> the
    # paired-end term is off, and the reference still dies in PEMaxLikModel when scipy raises (status -2), else calls
    hs = 128
    full, pref, rept = (np.zeros((len(cases), hs), np.int32) for _ in range(3))
    full[:, 15] = 10
    tl = np.full(5, 400, np.int32)
    units["n_target"], units["tl_off"] = 5, 0
    calls = np.zeros(len(cases), _lib.CALL_DTYPE)
    ctx.likelihood_grid(_lib.MEM_HOST, units, len(cases), hs, full, pref, rept, gl, len(gl), tl, len(tl), calls, None, None, None, 0)
    for i, w in enumerate(want):
        assert calls[i]["status"] == (-2 if w is None else 0), cases[i]


def test_fused_batch_matches_oracle_chain(ctx, loci):
    """SW -> tally -> grid on the GPU vs oracle classification + numpy likelihood, unit by unit."""
    _set_model(ctx)
    rng = np.random.default_rng(99)
    sel = [l for l in loci if l["name"] in ("HD", "DM1", "SCA10", "ULD", "FRDA", "AR", "OPMD")]
    p = synth.SynthParams(coverage=30, expanded_max=120, expanded_frac=0.3)
    b = synth.build_batch(rng, sel, 6, p, maxinsert=150)
    ctx.set_ladders(b.ladders)
    n, g = b.n_reads, b.n_units
    tag = np.zeros(n, np.uint8); h = np.zeros(n, np.int16); sc = np.zeros(n, np.int16)
    hs = b.hist_stride
    full = np.zeros((g, hs), np.int32); pref = np.zeros((g, hs), np.int32); rept = np.zeros((g, hs), np.int32)
    calls = np.zeros(g, _lib.CALL_DTYPE)
    ctx.genotype_batch(_lib.MEM_HOST, b.packed, b.read_off, b.read_len, n, b.unit_read_off, b.unit_ladder, b.units,
                       g, _lib.default_sw_params(max_read_len=150), None, b.global_lens, len(b.global_lens),
                       b.target_lens, len(b.target_lens), tag, h, sc, hs, full, pref, rept, calls)
    reads = [synth.decode(r) for r in b.codes]
    ls = po.LocusSet(b.ladders)
    read_locus = np.repeat(b.unit_ladder, np.diff(b.unit_read_off))
    cls = po.classify(reads, read_locus, ls, threads=8)
    assert np.array_equal(tag, cls[:, 0].astype(np.uint8))
    assert np.array_equal(h, cls[:, 1].astype(np.int16))
    n_called = 0
    for u in range(g):
        r0, r1 = b.unit_read_off[u], b.unit_read_off[u + 1]
        f, pp, rr = {}, {}, 0
        for t, hh, _ in cls[r0:r1]:
            if t == 1: f[int(hh)] = f.get(int(hh), 0) + 1
            elif t in (2, 3): pp[int(hh)] = pp.get(int(hh), 0) + 1
            elif t == 4: rr += 1
        assert {k: v for k, v in enumerate(full[u]) if v} == f
        assert {k: v for k, v in enumerate(pref[u]) if v} == pp
        assert rept[u].sum() == rr
        up = b.units[u]
        locus = sel[b.unit_ladder[u]]
        caller = lo.Caller(int(up["period"]), int(up["readlen"]), int(up["ploidy"]), 2 * float(up["half_depth"]), f, pp,
                           rr, b.global_lens[up["pe_off"]:up["pe_off"] + up["n_global"]],
                           b.target_lens[up["tl_off"]:up["tl_off"] + up["n_target"]], int(up["ref_len"]),
                           int(up["minpe"]), maxinsert=int(up["maxinsert"]))
        res = caller.evaluate()
        if res["status"] == 1:
            assert calls[u]["status"] == 1
            continue
        assert calls[u]["status"] == 0
        assert (calls[u]["h1"], calls[u]["h2"]) == tuple(res["alleles"]), (u, locus["name"])
        assert abs(calls[u]["lik"] - res["lik"]) <= ML_TOL
        assert tuple(calls[u]["ci"]) == tuple(res["CI"])
        ppv = lo.calc_PP(res["tot"], res["lik"], int(up["period"]), locus["cutoff_risk"],
                         locus["mutation_nature"] == "increase", locus["inheritance"][-1] == "R")
        assert abs(calls[u]["pp"] - ppv) <= 1e-9
        assert calls[u]["n_pairs"] == len(res["mls"])
        n_called += 1
    assert n_called >= g - 2


def test_wide_grids_beyond_512_columns(ctx):
    """--maxinsert above ~510 makes the extended axis wider than 512 entries: several column blocks per row
    group in the pairs kernel, a second column sweep in the reduce kernel, axis values beyond the 1000-bin
    pdfs (numpy's empty-slice semantics in pdf_spanning, models.py:160-166).  Checked against the numpy oracle."""
    _set_model(ctx)
    by_name = {c["name"]: c for c in CASES}
    cases = []
    for name, maxinsert in (("hd_expanded_rept_pe", 700), ("hd_100x", 640), ("hd_no_full", 560)):
        c = dict(by_name[name])
        c["maxinsert"] = maxinsert
        cases.append(c)
    hs = 128
    units, full, pref, rept, gl, tl = _case_inputs(cases, hs)
    n = len(cases)
    want = [oracle_caller(c).evaluate() for c in cases]
    goff = np.zeros(n + 1, np.int64)
    goff[1:] = np.cumsum([len(w["mls"]) for w in want])
    dump = np.zeros((int(goff[-1]), 6), np.float64)
    ms = 1024
    marg = np.zeros((n, 2, ms), np.float64)
    calls = np.zeros(n, _lib.CALL_DTYPE)
    ctx.likelihood_grid(_lib.MEM_HOST, units, n, hs, full, pref, rept, gl, len(gl), tl, len(tl), calls, goff, dump,
                        marg, ms)
    for i, (c, w) in enumerate(zip(cases, want)):
        call = calls[i]
        assert call["status"] == 0 and call["n_pairs"] == len(w["mls"]), (c["name"], call)
        got, exp = dump[goff[i]:goff[i + 1]], np.asarray(w["mls"], np.float64)
        assert np.array_equal(got[:, :2], exp[:, :2]), c["name"]
        assert np.abs(got[:, 2:] - exp[:, 2:]).max() <= ML_TOL, c["name"]
        assert (call["h1"], call["h2"]) == tuple(w["alleles"]), c["name"]
        assert abs(call["lik"] - w["lik"]) <= ML_TOL and tuple(call["ci"]) == tuple(w["CI"]), c["name"]
        l = c["locus_rec"]
        pp = lo.calc_PP(w["tot"], w["lik"], len(l["repeat"]), l["cutoff_risk"], l["mutation_nature"] == "increase",
                        l["inheritance"][-1] == "R")
        assert abs(call["pp"] - pp) <= 1e-9, c["name"]
        for which, name in enumerate(("P_h1", "P_h2")):     # un-normalised marginals, keyed by bp in the oracle
            m = marg[i, which]
            for h, v in w[name].items():
                assert abs(m[h // len(l["repeat"])] - v) <= 1e-9 * max(1.0, v), (c["name"], name, h)
    assert max(len(w["mls"]) for w in want) > 100000


def test_many_spanning_pairs_and_haploid_paired_end(ctx):
    """The paired-end term with more spanning pairs than one 32-wide register pass holds (40, 150: two and five
    passes over the rows, partial sums parked in the ml buffer) and for one allele (h2 = h1: both factors of a
    pair come from the row's own table entries), against the numpy oracle."""
    _set_model(ctx)
    base = next(c for c in CASES if c["name"] == "hd_expanded_rept_pe")
    rng = np.random.default_rng(5)
    cases = []
    for n_target, ploidy in ((40, 2), (150, 2), (33, 1), (150, 1)):
        c = dict(base)
        c["target_lens"] = [int(x) for x in rng.choice(base["target_lens"], n_target) + rng.integers(-40, 40, n_target)]
        c["ploidy"] = ploidy
        c["name"] = "{}_nt{}_p{}".format(base["name"], n_target, ploidy)
        cases.append(c)
    hs = 128
    units, full, pref, rept, gl, tl = _case_inputs(cases, hs)
    n = len(cases)
    want = [oracle_caller(c).evaluate() for c in cases]
    assert all(w["run_pe"] for w in want)
    goff = np.zeros(n + 1, np.int64)
    goff[1:] = np.cumsum([len(w["mls"]) for w in want])
    dump = np.zeros((int(goff[-1]), 6), np.float64)
    calls = np.zeros(n, _lib.CALL_DTYPE)
    ctx.likelihood_grid(_lib.MEM_HOST, units, n, hs, full, pref, rept, gl, len(gl), tl, len(tl), calls, goff, dump,
                        None, 0)
    for i, (c, w) in enumerate(zip(cases, want)):
        call = calls[i]
        assert call["status"] == 0 and call["run_pe"] == 1 and call["n_pairs"] == len(w["mls"]), (c["name"], call)
        got, exp = dump[goff[i]:goff[i + 1]], np.asarray(w["mls"], np.float64)
        assert np.array_equal(got[:, :2], exp[:, :2]), c["name"]
        assert np.abs(got[:, 2:] - exp[:, 2:]).max() <= ML_TOL, c["name"]
        assert (call["h1"], call["h2"]) == tuple(w["alleles"]) and tuple(call["ci"]) == tuple(w["CI"]), c["name"]
        assert abs(call["lik"] - w["lik"]) <= ML_TOL


def test_sparse_joint_matches_dense_dump(ctx):
    """tredgpu_likelihood_grid_joint: the triples {h1, h2, exp(ml - max)} >= e^-10 and the normaliser against the
    reference's P_h1h2 of the golden cases (and against the dense dump route), incl. a capacity that is too small."""
    _set_model(ctx)
    cases = [c for c in CASES if not c["expected"]["raised"] and c["expected"]["alleles"] != [-1, -1]]
    hs = 128
    units, full, pref, rept, gl, tl = _case_inputs(cases, hs)
    n = len(cases)
    for cap_each in (4096, 2):
        joff = np.arange(n + 1, dtype=np.int64) * cap_each
        trip = np.zeros((int(joff[-1]), 3), np.float64)
        jn = np.zeros(n, np.int32); jt = np.zeros(n, np.float64)
        calls = np.zeros(n, _lib.CALL_DTYPE)
        ctx.likelihood_grid_joint(_lib.MEM_HOST, units, n, hs, full, pref, rept, gl, len(gl), tl, len(tl), calls, None, 0,
                                  joff, trip, jn, jt)
        assert (calls["status"] == 0).all()
        for i, c in enumerate(cases):
            want = c["expected"]["P_h1h2"]
            period = len(c["locus_rec"]["repeat"])
            assert jn[i] == len(want), (c["name"], jn[i], len(want))          # count is reported even when truncated
            got = trip[joff[i]:joff[i] + min(jn[i], cap_each)]
            assert len(set((int(a), int(b)) for a, b, _ in got)) == len(got)  # distinct pairs
            for h1, h2, v in got:
                key = "{},{}".format(int(h1) // period, int(h2) // period)
                assert key in want and abs(v / jt[i] - want[key]) <= 1e-9, (c["name"], key)
            if cap_each >= jn[i]:
                assert len(got) == len(want)


def test_exact_tie_between_a_near_and_a_far_column(ctx):
    """Found by tools/fuzz_hist.py (seed 20261018, case 469): 300 repeat-only reads push the Poisson term of every
    long-allele pair onto its floor (log e^-100), the spanning / partial terms of a near column have already stopped
    depending on h2, so two pairs of one row have EXACTLY the same likelihood in the reference, which then keeps the
    first in enumeration order.  The near and the far table must therefore add their terms up in the same order (one
    lane-group size for both in grid_prepare_kernel): a last-bit difference picked (90, 111) instead of (90, 102)."""
    _set_model(ctx)
    locus = next(l for l in synth.load_loci() if l["name"] == "CCD")
    chrom, span = locus["repeat_location"].split(":")
    start, end = (int(x) for x in span.split("-"))
    c = {"locus_rec": locus, "readlen": 100, "ploidy": 2, "depth": 12.0, "maxinsert": 300, "fullsearch": False,
         "full": {24: 15, 30: 17},
         "partial": {30: 3, 11: 5, 29: 7, 20: 6, 3: 7, 34: 1, 9: 1, 27: 5, 14: 7, 33: 3, 21: 4, 22: 7, 26: 3, 25: 7, 28: 3,
                     32: 4, 1: 4, 12: 7, 6: 7, 5: 3},
         "rept": 300, "global_lens": [], "target_lens": [], "ref_len": end - start + 1, "minpe": end - start + 20}
    w = lo.Caller(len(locus["repeat"]), c["readlen"], c["ploidy"], c["depth"], c["full"], c["partial"], c["rept"],
                  c["global_lens"], c["target_lens"], c["ref_len"], c["minpe"], maxinsert=c["maxinsert"],
                  fullsearch=c["fullsearch"]).evaluate()
    mls = np.asarray(w["mls"], np.float64)
    tot = mls[:, 2:].sum(1)
    assert (tot == tot.max()).sum() >= 2, "the reference no longer has a tie here"
    units, full, pref, rept, gl, tl = _case_inputs([c], 128)
    calls = np.zeros(1, _lib.CALL_DTYPE)
    goff = np.array([0, len(mls)], np.int64)
    dump = np.zeros((len(mls), 6), np.float64)
    ctx.likelihood_grid(_lib.MEM_HOST, units, 1, 128, full, pref, rept, gl, len(gl), tl, len(tl), calls, goff, dump, None, 0)
    assert calls[0]["status"] == 0 and (int(calls[0]["h1"]), int(calls[0]["h2"])) == tuple(w["alleles"]) == (90, 102)
    got = dump[:, 2:].sum(1)
    assert (got == got.max()).sum() == (tot == tot.max()).sum()          # the tie is a tie on the device too


def _run_grid(ctx, cases, hs=128, ms=512):
    units, full, pref, rept, gl, tl = _case_inputs(cases, hs)
    n = len(cases)
    marg = np.zeros((n, 2, ms), np.float64)
    calls = np.zeros(n, _lib.CALL_DTYPE)
    ctx.likelihood_grid(_lib.MEM_HOST, units, n, hs, full, pref, rept, gl, len(gl), tl, len(tl), calls, None, None, marg, ms)
    return calls, marg


def test_batch_of_many_units_equals_units_one_by_one(ctx):
    """A unit's result does not depend on its batch: 120 units in shuffled order (the kernels hand units out through 16
    ticket queues, carve their tables from 16 sub-pools and build the paired-end KDEs in a kernel of their own) against
    the same cases run alone, bit for bit."""
    _set_model(ctx)
    alone = [_run_grid(ctx, [c]) for c in CASES]
    order = np.random.default_rng(5).permutation(np.repeat(np.arange(len(CASES)), 5))
    calls, marg = _run_grid(ctx, [CASES[i] for i in order])
    for k, i in enumerate(order):
        assert calls[k].tobytes() == alone[i][0][0].tobytes(), (k, CASES[i]["name"])
        assert np.array_equal(marg[k], alone[i][1][0]), (k, CASES[i]["name"])


def test_singular_pair_length_model_with_and_without_the_paired_end_term(ctx):
    """The reference builds the KDE whenever a unit has its paired-end model (models.py:131-132) and fails on a singular
    one whether or not the term is used afterwards.  With constant pair lengths scipy decides what "singular" means
    (test_kde_of_equal_pair_lengths_follows_scipy): for a value it refuses the unit gets status -2 in both kinds of unit,
    for a value it lets through the unit is called exactly as the oracle -- which asks scipy -- calls it; a single
    different length lifts the singularity."""
    import warnings
    from scipy.stats import gaussian_kde
    _set_model(ctx)
    with_pe = [c for c in CASES if c["kde"] is not None and not c["expected"]["raised"]]
    assert with_pe
    used = with_pe[0]
    unused = dict(used, partial={}, name=used["name"] + "/no PREF reads")   # no partial read: the term is not used
    n = len(used["global_lens"])
    refused = passed = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for c in range(300, 400):
            try:
                gaussian_kde([c] * n)
                passed = c if passed is None else passed
            except np.linalg.LinAlgError:
                refused = c if refused is None else refused
    assert refused is not None and passed is not None
    for base in (used, unused):
        flat_bad = dict(base, global_lens=[refused] * n)
        flat_ok = dict(base, global_lens=[passed] * n)
        bump = dict(flat_bad, global_lens=[refused] * (n - 1) + [refused + 1])
        calls, _ = _run_grid(ctx, [flat_bad, bump, base, flat_ok])
        assert calls[0]["status"] == -2, base["name"]
        assert calls[1]["status"] in (0, 1), base["name"]
        assert calls[2]["status"] in (0, 1), base["name"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = oracle_caller(flat_ok).evaluate()
        period = len(base["locus_rec"]["repeat"])
        assert calls[3]["status"] == want["status"], base["name"]
        if want["status"] == 0:
            assert (calls[3]["h1"], calls[3]["h2"]) == tuple(want["alleles"]) and abs(calls[3]["lik"] - want["lik"]) <= ML_TOL, base["name"]


def test_grid_kernels_are_timed_one_by_one(ctx):
    _set_model(ctx)
    ctx.reset_timing()
    _run_grid(ctx, CASES)
    total = ctx.get_timing(_lib.KERNEL_GRID)
    parts = [ctx.get_timing(k) for k in (_lib.KERNEL_GRID_KDE, _lib.KERNEL_GRID_PREPARE, _lib.KERNEL_GRID_PAIRS, _lib.KERNEL_GRID_REDUCE)]
    assert total[0] == 1 and all(p[0] == 1 for p in parts)
    assert all(p[1] > 0 for p in parts) and sum(p[1] for p in parts) <= total[1] * 1.01 + 0.05
    with pytest.raises(_lib.TredGpuError):
        ctx.get_timing(7)
