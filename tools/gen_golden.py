#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the REFERENCE ITSELF.

BUILD-CONTAINER ONLY (reads /root/reference through tools/refshim.py and oracle/_ref/libssw.so).
The outputs are plain data (inputs + expected outputs); no reference source is stored.

  sw_pairs.npz     (read, template) -> (score, ref_begin, ref_end, read_begin, read_end) from the
                   compiled reference ssw.c driven exactly as ssw_wrap.Aligner.align does
  classify.json    reads -> (tag, h) from the reference's BamParser._parseReadSW + tally_counts
  grid.npz/json    IntegratedCaller inputs -> ordered [(h1,h2,ml1..ml4)], alleles, lik, PP, CI,
                   P_h1/P_h2/P_h1h2, label from the reference's models.py
  kde.npz          global_lens -> PEMaxLikModel.pdf
"""
import json
import logging
import os
import sys
import types
from collections import defaultdict

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import refshim  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tredparse_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def locus_by_name(loci, name):
    return [l for l in loci if l["name"] == name][0]


def make_reads(rng, loci, names, readlen, n_units, **kw):
    """[(locus, reads[str])] drawn from the synthetic generator (+ a few adversarial reads)."""
    out = []
    p = synth.SynthParams(readlen=readlen, **kw)
    for nm in names:
        locus = locus_by_name(loci, nm)
        lb = synth.simulate_locus(rng, locus, n_units, p)
        reads = [synth.decode(r) for r in lb.reads]
        reads.append("N" * readlen)
        reads.append(synth.decode(rng.integers(0, 4, readlen).astype(np.uint8)))
        rep = locus["repeat"].replace("N", "C")
        reads.append((rep * (readlen // len(rep) + 1))[:readlen])          # pure repeat
        reads.append(po.rc((rep * (readlen // len(rep) + 1))[:readlen]))    # pure repeat, other strand
        out.append((locus, reads))
    return out


def gen_sw(loci):
    rng = np.random.default_rng(20260101)
    reads_all, refs_all, pr, pt, meta = [], [], [], [], []
    every = [l["name"] for l in loci]
    plan = [(150, ["HD", "DM1", "SCA10", "ULD", "OPMD", "BPES", "ALS", "DM2", "FRDA", "SCA8"], 2, 10, 40),
            (100, ["HD", "SCA36", "CCD"], 2, 8, 40), (250, ["HD", "DM2"], 1, 8, 40), (36, ["SCA3"], 1, 20, 40),
            (150, every, 1, 10, 5)]      # every locus of the table, a few reads each
    for readlen, names, n_units, cov, n_keep in plan:
        for locus, reads in make_reads(rng, loci, names, readlen, n_units, coverage=cov, sub=0.02,
                                       indel=0.004, nrate=0.01, min_units=1,
                                       max_units=max(4, readlen // len("CAG") + 10)):
            mu = -(-readlen // len(locus["repeat"]))
            if 36 + len(locus["repeat"]) * mu > 511:
                continue
            refs = [t for _, t in po.build_ladder(locus["prefix"], locus["repeat"], locus["suffix"], mu)]
            # subsample reads so the file stays small
            keep = rng.permutation(len(reads))[:n_keep]
            r0, t0 = len(reads_all), len(refs_all)
            reads_all += [reads[i] for i in keep]
            refs_all += refs
            for i in range(len(keep)):
                for j in range(len(refs)):
                    pr.append(r0 + i)
                    pt.append(t0 + j)
            meta.append((locus["name"], readlen, len(keep), len(refs)))
    res = po.ref_sw_pairs(reads_all, refs_all, pr, pt, threads=8)
    np.savez_compressed(os.path.join(GOLD, "sw_pairs.npz"), reads=np.array(reads_all), refs=np.array(refs_all),
                        pair_read=np.asarray(pr, np.int32), pair_ref=np.asarray(pt, np.int32),
                        result=res.astype(np.int16), scoring=np.asarray([1, 5, 7, 2], np.int32))
    print("sw_pairs:", len(pr), "pairs", meta)


class FakeRead:
    def __init__(self, name, seq):
        self.query_name, self.query_sequence = name, seq


def fake_input_params(ref, locus, readlen, clip=False, depth=30.0, ploidy_gender="Unknown"):
    tred = types.SimpleNamespace(
        repeat=locus["repeat"], chr="chrT", repeat_start=10000, alt=[],
        repeat_end=10000 + int(locus["repeat_location"].split(":")[1].split("-")[1]) -
        int(locus["repeat_location"].split(":")[1].split("-")[0]),
        prefix=locus["prefix"], suffix=locus["suffix"], is_xlinked=locus["inheritance"][0] == "X", ploidy=2,
        inheritance=locus["inheritance"], cutoff_risk=locus["cutoff_risk"], cutoff_prerisk=locus["cutoff_prerisk"],
        is_recessive=locus["inheritance"][-1] == "R", is_expansion=locus["mutation_nature"] == "increase",
        name=locus["name"])
    ip = types.SimpleNamespace(bam="none.bam", gender=ploidy_gender, depth=depth, READLEN=readlen, clip=clip,
                               alts=False, repeatpairs=True, ref="hg38", tred=tred,
                               getLogLevel=lambda *a: logging.INFO, kwargs={})
    return ip


def gen_classify(ref, loci):
    rng = np.random.default_rng(20260102)
    cases = []
    plan = [(150, ["HD", "DM1", "SCA10", "ULD", "OPMD", "XLMR"], False), (100, ["SCA2", "DM2"], False),
            (150, ["HD"], True)]
    for readlen, names, clip in plan:
        for locus, reads in make_reads(rng, loci, names, readlen, 1, coverage=25, min_units=3,
                                       max_units=readlen // 3 + 20):
            if clip:  # ragged lengths exercise the per-read REPT cut-off
                reads = [r[int(rng.integers(0, 40)):] for r in reads]
            bp = ref.bam_parser.BamParser(fake_input_params(ref, locus, readlen, clip=clip))
            db = bp._buildDB()
            per_read = []
            for i, seq in enumerate(reads):
                n_before = len(bp.details)
                hang_before = dict(bp.counts["HANG"])
                bp._parseReadSW("chrT", FakeRead("r{}".format(i), seq), db)
                tag, h = "", 0
                if len(bp.details) > n_before:
                    tag, h = bp.details[-1]["tag"], bp.details[-1]["h"]
                else:
                    for k, v in bp.counts["HANG"].items():
                        if v != hang_before.get(k, 0):
                            tag, h = "HANG", k
                per_read.append([tag, int(h)])
            bp.tally_counts()
            rept = sum(bp.counts["REPT"].values()) if bp.counts["REPT"] else 0
            cases.append({"locus": locus["name"], "readlen": readlen, "clip": clip, "max_units": bp.max_units,
                          "reads": reads, "expected": per_read,
                          "FULL": {str(k): v for k, v in sorted(bp.counts["FULL"].items())},
                          "PREF": {str(k): v for k, v in sorted(bp.counts["PREF"].items())},
                          "REPT": {str(k): v for k, v in sorted(bp.counts["REPT"].items())}, "rept": rept})
            print("classify:", locus["name"], readlen, clip, len(reads), "reads",
                  sum(1 for t, _ in per_read if t), "tagged")
    with open(os.path.join(GOLD, "classify.json"), "w") as fp:
        json.dump({"generator": "tools/gen_golden.py (reference bam_parser._parseReadSW)", "cases": cases}, fp)


def ref_caller(ref, locus, case):
    """Drive the reference's IntegratedCaller on plain inputs."""
    ip = fake_input_params(ref, locus, case["readlen"], depth=case["depth"])
    counts = {}
    counts["PREF"] = counts["POST"] = defaultdict(int)
    for tag in ("FULL", "REPT", "HANG"):
        counts[tag] = defaultdict(int)
    for k, v in sorted(case["full"].items(), key=lambda kv: int(kv[0])):
        counts["FULL"][int(k)] = v
    for k, v in sorted(case["partial"].items(), key=lambda kv: int(kv[0])):
        counts["PREF"][int(k)] = v
    bp = types.SimpleNamespace(tred=ip.tred, READLEN=case["readlen"], repeatSize=len(locus["repeat"]),
                               counts=counts, rept=case["rept"], ploidy=case["ploidy"], depth=case["depth"],
                               inputParams=ip)
    pe = types.SimpleNamespace(global_lens=list(case["global_lens"]), target_lens=list(case["target_lens"]),
                               ref=case["ref_len"], MINPE=case["minpe"])
    ref.models.PEextractor = lambda _bp: pe
    caller = ref.models.IntegratedCaller(bp, maxinsert=case["maxinsert"], fullsearch=case["fullsearch"])
    # capture the per-pair terms (the reference only logs them): wrap the four evaluators
    rec = []
    ev = {n: getattr(caller, n) for n in ("evaluate_spanning", "evaluate_partial", "evaluate_rept")}
    return caller, rec, ev


def gen_grid(ref, loci):
    rng = np.random.default_rng(20260103)
    cases = []

    def add(name, locus_name, readlen=150, ploidy=2, maxinsert=100, fullsearch=False, h=None, coverage=30,
            full=None, partial=None, rept=None, n_global=None, tweak=None, **kw):
        locus = locus_by_name(loci, locus_name)
        p = synth.SynthParams(coverage=coverage, readlen=readlen, **kw)
        lb = synth.simulate_locus(rng, locus, 1, p, h_pairs=[h] if h else None)
        reads = [synth.decode(r) for r in lb.reads]
        mu = -(-readlen // len(locus["repeat"]))
        ls = po.LocusSet([(locus["prefix"], locus["repeat"], locus["suffix"], mu)])
        cls = po.classify(reads, np.zeros(len(reads), np.int32), ls, threads=8)
        f, pp, r = defaultdict(int), defaultdict(int), 0
        for t, hh, _ in cls:
            if t == 1: f[int(hh)] += 1
            elif t in (2, 3): pp[int(hh)] += 1
            elif t == 4: r += 1
        span = locus["repeat_location"].split(":")[1].split("-")
        ref_len = int(span[1]) - int(span[0]) + 1
        gl = lb.global_lens if n_global is None else lb.global_lens[:n_global]
        case = {"name": name, "locus": locus_name, "readlen": readlen, "ploidy": ploidy, "maxinsert": maxinsert,
                "fullsearch": fullsearch, "depth": float(lb.depth[0]),
                "full": {str(k): v for k, v in sorted((full if full is not None else f).items())},
                "partial": {str(k): v for k, v in sorted((partial if partial is not None else pp).items())},
                "rept": r if rept is None else rept, "global_lens": [int(x) for x in gl],
                "target_lens": [int(x) for x in lb.target_lens], "ref_len": ref_len, "minpe": ref_len - 1 + 20,
                "h_true": [int(x) for x in lb.h_true[0]]}
        if tweak:
            tweak(case)
        cases.append(case)

    add("hd_typical", "HD", h=[15, 41], maxinsert=300)
    add("hd_close", "HD", h=[17, 19])
    add("hd_homo", "HD", h=[30, 30])
    add("hd_expanded_rept_pe", "HD", h=[20, 90], maxinsert=300)
    add("dm1_expanded_big", "DM1", h=[5, 200], maxinsert=300, coverage=40)
    add("hd_haploid", "HD", h=[22, 22], ploidy=1)
    add("hd_fullsearch", "HD", h=[15, 41], fullsearch=True, maxinsert=60)
    add("hd_no_full", "HD", h=[70, 80], maxinsert=120)
    add("hd_no_pe_model", "HD", h=[20, 90], n_global=50)
    add("ar_decrease", "AR", h=[6, 21])
    add("frda_recessive", "FRDA", h=[70, 80], maxinsert=120)
    add("uld_period12", "ULD", h=[2, 3])
    add("uld_period12_big", "ULD", h=[3, 40], maxinsert=80)
    add("sca10_period5", "SCA10", h=[12, 14])
    add("sca36_period6_rl100", "SCA36", readlen=100, h=[5, 9])
    add("dm2_period4_rl250", "DM2", readlen=250, h=[20, 75], maxinsert=120)
    add("opmd_nmotif", "OPMD", h=[10, 13])
    add("hd_dup_axis", "HD", h=[20, 90], full={"15": 3, "41": 2}, partial={"20": 2, "30": 1}, rept=3)
    add("hd_partial_only", "HD", full={}, partial={"12": 2, "33": 1}, rept=0, h=[15, 41])
    add("hd_rept_only_extended", "HD", full={}, partial={"45": 3, "47": 2}, rept=4, h=[20, 90], maxinsert=150)
    add("hd_empty", "HD", full={}, partial={}, rept=0, h=[15, 41])
    add("hd_low_cov", "HD", h=[15, 41], coverage=6)
    add("hd_100x", "HD", h=[18, 150], coverage=100, maxinsert=200)

    def singular(case):
        case["global_lens"] = [350] * 150
    add("hd_singular_kde", "HD", h=[20, 90], tweak=singular)

    out_arrays, out_cases = {}, []
    for ci, case in enumerate(cases):
        locus = locus_by_name(loci, case["locus"])
        exp = {}
        try:
            caller, rec, ev = ref_caller(ref, locus, case)
            terms = []
            orig_sp, orig_pa, orig_re = caller.evaluate_spanning, caller.evaluate_partial, caller.evaluate_rept
            pem = caller.pemodel
            state = {}

            def sp(obs, h1, h2): state["ml1"] = orig_sp(obs, h1, h2); return state["ml1"]
            def pa(obs, h1, h2): state["ml2"] = orig_pa(obs, h1, h2); return state["ml2"]

            def re_(n, h1, h2):
                state["ml3"] = orig_re(n, h1, h2)
                terms.append([h1, h2, state.get("ml1", 0), state.get("ml2", 0), state["ml3"], 0.0])
                state.pop("ml1", None); state.pop("ml2", None)
                return state["ml3"]
            caller.evaluate_spanning, caller.evaluate_partial, caller.evaluate_rept = sp, pa, re_
            if pem is not None:
                orig_pe = pem.evaluate

                def pev(h1, h2):
                    v = orig_pe(h1, h2)
                    terms[-1][5] = v
                    return v
                pem.evaluate = pev
            caller.call()
            exp["raised"] = ""
            exp["alleles"] = [int(x) for x in caller.alleles]
            exp["label"] = caller.label
            exp["CI"] = caller.CI
            exp["PP"] = float(caller.PP)
            exp["P_h1"] = caller.P_h1 if caller.P_h1 else {}
            exp["P_h2"] = caller.P_h2 if caller.P_h2 else {}
            exp["P_h1h2"] = caller.P_h1h2 if caller.P_h1h2 else {}
            exp["PEDP"], exp["PEG"], exp["PET"] = caller.PEDP, caller.PEG, caller.PET
            exp["P_PEG"], exp["P_PET"] = caller.P_PEG, caller.P_PET
            exp["n_pairs"] = len(terms)
            exp["pe_model"] = pem is not None
            out_arrays["mls_{}".format(ci)] = np.asarray(terms, np.float64).reshape(-1, 6)
            if pem is not None:
                out_arrays["kde_{}".format(ci)] = np.asarray(pem.pdf, np.float64)
        except Exception as e:  # the reference drops such a locus (tred.py:245-249)
            exp = {"raised": type(e).__name__}
        case["expected"] = exp
        out_cases.append(case)
        print("grid:", case["name"], exp.get("alleles"), exp.get("CI"), exp.get("PP"), exp.get("label"),
              exp.get("n_pairs"), exp.get("raised"))
    np.savez_compressed(os.path.join(GOLD, "grid.npz"), **out_arrays)
    with open(os.path.join(GOLD, "grid.json"), "w") as fp:
        json.dump({"generator": "tools/gen_golden.py (reference models.IntegratedCaller.call)",
                   "cases": out_cases}, fp)


class _Col(object):
    __slots__ = ("n",)

    def __init__(self, n):
        self.n = n


class PysamStandin(types.ModuleType):
    """pysam.AlignmentFile on top of tredparse_amd.bamio (pysam is not installable here)."""

    def __init__(self):
        types.ModuleType.__init__(self, "pysam")
        from tredparse_amd import bamio

        class AlignmentFile(bamio.PyAlignmentFile):
            def pileup(self, chrom, start, end):
                # htslib default pileup: every column covered by a read overlapping the region
                cov = {}
                for r in self.fetch(chrom, start, end):
                    if r.flag & (bamio.FUNMAP | bamio.FSECONDARY | bamio.FQCFAIL | bamio.FDUP):
                        continue
                    if r.reference_end is None:
                        continue
                    for p in range(r.pos, r.reference_end):
                        cov[p] = cov.get(p, 0) + 1
                for p in sorted(cov):
                    yield _Col(cov[p])
        self.AlignmentFile = AlignmentFile


def gen_e2e(loci):
    """The reference's own run() on its two test BAMs (tests/samples.csv), all 32 loci each."""
    import tempfile
    ref = refshim.load_reference(pysam_standin=PysamStandin(), full=True)
    logging.disable(logging.CRITICAL)
    out = {}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)      # a cwd without sites/ 
    try:
        repo = ref.meta.TREDsRepo(ref="hg38", toy=False, sites=os.path.join(tmp, "sites"))
        for samplekey, bam, treds in (("t001", "t001.bam", None), ("t002", "t002.bam", None)):
            bampath = os.path.join(refshim.REF, "tests", bam)
            names = treds or list(repo.names)
            res = ref.tred.run((samplekey, bampath, repo, names, 300, False, False, True, True, "INFO"))
            calls = res["tredCalls"]
            for k, v in list(calls.items()):
                if isinstance(v, (np.floating, np.integer)):
                    calls[k] = v.item()
            out[samplekey] = calls
            print("e2e:", samplekey, {k: calls[k] for k in calls if k.endswith((".1", ".2", ".label"))
                                      and calls[k] not in (-1, "missing")})
    finally:
        os.chdir(cwd)
    with open(os.path.join(GOLD, "run_t001_t002.json"), "w") as fp:
        json.dump({"generator": "tools/gen_golden.py: the reference's tredparse.tred.run() (v0.7.8, via "
                                "tools/refshim.py) on its own tests/t001.bam and tests/t002.bam, all 32 loci, "
                                "default flags; pysam replaced by tredparse_amd.bamio", "samples": out}, fp)


FLAG_LOCI = ("HD", "DM1", "SCA17", "AR", "FXS")        # two covered loci + an autosomal and two X-linked empty ones
SYNF_LOCI = ("HD", "DM1", "FXS", "SCA10")             # the loci tests/golden/bam/synf.bam carries reads for
SYNF_ALLELES = {"HD": (17, 95), "DM1": (5, 48), "SCA10": (12, 14), "FXS": (30, 120)}
# (name, sample, haploid contigs, maxinsert, fullsearch, clip, alts, repeatpairs): the non-default corners of
# tred.py:75-93 as they reach run() (tred.py:508-512) and repo.set_ploidy (tred.py:492).  The reference's two
# mini-BAMs hold no ALT-region reads, no pairs of repeat-only reads and only 150 bp reads, so three of the flags
# change nothing there; synf.bam (make_synf: expanded alleles, mates mismapped into ALT regions, unmapped mates
# inside the tract) is where they bite.
FLAG_CASES = [
    ("default", "synf", None, 300, False, False, True, True),
    ("clip", "synf", None, 300, False, True, True, True),
    ("noalts", "synf", None, 300, False, False, False, True),
    ("norepeatpairs", "synf", None, 300, False, False, True, False),
    ("noalts_norepeatpairs", "synf", None, 300, False, False, False, False),
    ("haploid", "synf", ["chrX", "chr19"], 300, False, False, True, True),
    ("fullsearch_maxinsert60", "synf", None, 60, True, False, True, True),
    ("maxinsert100", "synf", None, 100, False, False, True, True),
    ("clip_noalts_norepeatpairs", "t002", None, 300, False, True, False, False),
    ("haploid", "t001", ["chr4", "chrX"], 300, False, False, True, True),
    ("haploid", "t002", ["chr19"], 300, False, False, True, True),
    ("fullsearch_maxinsert60", "t001", None, 60, True, False, True, True),
    ("fullsearch_maxinsert60", "t002", None, 60, True, False, True, True),
    ("maxinsert100", "t002", None, 100, False, False, True, True),
    ("haploid_fullsearch", "t001", ["chr4"], 80, True, False, True, True),
]


def make_synf():
    """tests/golden/bam/synf.bam(.bai.gz): one synthetic 24x sample over SYNF_LOCI (tredparse_amd.synth_bam, fixed
    seed) -- test DATA for the flag goldens; the index is stored gzipped (it is 0.5 MB of mostly empty bins)."""
    import gzip
    import shutil
    from tredparse_amd import synth_bam
    chosen = [l for l in synth.load_loci() if l["name"] in SYNF_LOCI]
    recs, _ = synth_bam.simulate_sample(20260301, chosen, synth.SynthParams(coverage=24),
                                        h_pairs=[SYNF_ALLELES[l["name"]] for l in chosen], alt_rate=0.4)
    # every third tract-internal (unmapped, mate-anchored) read gets a secondary copy right behind it: two records
    # of one name that both come out repeat-only -- what --norepeatpairs removes (bam_parser.py:270-287)
    un = np.nonzero((recs.flag & synth_bam.FUNMAP) != 0)[0][::3]
    order = np.sort(np.concatenate([np.arange(len(recs)), un]), kind="stable")
    second = np.zeros(len(order), bool)
    second[1:] = order[1:] == order[:-1]
    recs = recs.take(order)
    recs.flag = np.where(second, recs.flag | synth_bam.FSEC, recs.flag).astype(recs.flag.dtype)
    path = os.path.join(GOLD, "bam", "synf.bam")
    for f in (path, path + ".bai.gz"):
        if os.path.exists(f):
            os.chmod(f, 0o644)
    synth_bam.write_bam(path, recs, sample="synf", level=9)
    with open(path + ".bai", "rb") as src, gzip.GzipFile(path + ".bai.gz", "wb", mtime=0) as dst:
        shutil.copyfileobj(src, dst)
    os.unlink(path + ".bai")
    print("synf:", len(recs), "records,", os.path.getsize(path), "+", os.path.getsize(path + ".bai.gz"), "bytes")


def _bam_of(sample, tmp):
    """Path of a flag-case BAM; synf is unpacked (with its index) into `tmp`."""
    import gzip
    import shutil
    if sample != "synf":
        return os.path.join(refshim.REF, "tests", sample + ".bam"), list(FLAG_LOCI)
    dst = os.path.join(tmp, "synf.bam")
    if not os.path.exists(dst):
        shutil.copy(os.path.join(GOLD, "bam", "synf.bam"), dst)
        with gzip.open(os.path.join(GOLD, "bam", "synf.bam.bai.gz"), "rb") as src, open(dst + ".bai", "wb") as out:
            shutil.copyfileobj(src, out)
    return dst, list(SYNF_LOCI)


def _plain(calls):
    for k, v in list(calls.items()):
        if isinstance(v, (np.floating, np.integer)):
            calls[k] = v.item()
    return calls


def _load_full_reference():
    ref = refshim.load_reference(pysam_standin=PysamStandin(), full=True)
    logging.disable(logging.CRITICAL)
    return ref


def gen_flags(loci):
    """The reference's run() under every non-default flag combination of FLAG_CASES."""
    import tempfile
    ref = _load_full_reference()
    cases = []
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        for name, sample, haploid, maxinsert, fullsearch, clip, alts, repeatpairs in FLAG_CASES:
            repo = ref.meta.TREDsRepo(ref="hg38", toy=False, sites=os.path.join(tmp, "sites"))
            repo.set_ploidy(haploid)
            bampath, names = _bam_of(sample, tmp)
            res = ref.tred.run((sample, bampath, repo, names, maxinsert, fullsearch, clip, alts,
                                repeatpairs, "INFO"))
            calls = _plain(res["tredCalls"])
            cases.append({"name": name, "sample": sample, "loci": names, "haploid": haploid,
                          "maxinsert": maxinsert, "fullsearch": fullsearch, "clip": clip, "alts": alts,
                          "repeatpairs": repeatpairs, "tredCalls": calls})
            print("flags:", name, sample, {k: calls[k] for k in calls if k.endswith((".1", ".2", ".CI", ".RR"))
                                           and calls[k] not in (-1, "", "missing")})
    finally:
        os.chdir(cwd)
    with open(os.path.join(GOLD, "run_flags.json"), "w") as fp:
        json.dump({"generator": "tools/gen_golden.py flags: the reference's tredparse.tred.run() (v0.7.8 via "
                                "tools/refshim.py) on its tests/t001.bam / t002.bam under non-default options; "
                                "pysam replaced by tredparse_amd.bamio", "cases": cases}, fp)


# Synthetic samples the reference is run on (default flags) -- NOT committed as files: the tests regenerate the identical
# record table from the seed (same numpy, same generator; its digest is checked) and write the BAM themselves.
#   name: (seed, loci or None = all 30 distinct ones, SynthParams keywords, alt_rate)
SYN_SAMPLES = {
    "synall": (20260777, None, dict(coverage=12.0, expanded_max=120, expanded_frac=0.25), 0.3),
    "syn100": (20260778, ["HD", "DM1", "SCA10", "DM2", "ULD", "FXS", "SCA36", "FRDA", "OPMD", "SCA3"],
               dict(coverage=20.0, readlen=100, ins_mean=300.0, ins_sd=50.0, expanded_max=90, expanded_frac=0.3), 0.3),
    "syn250": (20260779, ["HD", "DM1", "SCA10", "DM2", "ULD", "FXS", "SCA36", "FRDA", "OPMD", "SCA3"],
               dict(coverage=10.0, readlen=250, ins_mean=550.0, ins_sd=80.0, max_units=75, expanded_max=150, expanded_frac=0.3), 0.3),
    "syn100x": (20260780, ["HD", "DM1", "SCA1", "FXS"],
                dict(coverage=100.0, min_units=42, max_units=60, expanded_max=200, expanded_frac=0.8), 0.4),
    # whole-genome-shaped (synth_bam wgs_like): background reads over every alternative region's index window and the chrY
    # depth windows -- the reference's mate rescue has to walk through them to find the mismapped mates
    "synwgs": (20260781, ["HD", "DM1", "FXS", "SCA10"], dict(coverage=10.0, expanded_max=120, expanded_frac=0.5, wgs_like=True), 0.4),
}


def syn_sample(name):
    from tredparse_amd import synth_bam
    seed, names, kw, alt_rate = SYN_SAMPLES[name]
    loci = synth_bam.bench_loci() if names is None else [l for l in synth.load_loci() if l["name"] in names]
    kw = dict(kw)
    wgs_like = kw.pop("wgs_like", False)
    recs, h_true = synth_bam.simulate_sample(seed, loci, synth.SynthParams(**kw), alt_rate=alt_rate, wgs_like=wgs_like)
    return loci, recs, h_true


def records_digest(recs):
    """sha256 over the record table (what the BAM says, whatever the compression)."""
    import hashlib
    h = hashlib.sha256()
    for k in recs.FIELDS:
        h.update(np.ascontiguousarray(getattr(recs, k)).tobytes())
    return h.hexdigest()


def gen_synall(loci_unused, only=None):
    """The reference's run() with default flags on synthetic samples with reads at every locus listed: all 30 loci at
    150 bp (BASELINE configs[1] asks for all TRED loci; the reference's two mini-BAMs cover one locus each), ten loci at
    100 bp and at 250 bp (READLEN from the file, other ladder lengths), four loci at 100x with alleles up to 200 repeats
    (configs[4]: large grids, repeat-only reads, paired-end mode)."""
    import tempfile
    from tredparse_amd import synth_bam
    ref = _load_full_reference()
    out = {}
    path = os.path.join(GOLD, "run_synall.json")
    if only is not None and os.path.exists(path):         # (the other samples' records stay as they are)
        with open(path) as fp:
            out = json.load(fp)["samples"]
    for name in (SYN_SAMPLES if only is None else only):
        loci, recs, h_true = syn_sample(name)
        cwd = os.getcwd()
        tmp = tempfile.mkdtemp()
        os.chdir(tmp)
        try:
            bam = os.path.join(tmp, name + ".bam")
            synth_bam.write_bam(bam, recs, sample=name, level=1)
            repo = ref.meta.TREDsRepo(ref="hg38", toy=False, sites=os.path.join(tmp, "sites"))
            names = [l["name"] for l in loci]
            res = ref.tred.run((name, bam, repo, names, 300, False, False, True, True, "INFO"))
            calls = _plain(res["tredCalls"])
            for k in list(calls):
                if k.endswith(".details"):     # (the bases are in the regenerated BAM: keep what the reference decided)
                    calls[k] = [[d["id"], d["tag"], int(d["h"])] for d in calls[k]]
        finally:
            os.chdir(cwd)
        called = sum(1 for n in names if calls.get(n + ".1", -1) > 0)
        print(name + ":", len(recs), "records, readLen", calls.get("readLen"), ",", called, "of", len(names), "loci called;",
              sum(int(calls.get(n + ".1") == h[0]) for n, h in zip(names, h_true.tolist())), "short alleles as simulated;",
              "largest grid entries", max(len(calls.get(n + ".P_h1h2") or {}) for n in names))
        out[name] = {"seed": SYN_SAMPLES[name][0], "records_sha256": records_digest(recs), "loci": names,
                     "h_true": h_true.tolist(), "tredCalls": calls}
    with open(os.path.join(GOLD, "run_synall.json"), "w") as fp:
        json.dump({"generator": "tools/gen_golden.py synall: the reference's tredparse.tred.run() (v0.7.8 via tools/refshim.py) "
                                "on the synthetic samples of SYN_SAMPLES; `details` entries as [id, tag, h]",
                   "samples": out}, fp)


class _TextGzip(object):
    """gzip whose open(path, "w") is text mode: the py2 writers print str into it (tred.py:367-372)."""

    @staticmethod
    def open(path, mode="r", *a, **k):
        import gzip
        return gzip.open(path, mode + "t" if "b" not in mode and "t" not in mode else mode, *a, **k)


def gen_vcf(loci):
    """The reference's to_vcf text (tred.py:316-374) for its own run() results of both test BAMs, all loci, and for
    the flag cases that change a record (haploid)."""
    import gzip
    import tempfile
    ref = _load_full_reference()
    ref.tred.gzip = _TextGzip
    out = {}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        repo = ref.meta.TREDsRepo(ref="hg38", toy=False, sites=os.path.join(tmp, "sites"))
        gold = json.load(open(os.path.join(GOLD, "run_t001_t002.json")))["samples"]
        for sample in ("t001", "t002"):
            results = {"samplekey": sample, "bam": "tests/{}.bam".format(sample), "tredCalls": gold[sample]}
            ref.tred.to_vcf(results, "hg38", repo, treds=list(repo.names))
            text = gzip.open(sample + ".tred.vcf.gz", "rt").read()
            out[sample] = text.splitlines()
        # hg19 coordinates and a single-locus call list (the CSV's third column)
        repo19 = ref.meta.TREDsRepo(ref="hg19_nochr", toy=False, sites=os.path.join(tmp, "sites"))
        results = {"samplekey": "t001_hg19", "bam": "tests/t001.bam", "tredCalls": gold["t001"]}
        ref.tred.to_vcf(results, "hg19_nochr", repo19, treds=["HD", "DM1", "SCA1"])
        out["t001_hg19_nochr_3loci"] = gzip.open("t001_hg19.tred.vcf.gz", "rt").read().splitlines()
    finally:
        os.chdir(cwd)
    with open(os.path.join(GOLD, "vcf_t001_t002.json"), "w") as fp:
        json.dump({"generator": "tools/gen_golden.py vcf: the reference's tredparse.tred.to_vcf (v0.7.8 via "
                                "tools/refshim.py, gzip opened in text mode) on tests/golden/run_t001_t002.json; "
                                "the ##fileDate and ##source lines depend on the day and the install path",
                   "source_module_file": ref.tred.__file__, "vcf": out}, fp, indent=0)
    print("vcf:", {k: len(v) for k, v in out.items()})


def report_inputs():
    """Eight per-sample result files for the reporter: the reference's two run() results (without the bulky
    per-read and distribution entries, which the reporter never reads) and edited copies that reach the corners of
    tredreport.py:36-141 -- a male at an X-linked locus, a pre-risk sample, carriers of both mutation natures,
    a case below --minPP, a long evidence string, the AR exemption of the details file."""
    gold = json.load(open(os.path.join(GOLD, "run_t001_t002.json")))["samples"]
    drop = (".details", ".P_h1", ".P_h2", ".P_h1h2", ".P_PEG", ".P_PET")

    def base(sample):
        return {k: v for k, v in gold[sample].items() if not k.endswith(drop)}

    def called(calls, locus, a, b, label, pp, fr="", pr="", rr="", fdp=0, pdp=0, rdp=0, pedp=0):
        calls.update({locus + ".1": a, locus + ".2": b, locus + ".label": label, locus + ".PP": pp,
                      locus + ".FR": fr, locus + ".PR": pr, locus + ".RR": rr, locus + ".FDP": fdp,
                      locus + ".PDP": pdp, locus + ".RDP": rdp, locus + ".PEDP": pedp,
                      locus + ".CI": "{0}-{0}|{1}-{1}".format(a, b), locus + ".DP": 31.5})
        return calls

    out = {"t001": base("t001"), "t002": base("t002")}
    s = base("t001"); s["inferredGender"] = "Male"; s["depthY"] = 14.2
    called(s, "AR", 45, 45, "risk", 0.99, fr="45|9", fdp=9, pdp=4, pedp=11)
    called(s, "FXS", 30, 30, "ok", 0.0, fr="30|12", fdp=12)
    called(s, "HD", 17, 37, "prerisk", 0.31, fr="17|6;37|3", pr="5|1;12|2", fdp=9, pdp=3, pedp=20)
    out["m003"] = s
    s = base("t002"); s["inferredGender"] = "Female"
    called(s, "FXS", 30, 210, "risk", 0.97, fr="30|7", pr=";".join("{}|1".format(k) for k in range(3, 48)),
           rr="49|2;50|5", fdp=7, pdp=45, rdp=7, pedp=3)
    called(s, "HD", 19, 40, "risk", 0.42, fr="19|8", pr="40|1", fdp=8, pdp=1)          # below --minPP
    called(s, "SCA17", 36, 49, "ok", 0.2, fr="36|5;49|4", fdp=9)                       # carrier (>= cut-off, not risk)
    out["f004"] = s
    s = base("t001")
    called(s, "OPMD", 10, 13, "risk", 1.0, fr="10|6;13|5", fdp=11, pedp=8)
    called(s, "FRDA", 9, 80, "ok", 0.02, fr="9|7", pr="44|1", rr="50|1", fdp=7, pdp=1, rdp=1)   # recessive carrier
    called(s, "ULD", 2, 3, "ok", 0.0, fr="2|5;3|6", fdp=11)
    out["s005"] = s
    s = base("t002")
    called(s, "HD", 15, 15, "ok", 0.0, fr="15|11", fdp=11)
    called(s, "SCA17", 36, 37, "ok", 0.0, fr="36|4;37|5", fdp=9)
    called(s, "FRDA", 70, 85, "risk", 0.93, pr="44|2;46|1", rr="49|1;50|3", pdp=3, rdp=4, pedp=6)
    out["s006"] = s
    s = base("t001"); s["inferredGender"] = "Male"; s["depthY"] = 9.0
    called(s, "FXS", 230, 230, "risk", 0.88, pr="40|2;45|1", rr="50|6", pdp=3, rdp=6, pedp=2)
    called(s, "SBMA", 22, 22, "ok", 0.0, fr="22|8", fdp=8) if "SBMA.1" in s else None
    out["m007"] = s
    s = base("t002"); s["inferredGender"] = "Unknown"; s["depthY"] = -1
    out["u008"] = s
    return out


def gen_report(loci):
    """The reference's tredreport.main (tredreport.py:198-302) on report_inputs(): <tsv>, .cases.txt, .details.txt,
    .report.txt, with default options, with --columns PP,FR --minPP 0.3, and re-read from the TSV."""
    import tempfile
    import pandas as pd
    ref = _load_full_reference()

    # pandas >= 2 dropped two calls the py2-era reporter makes; both restated with their old semantics
    def _append(self, other, ignore_index=False, **k):
        other = pd.DataFrame(other if isinstance(other, list) else [other])
        return pd.concat([self, other], ignore_index=ignore_index, sort=True)      # unaligned columns came out sorted

    def _reindex_axis(self, labels, axis=0, **k):
        return self.reindex(columns=labels) if axis in (1, "columns") else self.reindex(labels)
    pd.DataFrame.append = _append
    pd.DataFrame.reindex_axis = _reindex_axis
    def py2_sorted(items, **k):                     # Python 2 orders numbers before strings (tredreport.py:107-108)
        try:
            return sorted(items, **k)
        except TypeError:
            key = k.get("key") or (lambda x: x)

            def rank(x):
                v = key(x)
                v0 = v[0] if isinstance(v, tuple) else v
                return (isinstance(v0, str), v)
            return sorted(items, key=rank)
    rep = refshim._load("tredparse/tredreport.py", "tredparse.tredreport", extra={"sorted": py2_sorted})
    inputs = report_inputs()
    out = {"inputs": inputs, "runs": []}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        files = []
        for key, calls in inputs.items():
            with open(key + ".json", "w") as fp:
                json.dump({"samplekey": key, "bam": key + ".bam", "tredCalls": calls}, fp)
            files.append(key + ".json")

        def collect(tsv):
            got = {}
            for suffix in ("", ".cases.txt", ".details.txt", ".report.txt"):
                with open(tsv + suffix) as fp:
                    got["tsv" + suffix] = fp.read()
            return got
        for name, argv in (("default", files + ["--tsv", "a.tsv"]),
                           ("columns_minpp", files + ["--tsv", "b.tsv", "--columns", "PP,FR", "--minPP", "0.3"]),
                           ("two_reference_samples", files[:2] + ["--tsv", "c.tsv"])):
            rep.main(argv)
            tsv = argv[argv.index("--tsv") + 1]
            out["runs"].append({"name": name, "files": [f for f in argv if f.endswith(".json")],
                                "options": [a for a in argv if not a.endswith(".json")], "outputs": collect(tsv)})
            print("report:", name, {k: len(v) for k, v in out["runs"][-1]["outputs"].items()})
    finally:
        os.chdir(cwd)
    with open(os.path.join(GOLD, "report.json"), "w") as fp:
        json.dump({"generator": "tools/gen_golden.py report: the reference's tredparse.tredreport.main (v0.7.8 via "
                                "tools/refshim.py; pandas {} with DataFrame.append / reindex_axis restated) on the "
                                "listed per-sample JSON inputs".format(pd.__version__), **out}, fp)


def gen_debug(loci):
    """What the reference prints under --log DEBUG for t001 / HD: its per-read tag lines (bam_parser.py:177-178) and
    per-pair `*** (h1, h2) ml1 ml2 ml3 ml4 ml` lines (models.py:270-272), captured from its own loggers."""
    import tempfile
    ref = refshim.load_reference(pysam_standin=PysamStandin(), full=True)
    lines = []

    class Grab(logging.Handler):
        def emit(self, record):
            lines.append((record.name, record.getMessage()))
    h = Grab()
    h.setLevel(logging.DEBUG)
    root = logging.getLogger()
    root.addHandler(h)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        repo = ref.meta.TREDsRepo(ref="hg38", toy=False, sites=os.path.join(tmp, "sites"))
        ref.tred.run(("t001", os.path.join(refshim.REF, "tests", "t001.bam"), repo, ["HD"], 300, False, False, True, True, "DEBUG"))
    finally:
        os.chdir(cwd)
        root.removeHandler(h)
    reads, pairs = [], []
    for name, msg in lines:
        head = msg.split(":")[0]
        if name == "BamParser" and head in ("FULL", "PREF", "POST", "REPT", "HANG") and ", seq=" in msg:
            reads.append([head, int(msg.split("h=")[1].split(",")[0]), msg.split("seq=")[1]])
        elif name == "IntegratedCaller" and msg.startswith("*** ("):
            inside, rest = msg[5:].split(")", 1)
            pairs.append([int(x) for x in inside.split(",")] + [float(x) for x in rest.split()])
    print("debug: {} read lines, {} pair lines (of {} records)".format(len(reads), len(pairs), len(lines)))
    with open(os.path.join(GOLD, "debug_t001_HD.json"), "w") as fp:
        json.dump({"generator": "tools/gen_golden.py debug: the reference's run() on tests/t001.bam, HD, log=DEBUG; lines of its "
                                "BamParser and IntegratedCaller loggers parsed into fields", "reads": reads, "pairs": pairs}, fp)


def main():
    os.makedirs(GOLD, exist_ok=True)
    loci = synth.load_loci()
    what = sys.argv[1:] or ["sw", "classify", "grid", "e2e"]
    if "sw" in what:
        gen_sw(loci)
    if "classify" in what or "grid" in what:
        ref = refshim.load_reference()
        logging.disable(logging.CRITICAL)
        if "classify" in what:
            gen_classify(ref, loci)
        if "grid" in what:
            gen_grid(ref, loci)
    if "e2e" in what:
        gen_e2e(loci)
    if "synf" in what:
        make_synf()
    if "flags" in what:
        gen_flags(loci)
    if "vcf" in what:
        gen_vcf(loci)
    if "report" in what:
        gen_report(loci)
    if "synall" in what:
        gen_synall(loci)
    if "synwgs" in what:
        gen_synall(loci, only=["synwgs"])
    if "debug" in what:
        gen_debug(loci)


if __name__ == "__main__":
    main()
