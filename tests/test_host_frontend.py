"""CPU: the host front end (BAM reader, read selection, pair extraction, locus table, CLI plumbing) against
numbers pinned by the reference run (SURVEY 9.4 / tests/golden/run_t001_t002.json).  No GPU needed."""
import json
import os
import types

import pytest

from tredparse_amd import bamio, tred as tredmod
from tredparse_amd.bam_parser import BamDepth, BamParser, BamReadLen, PEextractor, rc
from tredparse_amd.meta import TREDsRepo
from tredparse_amd.models import calc_label, histogram, mean_std
from tredparse_amd.utils import InputParams

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BAM1 = os.path.join(GOLD, "bam", "t001.bam")
BAM2 = os.path.join(GOLD, "bam", "t002.bam")
WANT = json.load(open(os.path.join(GOLD, "run_t001_t002.json")))["samples"]


def test_bam_reader_counts():
    f = bamio.AlignmentFile(BAM1)
    assert sum(1 for _ in f.fetch()) == 12309
    reads = list(f.fetch("chr4", 3074877 - 1000, 3074933 + 1000))
    assert len(reads) == 455 and sum(r.is_unmapped for r in reads) == 17   # SURVEY 3.2 probe numbers
    with pytest.raises(ValueError):
        list(f.fetch("chrNope", 1, 2))
    assert BamReadLen(BAM1, None).readlen == 150


def test_depth_and_pairs_match_reference_run():
    repo = TREDsRepo()
    for bam, name, sample in ((BAM1, "HD", "t001"), (BAM2, "DM1", "t002")):
        t = repo[name]
        depth = BamDepth(bam, "hg38", None).region_depth(t.chr, t.repeat_start - 1000, t.repeat_end + 1000)
        assert depth == WANT[sample][name + ".DP"]
        ip = InputParams(bam=bam, READLEN=150, tredName=name, repo=repo, depth=depth, alts=True, repeatpairs=True)
        bp = BamParser(ip)
        reads = bp.collect()
        assert len(reads) == {"HD": 68, "DM1": 177}[name]
        pe = PEextractor(bp)
        assert len(pe.target_lens) == WANT[sample][name + ".PEDP"]
        assert mean_std(pe.global_lens) == WANT[sample][name + ".PEG"]
        assert mean_std(pe.target_lens) == WANT[sample][name + ".PET"]
        assert histogram(pe.global_lens) == WANT[sample][name + ".P_PEG"]
        assert histogram(pe.target_lens) == WANT[sample][name + ".P_PET"]
        assert pe.MINPE == t.repeat_end - t.repeat_start + 20 and pe.ref == t.repeat_end - t.repeat_start + 1


def test_locus_table_and_labels():
    repo = TREDsRepo()
    assert len(repo.names) == 32 and repo["HD"].ref_copy == 19 and repo["DM1"].ref_copy == 20
    assert repo["AR"].is_expansion is False and repo["FRDA"].is_recessive and repo["FXS"].is_xlinked
    assert TREDsRepo(ref="hg19_nochr")["HD"].chr == "4"
    assert calc_label(repo["HD"], [15, 41]) == "risk" and calc_label(repo["HD"], [15, 37]) == "prerisk"
    assert calc_label(repo["HD"], [-1, -1]) == "missing" and calc_label(repo["AR"], [6, 21]) == "ok" and calc_label(repo["AR"], [6, 7]) == "risk"
    assert calc_label(repo["FRDA"], [20, 80]) == "ok" and calc_label(repo["FRDA"], [70, 80]) == "risk"
    assert rc("ACGTNacgtn") == "nacgtNACGT"
    repo.set_ploidy(["chrX"])
    assert repo["FXS"].ploidy == 1 and repo["HD"].ploidy == 2


def test_read_csv_modes(tmp_path):
    args = types.SimpleNamespace(workflow_execution_id=None, sample_id=None)
    assert tredmod.read_csv(BAM1, args) == [("t001", BAM1, None)]
    lst = tmp_path / "bams.txt"
    lst.write_text(BAM1 + "\n" + BAM2 + "\n")
    assert [x[0] for x in tredmod.read_csv(str(lst), args)] == ["t001", "t002"]
    csv = tmp_path / "s.csv"
    csv.write_text("#SampleKey,BAM,TRED\nA,{},HD\nB,{}\n".format(BAM1, BAM2))
    assert tredmod.read_csv(str(csv), args) == [("A", BAM1, "HD"), ("B", BAM2, None)]
    assert tredmod.counter_s({15: 4, 6: 1}) == "6|1;15|4"
    p = tredmod.set_argparse()
    a = p.parse_args([BAM1, "--tred", "HD", "--maxinsert", "100", "--norepeatpairs"])
    assert a.tred == ["HD"] and a.maxinsert == 100 and a.norepeatpairs and not a.fullsearch


def test_tredreport_on_reference_results(tmp_path):
    """tests.py:15-19 of the reference: tredreport on work/t001.json work/t002.json."""
    from tredparse_amd import tredreport
    files = []
    for s in ("t001", "t002"):
        f = tmp_path / (s + ".json")
        f.write_text(json.dumps({"samplekey": s, "bam": s + ".bam", "tredCalls": WANT[s]}))
        files.append(str(f))
    tsv = str(tmp_path / "work.tsv")
    total = tredreport.main(files + ["--tsv", tsv])
    assert total["risk"] == 2 and total["loci"] == 2          # HD 15/41 and DM1 5/66 are at-risk calls
    rows = open(tsv).read().splitlines()
    hdr = rows[0].split("\t")
    assert hdr[:2] == ["SampleKey", "inferredGender"] and "HD.calls" in hdr and "DM1.label" in hdr
    r1 = dict(zip(hdr, rows[1].split("\t")))
    assert r1["SampleKey"] == "t001" and r1["HD.calls"] == "15|41" and r1["HD.label"] == "risk"
    cases = open(tsv + ".cases.txt").read()
    assert "[HD] - Huntington" in cases and "[DM1]" in cases and "n_risk=1" in cases
    det = open(tsv + ".details.txt").read().splitlines()
    assert det[1].split("\t")[:5] == ["DM1", "AD", "t002", "Female", "5|66"]
    rep = open(tsv + ".report.txt").read()
    assert "{15:1,41:1}" in rep


def test_native_bam_layer_matches_python_layer():
    """libtredbam.so (csrc/bamread.cpp) against the pure-Python BGZF/BAM/BAI reader: same records in file order,
    same records for region queries through the index (incl. placed-unmapped mates, regions without reads,
    single-base regions), same pileup depth sums -- on both of the reference's test BAMs."""
    import random
    from tredparse_amd import bamio
    assert bamio._native() is not None, "libtredbam.so is not built"
    rng = random.Random(7)

    def key(r):
        return (r.tid, r.pos, r.mapq, r.flag, r.next_tid, r.next_pos, r.tlen, r.l_seq, r.query_name,
                tuple(r.cigartuples), r.query_sequence, r.reference_end, r.query_alignment_start,
                r.query_alignment_end, r.is_unmapped, r.is_reverse, r.query_length)

    for name in ("t001.bam", "t002.bam"):
        path = os.path.join(GOLD, "bam", name)
        a, b = bamio.NativeAlignmentFile(path), bamio.PyAlignmentFile(path)
        assert a.references == b.references and a.lengths == b.lengths
        ra, rb = [key(r) for r in a.fetch()], [key(r) for r in b.fetch()]
        assert ra == rb and len(ra) > 10000
        assert [key(r) for _, r in zip(range(101), a.fetch())] == ra[:101]       # early stop (BamReadLen)
        for tid in sorted(set(r[0] for r in ra if r[0] >= 0)):
            chrom = a.references[tid]
            starts = [r[1] for r in ra if r[0] == tid]
            for _ in range(40):
                s = max(0, rng.choice(starts) + rng.randint(-3000, 3000))
                e = s + rng.choice([1, 50, 1000, 2000, 20000])
                assert [key(r) for r in a.fetch(chrom, s, e)] == [key(r) for r in b.fetch(chrom, s, e)], (chrom, s, e)
                assert a.pileup_depth_sum(chrom, s, e) == b.pileup_depth_sum(chrom, s, e)
        with pytest.raises(ValueError):
            list(a.fetch("no_such_contig", 0, 10))
        a.close(); b.close()
    with pytest.raises(IOError):
        bamio.NativeAlignmentFile(os.path.join(GOLD, "bam", "missing.bam"))


def test_native_pe_lengths_match_python_pe_extractor(monkeypatch):
    """tredbam_pe_lengths (the whole PEextractor selection in one native call) against the Python loop over
    fetched records (bam_parser.py:316-369), for every locus on both test BAMs: same lists, same order."""
    from tredparse_amd import bam_parser as bpm
    repo = TREDsRepo("hg38")
    n_pairs = 0
    for name in ("t001.bam", "t002.bam"):
        path = os.path.join(GOLD, "bam", name)
        for tred in repo.names:
            ip = InputParams(bam=path, READLEN=150, tredName=tred, repo=repo, maxinsert=300, fullsearch=False,
                             gender="Unknown", depth=30, clip=False, alts=False, repeatpairs=True, log="ERROR")
            bp = BamParser(ip)
            monkeypatch.delenv("TREDBAM_PURE_PYTHON", raising=False)
            bamio._lib = None
            bpm._open_files.clear()
            nat = PEextractor(bp)
            monkeypatch.setenv("TREDBAM_PURE_PYTHON", "1")
            bamio._lib = None
            bpm._open_files.clear()
            py = PEextractor(bp)
            assert nat.global_lens == py.global_lens and nat.target_lens == py.target_lens, (name, tred)
            assert nat.MINPE == py.MINPE
            n_pairs += len(nat.global_lens) + len(nat.target_lens)
    monkeypatch.delenv("TREDBAM_PURE_PYTHON", raising=False)
    bamio._lib = None
    bpm._open_files.clear()
    assert n_pairs > 4000


def test_host_pool_collects_samples_in_worker_processes():
    """The host half of run() in forked workers (tred.host_pool, the reference's Pool over samples): what comes
    back through pickling equals an in-process collect_sample -- reads, pair lengths, depth, the shared PREF/POST
    dict -- and the workers never need a GPU."""
    repo = TREDsRepo("hg38")
    args = [(s, os.path.join(GOLD, "bam", s + ".bam"), repo, ["HD", "DM1", "SCA1"], 300, False, False, True, True, "ERROR")
            for s in ("t001", "t002")]
    pool = tredmod.host_pool(2, len(args))
    try:
        remote = pool.map(tredmod.collect_sample, args)
    finally:
        pool.close()
        pool.join()
    local = [tredmod.collect_sample(a) for a in args]
    for (r1, p1), (r2, p2) in zip(remote, local):
        assert r1 == r2 and len(p1) == len(p2) == 3
        for u1, u2 in zip(p1, p2):
            assert u1.tred == u2.tred and u1.depth == u2.depth and u1.bp.reads == u2.bp.reads
            assert u1.caller.pe.global_lens == u2.caller.pe.global_lens
            assert u1.caller.pe.target_lens == u2.caller.pe.target_lens
            assert u1.bp.counts["PREF"] is u1.bp.counts["POST"]
    assert tredmod.host_pool(1, 5) is None and tredmod.host_pool(4, 1) is None
