"""CPU: the host front end (BAM reader, read selection, pair extraction, locus table, CLI plumbing) against
numbers pinned by the reference run (SURVEY 9.4 / tests/golden/run_t001_t002.json).  No GPU needed."""
import json
import os
import types

import numpy as np
import pytest

from tredparse_amd import bamio, tred as tredmod
from tredparse_amd.bam_parser import BamDepth, BamParser, BamReadLen, InputParams, PEextractor, rc, scan_sample
from tredparse_amd.meta import TREDsRepo
from tredparse_amd.models import calc_label, histogram, mean_std

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
    assert rc("ACGTNacgtn") == "nacgtNACGT" and rc("AXC") == "GXT"
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


def test_fast_json_writer_is_byte_identical():
    """tred.dumps_result against the reference's json.dumps(sort_keys=True, indent=4, separators=(',', ': '))
    (tred.py:305-306): same bytes, for the golden runs and for awkward values."""
    for s in WANT:
        r = {"samplekey": s, "bam": "/data/" + s + ".bam", "tredCalls": WANT[s]}
        assert tredmod.dumps_result(r) == json.dumps(r, sort_keys=True, indent=4, separators=(",", ": "))
    odd = {"samplekey": 'a"q', "bam": "b", "tredCalls": {
        "X.details": [], "X.P_h1": "", "X.P_h2": {}, "X.PP": 0.1 + 0.2, "X.1": -1, "X.lik": float("-inf"),
        "X.d": [{"id": 'r,"1},\n                {', "h": 3, "seq": "AC", "tag": "FULL"}, {"id": "\u00e9", "h": 4, "seq": "", "tag": "PREF"}]}}
    assert tredmod.dumps_result(odd) == json.dumps(odd, sort_keys=True, indent=4, separators=(",", ": "))
    bare = {"samplekey": "k", "bam": "b", "tredCalls": {"inferredGender": "Unknown", "depthY": -1}}
    assert tredmod.dumps_result(bare) == json.dumps(bare, sort_keys=True, indent=4, separators=(",", ": "))


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


def test_tredreport_reads_vcf_and_accepts_cpus(tmp_path, monkeypatch):
    """The reference's reporter also takes the per-sample VCFs (tredreport.py:144-169, through PyVCF) and a --cpus
    flag: here the VCFs tred.to_vcf writes are parsed directly; the summary equals the JSON route's."""
    from tredparse_amd import tredreport
    repo = TREDsRepo()
    monkeypatch.chdir(tmp_path)
    files = []
    for s in ("t001", "t002"):
        tredmod.to_vcf({"samplekey": s, "bam": s + ".bam", "tredCalls": WANT[s]}, "hg38", repo, treds=repo.names)
        files.append(str(tmp_path / (s + ".tred.vcf.gz")))
    row = tredreport.read_vcf(files[0])
    assert row["SampleKey"] == "t001" and (row["HD.1"], row["HD.2"]) == (15, 41) and row["HD.label"] == "risk"
    assert row["HD.FR"] == WANT["t001"]["HD.FR"] and abs(row["HD.PP"] - WANT["t001"]["HD.PP"]) < 1e-3
    total = tredreport.main(files + ["--tsv", str(tmp_path / "v.tsv"), "--cpus", "2", "--columns", "PP"])
    assert total["risk"] == 2 and total["loci"] == 2
    hdr, first = [l.split("\t") for l in open(tmp_path / "v.tsv").read().splitlines()[:2]]
    assert hdr[0] == "SampleKey" and "inferredGender" not in hdr            # VCFs carry no sex
    assert dict(zip(hdr, first))["HD.calls"] == "15|41"
    # no files: the table written before is read back (it has calls, label and the PP asked for)
    again = tredreport.main(["--tsv", str(tmp_path / "v.tsv")])
    assert again["risk"] == 2


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


def _python_layer_selection(path, t, readlen, alts=True, strip=False):
    """The read selection, depth and pair lengths of one locus computed record by record over the pure-Python
    reader (bamio.PyAlignmentFile) -- the independent check of the native one-call scan."""
    f = bamio.PyAlignmentFile(path)
    lo, hi = max(0, t.repeat_start - 1000), t.repeat_end + 1000
    reads = []
    try:
        for r in f.fetch(t.chr, lo, hi):
            if r.is_unmapped or max(0, t.repeat_start - readlen) <= r.pos <= t.repeat_end + readlen:
                reads.append((r.query_name, r.query_sequence))
        if alts:
            for c, a, b in t.alt:
                try:
                    for r in f.fetch(c[3:] if strip else c, a, b):
                        if r.next_tid >= 0 and f.references[r.next_tid] == t.chr and lo <= r.next_pos <= hi:
                            reads.append((r.query_name, r.query_sequence))
                except ValueError:
                    pass
        depth = f.pileup_depth_sum(t.chr, lo, hi) / float(hi - lo + 1)
    except ValueError:
        return None
    first = {}
    for r in f.fetch(t.chr, max(t.repeat_start - 10000, 0), t.repeat_end + 10000):
        if r.is_paired and not r.is_unmapped and not r.is_duplicate:
            first.setdefault(r.query_name, []).append(r)
    gl, tl = [], []
    for pair in first.values():
        if len(pair) < 2 or pair[0].is_reverse or not pair[1].is_reverse:
            continue
        a, b = pair[:2]
        tlen = (b.reference_end + b.l_seq - b.query_alignment_end) - (a.pos - a.query_alignment_start)
        if tlen < 1000:
            (tl if a.pos < t.repeat_start - 9 and b.reference_end > t.repeat_end + 9 else gl).append(tlen)
    f.close()
    return reads, depth, gl, tl


def test_native_scan_matches_record_by_record_selection():
    """tredbam_scan (one native call per sample: depth, read selection incl. unmapped mates and ALT rescue, pair
    lengths, reads 2-bit packed) against the same selection done record by record over the pure-Python reader,
    for every locus on both test BAMs: same reads in the same order, same depth, same pair-length lists; the
    packed words equal libtredgpu's own packer on the decoded sequences."""
    from tredparse_amd import _lib
    repo = TREDsRepo("hg38")
    n_reads = n_pairs = 0
    for name in ("t001.bam", "t002.bam"):
        path = os.path.join(GOLD, "bam", name)
        s = scan_sample(path, repo, repo.names)
        assert s.opened and s.readlen == 150 and not s.dropped
        for k, tname in enumerate(repo.names):
            want = _python_layer_selection(path, repo[tname], 150)
            a, b = s.reads_of(k)
            got = [(s.name(i), s.sequence(i)) for i in range(a, b)]
            if want is None:                      # contig not in the file: no reads, depth falls back to 30
                assert got == [] and s.depth[k] == 30.0
                continue
            reads, depth, gl, tl = want
            assert got == reads, (name, tname)
            assert s.depth[k] == depth
            g, t = s.pair_lengths(k)
            assert g.tolist() == gl and t.tolist() == tl, (name, tname)
            n_reads += len(reads)
            n_pairs += len(gl) + len(tl)
            # name ids: equal names <-> equal ids inside the unit
            ids = s.name_id[a:b].tolist()
            assert [ids.index(x) for x in ids] == [[n for n, _ in reads].index(n) for n, _ in reads]
        if len(s.read_len):
            packed, woff, rlen = _lib.pack_reads([s.sequence(i) for i in range(len(s.read_len))])
            assert np.array_equal(packed[:woff[-1]], s.packed) and np.array_equal(woff, s.word_off)
            assert np.array_equal(rlen, s.read_len)
    assert n_reads > 200 and n_pairs > 4000


def test_scan_options_and_failures(tmp_path):
    repo = TREDsRepo("hg38")
    s = scan_sample(BAM1, repo, ["HD", "SCA6"], alts=False)
    assert s.unit["n_reads"].tolist() == [68, 0] and s.unit["n_global"][0] == 2805     # (ALT rescue: test_synth_bam)
    missing = scan_sample(str(tmp_path / "nope.bam"), repo, ["HD"])
    assert not missing.opened and missing.gender == "Unknown" and missing.ydepth == -1
    nochr = scan_sample(BAM1, TREDsRepo("hg38_nochr"), ["HD"])       # contig "4" is not in this file
    assert nochr.opened and nochr.unit["n_reads"][0] == 0 and nochr.depth[0] == 30.0
    # sex: no reads on chrY in the test BAM -> Female, depthY 0.0 (the reference run says the same)
    x = scan_sample(BAM1, repo, ["FXS"])
    assert x.gender == WANT["t001"]["inferredGender"] and x.ydepth == WANT["t001"]["depthY"]


def test_tally_follows_the_reference_bookkeeping():
    """counts / details / rept from per-read (tag, h): PREF and POST pooled, HANG counts every aligned read, details
    in BAM order without HANG reads; --norepeatpairs removes every read of a name that carries two REPT records."""
    from tredparse_amd import _lib
    from tredparse_amd.bam_parser import tally
    repo = TREDsRepo("hg38")
    s = scan_sample(BAM1, repo, ["HD"])
    n = int(s.unit["n_reads"][0])
    names = [s.name(i) for i in range(n)]
    mate = next(i for i in range(1, n) if names[i] in names[:i])           # second record of some pair
    first = names.index(names[mate])
    tags = np.zeros(n, np.uint8)
    hs = np.zeros(n, np.int16)
    tags[first], hs[first] = _lib.TAG_REPT, 50
    tags[mate], hs[mate] = _lib.TAG_REPT, 50
    other = [i for i in range(n) if names[i] != names[mate]][:4]
    for i, (t, h) in zip(other, ((_lib.TAG_FULL, 15), (_lib.TAG_PREF, 9), (_lib.TAG_POST, 9), (_lib.TAG_HANG, 3))):
        tags[i], hs[i] = t, h
    counts, details, rept = tally(s, 0, tags, hs, repeatpairs=True)
    assert counts["PREF"] is counts["POST"] and counts["PREF"] == {9: 2} and counts["FULL"] == {15: 1}
    assert counts["REPT"] == {50: 2} and rept == 2 and counts["HANG"] == {50: 2, 15: 1, 9: 2, 3: 1}
    order = sorted([first, mate] + other[:3])
    assert [(d["tag"], d["h"], d["id"], d["seq"]) for d in details] == \
        [(_lib.TAG_NAMES[int(tags[i])], int(hs[i]), names[i], s.sequence(i)) for i in order]
    counts2, details2, rept2 = tally(s, 0, tags, hs, repeatpairs=False)
    assert rept2 == 0 and counts2["REPT"] == {} and counts2["FULL"] == {15: 1} and len(details2) == len(details) - 2


def test_lazy_details_print_like_the_list():
    """bam_parser.Details (the pools' view of a locus' details) equals the list of dicts, and its natively written JSON
    is byte for byte what the driver's encoder prints for that list -- also for names with quotes and backslashes;
    names the native writer does not handle (control / non-ASCII bytes) go through the generic encoder."""
    from tredparse_amd import _lib
    from tredparse_amd.bam_parser import Details, SampleScan, tally
    repo = TREDsRepo("hg38")
    s = scan_sample(BAM1, repo, ["HD"])
    n = int(s.unit["n_reads"][0])
    rng = np.random.default_rng(5)
    tags = rng.integers(0, 6, n).astype(np.uint8)
    hs = rng.integers(0, 60, n).astype(np.int16)
    _, plain, _ = tally(s, 0, tags, hs)
    _, lazy, _ = tally(s, 0, tags, hs, lazy=True)
    assert isinstance(plain, list) and isinstance(lazy, Details) and len(lazy) == len(plain) > 10
    assert lazy == plain and plain == list(lazy) and lazy[3] == plain[3] and not (lazy != plain)
    text = lazy.json_text()
    assert text is not None and text == tredmod._flat_list(plain)
    res = {"samplekey": "x", "bam": "x.bam", "tredCalls": {"HD.1": 17, "HD.details": lazy, "HD.PP": 1.0}}
    ref = {"samplekey": "x", "bam": "x.bam", "tredCalls": {"HD.1": 17, "HD.details": plain, "HD.PP": 1.0}}
    assert tredmod.dumps_result(res) == json.dumps(ref, sort_keys=True, indent=4, separators=(",", ": "))
    _, none, _ = tally(s, 0, np.zeros(n, np.uint8), hs, lazy=True)
    assert len(none) == 0 and none.json_text() == "[]" == tredmod._flat_list([])
    # hand-made pools: odd and even read lengths, N and = codes, names that need escapes
    fake = SampleScan()
    names = [b'a"b', b"back\\slash", b"plain/1", b"x", b"tab\there", "smørre".encode("utf-8")]
    seqs = ["ACGTN", "=ACMGRSVTWYHKDBN", "", "T", "GG", "CA"]
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    blob, offs = bytearray(), [0]
    for q in seqs:
        nib = [code[c] for c in q] + [0]
        blob += bytes((nib[i] << 4) | nib[i + 1] for i in range(0, len(q), 2))
        offs.append(len(blob))
    fake.seq4, fake.seq4_off = np.frombuffer(bytes(blob), np.uint8), np.asarray(offs, np.int64)
    fake.read_len = np.asarray([len(q) for q in seqs], np.int32)
    fake.name_blob = b"".join(names)
    fake.name_off = np.cumsum([0] + [len(x) for x in names]).astype(np.int64)
    fake._text = None
    ok = Details(fake, np.asarray([2, 0, 1, 3], np.int64), np.asarray([1, 2, 4, 5], np.uint8), np.asarray([7, 0, 33, 120], np.int32))
    assert [d["seq"] for d in ok] == ["", "ACGTN", "=ACMGRSVTWYHKDBN", "T"] and ok[1]["id"] == 'a"b' and ok[2]["id"] == "back\\slash"
    assert ok.json_text() == tredmod._flat_list(ok.items())
    for bad in (4, 5):     # a tab, a non-ASCII letter: the generic encoder's business
        d = Details(fake, np.asarray([0, bad], np.int64), np.asarray([1, 1], np.uint8), np.asarray([1, 2], np.int32))
        assert d.json_text() is None
        out = tredmod.dumps_result({"samplekey": "k", "bam": "b", "tredCalls": {"X.details": d}})
        assert json.loads(out)["tredCalls"]["X.details"] == d.items()


def test_sparse_distributions_print_like_the_dicts():
    """models.SparseDist (P_h1 / P_h2 / P_h1h2 as arrays) reads like the dict and its natively written JSON is what the
    driver's encoder prints: keys sorted as strings ("10" < "9"), values in Python's repr(float) -- checked on its own
    over the exponent-form boundaries, subnormals and random bit patterns."""
    import ctypes as C
    import random
    import struct
    from tredparse_amd.models import SparseDist, sparsify_joint_triples, sparsify_marginal
    lib = bamio._native()
    lib.tredbam_float_repr.argtypes, lib.tredbam_float_repr.restype = [C.c_double, C.c_char_p], C.c_int
    buf = C.create_string_buffer(64)
    rnd = random.Random(11)
    xs = [0.0, -0.0, 1.0, 0.1, 1e-4, 9.999e-5, 1e-5, 1e15, 1e16, 9999999999999998.0, 1e22, 5e-324, 2.2250738585072014e-308,
          1.7976931348623157e308, 1 / 3, 100.0, 123.456, 4.5399929762484854e-05, -2.5e-7]
    xs += [rnd.random() for _ in range(20000)] + [rnd.random() * 10 ** rnd.randint(-12, 20) for _ in range(20000)]
    xs += [struct.unpack("<d", struct.pack("<Q", rnd.getrandbits(64)))[0] for _ in range(20000)]
    for x in xs:
        if x != x or x in (float("inf"), float("-inf")):
            assert lib.tredbam_float_repr(x, buf) == -1
            continue
        n = lib.tredbam_float_repr(x, buf)
        assert buf.raw[:n].decode() == repr(x), x
    rng = np.random.default_rng(4)
    P = rng.random(300) ** 8
    P[rng.random(300) < .5] = 0
    d, lazy = sparsify_marginal(P), sparsify_marginal(P, lazy=True)
    assert isinstance(d, dict) and isinstance(lazy, SparseDist) and lazy == d and d == dict(lazy) and len(lazy) == len(d) > 20
    assert "10" in lazy or "11" in lazy or "12" in lazy
    assert lazy.json_text(2) == tredmod._flat(d, 2)
    tr = np.stack([rng.integers(1, 60, 300) * 3, rng.integers(1, 300, 300) * 3, rng.random(300) ** 6], 1).astype(float)
    tr = tr[np.unique(tr[:, :2], axis=0, return_index=True)[1]]
    dj, lj = sparsify_joint_triples(tr, 7.3, 3), sparsify_joint_triples(tr, 7.3, 3, lazy=True)
    assert lj == dj and lj.json_text(2) == tredmod._flat(dj, 2) and lj.json_text(0) == tredmod._flat(dj, 0)
    assert SparseDist(np.zeros(0, np.int64), None, np.zeros(0)).json_text(2) == "{}"
    twice = SparseDist(np.array([3, 3]), None, np.array([.5, .25]))            # the generic encoder's business
    assert twice.json_text(2) is None and twice == {"3": .25}
    notfinite = SparseDist(np.array([1]), None, np.array([float("nan")]))
    assert notfinite.json_text(2) is None
    res = {"samplekey": "s", "bam": "b", "tredCalls": {"X.P_h1": lazy, "X.P_h1h2": lj, "X.P_h2": twice, "X.PP": 0.5, "X.1": 3}}
    ref = {"samplekey": "s", "bam": "b", "tredCalls": {"X.P_h1": d, "X.P_h1h2": dj, "X.P_h2": {"3": .25}, "X.PP": 0.5, "X.1": 3}}
    assert tredmod.dumps_result(res) == json.dumps(ref, sort_keys=True, indent=4, separators=(",", ": "))


def test_scans_run_in_host_threads():
    """run_many's host half: scans in worker threads give what a serial scan gives (the native call releases the
    GIL and every thread has its own file handle)."""
    from concurrent.futures import ThreadPoolExecutor
    repo = TREDsRepo("hg38")
    args = [(s, os.path.join(GOLD, "bam", s + ".bam"), repo, ["HD", "DM1", "SCA1"], 300, False, False, True, True, "ERROR")
            for s in ("t001", "t002", "t001", "t002")]
    with ThreadPoolExecutor(4) as ex:
        threaded = list(ex.map(tredmod.collect_sample, args))
    serial = [tredmod.collect_sample(a) for a in args]
    for a, b in zip(threaded, serial):
        assert np.array_equal(a.unit, b.unit) and np.array_equal(a.packed, b.packed)
        assert np.array_equal(a.global_lens, b.global_lens) and a.name_blob == b.name_blob
        assert np.array_equal(a.depth, b.depth) and a.readlen == b.readlen == 150


def test_long_read_drops_only_its_unit():
    """A read beyond the kernel's 480 bp (or a ladder beyond 511 columns) costs its own sample x locus unit, with an
    error log, like any failing locus of the reference -- not the sample, not the batch."""
    from tredparse_amd import bam_parser
    repo = TREDsRepo("hg38")
    s = scan_sample(BAM1, repo, ["SCA1", "HD", "DM1"])
    assert not s.dropped
    a, _ = s.reads_of(1)
    s.read_len[a] = 300                                # a MiSeq 2x300 read in HD's window is within the limit
    assert not bam_parser.admit(s)
    s.read_len[a] = 400                                # nor is a merged pair of 400 bp
    assert not bam_parser.admit(s)
    s.read_len[a] = 500                                # a longer one is
    assert list(bam_parser.admit(s)) == [1] and "500 bp" in s.dropped[1]
    s.read_len[a] = 150
    s.readlen = 500                                    # ladder of 18 + 3 * 167 + 18 columns
    assert sorted(bam_parser.admit(s)) == [0, 1, 2] and "ladder" in s.dropped[0]


def test_lazy_views_print_through_the_batched_native_calls():
    """A sample's `details` (bam_parser.Details over the scan's pools) and distributions (models.SparseDist) go
    through ONE native call each (tredbam_details_json_many / tredbam_sparse_json_many): same bytes as json.dumps of
    the plain lists and dicts, incl. an empty list, an empty distribution and a one-part next to a two-part one."""
    import numpy as np
    from tredparse_amd import bam_parser, models
    repo = TREDsRepo(ref="hg38", sites=os.path.join(GOLD, "no_sites"))
    scan = bam_parser.scan_sample(os.path.join(GOLD, "bam", "t001.bam"), repo, ["HD", "DM1", "SCA1"])
    rng = np.random.default_rng(3)
    calls, plain = {"inferredGender": "Female", "depthY": 0.0, "readLen": 150}, {}
    for k, name in enumerate(scan.names):
        a, b = scan.reads_of(k)
        n = b - a
        tags = rng.integers(0, 6, n).astype(np.uint8)
        hs = rng.integers(0, 51, n).astype(np.int16)
        _, det, _ = bam_parser.tally(scan, k, tags, hs, lazy=True)
        calls[name + ".details"] = det
        m = rng.random(60) * (rng.random(60) < 0.3)
        m[7] = 1e-7
        calls[name + ".P_h1"] = models.sparsify_marginal(m + 0.0, lazy=True) if k != 2 else models.SparseDist(
            np.zeros(0, np.int64), None, np.zeros(0))
        trip = np.array([[3 * i, 3 * j, rng.random()] for i in range(1, 9) for j in range(i, 9)])
        calls[name + ".P_h1h2"] = models.sparsify_joint_triples(trip, float(trip[:, 2].sum()), 3, lazy=True)
        calls[name + ".PP"] = float(rng.random())
    for key, v in calls.items():
        plain[key] = v.items() if isinstance(v, bam_parser.Details) else (v.as_dict() if isinstance(v, models.SparseDist) else v)
    assert sum(len(plain[n + ".details"]) for n in scan.names) > 20 and plain["SCA1.details"] == []
    lazy = {"samplekey": "t001", "bam": "x.bam", "tredCalls": calls}
    eager = {"samplekey": "t001", "bam": "x.bam", "tredCalls": plain}
    assert tredmod.dumps_result(lazy) == json.dumps(eager, sort_keys=True, indent=4, separators=(",", ": "))


def test_pair_stats_equal_the_reference_formulas():
    """tredbam_pair_stats (all loci of a sample in one native pass) against the reference's per-list formulas
    (models.py:87-98: mean_std = numpy mean / std, histogram = numpy.histogram over 40 bins of (0, 1000)) -- incl. an
    empty slice, a single value, values on bin edges and at 1000, and values outside the histogram's range."""
    rng = np.random.default_rng(9)
    lists = [[], [208], rng.integers(150, 999, 2000).tolist(), [0, 25, 24, 999, 1000, 975, 974], [1000, 1001, -3, 500],
             rng.integers(300, 420, 37).tolist()]
    pool = np.array([x for l in lists for x in l], np.int32)
    count = np.array([len(l) for l in lists], np.int32)
    first = np.concatenate([[0], np.cumsum(count)[:-1]]).astype(np.int64)
    mean, sd, hist = bamio.pair_stats(pool, first, count)
    for k, l in enumerate(lists):
        if not l:
            assert mean[k] == 0 and sd[k] == 0 and hist[k].sum() == 0
            continue
        assert "%.0f+/-%.0fbp" % (mean[k], sd[k]) == mean_std(l)
        assert abs(mean[k] - np.mean(l)) < 1e-9 and abs(sd[k] - np.std(l)) < 1e-9
        want = np.histogram(l, bins=40, range=(0, 1000))[0]
        assert hist[k].tolist() == want.tolist(), k
        assert ",".join("%d:%d" % (25 * j, c) for j, c in enumerate(hist[k])) == histogram(l)


def test_batched_json_calls_fall_back_per_item():
    """tredbam_sparse_json_many / tredbam_details_json_many: an item the native writer cannot print (a duplicate key, a
    value that is not finite, a name with a control character) comes back as None -- that item alone goes to the generic
    encoder -- and the others are unaffected; empty items print as {} / []."""
    from tredparse_amd import bam_parser
    a = np.array([3, 5, 7], np.int32)
    texts = bamio.sparse_json_many([(a, None, np.array([.1, .2, .3])), (a[:0], None, np.zeros(0)),
                                    (np.array([4, 4], np.int32), None, np.array([.5, .5])),
                                    (a[:2], np.array([9, 9], np.int32), np.array([1e-7, float("inf")])),
                                    (a[:1], a[:1], np.array([0.25]))], 2)
    assert texts[1] == "{}" and texts[2] is None and texts[3] is None
    assert json.loads(texts[0]) == {"3": .1, "5": .2, "7": .3} and json.loads(texts[4]) == {"3,3": 0.25}
    repo = TREDsRepo(ref="hg38", sites=os.path.join(GOLD, "no_sites"))
    s = scan_sample(BAM1, repo, ["HD"])
    a0, b0 = s.reads_of(0)
    reads = np.arange(a0, a0 + 3, dtype=np.int64)
    tags, hs = np.array([1, 2, 5], np.uint8), np.array([15, 7, 30], np.int32)
    ok = bamio.details_json_many(s.seq4, s.seq4_off, s.read_len, s.name_blob, s.name_off,
                                 [(reads, tags, hs), (reads[:0], tags[:0], hs[:0])])
    assert ok[1] == "[]" and [d["h"] for d in json.loads(ok[0])] == [15, 7, 30]
    blob = bytearray(s.name_blob)
    blob[int(s.name_off[a0 + 1])] = 7                       # a control character in the second read's name
    bad = bamio.details_json_many(s.seq4, s.seq4_off, s.read_len, bytes(blob), s.name_off,
                                  [(reads[:1], tags[:1], hs[:1]), (reads, tags, hs)])
    assert bad[1] is None and json.loads(bad[0])[0]["tag"] == "FULL"


def test_scan_over_blocks_inflated_elsewhere_equals_the_plain_scan():
    """tredbam_plan lists the BGZF blocks a scan will read, tredbam_plan_fill hands out their deflate payloads on 4-byte
    boundaries, tredbam_preload takes the inflated bytes back (zlib stands in for the GPU's batch decoder here): the scan
    then reads (nearly) all its blocks from there and returns exactly what the plain scan returns; a block that is
    missing from the set, refused by the decoder or damaged on the way is inflated by the scan itself / reported."""
    import zlib
    from tredparse_amd import bam_parser
    from tredparse_amd.meta import TREDsRepo
    repo = TREDsRepo(ref="hg38", sites=os.path.join(GOLD, "no_sites"))
    names = list(repo.names)
    for sample in ("t001", "t002"):
        path = os.path.join(GOLD, "bam", sample + ".bam")
        ref = bam_parser.scan_sample(path, repo, names)
        f = bam_parser.open_bam(path)
        sites, regions = bam_parser._site_arrays(repo, names, [repo[n] for n in names], f)
        n, cbytes, obytes = f.plan(sites, regions, ref.readlen)
        assert n > 30 and cbytes % 4 == 0 and obytes > cbytes
        comp, out = np.zeros(cbytes + 64, np.uint8), np.zeros(obytes + 64, np.uint8)
        coff, ooff = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
        f.plan_fill(comp.ctypes.data, 0, 0, coff, ooff)
        assert coff[-1] == cbytes and ooff[-1] == obytes and (coff % 4 == 0).all()
        for k in range(n):
            data = zlib.decompressobj(-15).decompress(bytes(comp[coff[k]:coff[k + 1]]))
            assert len(data) == ooff[k + 1] - ooff[k]
            out[ooff[k]:ooff[k + 1]] = np.frombuffer(data, np.uint8)
        status = np.zeros(n, np.int32)
        status[3] = -1                                       # one block the decoder refused: left to the scan
        assert f.preload(out.ctypes.data, ooff, status) == n - 1
        got = bam_parser.scan_sample(path, repo, names, handle=f, readlen=ref.readlen)
        hits, misses = f.preload_clear()
        assert hits > 10 * max(misses, 1) and misses >= 1
        for field in ("packed", "word_off", "read_len", "global_lens", "target_lens", "name_id", "unit", "depth"):
            assert np.array_equal(getattr(ref, field), getattr(got, field)), (sample, field)
        assert ref.name_blob == got.name_blob
        # a damaged block among the preloaded ones fails its CRC at first use, is dropped and inflated by the scan itself,
        # like a block the plan missed: only what the file holds can fail a scan
        status[:] = 0
        good = out.copy()
        out[ooff[:-1] + 7] ^= 0x20                            # (every block: whichever the walks read first)
        assert f.preload(out.ctypes.data, ooff, status) == 0  # preload_clear dropped the plan with the blocks: plan again
        assert f.plan(sites, regions, ref.readlen)[0] == n
        assert f.preload(out.ctypes.data, ooff, status) == n
        again = bam_parser.scan_sample(path, repo, names, handle=f, readlen=ref.readlen)
        hits, misses = f.preload_clear()
        assert hits == 0 and misses > 10
        for field in ("packed", "read_len", "global_lens", "target_lens", "unit", "depth"):
            assert np.array_equal(getattr(ref, field), getattr(again, field)), (sample, field)
        # with the decoder's own checksums (tredbam_preload_crc): blocks whose checksum equals the trailer's are taken as
        # verified -- the scan does not walk their bytes again, so even a byte damaged AFTER the checksum was taken is
        # not looked for --, a block whose checksum differs is not taken at all
        f.close()
        f = bam_parser.open_bam(path)                         # (a fresh handle: the last scan left its blocks in the old one's cache)
        assert f.plan(sites, regions, ref.readlen)[0] == n
        crc = np.array([zlib.crc32(bytes(good[ooff[k]:ooff[k + 1]])) for k in range(n)], np.uint32)
        crc[5] ^= 1
        assert f.preload(good.ctypes.data, ooff, status, crc) == n - 1
        got = bam_parser.scan_sample(path, repo, names, handle=f, readlen=ref.readlen)
        hits, misses = f.preload_clear()
        assert hits > 10 * max(misses, 1)
        for field in ("packed", "read_len", "global_lens", "target_lens", "unit", "depth"):
            assert np.array_equal(getattr(ref, field), getattr(got, field)), (sample, field)
        f.close()


def test_a_selected_record_without_a_sequence_is_reported_by_the_scan(tmp_path, caplog):
    """SEQ '*' (l_seq 0) on a record the selection takes: pysam gives query_sequence None, the reference's len(seq) raises
    (tredparse/bam_parser.py:129-133) and the locus is lost (tred.py:245-249).  Here: TREDBAM_UNIT_NO_SEQ on that unit,
    admit() drops it with the reference's kind of log line, the other loci stay; both file layers return None like pysam."""
    import logging
    import numpy as np
    from tredparse_amd import bam_parser, bamio, synth, synth_bam
    from tredparse_amd.meta import TREDsRepo
    loci = [l for l in synth.load_loci() if l["name"] in ("HD", "DM1", "SCA1")]
    recs, _ = synth_bam.simulate_sample(92, loci, synth.SynthParams(coverage=12))
    repo = TREDsRepo()
    t0 = repo["HD"]
    li = [l["name"] for l in loci].index("HD")
    inwin = np.nonzero((recs.locus == li) & (recs.pos >= t0.repeat_start - 100) & (recs.pos <= t0.repeat_end + 100) & ((recs.flag & 0x4) == 0))[0]
    far = np.nonzero((recs.locus == li) & (recs.pos < t0.repeat_start - 2000))[0]      # in the pair-length region only: harmless
    mask = np.zeros(len(recs), bool)
    mask[far[0]] = True
    path = str(tmp_path / "far.bam")
    synth_bam.write_bam(path, recs, sample="q", no_seq=mask)
    names = [l["name"] for l in loci]
    s = bam_parser.scan_sample(path, repo, names)
    assert s.dropped == {} and not (s.unit["status"] & bamio.UNIT_NO_SEQ).any()
    mask[inwin[0]] = True
    path = str(tmp_path / "noseq.bam")
    synth_bam.write_bam(path, recs, sample="q", no_seq=mask, split_records=True, block=5000)
    with caplog.at_level(logging.ERROR):
        s = bam_parser.scan_sample(path, repo, names)
    k = names.index("HD")
    assert list(s.dropped) == [k] and "without a sequence" in s.dropped[k]
    assert [bool(x & bamio.UNIT_NO_SEQ) for x in s.unit["status"]] == [n == "HD" for n in names]
    assert any("Exception on" in r.getMessage() and " HD " in r.getMessage() for r in caplog.records)
    a, b = s.reads_of(k)
    assert int(s.read_len[a:b].min()) == 0 and int((s.read_len[a:b] == 0).sum()) == 1
    for cls in (bamio.PyAlignmentFile, bamio.NativeAlignmentFile):
        f = cls(path)
        seqs = [r.query_sequence for r in f.fetch(t0.chr, t0.repeat_start - 150, t0.repeat_end + 150)]
        assert seqs.count(None) == 1 and all(x is None or len(x) == 150 for x in seqs)
        f.close()
