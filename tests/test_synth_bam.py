"""Synthetic BAMs (tredparse_amd/synth_bam.py; BASELINE configs 3-5 are "synthetic 30x 150 bp BAMs").

CPU: the writer produces files both readers accept, the .bai finds exactly the overlapping records, and the native
one-call scan selects what `expected_scan` -- numpy over the record table, written from the reference's rules
(bam_parser.py:196-243, 316-369, 404-411) -- says it must: unmapped mates, ALT-rescued reads, soft clips,
duplicates / QC-fail / secondary records included.
GPU: BAM -> tred.run_many gives the calls of the packed-batch path fed from the same simulation."""
import json
import os

import numpy as np
import pytest

from tredparse_amd import bamio, synth, synth_bam as sb
from tredparse_amd.bam_parser import scan_sample
from tredparse_amd.meta import TREDsRepo

NAMES = ("HD", "DM1", "SCA10", "FXS", "FRDA")


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    loci = [l for l in synth.load_loci() if l["name"] in NAMES]
    p = synth.SynthParams(coverage=30, expanded_max=120, expanded_frac=0.5)       # some tracts longer than a read
    recs, h_true = sb.simulate_sample(11, loci, p)
    path = str(tmp_path_factory.mktemp("synbam") / "s11.bam")
    sb.write_bam(path, recs, sample="s11")
    return loci, recs, h_true, path


def test_both_readers_parse_the_written_file(sample):
    loci, recs, _, path = sample
    names = recs.names("s11")
    for reader in (bamio.NativeAlignmentFile, bamio.PyAlignmentFile):
        f = reader(path)
        assert f.references == sb.CONTIGS
        got = list(f.fetch())
        assert len(got) == len(recs)
        for i in list(range(0, len(recs), 997)) + [len(recs) - 1]:
            r = got[i]
            assert (r.tid, r.pos, r.flag, r.next_tid, r.next_pos) == (recs.tid[i], recs.pos[i], recs.flag[i], recs.mtid[i], recs.mpos[i])
            assert r.query_name == names[i] and r.query_sequence == synth.decode(recs.codes[i])
            assert [(op, n) for op, n in r.cigartuples] == [(int(c & 15), int(c >> 4)) for c in recs.cig[i, :recs.n_cig[i]]]
            end = r.reference_end
            assert (end if end is not None else -1) == recs.ref_end[i]
        f.close()
    flags = recs.flag
    assert (flags & sb.FUNMAP).any() and (flags & sb.FDUP).any() and (flags & sb.FSEC).any() and (flags & sb.FQC).any()
    assert (recs.n_cig == 3).any() and ((recs.cig[:, 0] & 15) == sb.OP_S).any()


def test_index_finds_exactly_the_overlapping_records(sample):
    loci, recs, _, path = sample
    rng = np.random.default_rng(3)
    f = bamio.NativeAlignmentFile(path)
    rend = recs.ref_end
    epos = np.where(rend > recs.pos, rend, recs.pos + 1)
    for _ in range(200):
        i = int(rng.integers(0, len(recs)))
        lo = max(0, int(recs.pos[i]) + int(rng.integers(-3000, 3000)))
        hi = lo + int(rng.choice([1, 40, 300, 2000, 21000]))
        want = np.nonzero((recs.tid == recs.tid[i]) & (recs.pos < hi) & (epos > lo))[0]
        got = [(r.pos, r.flag, r.query_name) for r in f.fetch(sb.CONTIGS[recs.tid[i]], lo, hi)]
        names = recs.names("s11")
        assert got == [(recs.pos[k], recs.flag[k], names[k]) for k in want]
    f.close()


@pytest.mark.parametrize("alts", [True, False])
def test_native_scan_finds_what_the_rules_say(sample, alts):
    loci, recs, _, path = sample
    repo = TREDsRepo()
    names = recs.names("s11")
    scan = scan_sample(path, repo, [l["name"] for l in loci], alts=alts)
    assert scan.opened and scan.readlen == 150 and not scan.dropped
    n_unmapped = n_alt = 0
    for k, l in enumerate(loci):
        reads, depth, gl, tl = sb.expected_scan(recs, l, 150, alts=alts)
        a, b = scan.reads_of(k)
        assert [scan.name(i) for i in range(a, b)] == [names[i] for i in reads], l["name"]
        assert [scan.sequence(i) for i in range(a, b)] == [synth.decode(recs.codes[i]) for i in reads]
        assert scan.depth[k] == depth
        g, t = scan.pair_lengths(k)
        assert g.tolist() == gl and t.tolist() == tl and len(gl) > 1000 and len(tl) >= 5
        n_unmapped += int(((recs.flag[reads] & sb.FUNMAP) != 0).sum())
        n_alt += int((recs.tid[reads] != sb.CONTIGS.index(l["repeat_location"].split(":")[0])).sum())
    assert n_unmapped > 0 and (n_alt > 0) == alts


def test_whole_genome_shaped_sample_keeps_its_rescued_mates(tmp_path):
    """synth_bam wgs_like: the sample of tests/golden/run_synall.json["synwgs"] -- four loci plus 10x
    background reads over the 16 kb index windows of their ~300 alternative regions and the chrY depth windows -- regenerated
    from its seed (digest checked).  The background is most of the file, none of it is ever selected, and the mates the
    simulation mismapped into the alternative regions are still found behind it: by the rules (expected_scan), by the
    native scan, and by the reference itself -- the golden's `details` (the reference's run() through tools/gen_golden.py)
    list exactly the reads the scan selects and the kernels tag, which the GPU test of the same sample compares in full
    (test_flags_gpu.py::test_synthetic_samples_match_reference[synwgs])."""
    import hashlib
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "run_synall.json")))["samples"]["synwgs"]
    loci = [l for l in synth.load_loci() if l["name"] in gold["loci"]]
    assert [l["name"] for l in loci] == gold["loci"]
    p = synth.SynthParams(coverage=10.0, expanded_max=120, expanded_frac=0.5)
    recs, _ = sb.simulate_sample(gold["seed"], loci, p, alt_rate=0.4, wgs_like=True)
    h = hashlib.sha256()
    for k in recs.FIELDS:
        h.update(np.ascontiguousarray(getattr(recs, k)).tobytes())
    assert h.hexdigest() == gold["records_sha256"]
    plain, _ = sb.simulate_sample(gold["seed"], loci, p, alt_rate=0.4)
    background = recs.locus == sb.BACKGROUND_LOCUS
    assert background.sum() > 4 * len(plain) and len(recs) == len(plain) + background.sum()
    wins = sb.background_windows(loci)
    assert len(wins) > 150 and any(sb.CONTIGS[t] == "chrY" for t, _, _ in wins)          # FXS is X-linked
    path = str(tmp_path / "synwgs.bam")
    sb.write_bam(path, recs, sample="synwgs")
    names = recs.names("synwgs")
    scan = scan_sample(path, TREDsRepo(), gold["loci"])
    assert scan.opened and not scan.dropped and scan.gender in ("Male", "Female") and scan.ydepth > 1
    rescued = 0
    for k, l in enumerate(loci):
        reads, depth, gl, tl = sb.expected_scan(recs, l, 150)
        a, b = scan.reads_of(k)
        got = [scan.name(i) for i in range(a, b)]
        assert got == [names[i] for i in reads] and scan.depth[k] == depth, l["name"]
        assert not any(".{:02d}.".format(sb.BACKGROUND_LOCUS) in n for n in got)          # no background read is selected
        rescued += int((recs.tid[reads] != sb.CONTIGS.index(l["repeat_location"].split(":")[0])).sum())
        # the reads the reference reported for this locus are among the selected ones
        assert set(d[0] for d in gold["tredCalls"][l["name"] + ".details"]) <= set(got)
    assert rescued >= 4


@pytest.mark.gpu
def test_bam_path_equals_packed_path(tmp_path):
    """BAM -> CLI driver (native scan, PackedUnits) against the packed-batch path fed straight from the simulation's
    record table (read strings, depth and pair lengths from expected_scan): identical tredCalls."""
    from tredparse_amd import tred as tredmod
    from tredparse_amd.engine import Engine, Unit
    from tredparse_amd.models import format_call, pair_summary
    repo = TREDsRepo()
    loci = [l for l in synth.load_loci() if l["name"] in NAMES]
    p = synth.SynthParams(coverage=30, expanded_max=150, expanded_frac=0.4)
    made = sb.make_bams(str(tmp_path), 3, seed=500, loci=loci, p=p)
    engine = Engine(0)
    names = [l["name"] for l in loci]
    tasks = [(key, path, repo, names, 300, False, False, True, True, "ERROR") for key, path, _ in made]
    results = tredmod.run_many(tasks, engine, batch=2, threads=2)
    n_called = short_ok = 0
    for i, ((key, path, h_true), res) in enumerate(zip(made, results)):
        recs, h2 = sb.simulate_sample(500 + i, loci, p)
        assert np.array_equal(h_true, h2)
        units = []
        for l in loci:
            reads, depth, gl, tl = sb.expected_scan(recs, l, 150)
            units.append(Unit(repo[l["name"]], 150, [synth.decode(recs.codes[r]) for r in reads], depth, 2, gl, tl))
        direct = engine.genotype(units)
        calls = res["tredCalls"]
        assert calls["readLen"] == 150
        for l, u, r in zip(loci, units, direct):
            n = l["name"]
            want = format_call(repo[n], r)
            assert [calls[n + ".1"], calls[n + ".2"]] == want["alleles"], (key, n)
            for k in ("CI", "PP", "label", "P_h1", "P_h2", "P_h1h2"):
                assert calls[n + "." + k] == want[k], (key, n, k)
            assert calls[n + ".DP"] == u.depth
            for k, v in pair_summary(u.global_lens, u.target_lens).items():
                assert calls[n + "." + k] == v
            assert calls[n + ".RDP"] == r.rept and calls[n + ".FDP"] == sum(r.full.values())
            assert len(calls[n + ".details"]) == calls[n + ".FDP"] + calls[n + ".PDP"] + calls[n + ".RDP"]
            n_called += want["alleles"][0] > 0
            # the simulated short allele is recovered (the long one may exceed what 150 bp reads can size)
        short_ok += sum(calls[l["name"] + ".1"] == int(h) for l, h in zip(loci, h_true[:, 0]))
    assert n_called == 3 * len(loci)
    assert short_ok >= 0.6 * 3 * len(loci)         # the caller is right on ~80 % of such units (bench.py's check says the same)
    json.dumps(results)          # everything in the results is JSON-serialisable
    engine.close()


@pytest.mark.gpu
def test_cli_two_rank_processes_equal_one(tmp_path):
    """`tred.py --gpus 2` as real processes (both ranks on whatever devices are visible; one GPU is enough): every
    sample's JSON equals the one a single-process run writes, each file exactly once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    loci = [l for l in synth.load_loci() if l["name"] in NAMES]
    made = sb.make_bams(str(tmp_path / "bams"), 5, seed=900, loci=loci)
    (tmp_path / "list.txt").write_text("\n".join(p for _, p, _ in made) + "\n")
    base = [sys.executable, "-m", "tredparse_amd.tred", str(tmp_path / "list.txt"), "--cpus", "2", "--log", "INFO"]
    base += [x for n in NAMES for x in ("--tred", n)]
    env = dict(os.environ, PYTHONPATH=root)
    for name, extra in (("one", []), ("two", ["--gpus", "2"])):
        out = subprocess.run(base + ["--workdir", str(tmp_path / name)] + extra, cwd=str(tmp_path), env=env,
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.count('"samplekey"') == 5          # every sample's JSON echoed once
    files = sorted(os.listdir(tmp_path / "one"))
    assert files == sorted(os.listdir(tmp_path / "two")) and len(files) == 10
    for f in files:
        if f.endswith(".json"):
            a, b = (json.load(open(tmp_path / d / f)) for d in ("one", "two"))
            assert a == b, f
            assert set(n + ".1" for n in NAMES) <= set(a["tredCalls"])


@pytest.mark.gpu
def test_2x300bp_sample_keeps_every_locus(tmp_path):
    """A 2 x 300 bp sample through run(): no locus is dropped for its read length (round 2 lost every locus of such a
    sample), READLEN comes out as 300, and the per-read tags equal the oracle's for the reads the scan selected."""
    from oracle import pyoracle as po
    from tredparse_amd import _lib, bam_parser, tred as tredmod
    from tredparse_amd.engine import Engine
    repo = TREDsRepo()
    loci = [l for l in synth.load_loci() if l["name"] in NAMES]
    p = synth.SynthParams(coverage=20, readlen=300, ins_mean=500.0, ins_sd=60.0, max_units=70)
    made = sb.make_bams(str(tmp_path), 1, seed=930, loci=loci, p=p)
    key, path, h_true = made[0]
    names = [l["name"] for l in loci]
    engine = Engine(0)
    res = tredmod.run((key, path, repo, names, 300, False, False, True, True, "ERROR"), engine=engine)
    calls = res["tredCalls"]
    assert calls["readLen"] == 300
    assert all(n + ".1" in calls and calls[n + ".label"] != "missing" for n in names)
    assert sum(calls[n + ".1"] == int(h) for n, h in zip(names, h_true[:, 0])) >= 0.6 * len(names)
    scan = bam_parser.scan_sample(path, repo, names)
    assert not scan.dropped
    for k, l in enumerate(loci):
        a, b = scan.reads_of(k)
        reads = [scan.sequence(i) for i in range(a, b)]
        mu = -(-300 // len(l["repeat"]))
        cls = po.classify(reads, np.zeros(len(reads), np.int32), po.LocusSet([(l["prefix"], l["repeat"], l["suffix"], mu)]))
        want = sorted((int(h), _lib.TAG_NAMES[int(t)]) for t, h, _ in cls if int(t) in (1, 2, 3, 4))
        got = sorted((d["h"], d["tag"]) for d in calls[l["name"] + ".details"])
        assert got == want, l["name"]
    engine.close()


def test_optional_fields_and_stripped_sequences_do_not_change_what_a_scan_reads(tmp_path):
    """write_bam(aux=, no_seq=): the shapes real files have and the simulator's records lack -- optional fields behind every
    record, a record without its sequence -- leave every other record as it was: the scan of a file with optional fields equals
    the plain file's, field for field."""
    loci = [l for l in synth.load_loci() if l["name"] in ("HD", "DM1")]
    recs, _ = sb.simulate_sample(5, loci, synth.SynthParams(coverage=12))
    repo, names = TREDsRepo(), [l["name"] for l in loci]
    aux = b"NMC\x02MDZ75A74\x00ASC\x91RGZgroup1\x00"
    a, b = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    na = sb.write_bam(a, recs, sample="x")
    nb = sb.write_bam(b, recs, sample="x", aux=aux, split_records=True, block=777)
    assert nb == na + len(recs) * len(aux)
    sa, sc = scan_sample(a, repo, names), scan_sample(b, repo, names)
    assert np.array_equal(sa.read_len, sc.read_len) and np.array_equal(sa.seq4, sc.seq4) and sa.name_blob == sc.name_blob
    assert np.array_equal(sa.depth, sc.depth)
    for k in range(len(names)):
        assert all(np.array_equal(x, y) for x, y in zip(sa.pair_lengths(k), sc.pair_lengths(k)))


def test_trimmed_reads_round_trip_through_both_file_layers(tmp_path):
    """write_bam(lengths=) + trim_records: reads trimmed before alignment -- every record its own l_seq, its CIGAR's last
    operation shortened to match -- read back by both file layers: length, bases and query bases of the CIGAR agree."""
    loci = [l for l in synth.load_loci() if l["name"] in ("HD", "DM1")]
    recs, _ = sb.simulate_sample(5, loci, synth.SynthParams(coverage=10))
    rng = np.random.default_rng(3)
    want = np.where(rng.random(len(recs)) < 0.4, rng.integers(40, 151, len(recs)), 150)
    cut, ls = sb.trim_records(recs, want)
    assert (ls <= 150).all() and (ls < 150).sum() > len(recs) // 4 and recs.cig is not cut.cig
    path = str(tmp_path / "v.bam")
    sb.write_bam(path, cut, sample="v", lengths=ls, split_records=True, block=1234, aux=b"NMC\x01")
    for cls in (bamio.PyAlignmentFile, bamio.NativeAlignmentFile):
        f = cls(path)
        rows = list(f.fetch())
        f.close()
        assert len(rows) == len(recs)
        for i, r in enumerate(rows):
            assert r.query_length == ls[i] and r.query_sequence == synth.decode(recs.codes[i])[:ls[i]]
            assert not r.cigartuples or sum(n for op, n in r.cigartuples if op in (0, 1, 4, 7, 8)) == ls[i]     # (unmapped: no CIGAR)
    repo = TREDsRepo()
    s = scan_sample(path, repo, [l["name"] for l in loci])
    assert s.readlen == 150 and int(s.read_len.min()) < 150 and s.dropped == {}
