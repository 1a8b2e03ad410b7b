"""GPU: read selection, depth and 2-bit packing on the device (include/tredgpu.h section 5: select_kernel, pack_selected_kernel,
tredgpu_genotype_selected) against the host's scan (bamread.cpp scan_impl, which restates BamParser.parse's selection,
BamDepth and BamReadLen: tredparse/bam_parser.py:184-257, 372-411, and is itself pinned against the reference in
test_host_frontend.py / test_e2e_gpu.py): per sample the same sex, depth, reads in the same order with the same names and
sequences, the same pair-length slices -- and through the kernels the same tags, calls, marginals and joint entries,
bit for bit.  Then the product path: run_many with gpu_select against run_many on the host, result dict for result dict."""
import os
import sys

import numpy as np
import pytest

from tredparse_amd import synth, synth_bam, tred as t
from tredparse_amd.engine import Engine
from tredparse_amd.meta import TREDsRepo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def engine():
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def cohort(tmp_path_factory):
    root = tmp_path_factory.mktemp("select")
    loci = [l for l in synth.load_loci() if l["name"] in ("HD", "DM1", "SCA1", "AR", "FXS", "FRDA", "SCA17")]
    made = synth_bam.make_bams(str(root), 3, seed=77, loci=loci, p=synth.SynthParams(coverage=30, expanded_max=120, expanded_frac=0.3))
    repo = TREDsRepo(ref="hg38", sites=os.path.join(GOLD, "no_sites"))
    srepo = TREDsRepo()
    names = [l["name"] for l in loci]
    args = [(s, os.path.join(GOLD, "bam", s + ".bam"), repo, sorted(repo.names), 300, False, False, True, True, "ERROR") for s in ("t001", "t002")]
    args += [(key, path, srepo, names, 300, False, False, True, True, "ERROR") for key, path, _ in made]
    # blocks cut without regard to records (every record straddles the 300-byte ones), odd flags
    recs, _ = synth_bam.simulate_sample(78, loci[:4], synth.SynthParams(coverage=20, expanded_max=120, expanded_frac=0.3))
    rng = np.random.default_rng(78)
    recs.flag[rng.random(len(recs.flag)) < 0.03] |= 0x400
    recs.flag[rng.random(len(recs.flag)) < 0.02] |= 0x100
    recs.flag[rng.random(len(recs.flag)) < 0.02] |= 0x200
    recs.flag[rng.random(len(recs.flag)) < 0.05] ^= 0x10
    for block in (300, 20000):
        path = os.path.join(str(root), "cut{}.bam".format(block))
        synth_bam.write_bam(path, recs, sample="cut", block=block, split_records=True)
        args.append(("cut{}".format(block), path, srepo, names[:4], 300, False, False, True, True, "ERROR"))
    args.append(("noalts", os.path.join(GOLD, "bam", "t001.bam"), repo, ["HD", "DM1", "AR"], 300, False, False, False, True, "ERROR"))
    args.append(("clip", made[0][1], srepo, names, 300, False, True, True, True, "ERROR"))
    args.append(("full", made[1][1], srepo, names[:3], 60, True, False, True, True, "ERROR"))
    return args


sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import fuzz_select  # noqa: E402  (the comparison itself lives with the campaign: tools/fuzz_select.py)


def _device_scans(args, engine, batch):
    try:
        return fuzz_select.device_scans(args, engine, batch)
    finally:
        t.release_inflaters()


def test_selection_on_the_device_equals_the_host_scan(cohort, engine):
    for k in t.TIMING:
        t.TIMING[k] = 0
    got = _device_scans(cohort, engine, batch=4)
    assert t.TIMING["select_samples"] == len(cohort) and t.TIMING["select_declined"] == 0 and t.TIMING["walk_blocks_fetched"] == 0
    reads = units = 0
    for a, s, pieces in got:
        assert getattr(s, "device", None) is not None, a[0]
        bad, r, u = fuzz_select.compare(a, s, pieces, engine)
        assert not bad, (a[0], bad)
        reads += r
        units += u
    assert reads > 2500 and units > 80


def test_a_short_random_campaign(capsys, monkeypatch):
    """tools/fuzz_select.py, four rounds: random loci / coverage / read length / flags / options, blocks cut at random sizes
    without regard to records (the long campaign is profiles/r06_fuzz_select.json)."""
    import json
    monkeypatch.setattr(sys, "argv", ["fuzz_select.py", "4", "20261004"])
    fuzz_select.main()
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    # (the campaign strips a sequence or a CIGAR here and there: such samples come back to the host's scan -- most stay)
    assert out["samples"] >= 8 and out["on_device"] >= out["samples"] // 2 and out["units"] > 20
    assert out["mismatching_samples"] == 0, out["what"]


def test_what_the_device_cannot_serve_goes_to_the_host_scan(cohort, engine):
    """--norepeatpairs and --log DEBUG need more than the device path returns; a missing file, a file without the locus'
    contig: those samples are scanned on the host in the same chunk, the others stay on the device."""
    a0 = cohort[2]
    mixed = [cohort[2], cohort[2][:8] + (False, "ERROR"), cohort[3], cohort[3][:9] + ("DEBUG",),
             ("missing", os.path.join(os.path.dirname(cohort[2][1]), "no_such.bam")) + a0[2:]]
    got = _device_scans(mixed, engine, batch=5)
    where = [getattr(s, "device", None) is not None for _, s, _ in got]
    assert where == [True, False, True, False, False]
    assert got[4][1].opened is False
    (b0, i0, k0), = got[0][2]
    (b1, i1, k1), = got[1][2]
    n = len(k0)
    assert np.array_equal(b0.calls["h1"][i0:i0 + n], b1.calls["h1"][i1:i1 + n]) and np.array_equal(b0.calls["h2"][i0:i0 + n], b1.calls["h2"][i1:i1 + n])


@pytest.mark.parametrize("emit", [False, True])
def test_run_many_with_the_selection_on_the_device_gives_the_host_runs_results(cohort, engine, tmp_path, emit):
    """The product path: result dicts (and, with the native writer, the files' bytes) of run_many(gpu_select=True) against
    the host-only run_many, chunks of three so that chunks merge and inflaters rotate."""
    import hashlib
    args = list(cohort) + list(cohort[2:5])
    args = [(("s%02d_" % i) + a[0],) + a[1:] for i, a in enumerate(args)]
    if not emit:
        want = t.run_many(args, engine, batch=64, threads=2, lazy_details=False)
        got = t.run_many(args, engine, batch=3, threads=2, lazy_details=False, inflate_device=0, gpu_walk=True, gpu_select=True)
        assert len(got) == len(want) == len(args)
        for g, w in zip(got, want):
            assert g == w, g["samplekey"]
        return
    digests = []
    cwd = os.getcwd()
    for mode in ("host", "device"):
        work = tmp_path / mode
        work.mkdir()
        os.chdir(str(work))
        try:
            em = t.Emitter("hg38", args[2][2], args[2][3], workers=2)
            kw = dict(inflate_device=0, gpu_walk=True, gpu_select=True) if mode == "device" else {}
            # (one locus list per Emitter: the synthetic samples)
            t.run_many([a for a in args if a[3] == args[2][3]], engine, batch=3, threads=2, lazy_details=True, emit=em, **kw)
            em.close()
        finally:
            os.chdir(cwd)
        d = {}
        for name in sorted(os.listdir(str(work))):
            if name.endswith(".json"):
                d[name] = hashlib.sha256(open(str(work / name), "rb").read()).hexdigest()
        digests.append(d)
    assert len(digests[0]) >= 4 and digests[0] == digests[1]


def test_a_selected_record_without_a_sequence_drops_its_locus_on_either_path(engine, tmp_path):
    """SEQ '*' (l_seq 0): pysam's query_sequence is None and the reference's _parseReadSW dies in len(seq)
    (tredparse/bam_parser.py:129-133), which costs the sample that locus (tred.py:245-249).  The host's scan says so
    (TREDBAM_UNIT_NO_SEQ); the device's selection does not look for it, but the lengths it brings back show it, and such a
    sample is scanned by the host after all: the same result dicts, the locus missing from both."""
    loci = [l for l in synth.load_loci() if l["name"] in ("HD", "DM1", "SCA1", "FRDA")]
    recs, _ = synth_bam.simulate_sample(91, loci, synth.SynthParams(coverage=20))
    srepo = TREDsRepo()
    t0 = srepo["DM1"]
    inwin = np.nonzero((recs.locus == [l["name"] for l in loci].index("DM1")) & (recs.pos >= t0.repeat_start - 100) & (recs.pos <= t0.repeat_end + 100) &
                       ((recs.flag & 0x4) == 0))[0]
    assert len(inwin) > 3
    mask = np.zeros(len(recs), bool)
    mask[inwin[1]] = True
    plain, odd = str(tmp_path / "plain.bam"), str(tmp_path / "noseq.bam")
    synth_bam.write_bam(plain, recs, sample="p")
    synth_bam.write_bam(odd, recs, sample="p", no_seq=mask)
    names = [l["name"] for l in loci]
    args = [("plain", plain, srepo, names, 300, False, False, True, True, "ERROR"), ("noseq", odd, srepo, names, 300, False, False, True, True, "ERROR")]
    for k in t.TIMING:
        t.TIMING[k] = 0
    want = t.run_many(args, engine, batch=64, threads=2, lazy_details=False)
    got = t.run_many(args, engine, batch=2, threads=2, lazy_details=False, inflate_device=0, gpu_walk=True, gpu_select=True)
    t.release_inflaters()
    assert t.TIMING["select_samples"] == 2                      # (both were selected on the device; one came back to the host)
    assert got == want
    assert "DM1.1" in want[0]["tredCalls"] and "DM1.1" not in want[1]["tredCalls"]
    assert all(n + ".1" in want[1]["tredCalls"] for n in names if n != "DM1")
