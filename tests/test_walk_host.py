"""run_many(gpu_walk=True) on the host side, without a GPU: the feeder's plan -> decode + walk -> fetch -> scan pipeline with
tests/walk_model.ModelInflater in the inflater's place (zlib and a plain Python walk over the task tables).  Every scan
must equal the plain scan_sample of the same file field for field, having read only blocks the model was asked to fetch --
this pins tredbam_plan_walks / tredbam_plan_blocks / tredbam_scan_pe and the table arithmetic of _InflateFeeder; the
kernel's own parity is test_pairwalk_gpu.py's business."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from tredparse_amd import bamio, synth, synth_bam
from tredparse_amd import tred as t
from tredparse_amd.bam_parser import scan_sample
from tredparse_amd.meta import TREDsRepo

from .walk_model import ModelInflater

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIELDS = ("packed", "word_off", "read_len", "seq4", "seq4_off", "name_blob", "name_off", "name_id", "global_lens", "target_lens",
          "depth", "ploidy")


def _same(a, b):
    assert a.opened == b.opened and a.gender == b.gender and a.readlen == b.readlen
    if not a.opened:
        return
    assert a.dropped == b.dropped
    for key in a.unit.dtype.names:
        assert (a.unit[key] == b.unit[key]).all(), key
    for key in FIELDS:
        x, y = getattr(a, key), getattr(b, key)
        assert (x == y) if isinstance(x, bytes) else np.array_equal(x, y), key


@pytest.fixture(scope="module")
def cohort(tmp_path_factory):
    root = tmp_path_factory.mktemp("walkhost")
    loci = [l for l in synth.load_loci() if l["name"] in ("HD", "DM1", "SCA1", "AR")]
    made = synth_bam.make_bams(str(root), 2, seed=5, loci=loci, p=synth.SynthParams(coverage=12, expanded_max=120, expanded_frac=0.3))
    repo = TREDsRepo(ref="hg38", sites=os.path.join(GOLD, "no_sites"))
    srepo = TREDsRepo()
    args = [(s, os.path.join(GOLD, "bam", s + ".bam"), repo, sorted(repo.names), 300, False, False, True, True, "ERROR") for s in ("t001", "t002")]
    args += [(key, path, srepo, [l["name"] for l in loci], 300, False, False, True, True, "ERROR") for key, path, _ in made]
    # blocks cut without regard to records: every record straddles the 300-byte ones
    recs, _ = synth_bam.simulate_sample(6, loci[:2], synth.SynthParams(coverage=8, expanded_max=120, expanded_frac=0.3))
    rng = np.random.default_rng(6)
    recs.flag[rng.random(len(recs.flag)) < 0.03] |= 0x400
    recs.flag[rng.random(len(recs.flag)) < 0.05] ^= 0x10
    for block in (300, 20000):
        path = os.path.join(str(root), "cut{}.bam".format(block))
        synth_bam.write_bam(path, recs, sample="cut", block=block, split_records=True)
        args.append(("cut{}".format(block), path, srepo, [l["name"] for l in loci[:2]], 300, False, False, True, True, "ERROR"))
    # the kernel-side options that change what a scan reads: --noalts (no alternative loci), --useclippedreads (neither)
    args.append(("noalts", os.path.join(GOLD, "bam", "t001.bam"), repo, ["HD", "DM1", "AR"], 300, False, False, False, True, "ERROR"))
    args.append(("clip", made[0][1], srepo, [l["name"] for l in loci], 300, False, True, True, True, "ERROR"))
    args.append(("missing", os.path.join(str(root), "no_such.bam"), repo, ["HD"], 300, False, False, True, True, "ERROR"))
    return args


def test_feeder_with_walked_pair_lengths_gives_the_plain_scans(cohort, monkeypatch):
    if not hasattr(bamio.AlignmentFile(cohort[0][1]), "plan_walks"):
        pytest.skip("no native BAM layer")
    monkeypatch.setattr("tredparse_amd._lib.Inflater", ModelInflater)
    t.release_inflaters()
    del ModelInflater.made[:]
    for k in t.TIMING:
        t.TIMING[k] = 0
    chunks = [cohort[:2], cohort[2:4], cohort[4:6], cohort[6:8], cohort[8:]]
    ex = ThreadPoolExecutor(max_workers=2)
    feeder = t._InflateFeeder(chunks, ex, 0, walk=True)
    try:
        scans = [fut.result() for _ in chunks for fut in feeder.next()[1]]
    finally:
        feeder.close()
        ex.shutdown()
        t.release_inflaters()
    assert len(scans) == len(cohort)
    for a, s in zip(cohort, scans):
        o = t._options(a)
        _same(s, scan_sample(o["bam"], o["repo"], o["names"], clip=o["clip"], alts=o["alts"]))
    tm = t.TIMING
    assert tm["walk_regions"] == sum(len(a[3]) for a in cohort[:8]) and tm["walk_declined"] == 0
    assert 0 < tm["walk_blocks_fetched"] < tm["inflate_blocks"]
    assert tm["walk_alt_regions"] > 100 and tm["walk_alt_declined"] == 0   # the alternative loci were walked by the model too
    assert tm["inflate_misses"] == 0 and tm["inflate_hits"] > 0          # no scan inflated a block for itself
    assert sum(m.walks for m in ModelInflater.made) == 4                 # (the chunk with the missing file has nothing to decode)


def test_decoded_chunks_that_pile_up_share_a_genotyping_call(cohort, monkeypatch):
    """A driver whose genotyping call is slower than its decoder takes every chunk whose scans are already in along in the
    next call (run_many, MERGED_BATCH): fewer calls, every sample once, in cohort order -- and it never waits for a chunk
    to do so (with a fast engine nothing is merged)."""
    import time
    from tests.fake_engine import FakeEngine
    if not hasattr(bamio.AlignmentFile(cohort[0][1]), "plan_walks"):
        pytest.skip("no native BAM layer")
    monkeypatch.setattr("tredparse_amd._lib.Inflater", ModelInflater)

    class Slow(FakeEngine):
        def genotype_packed(self, b, dense=False):
            time.sleep(0.6)
            return FakeEngine.genotype_packed(self, b, dense)
    tasks = [a for a in cohort if a[0] != "missing"] * 2
    for engine, expect_merge in ((Slow(seed=3), True), (FakeEngine(seed=3), None)):
        t.release_inflaters()
        for k in t.TIMING:
            t.TIMING[k] = 0
        seen = []
        t.run_many(tasks, engine, batch=2, threads=2, lazy_details=True, sink=lambda r: seen.append(r["samplekey"]),
                   inflate_device=0, gpu_walk=True)
        assert seen == [a[0] for a in tasks]
        if expect_merge:
            assert t.TIMING["merged_chunks"] >= 2
    t.release_inflaters()


def test_regions_the_walker_declines_are_walked_by_the_scan(cohort, monkeypatch):
    """Every region comes back with a status (here: the model pretends a damaged block in each): the scans compute the
    pair lengths themselves, inflating what was not fetched, and nothing changes in the result."""
    class Declining(ModelInflater):
        def run_walk(self, n, bcoff, bclen, xcrc, tasks, chunks, pairs_per_task=2048, alt_tasks=None, alt_chunks=None, pool_pairs=None):
            status, crc, res, gp, tp, ares, need = ModelInflater.run_walk(self, n, bcoff, bclen, xcrc, tasks, chunks, alt_tasks=alt_tasks,
                                                                          alt_chunks=alt_chunks, pool_pairs=pool_pairs)
            res["status"][::2] = 2
            ares["status"][1::3] = 2
            return status, crc, res, gp, tp, ares, need

    monkeypatch.setattr("tredparse_amd._lib.Inflater", Declining)
    t.release_inflaters()
    ex = ThreadPoolExecutor(max_workers=2)
    feeder = t._InflateFeeder([cohort[:4]], ex, 0, walk=True)
    try:
        scans = [fut.result() for fut in feeder.next()[1]]
    finally:
        feeder.close()
        ex.shutdown()
        t.release_inflaters()
    for a, s in zip(cohort[:4], scans):
        o = t._options(a)
        _same(s, scan_sample(o["bam"], o["repo"], o["names"], clip=o["clip"], alts=o["alts"]))


def test_walk_need_marks_the_windows_blocks_and_what_the_flags_say():
    """bam_parser.walk_need: blocks between a walked site's window offsets (a position at a block's very start needs
    the block before it only), the extra regions' blocks always, the alternative loci's only when those were not walked
    elsewhere -- then the flags that walk returned instead."""
    from tredparse_amd.bam_parser import walk_need
    coff = np.array([100, 200, 300, 400, 500, 600], np.int64)
    host = np.array([1, 0, 0, 2, 0, 3], np.uint8)              # bit 0: alternative loci, bit 1: extra regions
    res = np.zeros(3, bamio.WALK_RESULT_DTYPE)
    res[0] = (0, 0, 0, 5, 0, 0, (200 << 16) | 17, (300 << 16) | 9)      # records in blocks 200 .. 300
    res[1] = (0, 0, 0, 2, 0, 0, (500 << 16) | 3, 600 << 16)             # ends exactly where block 600 starts: 500 only
    res[2] = (2, 0, 0, 7, 0, 0, (100 << 16), (600 << 16) | 1)           # declined: its offsets mean nothing
    assert walk_need(coff, host, res).tolist() == [1, 1, 1, 1, 1, 1]
    assert walk_need(coff, host, res[2:]).tolist() == [1, 0, 0, 1, 0, 1]
    alt_need = np.array([0, 0, 0, 0, 0, 0], np.uint8)
    assert walk_need(coff, host, res, alt_need).tolist() == [0, 1, 1, 1, 1, 1]
    alt_need[0] = 1
    assert walk_need(coff, host, res[:1], alt_need).tolist() == [1, 1, 1, 1, 0, 1]
    none = np.zeros(0, bamio.WALK_RESULT_DTYPE)
    assert walk_need(coff, host, none, np.zeros(6, np.uint8)).tolist() == [0, 0, 0, 1, 0, 1]


def test_pool_bound_covers_any_coverage():
    """the pair pools were 2 048 entries per task and overflowed silently from ~32x on.  The bound now comes
    from the call's inflated bytes: two records of at least MIN_PAIR_BYTES / 2 per pair, a byte in the regions of at most
    two loci -- whatever the coverage."""
    from tredparse_amd import _lib
    tasks = np.zeros(60, _lib.WALK_TASK_DTYPE)
    for cov in (30, 40, 100, 300):
        reads = int(20000 * cov / 150) * 60                     # 150 bp reads over 60 regions of +-10 kb
        out_bytes = reads * 290                                 # (a record of such a read: ~290 bytes)
        off = np.array([0, out_bytes], np.int64)
        assert _lib.walk_pool_pairs(tasks, off) >= reads // 2    # every read paired: the most pairs there can be
    assert _lib.walk_pool_pairs(tasks[:0], np.zeros(1, np.int64)) == 0


def test_plan_drivers_follows_the_rule_and_the_cohort_size():
    import argparse
    from tredparse_amd import shard
    a = argparse.Namespace(gpus=1, drivers="auto", cpus=None, gpu_inflate=True)
    assert t.plan_drivers(a, 5000, 16) == shard.driver_plan(16, 1) == (3, 5)
    assert t.plan_drivers(a, 2, 16) == (1, 16)                  # a handful of samples: one process, all the threads
    assert t.plan_drivers(a, 70, 16) == (2, 8)                  # never more drivers than blocks of 32 samples
    a.gpu_inflate = False
    assert t.plan_drivers(a, 5000, 16) == (3, 5)                # host decoding: a driver per five CPUs
    a.drivers, a.cpus = "2", 6
    assert t.plan_drivers(a, 5000, 16) == (2, 6)
    a8 = argparse.Namespace(gpus=8, drivers="auto", cpus=None, gpu_inflate=True)
    assert t.plan_drivers(a8, 8000, 128) == (3, 5)


def test_native_scan_checks_the_pair_length_slices_itself():
    """tredbam_scan_pe read pe_global + global_first without knowing how long the pool is; only the Python
    wrapper checked.  tredbam_pe_pool_sizes hands the lengths over and the library refuses a slice outside them."""
    import ctypes as C
    from tredparse_amd.bam_parser import DNAPE_ELONGATE, FLANKMATCH, SPAN, _site_arrays
    from tredparse_amd.meta import TREDsRepo
    repo = TREDsRepo("hg38", sites=os.path.join(GOLD, "no_sites"))
    f = bamio.AlignmentFile(os.path.join(GOLD, "bam", "t001.bam"))
    if not hasattr(f, "plan_walks"):
        pytest.skip("no native BAM layer")
    names = ["HD"]
    sites, regions = _site_arrays(repo, names, [repo[n] for n in names], f)
    res = np.zeros(1, bamio.WALK_RESULT_DTYPE)
    res[0] = (0, 5, 2, 0, 0, 0, 0, 0)                           # five global and two target lengths, "from elsewhere"
    gp, tp = np.arange(5, dtype=np.int32) + 300, np.arange(2, dtype=np.int32) + 400
    units, pools = f.scan(sites, regions, 150, pad=SPAN, flank=FLANKMATCH, pe_reach=DNAPE_ELONGATE, span=SPAN, pe=(res, gp, tp))
    assert pools["global_lens"].tolist() == gp.tolist() and pools["target_lens"].tolist() == tp.tolist()
    # the same call straight at the library with pools declared SHORTER than the slices: refused, nothing read
    o = bamio.ScanOpts(150, SPAN, FLANKMATCH, DNAPE_ELONGATE, SPAN, 1, 1, 1)
    out = np.zeros(1, bamio.SCAN_UNIT_DTYPE)
    alts = regions if len(regions) else np.zeros(1, bamio.REGION_DTYPE)
    assert f._lib.tredbam_pe_pool_sizes(f._h, 4, 2) == 0
    rc = f._lib.tredbam_scan_pe(f._h, sites.ctypes.data, 1, alts.ctypes.data, C.byref(o), res.ctypes.data, gp.ctypes.data, tp.ctypes.data,
                                out.ctypes.data)
    assert rc == -2 and "outside the pools" in f._err()
    assert f._lib.tredbam_pe_pool_sizes(f._h, 5, 2) == 0
    assert f._lib.tredbam_scan_pe(f._h, sites.ctypes.data, 1, alts.ctypes.data, C.byref(o), res.ctypes.data, gp.ctypes.data, tp.ctypes.data,
                                  out.ctypes.data) == 0
    f.close()


def test_feeder_with_the_selection_on_the_device_gives_the_plain_scans(cohort, monkeypatch):
    """run_many's gpu_select plumbing without a GPU: tests/walk_model.ModelInflater also models the device's read selection
    (csrc/walk.hip select_kernel: the window's picks and depth sum, the alternative regions' hits, the chrY region tasks) and
    the fake engine fills the reads in as tredgpu_genotype_selected does.  Every sample the device may serve comes back as a
    SampleScan whose sex, depths, read counts, pair-length slices, read lengths, sequences and names are the plain scan's;
    the others (--useclippedreads runs on the device too; the missing file does not) are scanned on the host, in the same chunk."""
    from tests.fake_engine import FakeEngine
    if not hasattr(bamio.AlignmentFile(cohort[0][1]), "plan_walks"):
        pytest.skip("no native BAM layer")
    monkeypatch.setattr("tredparse_amd._lib.Inflater", ModelInflater)
    t.release_inflaters()
    for k in t.TIMING:
        t.TIMING[k] = 0
    tasks = list(cohort) + [cohort[2][:8] + (False, "ERROR")]            # (--norepeatpairs: the host's)
    chunks = [tasks[:3], tasks[3:6], tasks[6:]]
    ex = ThreadPoolExecutor(max_workers=2)
    feeder = t._InflateFeeder(chunks, ex, 0, walk=True, select=True)
    engine = FakeEngine(seed=5, odd_units=False)
    got = []
    try:
        for _ in chunks:
            chunk, futs = feeder.next()
            scans = [f.result() for f in futs]
            picks, parts = t.genotype_scans(engine, chunk, scans)
            got += list(zip(chunk, scans))
            assert sorted(parts) == [si for si, s in enumerate(scans) if s.opened]
    finally:
        feeder.close()
        ex.shutdown()
        t.release_inflaters()
    on_device = [getattr(s, "device", None) is not None for _, s in got]
    assert on_device == [a[0] != "missing" and a[8] for a in tasks]
    assert t.TIMING["select_samples"] == sum(on_device) and t.TIMING["select_declined"] == 0
    for (a, s), dev in zip(got, on_device):
        o = t._options(a)
        plain = scan_sample(o["bam"], o["repo"], o["names"], clip=o["clip"], alts=o["alts"])
        if not dev:
            _same(s, plain)
            continue
        assert (s.gender, s.ydepth, s.readlen, s.dropped) == (plain.gender, plain.ydepth, plain.readlen, plain.dropped), a[0]
        for key in s.unit.dtype.names:
            assert (s.unit[key] == plain.unit[key]).all() or key in ("global_first", "target_first"), (a[0], key)
        for key in ("read_len", "seq4", "seq4_off", "name_blob", "name_off", "depth", "ploidy"):
            x, y = getattr(s, key), getattr(plain, key)
            assert (x == y) if isinstance(x, bytes) else np.array_equal(x, y), (a[0], key)
        for k in range(len(s.names)):
            for x, y in zip(s.pair_lengths(k), plain.pair_lengths(k)):
                assert np.array_equal(x, y), (a[0], k)


def test_a_failed_call_over_device_selected_reads_falls_back_to_the_host_scan(cohort, monkeypatch):
    """tredgpu_genotype_selected fails as a whole (here: the engine raises): the chunk's samples are scanned on the host after all
    and genotyped the host-packed way -- every sample still comes out, with the keys of the host-only run."""
    from tests.fake_engine import FakeEngine
    from tredparse_amd import _lib
    if not hasattr(bamio.AlignmentFile(cohort[0][1]), "plan_walks"):
        pytest.skip("no native BAM layer")
    monkeypatch.setattr("tredparse_amd._lib.Inflater", ModelInflater)
    t.release_inflaters()

    class Failing(FakeEngine):
        def genotype_selected(self, scans, **kw):
            raise _lib.TredGpuError("the device call failed")
    tasks = [a for a in cohort if a[0] != "missing"][:5]
    got = t.run_many(tasks, Failing(seed=2, odd_units=False), batch=3, threads=2, lazy_details=False, inflate_device=0, gpu_walk=True, gpu_select=True)
    want = t.run_many(tasks, FakeEngine(seed=2, odd_units=False), batch=3, threads=2, lazy_details=False)
    t.release_inflaters()
    assert [g["samplekey"] for g in got] == [a[0] for a in tasks]
    for g, w in zip(got, want):
        assert set(g["tredCalls"]) == set(w["tredCalls"]) and g["tredCalls"]["inferredGender"] == w["tredCalls"]["inferredGender"]
        for k, v in w["tredCalls"].items():
            if k.endswith((".DP", ".PEDP", ".PEG", ".PET")) or k in ("depthY", "readLen"):
                assert g["tredCalls"][k] == v, k
