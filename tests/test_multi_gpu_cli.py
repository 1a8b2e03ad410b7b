"""CPU: the product CLI's --gpus N fan-out (tredparse/tred.py:521-532 is the reference's per-sample Pool).
The parent fixes the sample list and who takes which sample (balanced by BAM size), hands both to N children in a
task file, child r takes its samples and writes
those samples' JSON / VCF.  Here the children run in-process with a stand-in engine (no GPU): every sample is
written exactly once, by the rank that owns it."""
import gzip
import json
import os

import numpy as np
import pytest

from tredparse_amd import _lib, shard, tred as tredmod
from tredparse_amd.engine import BatchResult

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class NoEvidenceEngine(object):
    """Stands in for engine.Engine: every unit comes back as "no evidence" (status 1)."""
    made = []

    def __init__(self, device=0):
        NoEvidenceEngine.made.append(device)

    def close(self):
        pass

    def genotype_packed(self, b):
        r = BatchResult()
        r.batch = b
        r.tag, r.h, r.score = np.zeros(b.n_reads, np.uint8), np.zeros(b.n_reads, np.int16), np.zeros(b.n_reads, np.int16)
        r.full = r.pref = r.rept = np.zeros((b.n_units, 4), np.int32)
        r.calls = np.zeros(b.n_units, _lib.CALL_DTYPE)
        r.calls["status"] = 1
        r.marg = np.zeros((b.n_units, 2, 4))
        r.joint = [None] * b.n_units
        return r


def test_two_ranks_write_every_sample_exactly_once(tmp_path, monkeypatch):
    monkeypatch.setattr("tredparse_amd.engine.Engine", NoEvidenceEngine)
    monkeypatch.chdir(tmp_path)
    bams = [os.path.join(GOLD, "bam", b) for b in ("t001.bam", "t002.bam")]
    rows = ["k{:02d},{}".format(i, bams[i % 2]) for i in range(5)]
    (tmp_path / "samples.csv").write_text("\n".join(rows) + "\n")
    work = tmp_path / "work"
    written = []

    def fake_spawn(cmd, world, devices, env=None, cwd=None, **kw):
        """Run the children one after the other in this process, each with its rank environment."""
        assert cmd[1:3] == ["-m", "tredparse_amd.tred"] and "--task-file" in cmd
        assert cwd == str(tmp_path)
        for r in range(world):
            renv = shard.rank_env(r, world, 1, r % devices, base={})
            assert renv["HIP_VISIBLE_DEVICES"] == str(r % devices)
            for k in ("RANK", "WORLD_SIZE", "TRED_SPAWNED_RANK"):
                monkeypatch.setenv(k, renv[k])
            before = set(os.listdir(work)) if work.exists() else set()
            os.chdir(cwd)
            tredmod.main(cmd[3:], quiet=True)
            written.append(sorted(set(os.listdir(work)) - before))
        for k in ("RANK", "WORLD_SIZE", "TRED_SPAWNED_RANK"):
            monkeypatch.delenv(k)
        return [0] * world

    monkeypatch.setattr(shard, "spawn_ranks", fake_spawn)
    monkeypatch.setattr(shard, "visible_gpus", lambda: 2)
    tredmod.main(["samples.csv", "--workdir", str(work), "--gpus", "2", "--tred", "HD", "--tred", "DM1", "--cpus", "2"],
                 quiet=True)
    # balanced by BAM size (t001.bam is the larger file: k00, k02, k04): rank 0 takes k00 and k04, rank 1 k02 and the
    # two smaller ones; nothing twice, nothing missing
    assert [sorted(set(f.split(".")[0] for f in w)) for w in written] == [["k00", "k04"], ["k01", "k02", "k03"]]
    assert sorted(os.listdir(work)) == sorted(k + s for k in ("k00", "k01", "k02", "k03", "k04")
                                              for s in (".json", ".tred.vcf.gz"))
    assert NoEvidenceEngine.made[-2:] == [0, 0]            # each child sees its one device as index 0
    js = json.load(open(work / "k03.json"))
    assert js["samplekey"] == "k03" and js["tredCalls"]["HD.label"] == "missing" and js["tredCalls"]["HD.1"] == -1
    assert js["tredCalls"]["DM1.DP"] > 0 and js["tredCalls"]["HD.DP"] == 0 and js["tredCalls"]["HD.PEDP"] == 0   # t002
    vcf = gzip.open(work / "k00.tred.vcf.gz", "rt").read().splitlines()
    assert [l.split("\t")[2] for l in vcf if not l.startswith("#")] == ["DM1", "HD"]       # chr19 sorts before chr4
    assert vcf[-1].split("\t")[4] == "." and vcf[-1].split("\t")[9].startswith("1/1:-1/-1:")


def test_checkexists_is_decided_by_the_parent(tmp_path, monkeypatch):
    """--checkexists with --gpus: the parent drops finished samples BEFORE partitioning, the children never re-check
    (a child seeing a sibling's fresh JSON must not shift the shard boundaries)."""
    monkeypatch.setattr("tredparse_amd.engine.Engine", NoEvidenceEngine)
    monkeypatch.chdir(tmp_path)
    bam = os.path.join(GOLD, "bam", "t001.bam")
    (tmp_path / "list.txt").write_text("\n".join([bam] * 1) + "\n")
    seen = {}

    def fake_spawn(cmd, world, devices, env=None, cwd=None, **kw):
        seen["tasks"] = json.load(open(cmd[cmd.index("--task-file") + 1]))
        return [0] * world

    monkeypatch.setattr(shard, "spawn_ranks", fake_spawn)
    monkeypatch.setattr(shard, "visible_gpus", lambda: 1)
    (tmp_path / "t001.json").write_text("{}")
    tredmod.main(["list.txt", "--gpus", "2", "--checkexists"], quiet=True)
    assert "tasks" not in seen                              # nothing left to do: no children started
    os.remove(tmp_path / "t001.json")
    tredmod.main(["list.txt", "--gpus", "2", "--checkexists"], quiet=True)
    assert seen["tasks"] == {"samples": [["t001", bam, None]], "owner": [0]}


def test_cleanup_with_gpus_is_the_parents(tmp_path, monkeypatch, capsys):
    """--gpus N --cleanup: the children never see --cleanup (the first one to finish would remove the directory its
    siblings are still writing to); the parent echoes the JSONs and only then removes the working directory."""
    monkeypatch.setattr("tredparse_amd.engine.Engine", NoEvidenceEngine)
    monkeypatch.chdir(tmp_path)
    bams = [os.path.join(GOLD, "bam", b) for b in ("t001.bam", "t002.bam")]
    (tmp_path / "samples.csv").write_text("".join("c{},{}\n".format(i, bams[i % 2]) for i in range(3)))
    work = tmp_path / "work"
    seen = []

    def fake_spawn(cmd, world, devices, env=None, cwd=None, **kw):
        assert "--cleanup" not in cmd
        parent_cwd = os.getcwd()                     # real children are processes: the parent's directory stays put
        for r in range(world):
            for k, v in (("RANK", str(r)), ("WORLD_SIZE", str(world)), ("TRED_SPAWNED_RANK", "1")):
                monkeypatch.setenv(k, v)
            os.chdir(cwd)
            tredmod.main(cmd[3:], quiet=True)
            assert work.exists()                     # a finished child leaves the directory alone
            seen.append(sorted(os.listdir(work)))
        for k in ("RANK", "WORLD_SIZE", "TRED_SPAWNED_RANK"):
            monkeypatch.delenv(k)
        os.chdir(parent_cwd)
        return [0] * world

    monkeypatch.setattr(shard, "spawn_ranks", fake_spawn)
    monkeypatch.setattr(shard, "visible_gpus", lambda: 2)
    tredmod.main(["samples.csv", "--workdir", str(work), "--gpus", "2", "--tred", "HD", "--cleanup"])
    assert len(seen) == 2 and len(seen[1]) == 6      # both ranks' files were all there before the parent cleaned up
    echoed = capsys.readouterr().out
    assert [echoed.count('"samplekey": "c{}"'.format(i)) for i in range(3)] == [1, 1, 1]
    assert not work.exists()


def test_child_invoked_with_cleanup_keeps_the_directory(tmp_path, monkeypatch):
    """Belt and braces: even a child that is handed --cleanup (an older parent) does not remove the directory."""
    monkeypatch.setattr("tredparse_amd.engine.Engine", NoEvidenceEngine)
    monkeypatch.chdir(tmp_path)
    bam = os.path.join(GOLD, "bam", "t001.bam")
    tasks = tmp_path / "tasks.json"
    tasks.write_text(json.dumps([["z0", bam, None]]))
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1")):
        monkeypatch.setenv(k, v)
    work = tmp_path / "w"
    tredmod.main([bam, "--workdir", str(work), "--tred", "HD", "--cleanup", "--task-file", str(tasks)], quiet=True)
    assert (work / "z0.json").exists()


def test_rank_devices_follow_an_inherited_visibility_mask():
    """A parent confined to HIP_VISIBLE_DEVICES=2,3 (or CUDA_VISIBLE_DEVICES) counts 2 devices; its children must get
    entries 2 and 3 of that mask, not the absolute devices 0 and 1."""
    env = shard.rank_env(0, 2, 1234, 0, base={"HIP_VISIBLE_DEVICES": "2,3"})
    assert env["HIP_VISIBLE_DEVICES"] == "2" and env["TRED_RANK_DEVICE"] == "2" and env["LOCAL_RANK"] == "0"
    assert shard.rank_env(1, 2, 1234, 1, base={"HIP_VISIBLE_DEVICES": "2,3"})["HIP_VISIBLE_DEVICES"] == "3"
    assert shard.rank_env(2, 4, 1234, 2 % 2, base={"HIP_VISIBLE_DEVICES": "2,3"})["HIP_VISIBLE_DEVICES"] == "2"
    env = shard.rank_env(1, 2, 1234, 1, base={"CUDA_VISIBLE_DEVICES": "5, 7"})
    assert env["HIP_VISIBLE_DEVICES"] == "7" and "CUDA_VISIBLE_DEVICES" not in env
    env = shard.rank_env(1, 2, 1234, 1, base={"HIP_VISIBLE_DEVICES": "4,6", "CUDA_VISIBLE_DEVICES": "0,1"})
    assert env["HIP_VISIBLE_DEVICES"] == "6" and "CUDA_VISIBLE_DEVICES" not in env
    env = shard.rank_env(3, 8, 1234, 3, base={"ROCR_VISIBLE_DEVICES": "0,1,2,3"})
    assert env["HIP_VISIBLE_DEVICES"] == "3" and env["ROCR_VISIBLE_DEVICES"] == "0,1,2,3"
    assert shard.rank_env(1, 2, 1234, None, base={"HIP_VISIBLE_DEVICES": "2,3"})["HIP_VISIBLE_DEVICES"] == "2,3"


def test_tasks_with_different_options_go_in_separate_gpu_batches(tmp_path, monkeypatch):
    """finish_batch groups a batch's tasks by the kernel-side options (maxinsert, fullsearch, clip, repeatpairs)
    instead of applying the first task's to all."""
    from tredparse_amd.meta import TREDsRepo
    seen = []

    class Recording(NoEvidenceEngine):
        def genotype_packed(self, b):
            seen.append((int(b.params["maxinsert"][0]), int(b.params["fullsearch"][0]), b.n_units))
            assert len(set(b.params["maxinsert"].tolist())) == 1 and len(set(b.params["fullsearch"].tolist())) == 1
            return NoEvidenceEngine.genotype_packed(self, b)

    repo = TREDsRepo(ref="hg38", sites=str(tmp_path / "no_sites"))
    bam = os.path.join(GOLD, "bam", "t001.bam")
    args = [("a", bam, repo, ["HD", "DM1"], 300, False, False, True, True, "INFO"),
            ("b", bam, repo, ["HD"], 60, True, False, True, True, "INFO"),
            ("c", bam, repo, ["HD", "DM1"], 300, False, False, True, True, "INFO")]
    scans = [tredmod.collect_sample(a) for a in args]
    res = tredmod.finish_batch(Recording(), args, scans)
    assert [r["samplekey"] for r in res] == ["a", "b", "c"]
    assert sorted(seen) == [(60, 1, 1), (300, 0, 4)]


def test_results_reach_the_sink_in_order_from_the_writer_thread(tmp_path, monkeypatch):
    """run_many(background_sink=True): the sink runs on one writer thread, every result once and in task order; an
    exception in the sink surfaces in the driver."""
    import threading
    from tredparse_amd.meta import TREDsRepo
    repo = TREDsRepo(ref="hg38", sites=str(tmp_path / "no_sites"))
    bams = [os.path.join(GOLD, "bam", b) for b in ("t001.bam", "t002.bam")]
    args = [("w{}".format(i), bams[i % 2], repo, ["HD", "DM1"], 300, False, False, True, True, "INFO") for i in range(7)]
    seen, threads = [], set()

    def sink(r):
        seen.append(r["samplekey"])
        threads.add(threading.current_thread().name)
    out = tredmod.run_many(args, NoEvidenceEngine(), batch=3, sink=sink, threads=2, background_sink=True)
    assert out == [] and seen == ["w{}".format(i) for i in range(7)] and threads == {"tred-writer"}

    def bad(r):
        raise OSError("disk full")
    with pytest.raises(OSError, match="disk full"):
        tredmod.run_many(args[:2], NoEvidenceEngine(), batch=2, sink=bad, threads=1, background_sink=True)


def test_inflate_feeder_stops_when_the_consumer_fails(tmp_path, monkeypatch):
    """run_many(inflate_device=...) feeds chunks through plan -> GPU inflate -> scan on threads of its own.
    When the consumer dies half way (here: the sink raises on the second sample) the feeder must stop at once -- not sit
    in a full queue until a timeout --, every planned handle must be closed, the inflaters released after the scans that
    read them, and the exception that reaches the caller must be the consumer's own.  A stand-in inflater (no GPU)
    counts what happens to it."""
    import threading
    import time
    from tredparse_amd import tred as t
    from tredparse_amd.meta import TREDsRepo

    class FakeInflater(object):
        made, closed, runs = [], [], []

        def __init__(self, device=0, host_out=True):
            FakeInflater.made.append(self)
            self.comp_addr = self.out_addr = 0
            self.host_out = host_out

        def reserve(self, cb, ob, n):
            self.bufs = (np.zeros(cb + 64, np.uint8), np.zeros(ob + 64, np.uint8), np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64))
            self.comp_addr, self.out_addr = self.bufs[0].ctypes.data, self.bufs[1].ctypes.data
            return self.bufs[0][:cb], self.bufs[1][:ob], self.bufs[2], self.bufs[3]

        def run(self, n, crc=False):
            FakeInflater.runs.append(n)
            time.sleep(0.05)
            status = np.full(n, -1, np.int32)                  # every block "refused": the scans inflate for themselves
            return (status, np.zeros(n, np.uint32)) if crc else status

        def close(self):
            FakeInflater.closed.append(self)

    monkeypatch.setattr("tredparse_amd._lib.Inflater", FakeInflater)
    t.release_inflaters()                                       # (none of another test's in the process's pool)
    repo = TREDsRepo("hg38", sites=os.path.join(GOLD, "no_sites"))
    bams = [os.path.join(GOLD, "bam", b) for b in ("t001.bam", "t002.bam")]
    tasks = [("s{:02d}".format(i), bams[i % 2], repo, ["HD", "DM1"], 300, False, False, True, True, "ERROR") for i in range(40)]
    seen = []

    def sink(result):
        seen.append(result["samplekey"])
        if len(seen) == 2:
            raise RuntimeError("the consumer failed")

    before = threading.active_count()
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="the consumer failed"):
        t.run_many(tasks, NoEvidenceEngine(), batch=4, sink=sink, threads=3, inflate_device=0)
    assert time.perf_counter() - t0 < 20
    # the three inflaters went back to the process's pool (their pinned staging is reused by the next cohort), unharmed
    assert len(FakeInflater.made) == 3 and len(FakeInflater.closed) == 0 and 1 <= len(FakeInflater.runs) < 10
    deadline = time.time() + 10
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.05)
    assert threading.active_count() <= before                  # feeder, decode and pool threads are gone
    # and the good path: the same cohort through the same stand-in, every sample once
    del seen[:]
    out = t.run_many(tasks, NoEvidenceEngine(), batch=4, threads=3, inflate_device=0)
    assert [r["samplekey"] for r in out] == [a[0] for a in tasks]
    assert len(FakeInflater.made) == 3                          # the same three again
    t.release_inflaters()
    assert len(FakeInflater.closed) == 3
