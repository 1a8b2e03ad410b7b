"""CPU: the product CLI's --gpus N fan-out (tredparse/tred.py:521-532 is the reference's per-sample Pool).
The parent fixes the sample list, hands it to N children in a task file, child r takes shard_range(r) and writes
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
    # rank 0 owns k00..k02, rank 1 owns k03..k04 (block partition); nothing twice, nothing missing
    assert [sorted(set(f.split(".")[0] for f in w)) for w in written] == [["k00", "k01", "k02"], ["k03", "k04"]]
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
    assert seen["tasks"] == [["t001", bam, None]]
