"""CPU: bench.py's launcher (started directly it never touches a GPU; it starts one child process per rank) with
stub ranks -- every rank runs, the ranks meet over gloo, rank 0's record carries the whole-job value
(sum of units / max of elapsed), and the sweep list has one entry per rank count."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--samples", "10", "--steps", "2",
                          "--no-cpu-baseline"] + list(extra), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_launcher_runs_n_ranks_and_sums_their_units():
    rec = _run("--gpus", "2")
    assert rec["n_gpus"] == 2 and rec["stub"] is True
    # stub ranks: 10 samples x 30 loci x 2 steps each, rank r takes 0.05 * (r + 1) s -> 1200 units / 0.1 s
    assert rec["value"] == 12000.0
    sweep = {s["n"]: s for s in rec["scaling_sweep"]}
    assert sorted(sweep) == [1, 2]
    assert [r["rank"] for r in sweep[2]["ranks"]] == [0, 1]
    assert [r["units"] for r in sweep[2]["ranks"]] == [600, 600]
    assert sweep[2]["oversubscribed"] is True and "oversubscribed" not in sweep[1]


def test_no_sweep_runs_only_the_requested_rank_count():
    rec = _run("--gpus", "3", "--no-sweep")
    assert rec["n_gpus"] == 3 and [s["n"] for s in rec["scaling_sweep"]] == [3]
    assert len(rec["scaling_sweep"][0]["ranks"]) == 3


def test_sweep_counts():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.sweep_counts(1, 1, True) == [1, 2]          # 1-GPU box: the two-rank launcher check rides along
    assert bench.sweep_counts(1, 8, True) == [1, 2, 4, 8]
    assert bench.sweep_counts(8, 8, True) == [1, 2, 4, 8]
    assert bench.sweep_counts(4, 8, False) == [4]
    assert bench.sweep_counts(3, 1, True) == [1, 2, 3]


def test_rank_env_pins_one_device_per_rank():
    from tredparse_amd import shard
    e = shard.rank_env(3, 4, 29500, 1, base={})
    assert e["RANK"] == "3" and e["WORLD_SIZE"] == "4" and e["LOCAL_RANK"] == "0"
    assert e["HIP_VISIBLE_DEVICES"] == "1" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29500"
    e = shard.rank_env(1, 2, 1, None, base={})
    assert "HIP_VISIBLE_DEVICES" not in e and e["LOCAL_RANK"] == "1"
