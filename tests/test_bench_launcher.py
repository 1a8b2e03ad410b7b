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


def test_eight_ranks_of_a_thousand_samples_each_configs3_shape():
    """BASELINE configs[3] (8 000 samples x 30 loci over 8 GPUs) as the launcher would run it: eight rank processes with
    1 000 samples each, weak scaling, the whole-job value = all ranks' units over the slowest rank's time."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--samples", "1000", "--steps", "1",
                          "--no-cpu-baseline", "--gpus", "8", "--no-sweep"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 8 and rec["stub"] is True
    ranks = rec["scaling_sweep"][0]["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8)) and all(r["units"] == 30000 for r in ranks)
    # stub rank r takes 0.05 * (r + 1) s per step: 240 000 units / 0.4 s
    assert rec["value"] == 600000.0


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


def test_end_to_end_legs_spread_over_the_devices(tmp_path, monkeypatch):
    """run_e2e on an 8-GPU box (stub ranks): one set of BAMs, a leg per device count of the sweep, ranks = drivers per
    GPU x GPUs with rank r on device r mod n and an equal share of the CPUs each; on a 1-GPU box the plan is the one of
    earlier rounds with one core left to the drivers (1 driver x 15 threads, 2 x 7, 3 x 5 on 16 CPUs)."""
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    from tredparse_amd import shard
    assert bench.e2e_plan(1, 16) == [(1, 15), (2, 7), (3, 5)]
    assert bench.e2e_plan(8, 128) == [(8, 15), (16, 7), (24, 5)]
    assert bench.e2e_plan(8, 16) == [(8, 1)]
    assert bench.e2e_plan(2, 16, drivers_opt=2, threads_opt=3) == [(2, 3), (4, 3)]
    monkeypatch.setattr(shard, "usable_cpus", lambda: 64)
    made, spawned = [], []

    def fake_bams(root, n, seed=0, workers=1):
        made.append(n)
        import numpy as np
        return [("s{:04d}".format(i), os.path.join(root, "s{:04d}.bam".format(i)), np.zeros((30, 2), int)) for i in range(n)]

    def fake_spawn(argv, world, n_devices, timeout=None, env=None, stdout=None, cwd=None):
        limit = int(argv[argv.index("--e2e-limit") + 1])
        threads = int(argv[argv.index("--e2e-threads") + 1])
        spawned.append((world, n_devices, limit, threads))
        for r in range(world):
            lo, hi = shard.shard_range(limit, r, world)
            dev = shard.rank_env(r, world, 1, r % n_devices, base={})["TRED_RANK_DEVICE"]
            with open(os.path.join(env["TREDBENCH_OUT"], "e2e_rank{}.json".format(r)), "w") as fp:
                json.dump({"rank": r, "device": dev, "units": 30 * (hi - lo), "seconds": 1.0 + 0.01 * r, "samples": hi - lo,
                           "host_threads": threads, "driver_seconds": {"gpu": 0.1}, "short_ok": 1, "short_n": 1,
                           "bam_bytes": 1000 * (hi - lo)}, fp)
        return [0] * world

    args = argparse.Namespace(e2e_samples=128, seed=1, e2e_drivers=0, e2e_threads=0, e2e_batch=16, rank_timeout=60,
                              e2e_gpu_inflate="0", e2e_inflate_batch=32, e2e_repeat=1)
    recs = bench.run_e2e(args, [1, 2, 8], spawn=fake_spawn, make_bams=fake_bams)
    assert made == [512]                                            # one set of files: 8 GPUs x 64 (capped at 512)
    assert sorted(recs) == [1, 2, 8]
    # (ranks, devices, files, threads per rank): 64 CPUs -> 8 and 12 drivers per GPU at one GPU, 4 and 6 at two, 1 at eight
    assert spawned == [(1, 1, 128, 63), (8, 1, 128, 7), (12, 1, 128, 5), (2, 2, 256, 31), (8, 2, 256, 7), (12, 2, 256, 5),
                       (8, 8, 512, 7)]
    eight = recs[8]
    assert eight["devices"] == 8 and eight["drivers"] == 8 and eight["samples"] == 512
    assert sorted(d["device"] for d in eight["per_driver"]) == [str(i) for i in range(8)]
    assert abs(eight["value"] - 30 * 512 / 1.07) < 1e-6              # all ranks' units / the slowest rank's time


    # with the GPU-inflate legs: every plan twice, the second time with the larger batch and the child flag
    del spawned[:]
    flags = []
    plain_spawn = fake_spawn

    def flag_spawn(argv, *a, **k):
        flags.append((argv[argv.index("--e2e-gpu-inflate") + 1], argv[argv.index("--e2e-batch") + 1]))
        return plain_spawn(argv, *a, **k)
    args.e2e_gpu_inflate = "both"
    recs = bench.run_e2e(args, [1], spawn=flag_spawn, make_bams=fake_bams)
    assert flags == [("0", "16")] * 3 + [("1", "32")] * 7           # (the GPU legs add plans with a driver per four, three and 2.7 CPUs)
    assert [l["gpu_inflate"] for l in recs[1]["legs"]] == [False] * 3 + [True] * 7
    assert [l["gpu_walk"] for l in recs[1]["legs"]] == [False] * 3 + [True] * 6 + [False]     # (the last plan again, walks on the host)
    assert bench.e2e_plan(1, 16, dense=True) == [(1, 15), (2, 7), (3, 5), (4, 3), (5, 3), (6, 3)]


def test_pmc_traffic_is_only_cited_for_the_build_it_was_taken_from(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "PMC_GLOB", str(tmp_path / "r*_pmc_summary.json"))
    (tmp_path / "r02_pmc_summary.json").write_text(json.dumps(
        {"kernels": {"sw_cont_kernel": {"hbm_bytes_per_launch": 111.0}}}))                      # no build recorded
    (tmp_path / "r03_pmc_summary.json").write_text(json.dumps(
        {"library_version": "tredgpu 0.3 (gfx950) src aaaa", "kernels": {"sw_cont_kernel": {"hbm_bytes_per_launch": 222.0}}}))
    got, src = bench.pmc_traffic("sw_cont_kernel", "tredgpu 0.3 (gfx950) src aaaa")
    assert got == 222.0 and "r03_pmc_summary.json" in src and "src aaaa" in src
    got, src = bench.pmc_traffic("sw_cont_kernel", "tredgpu 0.3 (gfx950) src bbbb")
    assert got is None and "no PMC summary of this build" in src and "r02_pmc_summary.json" in src
