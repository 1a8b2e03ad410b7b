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
    line = json.loads(out.stdout.strip().splitlines()[-1])
    detail = json.loads([l for l in out.stderr.splitlines() if l.startswith("bench detail: ")][-1][len("bench detail: "):])
    assert abs(detail["value"] - line["value"]) < 1e-3 and detail["n_gpus"] == line["n_gpus"]
    return detail


def test_launcher_runs_n_ranks_and_sums_their_units():
    rec = _run("--gpus", "2")
    assert rec["n_gpus"] == 2 and rec["stub"] is True
    # stub ranks: 10 samples x 30 loci x 2 steps each, rank r takes 0.05 * (r + 1) s -> 1200 units / 0.1 s
    assert abs(rec["value"] - 12000.0) < 1e-6
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
    assert json.loads(out.stdout.strip().splitlines()[-1])["value"] == 600000.0
    rec = json.loads([l for l in out.stderr.splitlines() if l.startswith("bench detail: ")][-1][len("bench detail: "):])
    assert rec["n_gpus"] == 8 and rec["stub"] is True
    ranks = rec["scaling_sweep"][0]["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8)) and all(r["units"] == 30000 for r in ranks)
    # stub rank r takes 0.05 * (r + 1) s per step: 240 000 units / 0.4 s
    assert abs(rec["value"] - 600000.0) < 1e-6


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


def _fake_e2e(monkeypatch, usable=64):
    """Stand-ins for run_e2e's helpers: no BAMs, no processes -- every driver 'finishes' a sample every 10 ms from 0.5 s
    after the common start, for as long as its leg is told to run."""
    sys.path.insert(0, ROOT)
    import bench
    import numpy as np
    from tredparse_amd import shard
    monkeypatch.setattr(shard, "usable_cpus", lambda: usable)
    made, spawned, legs = [], [], {}

    def fake_bams(root, n, seed=0, workers=1):
        made.append(n)
        out = []
        for i in range(n):
            key = "s{:04d}".format(i)
            for ext in (".bam", ".bam.bai"):
                open(os.path.join(root, key + ext), "w").close()
            out.append((key, os.path.join(root, key + ".bam"), np.zeros((30, 2), int)))
        return out

    def fake_spawn(argv, world, n_devices, timeout=None, env=None, stdout=None, cwd=None):
        get = lambda flag: argv[argv.index(flag) + 1]
        limit, threads, seconds = int(get("--e2e-limit")), int(get("--e2e-threads")), float(get("--e2e-seconds"))
        spawned.append((world, n_devices, limit, threads, get("--e2e-gpu-inflate"), get("--e2e-gpu-walk") + get("--e2e-gpu-select"),
                        get("--e2e-batch")))
        ranks, logs = [], []
        names = sorted(f[:-4] for f in os.listdir(get("--e2e-child")) if f.endswith(".bam"))[:limit]
        for r in range(world):
            lo, hi = shard.shard_range(limit, r, world)
            dev = shard.rank_env(r, world, 1, r % n_devices, base={})["TRED_RANK_DEVICE"]
            # (the k-th leg started is 1 % slower per sample than the one before: repeats of a leg differ a little)
            t = 1000.0 + 0.5 + 0.01 * (1 + 0.01 * len(legs)) * np.arange(1, int(seconds * 100) + 1)
            logs.append(np.stack([t, np.full(len(t), 30.0), np.full(len(t), 25.0), np.full(len(t), 30.0)], axis=1))
            ranks.append({"rank": r, "device": dev, "t_process": 990.0, "t_begin": 1000.0, "t_end": float(t[-1]), "files": hi - lo,
                          "host_threads": threads, "first_chunk": 3, "driver_seconds": {"gpu": 0.1}, "bam_bytes": 1000 * (hi - lo),
                          "digests": {k: "h" + k for k in names[lo:hi]}})
        legs[env["TREDBENCH_OUT"]] = (ranks, logs)
        return [0] * world

    return bench, made, spawned, fake_bams, fake_spawn, (lambda out_dir, drivers: legs[out_dir])


def _e2e_args(**kw):
    import argparse
    base = dict(e2e_samples=128, e2e_distinct=128, seed=1, e2e_drivers=0, e2e_threads=0, e2e_batch=16, rank_timeout=60,
                e2e_gpu_inflate="1", e2e_gpu_walk="1", e2e_gpu_select="1", e2e_inflate_batch=32, e2e_seconds=12.0, e2e_sweep=False,
                e2e_repeats=3)
    base.update(kw)
    return argparse.Namespace(**base)


def test_end_to_end_legs_spread_over_the_devices(tmp_path, monkeypatch):
    """run_e2e on an 8-GPU box (stub ranks): one set of BAMs; per device count of the sweep TWO legs, both fixed before
    anything runs -- one host-only driver per GPU, and the plan of shard.driver_plan(usable CPUs, GPUs) with inflate
    and pair walks on the GPU -- over a cohort of the SAME size per GPU at every device count (n = 8: 8 x 128 files,
    the ones beyond the distinct 128 hard-linked under their own keys)."""
    from tredparse_amd import shard
    bench, made, spawned, fake_bams, fake_spawn, read_leg = _fake_e2e(monkeypatch, usable=64)
    per_gpu, threads = shard.driver_plan(64, 8)
    assert bench.e2e_rule(8, 64) == (8 * per_gpu, threads)
    assert shard.driver_plan(16, 1) == (3, 5) and shard.driver_plan(128, 8) == (3, 5) and shard.driver_plan(4, 1) == (2, 1)
    assert shard.driver_plan(16, 1, gpu_inflate=False) == (3, 5) and shard.driver_plan(256, 8) == (3, 8)
    recs = bench.run_e2e(_e2e_args(), [1, 2, 8], spawn=fake_spawn, make_bams=fake_bams, read_leg=read_leg)
    assert made == [128]                                            # the distinct files are made once
    assert sorted(recs) == [1, 2, 8]
    # (ranks, devices, files, threads per rank, gpu_inflate, gpu_walk + gpu_select, batch): files = devices x 128, always; the
    # planned leg (inflate, walks and read selection on the GPU) three times over
    want = []
    for n in (1, 2, 8):
        d, t = shard.driver_plan(64, n)
        want += [(n, n, 128 * n, 63 // n, "0", "00", "16")] + [(d * n, n, 128 * n, t, "1", "11", "32")] * 3
    assert spawned == want
    eight = recs[8]
    assert eight["devices"] == 8 and eight["drivers"] == 8 * per_gpu and eight["files"] == 1024
    assert eight["gpu_inflate"] and eight["gpu_walk"] and eight["gpu_select"] and eight["outputs_identical"] is True and eight["outputs"] == 1024
    assert [l["role"] for l in eight["legs"]] == ["host_only_one_driver_per_gpu", "plan", "plan", "plan"]
    # steady state: every driver finishes ~100 samples x 30 units a second; the window opens at the slowest driver's
    # third sample (its first chunk) and closes at the first driver's last.  The record is the MEDIAN repeat's, min and max beside it
    assert len(eight["repeats"]) == 3 and eight["min"] < eight["value"] < eight["max"] and eight["repeat"] == 1
    assert sorted(eight["repeats"])[1] == round(eight["value"], 1) and eight["repeats"][0] > eight["repeats"][1] > eight["repeats"][2]
    assert abs(eight["value"] - 8 * per_gpu * 3000.0) < 0.15 * 8 * per_gpu * 3000.0
    assert eight["host_only_one_driver_per_gpu"]["value"] > 0 and "per_driver" not in eight


def test_end_to_end_sweep_is_behind_its_flag(tmp_path, monkeypatch):
    bench, made, spawned, fake_bams, fake_spawn, read_leg = _fake_e2e(monkeypatch, usable=16)
    recs = bench.run_e2e(_e2e_args(e2e_sweep=True), [1], spawn=fake_spawn, make_bams=fake_bams, read_leg=read_leg)
    roles = [l["role"] for l in recs[1]["legs"]]
    assert roles[:4] == ["host_only_one_driver_per_gpu", "plan", "plan", "plan"] and set(roles[4:]) == {"sweep"} and len(roles) > 6
    assert recs[1]["role"] == "plan"                  # the record is the plan's whatever the sweep finds
    assert bench.e2e_plan(1, 16) == [(1, 15), (2, 7), (3, 5)]
    assert bench.e2e_plan(1, 16, dense=True) == [(1, 15), (2, 7), (3, 5), (4, 3), (5, 3), (6, 3)]


def test_outputs_that_differ_between_legs_are_reported(tmp_path, monkeypatch):
    bench, made, spawned, fake_bams, fake_spawn, read_leg = _fake_e2e(monkeypatch, usable=16)

    def one_differs(out_dir, drivers):
        ranks, logs = read_leg(out_dir, drivers)
        if drivers > 1:
            ranks[0]["digests"][sorted(ranks[0]["digests"])[0]] = "other"
        return ranks, logs
    recs = bench.run_e2e(_e2e_args(), [1], spawn=fake_spawn, make_bams=fake_bams, read_leg=one_differs)
    assert recs[1]["outputs_identical"] is False


def test_final_line_is_compact_and_parses():
    """The driver reads bench.py's LAST stdout line: it must stay under 4 KB whatever the legs carry (round 4's had grown
    to 29 KB and the driver's record came out unparsed), round-trip through JSON, and hold the contract's keys together with
    roofline.frac and cpu_baseline.value."""
    sys.path.insert(0, ROOT)
    import bench
    drivers = [{"seconds": 12.1, "device": "0", **{k: 1.234567 for k in ("scan_wait", "gpu", "format", "write", "inflate",
                "inflate_blocks", "inflate_failed", "inflate_hits", "inflate_misses", "inflate_gpu", "walk_regions", "walk_declined")}}
               for _ in range(48)]
    leg = {"drivers": 48, "devices": 8, "gpu_inflate": True, "gpu_walk": True, "value": 201234.5678, "unit": "genotypes/s",
           "first_pass_value": 190000.123, "whole_run_value": 195000.0, "startup_s": 0.7, "seconds": 11.2, "samples": 75000,
           "files": 4096, "host_threads_per_driver": 3, "outputs_identical": True, "per_driver": drivers, "what": "x" * 900,
           "page_cache": "y" * 400, "cohort": "z" * 100}
    out = {"metric": "sample x TRED genotypes/sec at 30x 150bp", "value": 11234567.891, "unit": "genotypes/s", "n_gpus": 8,
           "steps": 20, "warmup": 5, "ms_per_step": 21.2345678, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int32 (SW) + f64 (likelihood)", "data": "synthetic", "library": "tredgpu 0.5 (gfx950) src 0123abcd",
           "config": {"workload": "1000 synthetic 30x samples x 30 TRED loci per GPU (BASELINE configs[2]) at 150 bp; fused "
                                  "SW+tagging -> histograms -> (h1,h2) grid, inputs resident in HBM", "name": "config3",
                      "units_per_step_per_gpu": 30000, "reads_per_step_per_gpu": 2400000, "coverage": 30.0, "readlen": 150,
                      "maxinsert": 300, "alleles": "uniform", "parallelism": "sample-sharded x8 (no collective)"},
           "roofline": {"kernel": "sw_cont_kernel<10,4>", "bound": "valu", "achieved": 2.7812345, "peak": 7.8643, "unit": "TCUPS",
                        "frac": 0.35365, "mix_ceiling_frac": 0.39, "traffic": 450e6, "traffic_source": "s" * 300,
                        "traffic_over_algorithmic": 2.0, "note": "n" * 900, "avg_launch_ms": 16.01, "effective_TCUPS": 275.0,
                        "sw_counters": {"c{}".format(i): 123456789 for i in range(40)}, "hbm": {"frac": 0.001}},
           "kernels_ms_per_step": {k: 1.23456 for k in ("sw_ladder", "tally", "grid", "grid_kde", "grid_prepare", "grid_pairs", "grid_reduce")},
           "check": {"grid_pairs_percentiles": {str(q): q for q in range(10)}},
           "scaling_sweep": [{"n": n, "value": 1.4e6 * n, "unit": "genotypes/s", "ms_per_step": 21.0, "devices": n,
                              "ranks": [{"rank": r, "device": str(r), "units": 600000, "elapsed_s": 0.42} for r in range(n)],
                              "end_to_end": {"value": 25000.0 * n}} for n in (1, 2, 4, 8)],
           "gpus_visible": 8, "end_to_end": dict(leg, legs=[dict(leg) for _ in range(12)],
                                                 host_only_one_driver_per_gpu={"value": 6500.0, "first_pass_value": 6000.0, "seconds": 5.9, "samples": 1300, "startup_s": 0.4}),
           "legs": [{"leg": l, "value": 1.2e6, "ms_per_step": 12.3, "frac": 0.31, "workload": "w" * 200, "kernels_ms_per_step": {"a": 1.0},
                     "roofline": {"frac": 0.31}} for l in ("streamed", "config5:150:200", "config3:100:500", "config3:250:500")],
           "cpu_baseline": {"value": 59.9, "unit": "genotypes/s", "cores": 16, "kind": "reference", "sample": "q" * 400, "host_cpus": 256},
           "cpu_baseline_1core": {"value": 2.68, "unit": "genotypes/s", "cores": 1, "kind": "reference", "sample": "q" * 400}}
    assert len(json.dumps(out)) > 20000
    text = bench.compact_line(out)
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "end_to_end", "legs"):
        assert key in line, key
    assert line["roofline"]["frac"] == 0.35365 and line["roofline"]["mix_ceiling_frac"] == 0.39 and line["roofline"]["bound"] == "valu"
    assert line["cpu_baseline"]["value"] == 59.9 and line["cpu_baseline"]["cores"] == 16 and line["cpu_baseline"]["kind"] == "reference"
    assert line["config"]["workload"].startswith("1000 synthetic") and "model" not in line["config"]
    assert line["end_to_end"]["value"] == 201234.568 and line["end_to_end"]["outputs_identical"] is True
    assert [l["leg"] for l in line["legs"]] == ["streamed", "config5:150:200", "config3:100:500", "config3:250:500"]
    assert all(set(l) <= {"leg", "value", "ms_per_step", "frac", "error"} for l in line["legs"])
    assert [s["n"] for s in line["scaling_sweep"]] == [1, 2, 4, 8] and line["scaling_sweep"][3]["end_to_end"] == 200000.0
    # a record with even more in it still fits: the optional parts go first
    out["legs"] = out["legs"] * 40
    assert len(bench.compact_line(out)) < 4096


def test_stub_launcher_prints_the_compact_line_last_and_writes_the_detail(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--samples", "10", "--steps", "2",
                          "--no-cpu-baseline", "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096          # stdout carries the compact line alone
    line = json.loads(lines[0])
    assert line["detail"] == "bench_detail.json" and "bench detail: " in out.stderr
    with open(os.path.join(ROOT, "bench_detail.json")) as fp:
        detail = json.load(fp)
    assert abs(detail["value"] - line["value"]) < 1e-3 and "ranks" in detail["scaling_sweep"][0]


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


def test_pmc_traffic_comes_from_the_headline_summary_not_from_another_legs(tmp_path, monkeypatch):
    """profiles/ holds one summary per profiled leg, all of the same build and all with sw_cont_kernel in them (walk16 is a
    16-sample call: 8 MB per launch where the headline batch moves 450): only <round>_pmc_summary.json is the headline's."""
    sys.path.insert(0, ROOT)
    import bench
    assert os.path.basename(bench.PMC_GLOB) == "r[0-9][0-9]_pmc_summary.json"
    monkeypatch.setattr(bench, "PMC_GLOB", str(tmp_path / os.path.basename(bench.PMC_GLOB)))
    lib = "tredgpu 0.6 (gfx950) src cccc"
    for name, b in (("r06_pmc_summary.json", 445e6), ("r06_walk16_pmc_summary.json", 8e6), ("r06_len250_pmc_summary.json", 1.6e9)):
        (tmp_path / name).write_text(json.dumps({"library_version": lib, "kernels": {"sw_cont_kernel": {"hbm_bytes_per_launch": b}}}))
    got, src = bench.pmc_traffic("sw_cont_kernel", lib)
    assert got == 445e6 and "profiles/r06_pmc_summary.json" in src


def test_mix_ceiling_is_computed_from_the_builds_census_and_the_kernels_counters():
    """roofline.mix_ceiling_frac: no literal -- the build's ISA census of sw_cont_kernel's column blocks
    (tools/isa_census.py -> tredparse_amd/data/sw_isa_census.json, taken from THIS tree's sw_ladder.hip) times the columns the
    kernel's counters report; it differs between the instantiations and moves with the counters, and is null without a census."""
    sys.path.insert(0, ROOT)
    import bench
    census = bench.load_census()
    assert census is not None, "tredparse_amd/data/sw_isa_census.json is missing or stale: make -C tredparse_amd/csrc"
    assert sorted(set(k["rows_per_lane"] for k in census["kernels"].values())) == [4, 7, 10, 16, 20, 32] and len(census["kernels"]) == 12
    assert bench.census_entry(census, 10, False)["waves_per_simd"] == 4 and bench.census_entry(census, 16, True)["generic"] is True
    for k in census["kernels"].values():
        c = k["trunk_column"]
        assert k["trunk_cycles_per_column"] >= k["free_cycles_per_column"] > 0 and c["max3"] >= 2 * k["rows_per_lane"]
        assert c["valu4"] >= 4 * k["rows_per_lane"] and c["valu2"] >= 3 * k["rows_per_lane"]      # 7 VALU per cell, 4 of them slow
    cnt = {"trunk_cols": 5.0e7, "continuation_cols": 2.5e7}
    cells = {7: 4 * 100 * 7.5e7, 10: 4 * 150 * 7.5e7, 16: 4 * 250 * 7.5e7}
    got = {r: bench.mix_ceiling(census, r, False, cnt, 1, cells[r])[0] for r in (7, 10, 16)}
    assert len(set(round(v, 4) for v in got.values())) == 3 and all(0.3 < v < 1.0 for v in got.values()), got
    more = bench.mix_ceiling(census, 10, False, {"trunk_cols": 6.0e7, "continuation_cols": 2.5e7}, 1, cells[10])[0]
    assert more < got[10]
    assert bench.mix_ceiling(None, 10, False, cnt, 1, cells[10])[0] is None
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "MIX_CEILING_FRAC" not in src and "0.39" not in src


def test_whole_genome_shaped_leg_runs_the_same_two_legs_over_its_own_files(tmp_path, monkeypatch):
    """run_e2e_wgs: a small cohort of wgs_like files (8 distinct ones hard-linked to 64), the host-only leg and the planned leg
    once, four samples per decode call; blocks per sample from the drivers' counters."""
    bench, made, spawned, fake_bams, fake_spawn, read_leg = _fake_e2e(monkeypatch, usable=16)

    def leg_with_counters(out_dir, drivers):
        ranks, logs = read_leg(out_dir, drivers)
        for r in ranks:
            r["driver_seconds"] = {"gpu": 0.1, "inflate_blocks": 46000.0, "select_samples": 10.0, "select_declined": 0.0, "walk_call": 1.5}
        return ranks, logs
    rec = bench.run_e2e_wgs(_e2e_args(e2e_wgs_samples=64, e2e_wgs_distinct=8), spawn=fake_spawn, make_bams=fake_bams, read_leg=leg_with_counters)
    assert made == [8] and [w[:3] for w in spawned] == [(1, 1, 64), (3, 1, 64)] and spawned[1][5:] == ("11", "4")
    assert [l["role"] for l in rec["legs"]] == ["host_only_one_driver_per_gpu", "plan"] and rec["outputs_identical"] is True
    assert rec["blocks_per_sample"] == 4600.0 and rec["walk_call_seconds_per_driver"] == [1.5, 1.5, 1.5] and rec["value"] > 0
    line = json.loads(bench.compact_line({"metric": "m", "value": 1.0, "end_to_end_wgs": rec}))
    assert line["end_to_end_wgs"]["blocks_per_sample"] == 4600.0 and line["end_to_end_wgs"]["host_only"] > 0
