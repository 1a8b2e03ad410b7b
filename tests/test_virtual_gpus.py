"""TRED_VIRTUAL_GPUS: the rehearsal of an 8-GPU run on a box with one GPU.  shard.virtual_gpus makes
the launcher count the box's devices as V: bench.py --gpus 8 then starts eight real ranks (gloo barrier, sum / max
reduction), and its end-to-end legs real drivers over a hard-linked cohort of 8 x --e2e-samples files with the NUMA code
reading the box's sysfs -- every record marked `oversubscribed`.  CPU: the device arithmetic; GPU: the whole command."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_virtual_devices_map_onto_the_physical_ones():
    from tredparse_amd import shard
    env = {"TRED_VIRTUAL_GPUS": "8"}
    assert shard.virtual_gpus(1, env) == 8 and env["TRED_REAL_GPUS"] == "1" and shard.real_gpus(8, env) == 1
    assert [shard.device_entry(k, env) for k in range(8)] == ["0"] * 8
    env = {"TRED_VIRTUAL_GPUS": "8"}
    assert shard.virtual_gpus(2, env) == 8 and [shard.device_entry(k, env) for k in range(4)] == ["0", "1", "0", "1"]
    env = {"TRED_VIRTUAL_GPUS": "8", "HIP_VISIBLE_DEVICES": "4,6"}
    assert shard.virtual_gpus(2, env) == 8 and [shard.device_entry(k, env) for k in range(4)] == ["4", "6", "4", "6"]
    assert shard.rank_env(5, 8, 1, 5, base=env)["HIP_VISIBLE_DEVICES"] == "6"
    env = {"TRED_REAL_GPUS": "3"}
    assert shard.virtual_gpus(4, env) == 4 and "TRED_REAL_GPUS" not in env and shard.real_gpus(4, env) == 4     # not asked for: nothing virtual
    assert shard.virtual_gpus(0, {"TRED_VIRTUAL_GPUS": "8"}) == 0                                                  # no GPU stays no GPU
    # the NUMA node of a virtual device is its physical device's
    sets = shard.rank_cpusets(8, 8, allowed=list(range(16)), gpu_nodes=[1], node_cpus={0: list(range(8)), 1: list(range(8, 16))},
                              visible=[0] * 8)
    assert all(set(c) & set(range(8, 16)) for c in sets)


@pytest.mark.gpu
def test_bench_with_eight_virtual_gpus(tmp_path):
    """The driver's `bench.py --gpus 8` on this box: eight ranks on the one GPU for the kernel path, and for every device
    count of the sweep the host-only leg and the planned leg (inflate, walks, selection on the GPU) over 8 files per GPU;
    the last stdout line parses, stays under 4 KB, every leg's outputs agree, and the whole command stays far below the
    driver's limit."""
    import time
    env = dict(os.environ, TRED_VIRTUAL_GPUS="8")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--samples", "40",
                        "--e2e-samples", "8", "--e2e-distinct", "8", "--e2e-seconds", "2", "--e2e-repeats", "2", "--e2e-inflate-batch", "4",
                        "--legs", "", "--no-cpu-baseline", "--e2e-wgs-samples", "0"], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    seconds = time.time() - t0
    assert p.returncode == 0, p.stderr[-3000:]
    last = p.stdout.strip().splitlines()[-1]
    assert len(last) < 4000
    line = json.loads(last)
    assert line["n_gpus"] == 8 and line["gpus_visible"] == 8 and line["virtual_gpus"]["physical"] >= 1 and line["value"] > 0
    sweep = {s["n"]: s for s in line["scaling_sweep"]}
    assert sorted(sweep) == [1, 2, 4, 8] and all(sweep[n].get("oversubscribed") for n in (2, 4, 8) if n > line["virtual_gpus"]["physical"])
    e = line["end_to_end"]
    assert e["devices"] == 8 and e["files"] == 64 and e["outputs_identical"] is True and e["value"] > 0 and e.get("oversubscribed") is True
    assert len(e["repeats"]) == 2
    with open(os.path.join(ROOT, "bench_detail.json")) as fp:
        detail = json.load(fp)
    legs = detail["end_to_end"]["legs"]
    assert [l["role"] for l in legs] == ["host_only_one_driver_per_gpu", "plan", "plan"] and all("error" not in l for l in legs)
    assert all(l["outputs"] == 64 for l in legs) and all(d["device"] == "0" for l in legs for d in l["per_driver"]) or line["virtual_gpus"]["physical"] > 1
    assert legs[1]["pinned_MB_per_gpu"] > 0
    assert seconds < 600, seconds
    with open(os.path.join(ROOT, "gpurun_out", "r06_virtual8.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else str(tmp_path / "r06_virtual8.json"), "w") as fp:
        json.dump({"seconds": round(seconds, 1), "line": line}, fp)
