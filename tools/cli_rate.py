#!/usr/bin/env python3
"""The product's command line at cohort scale (GPU box): N sample keys over D distinct synthetic 30x BAMs (hard links), then

    python -m tredparse_amd.tred samples.csv --workdir w --gpu-inflate --gpu-walk --gpu-select

with NO process or thread count -- `--drivers auto` takes shard.driver_plan -- timed as a whole (start-up of the driver
processes included) and between the first and the last output file.  VERDICT r4 item 3: within 10 % of bench.py's plan.

usage: python tools/cli_rate.py [samples = 12288] [distinct = 512] [extra tred.py arguments ...]
"""
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def _version():
    """tredgpu_version() of the tree's library, asked in a child (this process stays off the GPU runtime)."""
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from tredparse_amd import _lib; print(_lib.version())" % ROOT],
                         capture_output=True, text=True)
    return out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""


def main():
    from tredparse_amd import shard, synth_bam
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
    distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    extra = sys.argv[3:]
    root = tempfile.mkdtemp(prefix="tred_cli_rate_")
    made = synth_bam.make_bams(root, min(n, distinct), seed=20260101, workers=shard.usable_cpus())
    rows = []
    for i in range(n):
        key, path, _ = made[i % len(made)]
        if i >= len(made):
            new = "{}x{}".format(key, i // len(made))
            for ext in (".bam", ".bam.bai"):
                os.link(os.path.join(root, key + ext), os.path.join(root, new + ext))
            key, path = new, os.path.join(root, new + ".bam")
        rows.append("{},{}".format(key, path))
    csv = os.path.join(root, "samples.csv")
    with open(csv, "w") as fp:
        fp.write("\n".join(rows) + "\n")
    work = os.path.join(root, "work")
    names = [l["name"] for l in synth_bam.bench_loci()]
    argv = [sys.executable, "-m", "tredparse_amd.tred", csv, "--workdir", work, "--gpu-inflate", "--gpu-walk", "--gpu-select"]
    for t in names:
        argv += ["--tred", t]
    env = dict(os.environ)
    timeline = "--timeline" in extra                  # (the drivers' marks, tredparse_amd/runtime.py: seconds since the command began)
    if timeline:
        extra.remove("--timeline")
        env["TRED_TIMELINE"] = root
    t0 = time.time()
    out = subprocess.run(argv + extra, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
    dt = time.time() - t0
    lines = {}
    if timeline:
        for f in sorted(glob.glob(os.path.join(root, "timeline_*.json"))):
            with open(f) as fp:
                ev = json.load(fp)
            lines[os.path.basename(f)] = [[round(t - t0, 3), e, kw] for t, e, kw in ev[:90]] + ["..."] + [[round(t - t0, 3), e, kw] for t, e, kw in ev[-6:]]
    files = glob.glob(os.path.join(work, "*.json"))
    times = sorted(os.path.getmtime(f) for f in files)
    span = times[-1] - times[0] if len(times) > 1 else dt
    mid = times[len(times) // 10:] if len(times) > 100 else times          # (behind the first tenth: the pipeline is full)
    steady = (len(mid) - 1) * len(names) / max(mid[-1] - mid[0], 1e-9) if len(mid) > 1 else 0.0
    print(json.dumps({"samples": n, "distinct_bams": len(made), "loci": len(names), "exit_code": out.returncode, "json_files": len(files),
                      "elapsed_s": round(dt, 2), "genotypes_per_s_whole_command": round(len(files) * len(names) / dt, 1),
                      "first_file_after_s": round(times[0] - t0, 2) if times else None, "first_to_last_file_s": round(span, 2), "genotypes_per_s_behind_the_first_tenth": round(steady, 1),
                      "plan": shard.driver_plan(shard.usable_cpus(), 1), "library": _version(), "extra_arguments": extra, "usable_cpus": shard.usable_cpus(),
                      "stderr_tail": out.stderr[-300:], **({"timeline": lines} if timeline else {})}))
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
