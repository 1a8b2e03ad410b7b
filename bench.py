#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X STR-genotyping hot path.

Metric (BASELINE.json): sample x TRED genotypes / second at 30x 150 bp; 1/2/4/8 MI355X + host-CPU baseline.
One "step" = one pass of the whole hot path (template SW + tagging -> histograms -> (h1,h2) likelihood grid) over
one resident batch of `--samples` synthetic samples x 30 loci per GPU (BASELINE.json configs[2]: "1k synthetic
30x 150 bp BAMs x 30 TREDs on 1 GPU").  Inputs are packed and already in HBM when the timed region starts;
results stay on the device.

How it runs
  * under torch.distributed.run (RANK in the environment): this process is one rank = one GPU; RCCL carries the
    barrier and the max-over-ranks wall time only -- samples are independent, there is no data-path collective;
  * started directly: this process is a launcher that NEVER touches a GPU.  For every rank count n of the sweep
    (1, 2, 4, 8 up to the GPUs visible, and --gpus itself) it starts n child ranks (tredparse_amd.shard.spawn_ranks:
    HIP_VISIBLE_DEVICES = one device per child, rank r on device r mod visible), collects their per-rank records,
    runs the end-to-end legs from BAM files (run_e2e: the host-only leg and THE plan of shard.driver_plan, steady state),
    the extra one-GPU legs (streamed, configs[4], 100 / 250 bp), and times the CPU baseline on 1 core and on all
    host cores.

Output: the LAST stdout line is one compact JSON object (< 4 KB, compact_line): the contract's keys, `roofline`,
`cpu_baseline`, `end_to_end`, one small object per leg.  The full record -- per-rank and per-driver detail, counters,
notes, every plan of an --e2e-sweep -- goes to bench_detail.json next to this script (and to stderr).

Roofline of the dominant kernel (sw_cont_kernel, integer VALU bound; HIP-event timed on the context's own stream):
`achieved` = DP cells the kernel really swept (read rows x columns, padding rows and empty quad slots excluded) per
second, `peak` = int32 VALU lane-ops/s / 10 ops per cell, `mix_ceiling_frac` = what the column blocks' own instruction
mix allows of that peak for the columns this launch swept (mix_ceiling: the build's ISA census x the kernel's counters).  The brute-force cell count of SURVEY 8(d) over the same time is
`effective_TCUPS` (it exceeds the hardware peak because of the exact shortcuts; it is not a hardware rate).
`traffic` = HBM bytes per launch from the rocprofv3 PMC passes summarised in profiles/ (FETCH_SIZE + WRITE_SIZE), cited
only when the summary was taken on THIS build.
"""
import argparse
import json
import os
import sys
import tempfile
import time

if "RANK" not in os.environ:
    # launcher process (it also runs the CPU baseline): one BLAS / OpenMP thread per process, set before numpy
    # loads, so that the worker processes of the all-core leg do not each start a thread pool of their own
    for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(_v, "1")

# (the host driver of this pool supports dmabuf IPC only: without this RCCL's set-up of a multi-rank job fails in hipIpcGetMemHandle;
#  the GPU boxes export it already -- a rank started by anything else gets it here, before torch loads)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9      # int32 VALU lane-ops/s: 256 CU x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md)
OPS_PER_CELL = 10.0                        # minimum VALU ops of one affine-gap local-alignment cell
PEAK_TCUPS = PEAK_LANE_OPS / OPS_PER_CELL / 1e12
HBM_PEAK_GBS = 8000.0
# (the HEADLINE configuration's summary only -- tools/collect_profiles.py copies it to <round>_pmc_summary.json; the other legs'
#  files, <round>_<leg>_pmc_summary.json, hold the same kernels over other workloads: walk16 is a 16-sample call)
PMC_GLOB = os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.json")


def pmc_traffic(kernel, library_version):
    """(HBM bytes per launch of `kernel`, source) from the rocprofv3 PMC summaries under profiles/ -- only from a
    summary taken on THIS build: tools/pmc_to_json.py records the library's tredgpu_version() (which carries a hash
    of the kernel sources), and a summary of any other build is not evidence for the kernels being timed."""
    import glob
    seen = []
    for path in sorted(glob.glob(PMC_GLOB), reverse=True):
        try:
            with open(path) as fp:
                pm = json.load(fp)
        except (OSError, ValueError):
            continue
        seen.append(os.path.basename(path))
        if pm.get("library_version") != library_version:
            continue
        k = pm.get("kernels", {}).get(kernel, {})
        if "hbm_bytes_per_launch" in k:
            return k["hbm_bytes_per_launch"], "profiles/{} ({}; build {})".format(
                os.path.basename(path), pm.get("how", "rocprofv3 --pmc passes"), library_version)
    return None, "no PMC summary of this build ({}) under profiles/ (found: {})".format(library_version,
                                                                                       ", ".join(seen) or "none")


def bench_loci(loci):
    """The 30 loci with distinct coordinates (FXTAS/FXS and SBMA/AR share a region)."""
    return [l for l in loci if l["name"] not in ("FXTAS", "AR")]


def algorithmic_cells(batch):
    """R * L * 2 * sum_u(|prefix| + |suffix| + p*u), the brute-force forward cells."""
    total = 0
    n_reads_unit = np.diff(batch.unit_read_off).astype(np.int64)
    for li, (prefix, repeat, suffix, mu) in enumerate(batch.ladders):
        cols = 2 * sum(len(prefix) + len(suffix) + len(repeat) * u for u in range(1, mu + 1))
        r = int(n_reads_unit[batch.unit_ladder == li].sum())
        total += r * batch.readlen * cols
    return total


def ladder_cells(batch):
    """Cells of the shared-prefix ladder (trunk + every branch, both strands) before any pruning."""
    total = 0
    n_reads_unit = np.diff(batch.unit_read_off).astype(np.int64)
    for li, (prefix, repeat, suffix, mu) in enumerate(batch.ladders):
        cols = (len(prefix) + len(repeat) * mu + mu * len(suffix)) + (len(suffix) + len(repeat) * mu + mu * len(prefix))
        r = int(n_reads_unit[batch.unit_ladder == li].sum())
        total += r * batch.readlen * cols
    return total


WORKLOADS = {
    # BASELINE.json configs[2] (the headline) and configs[4]; `samples` is the default per-GPU batch
    "config3": {"what": "synthetic 30x samples x 30 TRED loci per GPU (BASELINE configs[2])",
                "alleles": "uniform 5..60 units (SURVEY 8d)", "synth": {}},
    "config5": {"what": "synthetic 100x samples x 30 TRED loci per GPU, one allele expanded up to 200 repeats in "
                        "80 % of the units (BASELINE configs[4]: large (h1,h2) grids, repeat-only reads)",
                "alleles": "uniform 5..60 units, the longer one 60..200 in 80 % of the units",
                "synth": {"coverage": 100.0, "expanded_max": 200, "expanded_frac": 0.8}},
}


def rows_per_lane(readlen):
    """The sw_cont_kernel instantiation a maximum read length selects (csrc/capi.hip rows_for)."""
    return 4 if readlen <= 64 else 7 if readlen <= 112 else 10 if readlen <= 160 else 16 if readlen <= 256 else 20 if readlen <= 320 else 32


def make_batch(args, rank, world):
    from tredparse_amd import synth
    loci = bench_loci(synth.load_loci())
    kw = dict(coverage=args.coverage, readlen=args.readlen)
    kw.update(WORKLOADS[args.workload]["synth"])
    if args.coverage_set:
        kw["coverage"] = args.coverage
    p = synth.SynthParams(**kw)
    args.coverage_used = p.coverage
    from tredparse_amd import shard
    # TRED_BENCH_WORKERS=1 (tools/profile_round.sh): under rocprofv3 the profiler's preloaded library has initialised
    # the GPU before Python starts, and a GPU-initialised process must not fork workers -- build the batch in-process
    workers = int(os.environ.get("TRED_BENCH_WORKERS", "0")) or max(1, min(len(loci), shard.usable_cpus() // max(1, world)))
    return loci, synth.build_batch(args.seed + rank, loci, args.samples, p, workers=workers)


# ---- CPU baseline (launcher process only: it never initialises a GPU, so it may fork workers) ---------------
_CPU = {}


def _cpu_unit(u):
    """One sample x locus unit on the CPU the way the reference computes it: every (read, template) alignment by the
    reference's own ssw.c (oracle/_ref: one ssw_init + ssw_align per pair as ssw_wrap.py does) or, without it, the
    C restatement; then the numpy/scipy likelihood oracle.  Returns (seconds in SW, seconds in the likelihood)."""
    from oracle import lik_oracle as lo
    from tredparse_amd import synth
    batch, loci, classify, ls = _CPU["batch"], _CPU["loci"], _CPU["classify"], _CPU["ls"]
    t0 = time.perf_counter()
    r0, r1 = int(batch.unit_read_off[u]), int(batch.unit_read_off[u + 1])
    reads = [synth.decode(r) for r in batch.codes[r0:r1]]
    lad = int(batch.unit_ladder[u])
    cls = classify(reads, np.full(len(reads), lad, np.int32), ls, threads=1)
    t1 = time.perf_counter()
    f, pp, rr = {}, {}, 0
    for t, hh, _ in cls:
        if t == 1: f[int(hh)] = f.get(int(hh), 0) + 1
        elif t in (2, 3): pp[int(hh)] = pp.get(int(hh), 0) + 1
        elif t == 4: rr += 1
    up = batch.units[u]
    try:
        res = lo.Caller(int(up["period"]), int(up["readlen"]), int(up["ploidy"]), 2 * float(up["half_depth"]), f,
                        pp, rr, batch.global_lens[up["pe_off"]:up["pe_off"] + up["n_global"]],
                        batch.target_lens[up["tl_off"]:up["tl_off"] + up["n_target"]], int(up["ref_len"]),
                        int(up["minpe"]), maxinsert=int(up["maxinsert"])).evaluate()
        if res["status"] == 0:
            locus = loci[lad]
            lo.calc_PP(res["tot"], res["lik"], int(up["period"]), locus["cutoff_risk"],
                       locus["mutation_nature"] == "increase", locus["inheritance"][-1] == "R")
    except Exception:
        pass
    return t1 - t0, time.perf_counter() - t1


def cpu_baselines(batch, loci, budget_s, cores):
    """(all-core record, 1-core record) on a bounded sample of the batch: units taken round-robin over the 30 loci
    so that the period mix matches the whole workload."""
    from oracle import pyoracle as po
    kind = "reference" if po.have_ref() else "port"
    _CPU.update(batch=batch, loci=loci, classify=po.ref_classify if kind == "reference" else po.classify,
                ls=po.LocusSet(batch.ladders))
    try:                                    # numpy/scipy BLAS and OpenMP pools: one thread per process, so that
        import threadpoolctl                # "cores" is what is really used
        threadpoolctl.threadpool_limits(1)
    except ImportError:
        pass
    n_samples = batch.n_units // len(batch.ladders)
    order = [li * n_samples + s for s in range(n_samples) for li in range(len(batch.ladders))]
    how = ("SW by the reference's ssw.c via oracle/_ref (C driver, no Python per alignment)" if kind == "reference"
           else "SW by the C restatement oracle/sw_oracle.c") + ", likelihood by the numpy/scipy oracle"
    # one core
    done, sw_s, lik_s, t0 = 0, 0.0, 0.0, time.perf_counter()
    for u in order:
        a, b = _cpu_unit(u)
        sw_s, lik_s, done = sw_s + a, lik_s + b, done + 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    one = {"value": done / dt, "unit": "genotypes/s", "cores": 1, "kind": kind,
           "sample": "{} units (round-robin over the 30 loci) of the same batch, {:.1f} s; {}; {:.0f} % of the time "
                     "in SW".format(done, dt, how, 100 * sw_s / max(sw_s + lik_s, 1e-9))}
    # all cores: a pool of worker processes over units, as the reference's Pool over samples (tred.py:521-532);
    # bounded by time, not by count: whatever finished when the budget is over is the sample
    cores = max(1, cores)
    import multiprocessing
    pool = multiprocessing.get_context("fork").Pool(cores)
    try:
        pool.map(_cpu_unit, order[:cores], chunksize=1)          # warm the workers (imports, page-in)
        n, t0 = 0, time.perf_counter()
        for _ in pool.imap_unordered(_cpu_unit, order, chunksize=1):
            n += 1
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    finally:
        pool.terminate()
        pool.join()
    many = {"value": n / dt, "unit": "genotypes/s", "cores": cores, "kind": kind,
            "sample": "{} units (round-robin over the 30 loci) of the same batch finished by {} worker processes in "
                      "{:.1f} s; {}".format(n, cores, dt, how)}
    return many, one


# ---- one rank = one GPU --------------------------------------------------------------------------------------
def rank_main(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    spawned = os.environ.get("TRED_SPAWNED_RANK") == "1"       # started by this script's launcher, not by torchrun
    loci, batch = make_batch(args, rank, world)                 # process pool first, before this process touches the GPU
    stream_batches = []
    if getattr(args, "streamed", 0) > 0:                        # distinct batches of the streamed leg: other seeds
        seed0 = args.seed
        for i in range(args.streamed):
            args.seed = seed0 + 7919 * (i + 1)
            stream_batches.append(make_batch(args, rank, world)[1])
        args.seed = seed0

    import torch
    from tredparse_amd import _lib
    from tredparse_amd.engine import load_model
    n_dev = max(1, torch.cuda.device_count())
    local_rank %= n_dev                                         # (more ranks than devices: ranks share them)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if spawned or torch.cuda.device_count() < world:
            dist.init_process_group("gloo", rank=rank, world_size=world)       # ranks may share a device
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    red_dev = dev if dist is not None and dist.get_backend() == "nccl" else torch.device("cpu")

    ctx = _lib.Context(local_rank)
    ctx.set_ladders(batch.ladders)
    step, w = load_model()
    ctx.set_model(step, w)

    def dv(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    n, g, hs = batch.n_reads, batch.n_units, batch.hist_stride
    d_packed, d_roff, d_rlen = dv(batch.packed.view(np.int32)), dv(batch.read_off), dv(batch.read_len)
    d_uoff, d_ulad = dv(batch.unit_read_off), dv(batch.unit_ladder)
    d_units = dv(batch.units.view(np.uint8))
    d_gl = dv(batch.global_lens if len(batch.global_lens) else np.zeros(1, np.int32))
    d_tl = dv(batch.target_lens if len(batch.target_lens) else np.zeros(1, np.int32))
    d_tag = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_h = torch.zeros(n, dtype=torch.int16, device=dev)
    d_sc = torch.zeros(n, dtype=torch.int16, device=dev)
    d_full = torch.zeros((g, hs), dtype=torch.int32, device=dev)
    d_pref = torch.zeros((g, hs), dtype=torch.int32, device=dev)
    d_rept = torch.zeros((g, hs), dtype=torch.int32, device=dev)
    d_calls = torch.zeros(g * _lib.CALL_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    params = _lib.default_sw_params(max_read_len=args.readlen)
    torch.cuda.synchronize()

    def one_step():
        ctx.genotype_batch(_lib.MEM_DEVICE, d_packed, d_roff, d_rlen, n, d_uoff, d_ulad, d_units, g, params, None,
                           d_gl, len(batch.global_lens), d_tl, len(batch.target_lens), d_tag, d_h, d_sc, hs,
                           d_full, d_pref, d_rept, d_calls)

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        one_step()
    barrier()
    ctx.reset_timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    ctx.sync()
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    elapsed, units_all = elapsed_local, g * args.steps
    if dist is not None:
        t = torch.tensor([elapsed_local], dtype=torch.float64, device=red_dev)
        u = torch.tensor([g * args.steps], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        elapsed, units_all = float(t.item()), int(u.item())
        dist.barrier()

    sw_n, sw_ms = ctx.get_timing(_lib.KERNEL_SW)
    gr_n, gr_ms = ctx.get_timing(_lib.KERNEL_GRID)
    ta_n, ta_ms = ctx.get_timing(_lib.KERNEL_TALLY)
    streamed = streamed_leg(ctx, torch, dev, stream_batches, args, params, g * args.steps / elapsed_local) if stream_batches else None
    calls = np.frombuffer(d_calls.cpu().numpy().tobytes(), _lib.CALL_DTYPE)
    ok = int((calls["status"] == 0).sum())
    # sanity: the genotypes are real (most simulated alleles recovered exactly on the short allele)
    short_ok = float(np.mean((calls["h1"] // batch.units["period"]) == batch.h_true[:, 0]))
    per_rank = {"rank": rank, "device": os.environ.get("TRED_RANK_DEVICE", str(local_rank)),
                "units": g * args.steps, "elapsed_s": elapsed_local, "units_ok_last_step": ok}
    out_dir = os.environ.get("TREDBENCH_OUT")
    if out_dir:
        with open(os.path.join(out_dir, "rank{}.json".format(rank)), "w") as fp:
            json.dump(per_rank, fp)

    if rank == 0:
        value = units_all / elapsed
        alg = algorithmic_cells(batch)
        sw_s = sw_ms / 1e3 / max(sw_n, 1)
        cnt = ctx.get_sw_counters()
        launches = max(sw_n, 1)
        cols = (cnt["trunk_cols"] + cnt["continuation_cols"]) / launches
        # a swept column = one DP column of every read of the quad: read rows only (no padding rows, no empty slots)
        swept = cnt["read_cols"] / launches * batch.readlen
        rpl = rows_per_lane(args.readlen)
        lanes = cols * 4 * 16 * rpl                # cells the wavefronts occupy: 4 read slots x 16 lanes x R rows
        # (the generic variant runs when a ladder's branch alone can pass the score filter: csrc/capi.hip run_sw_device)
        generic = any(max(len(k[0]), len(k[2])) >= 30 for k in batch.ladders)
        census = load_census()
        ceiling, ceiling_basis = mix_ceiling(census, rpl, generic, cnt, launches, swept)
        waves = (census_entry(census, rpl, generic) or {}).get("waves_per_simd", "?")
        alg_bytes = int(batch.packed.nbytes + n * 12 + n * 5)  # packed reads + offsets/lengths in, tag/h/score out
        traffic, traffic_src = (None, "counters of the 1 000-sample 150 bp config3 batch only") \
            if (args.workload, args.readlen, args.samples) != ("config3", 150, 1000) else pmc_traffic("sw_cont_kernel", _lib.version())
        out = {
            "metric": "sample x TRED genotypes/sec at {:g}x {}bp".format(args.coverage_used, args.readlen),
            "value": value, "unit": "genotypes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32 (SW) + f64 (likelihood)", "data": "synthetic",
            "library": _lib.version(),
            "config": {"workload": "{} {} at {} bp; fused SW+tagging -> histograms -> (h1,h2) grid, inputs resident "
                                   "in HBM".format(args.samples, WORKLOADS[args.workload]["what"], args.readlen),
                       "name": args.workload,
                       "units_per_step_per_gpu": g, "reads_per_step_per_gpu": n, "coverage": args.coverage_used,
                       "readlen": args.readlen, "maxinsert": 300, "alleles": WORKLOADS[args.workload]["alleles"],
                       "parallelism": "sample-sharded x{} (no collective)".format(world)},
            "roofline": {"kernel": "sw_cont_kernel<{},{}>".format(rpl, waves), "bound": "valu",
                         "achieved": swept / sw_s / 1e12, "peak": PEAK_TCUPS, "unit": "TCUPS",
                         "frac": swept / sw_s / 1e12 / PEAK_TCUPS, "mix_ceiling_frac": ceiling, "mix_ceiling_basis": ceiling_basis,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                         "note": "achieved = DP cells really swept per launch (read rows x columns of every read; "
                                 "padding rows and empty quad slots excluded) / HIP-event launch time; peak = int32 "
                                 "VALU lane-ops/s / 10 ops per cell (2-cycle issue; integer max/max3/add3 issue at 4 "
                                 "cycles on gfx950, profiles/r02_ubench_valu.txt).  effective_TCUPS = SURVEY 8(d) "
                                 "brute-force cells / the same time: {:.1f}x more cells than are swept, the effect "
                                 "of the exact shortcuts (shared-prefix ladder, suffix continuation vectors, 6-mer "
                                 "strand filter, score-bound pruning, steady-state strand exit), not a hardware rate".format(alg / max(swept, 1)),
                         "avg_launch_ms": sw_s * 1e3, "swept_cells_per_launch": swept,
                         "lane_cells_per_launch": lanes, "lane_occupancy": swept / max(lanes, 1),
                         "algorithmic_cells_per_launch": alg, "effective_TCUPS": alg / sw_s / 1e12,
                         "ladder_cells_per_launch": ladder_cells(batch), "sw_counters": cnt,
                         "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                                 "achieved_GBps": alg_bytes / sw_s / 1e9, "peak_GBps": HBM_PEAK_GBS,
                                 "frac": alg_bytes / sw_s / 1e9 / HBM_PEAK_GBS}},
            "kernels_ms_per_step": {"sw_ladder": sw_ms / max(sw_n, 1), "tally": ta_ms / max(ta_n, 1),
                                    "grid": gr_ms / max(gr_n, 1),
                                    "grid_kde": ctx.get_timing(_lib.KERNEL_GRID_KDE)[1] / max(gr_n, 1),
                                    "grid_prepare": ctx.get_timing(_lib.KERNEL_GRID_PREPARE)[1] / max(gr_n, 1),
                                    "grid_pairs": ctx.get_timing(_lib.KERNEL_GRID_PAIRS)[1] / max(gr_n, 1),
                                    "grid_reduce": ctx.get_timing(_lib.KERNEL_GRID_REDUCE)[1] / max(gr_n, 1)},
            "check": {"units_ok": ok, "units": g, "short_allele_exact_frac": short_ok,
                      "mean_grid_pairs": float(calls["n_pairs"].mean()), "max_grid_pairs": int(calls["n_pairs"].max()),
                      # how the (h1, h2) rectangles are distributed: units whose grid would fit a 32 KB / 128 KB LDS slice, and
                      # the share of all pairs those units hold (why the ml rectangle stays in HBM)
                      "grid_pairs_percentiles": {str(q): int(np.percentile(calls["n_pairs"], q)) for q in (10, 50, 75, 90, 99)},
                      "units_le_4096_pairs": float(np.mean(calls["n_pairs"] <= 4096)), "pairs_in_units_le_4096": float(calls["n_pairs"][calls["n_pairs"] <= 4096].sum() / max(1, calls["n_pairs"].sum())),
                      "units_le_16384_pairs": float(np.mean(calls["n_pairs"] <= 16384)), "pairs_in_units_le_16384": float(calls["n_pairs"][calls["n_pairs"] <= 16384].sum() / max(1, calls["n_pairs"].sum())),
                      "run_pe_frac": float(calls["run_pe"].mean())},
        }
        if streamed is not None:
            out["streamed"] = streamed
        if n_dev < world:                             # (ranks share devices: a rehearsal or a stub, never a scaling point)
            out["devices"], out["oversubscribed"] = n_dev, True
        if out_dir:
            with open(os.path.join(out_dir, "line.json"), "w") as fp:
                json.dump(out, fp)
        else:
            print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def streamed_leg(ctx, torch, dev, batches, args, params, resident_rate):
    """The same step fed from the host: K distinct batches sit in PINNED host memory; while batch k runs on the
    context's stream, batch k + 1 is copied in on a copy stream into the other of two device buffer sets, and the calls
    of batch k - 1 are copied back.  Every step moves its whole input over PCIe (the resident headline replays one
    batch that is already in HBM); the calls of every step are checked against the simulated alleles.  Returns the
    record of the `streamed` leg (value = this rank's genotypes/s)."""
    from tredparse_amd import _lib
    K = len(batches)
    hs = max(b.hist_stride for b in batches)
    names = ("packed", "read_off", "read_len", "unit_read_off", "unit_ladder", "units", "global_lens", "target_lens")

    def arrays(b):
        return (b.packed.view(np.int32), b.read_off, b.read_len, b.unit_read_off, b.unit_ladder, b.units.view(np.uint8),
                b.global_lens if len(b.global_lens) else np.zeros(1, np.int32),
                b.target_lens if len(b.target_lens) else np.zeros(1, np.int32))
    pinned = [[torch.from_numpy(np.ascontiguousarray(a)).pin_memory() for a in arrays(b)] for b in batches]
    sets = []
    for _ in range(2):
        bufs = [torch.empty(max(p[k].numel() for p in pinned), dtype=pinned[0][k].dtype, device=dev) for k in range(len(names))]
        nmax, gmax = max(b.n_reads for b in batches), max(b.n_units for b in batches)
        outs = {"tag": torch.zeros(nmax, dtype=torch.uint8, device=dev), "h": torch.zeros(nmax, dtype=torch.int16, device=dev),
                "sc": torch.zeros(nmax, dtype=torch.int16, device=dev),
                "full": torch.zeros((gmax, hs), dtype=torch.int32, device=dev), "pref": torch.zeros((gmax, hs), dtype=torch.int32, device=dev),
                "rept": torch.zeros((gmax, hs), dtype=torch.int32, device=dev),
                "calls": torch.zeros(gmax * _lib.CALL_DTYPE.itemsize, dtype=torch.uint8, device=dev)}
        sets.append((bufs, outs))
    h_calls = [torch.zeros(b.n_units * _lib.CALL_DTYPE.itemsize, dtype=torch.uint8).pin_memory() for b in batches]
    copy = torch.cuda.Stream(device=dev)
    compute = torch.cuda.ExternalStream(ctx.stream, device=dev)
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    done = [torch.cuda.Event(), torch.cuda.Event()]
    in_bytes = [sum(t.numel() * t.element_size() for t in p) for p in pinned]

    def copy_in(step):
        i, slot = step % K, step % 2
        with torch.cuda.stream(copy):
            if step >= 2:
                copy.wait_event(done[slot])            # the step that read this buffer set has finished
            for k in range(len(names)):
                sets[slot][0][k][:pinned[i][k].numel()].copy_(pinned[i][k], non_blocking=True)
            ready[slot].record(copy)

    def run(step):
        i, slot = step % K, step % 2
        b, (bufs, o) = batches[i], sets[slot]
        compute.wait_event(ready[slot])
        ctx.genotype_batch(_lib.MEM_DEVICE, bufs[0], bufs[1], bufs[2], b.n_reads, bufs[3], bufs[4], bufs[5], b.n_units, params, None,
                           bufs[6], len(b.global_lens), bufs[7], len(b.target_lens), o["tag"], o["h"], o["sc"], hs,
                           o["full"], o["pref"], o["rept"], o["calls"])
        done[slot].record(compute)
        with torch.cuda.stream(copy):                  # the calls go back behind the next batch's copy-in
            copy.wait_event(done[slot])
            h_calls[i].copy_(o["calls"][:h_calls[i].numel()], non_blocking=True)

    def loop(steps):
        copy_in(0)
        for s in range(steps):
            if s + 1 < steps:
                copy_in(s + 1)
            run(s)
        ctx.sync()
        torch.cuda.synchronize()

    loop(min(K, 2))                                    # warm-up (allocations inside the library, first touches)
    steps = max(args.steps, K)            # (the first batch's copy-in is inside the timed region, un-overlapped: the pipeline's fill)
    t0 = time.perf_counter()
    loop(steps)
    dt = time.perf_counter() - t0
    units = sum(batches[s % K].n_units for s in range(steps))
    exact = []
    for b, hc in zip(batches, h_calls):
        calls = np.frombuffer(hc.numpy().tobytes(), _lib.CALL_DTYPE)
        exact.append(float(np.mean((calls["h1"] // b.units["period"]) == b.h_true[:, 0])))
    return {"what": "{} distinct batches in pinned host memory, two device buffer sets: copy-in of batch k + 1 and copy-back "
                    "of the calls of batch k - 1 on a copy stream while batch k runs; every step's input crosses PCIe".format(K),
            "value": units / dt, "unit": "genotypes/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
            "distinct_batches": K, "input_MB_per_step": float(np.mean(in_bytes)) / 1e6,
            "resident_value_same_run": resident_rate, "fraction_of_resident": units / dt / resident_rate,
            "short_allele_exact_frac_per_batch": exact}


def stub_rank_main(args):
    """The rank protocol without a GPU (tests of the launcher): same files, same reduction over gloo."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    units, elapsed_local = args.samples * 30 * args.steps, 0.05 * (rank + 1)
    elapsed, units_all = elapsed_local, units
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
        t = torch.tensor([elapsed_local], dtype=torch.float64)
        u = torch.tensor([units], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        elapsed, units_all = float(t.item()), int(u.item())
        dist.destroy_process_group()
    out_dir = os.environ["TREDBENCH_OUT"]
    with open(os.path.join(out_dir, "rank{}.json".format(rank)), "w") as fp:
        json.dump({"rank": rank, "device": os.environ.get("TRED_RANK_DEVICE", "-"), "units": units,
                   "elapsed_s": elapsed_local}, fp)
    if rank == 0:
        with open(os.path.join(out_dir, "line.json"), "w") as fp:
            json.dump({"metric": "sample x TRED genotypes/sec at 30x 150bp", "value": units_all / elapsed,
                       "unit": "genotypes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                       "ms_per_step": elapsed / args.steps * 1e3, "stub": True}, fp)


# ---- end to end from BAM files (driver processes share the GPUs; the launcher made the files) -------------------------
def _output_digests(work):
    """{samplekey: sha256 over the sample's JSON bytes and its VCF text} of the files a driver wrote (the VCF is hashed
    decompressed: the gzip header carries the time of writing)."""
    import gzip
    import hashlib
    out = {}
    for name in sorted(os.listdir(work)):
        if not name.endswith(".json"):
            continue
        key = name[:-5]
        h = hashlib.sha256()
        with open(os.path.join(work, name), "rb") as fp:
            h.update(fp.read())
        vcf = os.path.join(work, key + ".tred.vcf.gz")
        if os.path.exists(vcf):
            with open(vcf, "rb") as fp:
                h.update(gzip.decompress(fp.read()))
        out[key] = h.hexdigest()
    return out


def e2e_main(args):
    """One driver process of an end-to-end leg: the product path over synthetic BAMs -- tred.run_many = native scans in
    host threads -> GPU batches -> tredCalls -> <key>.json + <key>.tred.vcf.gz in a scratch directory.  The drivers of a
    leg each take a block of the BAMs (shard_range), start together (barrier) and go over their block again and again
    until `--e2e-seconds` have passed (at least once in full); every finished sample is logged with its wall-clock time,
    and the launcher reads the steady-state rate off those logs (steady_state)."""
    import glob
    import shutil
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    out_dir = os.environ["TREDBENCH_OUT"]
    t_proc = time.time()
    with open(os.path.join(args.e2e_child, "truth.json")) as fp:
        truth = json.load(fp)
    bams = sorted(glob.glob(os.path.join(args.e2e_child, "*.bam")))
    if args.e2e_limit > 0:
        bams = bams[:args.e2e_limit]
    import torch
    torch.cuda.init()
    from tredparse_amd import shard, synth_bam, tred
    from tredparse_amd.engine import Engine
    from tredparse_amd.meta import TREDsRepo
    lo, hi = shard.shard_range(len(bams), rank, world)
    repo = TREDsRepo("hg38", sites=os.path.join(args.e2e_child, "no_sites"))
    names = [l["name"] for l in synth_bam.bench_loci()]
    mine = [(os.path.basename(b)[:-4], b, repo, names, 300, False, False, True, True, "ERROR") for b in bams[lo:hi]]
    pinned = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else shard.usable_cpus()
    # never more scan threads than this rank's CPU set holds (spawn_ranks pins a rank to its share of its GPU's node)
    threads = max(1, min(args.e2e_threads or max(1, (shard.usable_cpus() - world) // world), max(len(mine), 1), max(1, pinned)))
    engine = Engine(0)
    work = os.path.join(args.e2e_child, "work{}".format(rank))
    os.makedirs(work, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(work)
    log = []                                     # (wall-clock time, units, short alleles right, short alleles) per sample
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    gpu_inflate, gpu_walk = args.e2e_gpu_inflate == "1", args.e2e_gpu_walk == "1"
    gpu_select = gpu_inflate and gpu_walk and args.e2e_gpu_select == "1"
    sink_lock = __import__("threading").Lock()

    def on_sample(done):                          # from the native writer's threads: the sample's files are written
        tr = truth[done["samplekey"]]
        hits = sum(1 for k, (p, a) in enumerate(zip(done["printed"], done["first_allele"])) if p and a == tr[k][0])
        with sink_lock:
            log.append((time.time(), sum(done["printed"]), hits, len(names)))

    def sink(result):                             # --e2e-python-writer: the dict path (format_scans + to_json / to_vcf)
        tred.write_vcf_json(result, "hg38", repo, names, quiet=True)
        calls, tr = result["tredCalls"], truth[result["samplekey"]]
        units = sum(1 for n in names if n + ".1" in calls)
        hits = sum(1 for k, n in enumerate(names) if calls.get(n + ".1") == tr[k][0])
        with sink_lock:
            log.append((time.time(), units, hits, len(names)))

    deadline = [None]

    def cohort():
        """The driver's block of samples, over and over: at least once, and until the deadline."""
        passes = 0
        while passes == 0 or time.time() < deadline[0]:
            for t in mine:
                if passes > 0 and time.time() >= deadline[0]:
                    return
                yield t
            passes += 1
    kw = dict(lazy_details=True, inflate_device=0 if gpu_inflate else None, gpu_walk=gpu_walk, gpu_select=gpu_select)
    emit = None
    if args.e2e_python_writer:
        kw.update(sink=sink, background_sink=2 if gpu_inflate else 1)
    else:
        # (never more results in flight than the block has files: a driver goes over its block again and again, and two
        #  writer threads must not meet in one <key>.json -- ADVICE r5)
        emit = kw["emit"] = tred.Emitter("hg38", repo, names, workers=3 if gpu_select else 2, on_sample=on_sample,
                                         depth=max(1, min(96, len(mine))))
    try:
        # warm-up: HIP context, ladders, caches -- and, for the GPU-inflate legs, one full chunk through each inflater, whose
        # pinned staging (45 MB per sample of a chunk, three inflaters) stays with the process for the timed cohort
        warm = mine[:2] if not gpu_inflate else mine[:min(len(mine), 3 * args.e2e_batch)]
        tred.run_many(warm, engine, batch=2 if not gpu_inflate else args.e2e_batch, threads=max(2, threads), **kw)
        if emit is not None:
            emit.drain()
        with sink_lock:
            del log[:]
        for k in tred.TIMING:
            tred.TIMING[k] = 0.0
        if dist is not None:
            dist.barrier()
        t0 = time.time()
        cpu0 = time.process_time()
        deadline[0] = t0 + args.e2e_seconds
        tred.run_many(cohort(), engine, batch=args.e2e_batch, threads=threads, **kw)
        if emit is not None:
            emit.close()
        t1 = time.time()
    finally:
        os.chdir(cwd)
    log.sort()
    np.save(os.path.join(out_dir, "e2e_log{}.npy".format(rank)), np.array(log, np.float64).reshape(-1, 4))
    cpu_s = time.process_time() - cpu0
    rec = {"rank": rank, "device": os.environ.get("TRED_RANK_DEVICE", "0"), "t_process": t_proc, "t_begin": t0, "t_end": t1,
           "files": len(mine), "host_threads": threads, "first_chunk": min(args.e2e_batch, threads, max(1, len(mine))),
           "driver_seconds": {k: round(v, 4) for k, v in tred.TIMING.items()},
           "bam_bytes": sum(os.path.getsize(b) for b in bams[lo:hi]), "digests": _output_digests(work),
           "pinned_MB": round(tred.pinned_bytes() / 1e6, 1), "cpu_seconds": round(cpu_s, 3), "samples_logged": len(log)}
    with open(os.path.join(out_dir, "e2e_rank{}.json".format(rank)), "w") as fp:
        json.dump(rec, fp)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    shutil.rmtree(work, ignore_errors=True)


def steady_state(ranks, logs):
    """The rates of one end-to-end leg from its drivers' records and per-sample logs (rows: time, units, hits, loci).
      value            units finished inside the window in which EVERY driver is past its first chunk and none has
                       finished, over that window's length: no pipeline fill, no drain
      startup_s        from the common start (the barrier) to the window's start
      whole_run_value  all units over (last finish - common start): fill and drain included
      first_pass_value the units of every driver's first pass over its files, over the time the slowest driver took"""
    # (a driver that was given no file -- more drivers than files -- has an empty log: it takes no part in the rates)
    live = [(r, l) for r, l in zip(ranks, logs) if len(l)]
    if not live:
        return {"error": "no driver finished a sample"}
    ranks, logs = [r for r, _ in live], [l for _, l in live]
    t0 = min(r["t_begin"] for r in ranks)
    w0 = max(float(l[min(r["first_chunk"], len(l)) - 1, 0]) for r, l in zip(ranks, logs))
    w1 = min(float(l[-1, 0]) for l in logs)
    every = np.concatenate(logs)
    inside = (every[:, 0] > w0) & (every[:, 0] <= w1)
    units_in = float(every[inside, 1].sum())
    last = max(float(l[-1, 0]) for l in logs)
    fp_units = sum(float(l[:r["files"], 1].sum()) for r, l in zip(ranks, logs))
    fp_end = max(float(l[min(r["files"], len(l)) - 1, 0]) for r, l in zip(ranks, logs))
    return {"value": units_in / max(w1 - w0, 1e-9), "seconds": w1 - w0, "units": units_in, "samples": int(inside.sum()),
            "startup_s": w0 - t0, "whole_run_value": float(every[:, 1].sum()) / max(last - t0, 1e-9),
            "whole_run_seconds": last - t0, "first_pass_value": fp_units / max(fp_end - t0, 1e-9),
            "short_allele_exact_frac": float(every[:, 2].sum()) / max(1.0, float(every[:, 3].sum()))}


def e2e_rule(n_devices, usable):
    """THE plan of the end-to-end leg, fixed before anything runs, from (usable host CPUs, GPUs) alone -- the same rule
    the product's command line applies when --drivers is left to it (shard.driver_plan): (ranks, threads per rank)."""
    from tredparse_amd import shard
    per_gpu, threads = shard.driver_plan(usable, n_devices)
    return per_gpu * max(1, n_devices), threads


def e2e_plan(n_devices, usable, drivers_opt=0, threads_opt=0, dense=False):
    """--e2e-sweep only: [(ranks, threads per rank)] of the plans tried beside the rule's -- one driver per GPU, and as
    many drivers per GPU as keep ~7 or ~4 scan threads busy each; dense: also one driver per four, per three and per 2.7
    CPUs (the legs whose BGZF blocks the GPU inflates leave the host a quarter of the work per sample).
    Rank r works on device r mod n_devices; every rank gets an equal share of the CPUs."""
    g = max(1, n_devices)
    tried = [drivers_opt] if drivers_opt else [max(1, usable // (8 * g)), max(1, usable // (5 * g))]
    if dense and not drivers_opt:
        tried += [max(1, usable // (4 * g)), max(1, usable // (3 * g)), max(1, (3 * usable) // (8 * g))]
    plans = []
    for dpg in sorted(set([1] + tried)):
        ranks = dpg * n_devices
        threads = threads_opt or max(1, (usable - 1) // ranks)
        if dense and not threads_opt and not drivers_opt and dpg == max(1, (3 * usable) // (8 * g)) and dpg > usable // (3 * g):
            threads = max(threads, 3)
        plans.append((ranks, threads))
    return plans


def e2e_cohort(root, made, n_files):
    """The first n_files names of a cohort of n_files BAMs made out of the `made` distinct ones: beyond them, hard
    links under new sample keys (one inode, so one copy in the page cache; outputs are per key).  Returns
    {key: true alleles} of the added names."""
    extra = {}
    k = 0
    while len(made) + len(extra) < n_files:
        key, path, h = made[k % len(made)]
        new = "{}x{}".format(key, k // len(made) + 1)
        for ext in (".bam", ".bam.bai"):
            src, dst = os.path.join(root, key + ext), os.path.join(root, new + ext)
            if os.path.exists(src) and not os.path.exists(dst):
                os.link(src, dst)
        extra[new] = h
        k += 1
    return extra


def run_e2e(args, device_counts=(1,), spawn=None, make_bams=None, read_leg=None):
    """Launcher side of the end-to-end legs: make the BAMs once (process pool, no GPU), then for every device count n:
    the host-only leg with one driver per GPU (the one that compares across rounds) and THE plan of e2e_rule (BGZF
    inflate and pair walks on the GPU) over n * per_gpu files -- a constant cohort per GPU; with --e2e-sweep also the
    other plans of e2e_plan.  Every leg's outputs are hashed and must agree.  Returns {n: record}.
    (spawn / make_bams / read_leg: stand-ins for the CPU test of the launcher.)"""
    from tredparse_amd import shard
    import hashlib
    import shutil
    if make_bams is None:
        from tredparse_amd import synth_bam
        make_bams = synth_bam.make_bams
    spawn = spawn or shard.spawn_ranks
    read_leg = read_leg or _read_leg
    root = tempfile.mkdtemp(prefix="tredbench_e2e_")
    device_counts = sorted(set(device_counts))
    usable = shard.usable_cpus()
    per_gpu = max(1, args.e2e_samples)
    distinct = min(per_gpu * max(device_counts), max(1, args.e2e_distinct))
    try:
        t0 = time.perf_counter()
        made = make_bams(root, distinct, seed=args.seed, workers=usable)
        gen_s = time.perf_counter() - t0
        truth = {key: np.asarray(h).tolist() for key, _, h in made}
        truth.update({k: np.asarray(h).tolist() for k, h in e2e_cohort(root, made, per_gpu * max(device_counts)).items()})
        with open(os.path.join(root, "truth.json"), "w") as fp:
            json.dump(truth, fp)
        out = {}
        for n_devices in device_counts:
            n_files = n_devices * per_gpu
            # (ranks, threads, gpu_inflate, gpu_walk, seconds): the continuity leg, then the rule's plan
            plans = [(n_devices, max(1, (usable - 1) // n_devices), False, False, args.e2e_seconds / 2, False)]
            rule = e2e_rule(n_devices, usable)
            if args.e2e_drivers:
                rule = (args.e2e_drivers * n_devices, args.e2e_threads or max(1, (usable - 1) // (args.e2e_drivers * n_devices)))
            on = args.e2e_gpu_inflate != "0"
            walk = on and args.e2e_gpu_walk == "1"
            select = walk and args.e2e_gpu_select == "1"
            plans.append(rule + (on, walk, args.e2e_seconds, select))
            if args.e2e_sweep:
                plans += [(d, t, False, False, args.e2e_seconds / 2, False) for d, t in e2e_plan(n_devices, usable) if d != n_devices]
                plans += [(d, t, True, True, args.e2e_seconds / 2, select) for d, t in e2e_plan(n_devices, usable, dense=True) if (d, t) != rule]
                plans.append(rule + (True, False, args.e2e_seconds / 2, False))
                if select:
                    plans.append(rule + (True, True, args.e2e_seconds / 2, False))      # (the round-5 plan: the selection on the host)
            legs = []
            # the planned leg is run several times over (one 12-s leg cannot tell a 5 % change from the box-to-box
            # and run-to-run spread of +-12 %): the record's value is the MEDIAN of the repeats, min and max beside it
            repeats = max(1, int(getattr(args, "e2e_repeats", 1)))
            runs = [(li, p, rep) for li, p in enumerate(plans) for rep in range(repeats if li == 1 else 1)]
            for li, (drivers, threads, gpu_inflate, gpu_walk, seconds, gpu_select), rep in runs:
                batch = (args.e2e_inflate_batch if gpu_select else min(args.e2e_inflate_batch, 12)) if gpu_inflate else args.e2e_batch
                argv = [sys.executable, os.path.abspath(__file__), "--e2e-child", root, "--e2e-batch", str(batch),
                        "--e2e-threads", str(threads), "--e2e-limit", str(n_files), "--e2e-gpu-inflate", "1" if gpu_inflate else "0",
                        "--e2e-seconds", str(seconds), "--e2e-gpu-walk", "1" if gpu_walk else "0", "--e2e-gpu-select",
                        "1" if gpu_select else "0"]
                if getattr(args, "e2e_python_writer", False):
                    argv.append("--e2e-python-writer")
                out_dir = os.path.join(root, "out{}_{}_{}".format(n_devices, li, rep))
                os.makedirs(out_dir)
                env = dict(os.environ, TREDBENCH_OUT=out_dir)
                leg = {"drivers": drivers, "devices": n_devices, "gpu_inflate": gpu_inflate, "gpu_walk": gpu_walk, "gpu_select": gpu_select,
                       "host_threads_per_driver": threads, "samples_per_gpu_batch": batch, "files": n_files,
                       "role": "host_only_one_driver_per_gpu" if li == 0 else "plan" if li == 1 else "sweep", "repeat": rep}
                codes = spawn(argv, drivers, n_devices, timeout=args.rank_timeout, env=env, stdout=sys.stderr)
                if any(codes):
                    leg["error"] = "exit codes {}".format(codes)
                    legs.append(leg)
                    continue
                ranks, logs = read_leg(out_dir, drivers)
                leg.update(steady_state(ranks, logs))
                leg["unit"] = "genotypes/s"
                leg["host_threads_per_driver"] = ranks[0]["host_threads"]
                leg["warmup_s"] = max(r["t_begin"] - r["t_process"] for r in ranks)
                digests = {}
                for r in ranks:
                    digests.update(r.pop("digests", {}))
                leg["outputs"] = len(digests)
                leg["outputs_sha256"] = hashlib.sha256(json.dumps(sorted(digests.items())).encode()).hexdigest()
                leg["per_driver"] = [{"seconds": round(r["t_end"] - r["t_begin"], 3), "device": r.get("device", "0"),
                                      **r["driver_seconds"]} for r in ranks]
                leg["bam_MB"] = sum(r["bam_bytes"] for r in ranks) / 1e6
                leg["pinned_MB_per_gpu"] = round(sum(r.get("pinned_MB", 0.0) for r in ranks) / max(1, n_devices), 1)
                if all("cpu_seconds" in r for r in ranks):      # process CPU time (all threads, the HIP runtime's included) per sample
                    leg["host_cpu_ms_per_sample"] = round(1e3 * sum(r["cpu_seconds"] for r in ranks) / max(1, sum(r["samples_logged"] for r in ranks)), 3)
                legs.append(leg)
            good = [l for l in legs if "value" in l]
            plan = sorted([l for l in good if l["role"] == "plan"], key=lambda l: l["value"])
            rec = dict(plan[(len(plan) - 1) // 2]) if plan else {"error": "the planned end-to-end leg did not finish"}
            rec.pop("per_driver", None)
            if plan:
                rec["repeats"] = [round(l["value"], 1) for l in sorted(plan, key=lambda l: l["repeat"])]
                rec["min"], rec["max"] = plan[0]["value"], plan[-1]["value"]
                rec["value_is"] = "median of {} repeats of {:g} s".format(len(plan), args.e2e_seconds)
            rec["legs"] = legs
            rec["devices"] = n_devices
            rec["outputs_identical"] = bool(good) and len(set((l["outputs"], l["outputs_sha256"]) for l in good)) == 1 \
                and all(l["outputs"] == n_files for l in good)
            rec["bam_generation_seconds"] = gen_s
            rec["cohort"] = "{} BAM files per GPU ({} distinct, the rest hard links under their own sample keys), gone over " \
                            "repeatedly for {:g} s".format(per_gpu, distinct, args.e2e_seconds)
            rec["page_cache"] = "the files were written moments before the legs by this run: every pass reads them from the " \
                                "page cache and rewrites the same output files; not a cold-storage number"
            host_one = [l for l in good if l["role"] == "host_only_one_driver_per_gpu"]
            if host_one:
                rec["host_only_one_driver_per_gpu"] = {k: host_one[0][k] for k in ("value", "first_pass_value", "seconds", "samples", "startup_s")}
            rec["what"] = ("synthetic 30x 150bp BAMs (tredparse_amd/synth_bam.py: +-10.5 kb around each of the 30 loci) -> native "
                           "scan (BGZF inflate and the pair-length walks on the GPU in the planned leg, on the host in the "
                           "host-only leg; BAI queries, read selection, depth) in host threads -> GPU batches -> tredCalls -> "
                           "JSON + VCF files; `drivers` processes over `devices` GPUs (rank r on device r mod devices).  "
                           "`value` = units finished between every driver's first chunk and the first driver's last sample, "
                           "over that time; the plan is e2e_rule(GPUs, usable CPUs), fixed before the run")
            out[n_devices] = rec
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def run_e2e_wgs(args, spawn=None, make_bams=None, read_leg=None):
    """The whole-genome-shaped leg (a stated second number, not the headline): the same two legs -- host
    only, and the plan with inflate, walks and selection on the GPU -- over synthetic BAMs that also hold what a whole-genome
    file makes this path read through: 30x reads over the 16 kb index windows of every alternative region and the chrY depth
    windows (synth_bam.background_windows: ~570 stretches, 4.7 Mb, ~8 x the blocks of a locus-only sample).  The mate rescue
    of bam_parser.py:217-243 is free on the locus-only files (every alternative-locus fetch is empty there) and is most of a
    sample's blocks here.  A few distinct files hard-linked up to a small cohort; 4 samples per decode call (a sample is
    300 MB inflated)."""
    import functools
    from tredparse_amd import shard
    if make_bams is None:
        from tredparse_amd import synth_bam
        make_bams = functools.partial(synth_bam.make_bams, wgs_like=True)
    a = argparse.Namespace(**vars(args))
    a.e2e_samples, a.e2e_distinct = args.e2e_wgs_samples, min(args.e2e_wgs_distinct, args.e2e_wgs_samples)
    a.e2e_repeats, a.e2e_sweep, a.e2e_inflate_batch, a.e2e_batch = 1, False, 4, 4

    def few_workers(root, n, seed=0, workers=1):       # (a worker holds ~1 GB while it writes a file)
        return make_bams(root, n, seed=seed, workers=max(1, min(workers, 4)))
    rec = run_e2e(a, [1], spawn=spawn, make_bams=few_workers, read_leg=read_leg)[1]
    plan = [l for l in rec.get("legs", []) if l.get("role") == "plan" and "value" in l]
    if plan:
        d = plan[0].get("per_driver", [])
        blocks = sum(x.get("inflate_blocks", 0) for x in d)
        samples = max(1.0, sum(x.get("select_samples", 0) + x.get("select_declined", 0) for x in d) or plan[0].get("samples", 0))
        rec["blocks_per_sample"] = blocks / samples if blocks else None
        rec["walk_call_seconds_per_driver"] = [round(x.get("walk_call", 0.0), 3) for x in d]
    rec["what"] = ("whole-genome-shaped synthetic BAMs (synth_bam.make_bams(wgs_like=True): the 30 loci's +-10.5 kb AND 30x reads over "
                   "every alternative region's index window and the chrY depth windows); same legs and rule as end_to_end")
    return rec


def _read_leg(out_dir, drivers):
    ranks, logs = [], []
    for r in range(drivers):
        with open(os.path.join(out_dir, "e2e_rank{}.json".format(r))) as fp:
            ranks.append(json.load(fp))
        logs.append(np.load(os.path.join(out_dir, "e2e_log{}.npy".format(r))))
    return ranks, logs


# ---- launcher ------------------------------------------------------------------------------------------------
def run_ranks(args, n, n_devices):
    """Start n ranks of this script (rank r on device r mod n_devices) and return (rank 0's line, per-rank records)."""
    from tredparse_amd import shard
    argv = [sys.executable, os.path.abspath(__file__), "--gpus", str(n), "--steps", str(args.steps), "--warmup",
            str(args.warmup), "--samples", str(args.samples), "--seed", str(args.seed), "--readlen", str(args.readlen),
            "--workload", args.workload]
    if args.coverage_set:
        argv += ["--coverage", str(args.coverage)]
    if getattr(args, "streamed", 0) > 0:
        argv += ["--streamed", str(args.streamed)]
    if args.stub:
        argv.append("--stub")
    with tempfile.TemporaryDirectory(prefix="tredbench_") as out_dir:
        env = dict(os.environ, TREDBENCH_OUT=out_dir)
        codes = shard.spawn_ranks(argv, n, 0 if args.stub else n_devices, timeout=args.rank_timeout, env=env,
                                  stdout=sys.stderr)
        if any(codes):
            raise RuntimeError("rank exit codes {}".format(codes))
        with open(os.path.join(out_dir, "line.json")) as fp:
            line = json.load(fp)
        ranks = []
        for r in range(n):
            with open(os.path.join(out_dir, "rank{}.json".format(r))) as fp:
                ranks.append(json.load(fp))
    return line, ranks


def sweep_counts(n_target, n_devices, sweep):
    if not sweep:
        return [n_target]
    top = max(n_target, n_devices)
    ns = sorted(set([n for n in (1, 2, 4, 8) if n <= top] + [n_target]))
    if n_devices <= 1 and ns == [1]:
        ns.append(2)           # 1-GPU box: two ranks on the one GPU, to exercise the multi-rank path
    return ns


SIMDS, CLOCK_HZ = 256 * 4, 2.4e9


def load_census():
    """tredparse_amd/data/sw_isa_census.json (tools/isa_census.py, written by the build next to the library): the VALU
    census of sw_cont_kernel's column blocks per instantiation.  None when it is missing or was not taken from the kernel
    source this tree holds (then mix_ceiling_frac is null: nothing is typed in its place)."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "tredparse_amd", "data", "sw_isa_census.json")
    try:
        with open(path) as fp:
            rec = json.load(fp)
        h = hashlib.sha256()
        for name in ("sw_ladder.hip", "tredgpu_internal.h"):
            with open(os.path.join(here, "tredparse_amd", "csrc", name), "rb") as fp:
                h.update(fp.read())
        return rec if rec.get("source_sha16") == h.hexdigest()[:16] else None
    except (OSError, ValueError):
        return None


def census_entry(census, rpl, generic):
    """The census of sw_cont_kernel<rpl, W, generic> (W: whatever the build compiled it for)."""
    for k in (census or {}).get("kernels", {}).values():
        if k["rows_per_lane"] == rpl and bool(k["generic"]) == bool(generic):
            return k
    return None


def mix_ceiling(census, rpl, generic, counters, launches, swept_cells_per_launch):
    """roofline.mix_ceiling_frac: the fraction of `peak` (2-cycle issue, 10 ops per cell) this launch could have reached had
    the kernel done nothing but sweep the columns it swept, each at the issue cost of its own column block: the kernel's work
    counters say how many trunk and continuation columns the wavefronts went through (tredgpu_get_sw_counters), the build's
    census (tools/isa_census.py) what a column of either kind costs a SIMD in issue cycles -- 2- and 4-cycle VALU
    instructions and the DPP wait states of sw_cont_kernel<R>'s own assembly, at the rates of tools/ubench_valu.hip.
    Everything outside the column blocks (profiles, template ends, best-cell notes: half of the kernel's VALU at 150 bp,
    profiles/r05_sw_isa_column.txt) and every stall is what separates `frac` from it.  (null, reason) without a census."""
    if census is None:
        return None, "no census of this build's sw_ladder.hip (tools/isa_census.py)"
    k = census_entry(census, rpl, generic)
    if k is None or launches <= 0 or swept_cells_per_launch <= 0:
        return None, "no census entry for this instantiation"
    cycles = (counters["trunk_cols"] * k["trunk_cycles_per_column"] + counters["continuation_cols"] * k["free_cycles_per_column"]) / launches
    if cycles <= 0:
        return None, "no columns counted"
    seconds = cycles / (SIMDS * CLOCK_HZ)
    return swept_cells_per_launch / seconds / 1e12 / PEAK_TCUPS, \
        "{:.3g} trunk + {:.3g} continuation wave-columns per launch at {:g} / {:g} issue cycles each (census of sw_cont_kernel<{}>)".format(
            counters["trunk_cols"] / launches, counters["continuation_cols"] / launches, k["trunk_cycles_per_column"],
            k["free_cycles_per_column"], rpl)


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def compact_line(out):
    """The ONE line the driver parses (<= 4 KB): the contract's keys, the roofline and CPU-baseline objects, one small
    object per leg.  Everything else -- per-driver stage times, counters, prose, every plan of a sweep -- is in
    bench_detail.json next to this script (and on stderr)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "library", "stub", "gpus_visible", "virtual_gpus")
    line = {k: _r(out[k]) for k in keep if k in out}
    cfg = out.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "name", "units_per_step_per_gpu", "reads_per_step_per_gpu", "parallelism") if k in cfg}
    r = out.get("roofline")
    if r:
        line["roofline"] = {k: _r(r.get(k), 5) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "mix_ceiling_frac",
                                                      "traffic", "traffic_over_algorithmic", "avg_launch_ms", "effective_TCUPS")}
    if "kernels_ms_per_step" in out:
        line["kernels_ms_per_step"] = {k: _r(v, 3) for k, v in out["kernels_ms_per_step"].items()}
    for key in ("cpu_baseline", "cpu_baseline_1core"):
        c = out.get(key)
        if c:
            line[key] = {k: _r(c[k], 3) for k in ("value", "unit", "cores", "kind") if k in c}
            if key == "cpu_baseline":
                line[key]["sample"] = c.get("sample", "")[:160]
    e = out.get("end_to_end")
    if e:
        line["end_to_end"] = {k: _r(e[k], 3) for k in ("value", "unit", "first_pass_value", "whole_run_value", "startup_s", "drivers",
                                                       "devices", "host_threads_per_driver", "gpu_inflate", "gpu_walk", "gpu_select", "seconds", "samples",
                                                       "files", "pinned_MB_per_gpu", "host_cpu_ms_per_sample", "outputs_identical", "repeats", "min", "max", "oversubscribed", "error") if k in e}
        h = e.get("host_only_one_driver_per_gpu")
        if h:
            line["end_to_end"]["host_only_one_driver_per_gpu"] = {k: _r(h[k], 3) for k in ("value", "first_pass_value", "seconds")}
    w = out.get("end_to_end_wgs")
    if w:
        line["end_to_end_wgs"] = {k: _r(w[k], 3) for k in ("value", "unit", "drivers", "files", "seconds", "samples", "blocks_per_sample",
                                                           "outputs_identical", "error") if k in w}
        if "host_only_one_driver_per_gpu" in w:
            line["end_to_end_wgs"]["host_only"] = _r(w["host_only_one_driver_per_gpu"]["value"], 1)
    if "legs" in out:
        line["legs"] = [{k: _r(l[k], 4) for k in ("leg", "value", "ms_per_step", "frac", "mix_ceiling_frac", "error") if k in l} for l in out["legs"]]
    if "scaling_sweep" in out:
        line["scaling_sweep"] = [dict({k: _r(s_[k], 3) for k in ("n", "value", "ms_per_step", "devices", "oversubscribed") if k in s_},
                                      **({"end_to_end": _r(s_["end_to_end"]["value"], 1)} if "value" in s_.get("end_to_end", {}) else {}))
                                 for s_ in out["scaling_sweep"]]
    line["detail"] = "bench_detail.json"
    text = json.dumps(line)
    if len(text) > 4000:                  # never let the line grow past what the driver reads: drop the optional parts
        for key in ("scaling_sweep", "legs", "kernels_ms_per_step", "cpu_baseline_1core", "end_to_end_wgs"):
            line.pop(key, None)
            text = json.dumps(line)
            if len(text) <= 4000:
                break
    return text


def launcher_main(args):
    from tredparse_amd import shard
    n_devices = 1 if args.stub else shard.visible_gpus()
    if n_devices < 1:
        raise SystemExit("bench.py: no HIP device visible (the hot path has no CPU fallback)")
    if args.e2e_only:                       # tuning runs: the end-to-end legs alone, one line per leg
        rec = run_e2e(args, [min(args.gpus, n_devices)])[min(args.gpus, n_devices)]
        for l in rec.get("legs", []):
            print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in l.items() if k != "per_driver"}), flush=True)
            for d in l.get("per_driver", [])[:2]:
                print("    " + json.dumps(d), flush=True)
        print(json.dumps({"outputs_identical": rec.get("outputs_identical")}), flush=True)
        return
    lines = {}
    scaling = []
    # TRED_VIRTUAL_GPUS=V (shard.virtual_gpus): the box's devices counted as V -- the rehearsal of an 8-GPU run on one GPU
    real = 1 if args.stub else shard.real_gpus(n_devices)
    for n in sweep_counts(args.gpus, n_devices, not args.no_sweep):
        line, ranks = run_ranks(args, n, n_devices)
        lines[n] = line
        rec = {"n": n, "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
               "devices": min(n, n_devices), "ranks": [{k: r[k] for k in ("rank", "device", "units", "elapsed_s")}
                                                        for r in ranks]}
        if n > real:
            rec["oversubscribed"] = True     # several ranks per GPU: exercises the launcher, not a scaling point
        scaling.append(rec)
    out = lines[args.gpus]
    out["scaling_sweep"] = scaling
    out["gpus_visible"] = n_devices
    if real < n_devices:
        out["virtual_gpus"] = {"counted": n_devices, "physical": real,
                               "note": "TRED_VIRTUAL_GPUS: every rank and driver is real, the devices are not -- a rehearsal, no scaling point"}
    if args.e2e_samples > 0 and not args.stub:
        # the product path from BAM files at every device count of the sweep that the box really has (a 1-GPU box:
        # n = 1 only); the line's own record is that of --gpus
        counts = sorted(set(r["n"] for r in scaling if not r.get("oversubscribed") or (real < n_devices and r["n"] <= n_devices)) |
                        {min(args.gpus, n_devices)})
        e2e = run_e2e(args, counts)
        out["end_to_end"] = e2e[min(args.gpus, n_devices)]
        for n, rec in e2e.items():
            if n > real:
                rec["oversubscribed"] = True
        for r in scaling:
            if r["n"] in e2e and (not r.get("oversubscribed") or real < n_devices):
                rec = e2e[r["n"]]
                r["end_to_end"] = {k: rec[k] for k in ("value", "unit", "drivers", "devices", "samples", "seconds",
                                                       "host_threads_per_driver", "outputs_identical") if k in rec}
    if args.e2e_samples > 0 and args.e2e_wgs_samples > 0 and not args.stub:
        try:
            out["end_to_end_wgs"] = run_e2e_wgs(args)
        except Exception as e:                  # (a stated second number: its failure must not take the line down)
            out["end_to_end_wgs"] = {"error": str(e)}
    if args.legs and not args.stub:
        # the configurations that otherwise only have correctness tests, one rank each on device 0: BASELINE
        # configs[4] and the other read lengths (their own sw_cont_kernel instantiations)
        out["legs"] = []
        for spec in args.legs.split(","):
            if spec.startswith("streamed"):
                # the headline's step with its input arriving over PCIe: distinct batches, copies beside the kernels
                leg_args = argparse.Namespace(**vars(args))
                leg_args.streamed = int(spec.split(":")[1]) if ":" in spec else 4
                leg_args.steps, leg_args.warmup = max(args.steps, 4 * leg_args.streamed), 1
                try:
                    line, _ = run_ranks(leg_args, 1, n_devices)
                    rec = dict(line["streamed"])
                    rec.update(leg="streamed", metric=line["metric"], workload=line["config"]["workload"],
                               units_per_step=line["config"]["units_per_step_per_gpu"])
                    out["legs"].append(rec)
                except Exception as e:
                    out["legs"].append({"leg": spec, "error": str(e)})
                continue
            workload, readlen, samples = spec.split(":")
            leg_args = argparse.Namespace(**vars(args))
            leg_args.workload, leg_args.readlen, leg_args.samples = workload, int(readlen), int(samples)
            leg_args.steps, leg_args.warmup, leg_args.coverage_set = min(args.steps, 5), 1, False
            leg_args.coverage = WORKLOADS[workload]["synth"].get("coverage", 30.0)
            try:
                line, _ = run_ranks(leg_args, 1, n_devices)
            except Exception as e:
                out["legs"].append({"leg": spec, "error": str(e)})
                continue
            r = line["roofline"]
            out["legs"].append({"leg": spec, "metric": line["metric"], "value": line["value"], "unit": line["unit"],
                                "ms_per_step": line["ms_per_step"], "steps": line["steps"], "frac": r["frac"],
                                "workload": line["config"]["workload"],
                                "units_per_step": line["config"]["units_per_step_per_gpu"],
                                "reads_per_step": line["config"]["reads_per_step_per_gpu"],
                                "kernels_ms_per_step": line["kernels_ms_per_step"],
                                "mean_grid_pairs": line["check"]["mean_grid_pairs"],
                                "max_grid_pairs": line["check"]["max_grid_pairs"], "units_ok": line["check"]["units_ok"],
                                "mix_ceiling_frac": r.get("mix_ceiling_frac"),
                                "roofline": {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "mix_ceiling_frac",
                                                               "avg_launch_ms", "lane_occupancy", "effective_TCUPS")}})
    if not args.no_cpu_baseline and not args.stub:
        loci, batch = make_batch(args, 0, 1)
        from tredparse_amd import shard as _shard
        cores = args.cpu_cores or _shard.usable_cpus()      # the cgroup quota, not the CPUs merely visible
        many, one = cpu_baselines(batch, loci, args.cpu_budget, cores)
        out["cpu_baseline"] = many
        out["cpu_baseline"]["host_cpus"] = os.cpu_count()
        out["cpu_baseline"]["usable_cpus"] = cores
        out["cpu_baseline_1core"] = one
    # the full record: a file next to the script (and gpurun_out/ when there is one: that directory travels back from
    # the GPU box) and stderr; stdout carries the compact line alone, last
    detail = json.dumps(out)
    for path in (os.path.join(ROOT, "bench_detail.json"),) + (() if args.stub else (os.path.join(ROOT, "gpurun_out", "bench_detail.json"),)):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as fp:
                    fp.write(detail + "\n")
        except OSError:
            pass
    print("bench detail: " + detail, file=sys.stderr, flush=True)
    print(compact_line(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=1000, help="synthetic samples per GPU (x 30 loci)")
    ap.add_argument("--coverage", type=float, default=None, help="override the workload's coverage (30x / 100x)")
    ap.add_argument("--readlen", type=int, default=150, choices=(100, 150, 250, 300, 400),
                    help="read length: selects the sw_cont_kernel instantiation (R = 7 / 10 / 16 / 20 / 32 rows per lane)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="config3",
                    help="config3 = BASELINE configs[2] (the headline); config5 = configs[4] (100x, expanded alleles)")
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds per CPU baseline leg")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker processes of the all-core leg (0: all)")
    ap.add_argument("--no-sweep", action="store_true", help="only --gpus ranks, no 1/2/4/8 sweep")
    ap.add_argument("--rank-timeout", type=float, default=1500.0)
    ap.add_argument("--stub", action="store_true", help="launcher self-test: ranks do no GPU work")
    ap.add_argument("--e2e-samples", type=int, default=4096,
                    help="BAM files per GPU of the end-to-end legs (0: skip them): the same number at every device count")
    ap.add_argument("--e2e-distinct", type=int, default=512,
                    help="distinct synthetic BAMs made; a larger cohort gets the rest as hard links under their own sample keys")
    ap.add_argument("--e2e-wgs-samples", type=int, default=64,
                    help="BAM files of the whole-genome-shaped leg (0: skip it): the loci AND 30x background over every alternative "
                         "region's index window")
    ap.add_argument("--e2e-wgs-distinct", type=int, default=8, help="distinct files among them (the rest are hard links)")
    ap.add_argument("--e2e-repeats", type=int, default=3, help="how often the planned end-to-end leg is run (value = the median)")
    ap.add_argument("--e2e-seconds", type=float, default=8.0,
                    help="how long the planned end-to-end leg's drivers keep going over their files (the host-only leg: half)")
    ap.add_argument("--e2e-sweep", action="store_true", help="also run the other driver plans (they go to bench_detail.json)")
    ap.add_argument("--e2e-batch", type=int, default=16, help="samples per GPU batch in the end-to-end leg")
    ap.add_argument("--e2e-threads", type=int, default=0, help="host threads per driver in the end-to-end leg (0: cores / drivers)")
    ap.add_argument("--e2e-drivers", type=int, default=0, help="driver processes per GPU in the planned end-to-end leg (0: shard.driver_plan's rule)")
    ap.add_argument("--legs", default="streamed:4,config5:150:200,config3:100:500,config3:250:500",
                    help="extra one-GPU legs workload:readlen:samples, comma separated ('' for none)")
    ap.add_argument("--e2e-gpu-inflate", choices=("0", "1"), default="1",
                    help="the planned end-to-end leg has the BAMs' BGZF blocks inflated on the GPU (tred.run_many inflate_device)")
    ap.add_argument("--e2e-gpu-select", choices=("0", "1"), default="1",
                    help="GPU-walk legs: read selection, depth and packing run on the GPU too (tred.run_many gpu_select): no block "
                         "comes back to the host, no host scan runs")
    ap.add_argument("--e2e-gpu-walk", choices=("0", "1"), default="1",
                    help="GPU-inflate legs: the pair-length walks run on the GPU too (tred.run_many gpu_walk), and only the "
                         "blocks of the loci's windows and alternative loci come back")
    ap.add_argument("--e2e-inflate-batch", type=int, default=36, help="samples per GPU batch (= per decode + walk + select call) in the planned leg: 36 fills the device (55.8 / 60.6 / 64.0 k genotypes/s at 12 / 24 / 36; 3.8 GB of pinned staging per GPU); legs without the selection on the device keep 12")
    ap.add_argument("--streamed", type=int, default=0,
                    help="also time the step fed from pinned host memory: this many distinct batches, double-buffered "
                         "copy-in beside the kernels (the default run adds it as the `streamed` leg with 4 batches)")
    ap.add_argument("--e2e-python-writer", action="store_true",
                    help="end-to-end legs through the Python result dicts and writers instead of the native writer (A/B)")
    ap.add_argument("--e2e-only", action="store_true", help="run the end-to-end legs alone and print one line per leg (tuning)")
    ap.add_argument("--e2e-child", help=argparse.SUPPRESS)
    ap.add_argument("--e2e-limit", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    args.coverage_set = args.coverage is not None
    if args.coverage is None:
        args.coverage = WORKLOADS[args.workload]["synth"].get("coverage", 30.0)
    args.coverage_used = args.coverage
    if args.e2e_child or ("RANK" in os.environ and "WORLD_SIZE" in os.environ):
        # a rank: onto the CPUs of its GPU's NUMA node (TRED_CPUSET from shard.spawn_ranks) before anything touches the
        # GPU or allocates pinned memory
        from tredparse_amd import shard as _sh
        _sh.apply_rank_cpuset()
    if args.e2e_child:
        return e2e_main(args)
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:     # a rank (torch.distributed.run or our launcher)
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            args.gpus = world
        if args.stub:
            stub_rank_main(args)
        else:
            rank_main(args)
    else:
        launcher_main(args)


if __name__ == "__main__":
    main()
