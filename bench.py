#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X STR-genotyping hot path.

Metric (BASELINE.json): sample x TRED genotypes / second at 30x 150 bp.
One "step" = one pass of the whole hot path (template SW + tagging -> histograms -> (h1,h2)
likelihood grid) over one resident batch of `--samples` synthetic samples x 30 loci
(BASELINE.json configs[2]: "1k synthetic 30x 150 bp BAMs x 30 TREDs on 1 GPU").  Inputs are packed
and already in HBM when the timed region starts; results stay on the device.

Multi-GPU: one process per GPU (torchrun), every rank owns its own `--samples` samples (weak scaling,
samples are independent -- no data-path collective); torch.distributed is used only for the
barrier and the max-over-ranks reduction of the wall time.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel = sw_ladder, VALU-bound; HIP-event
timed inside this script) and, at N=1, `cpu_baseline` (the reference's own ssw.c compiled in
oracle/_ref driven natively + the numpy likelihood oracle, on a bounded sample of the same batch).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9      # int32 VALU lane-ops/s: 256 CU x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md)
OPS_PER_CELL = 10.0                        # minimum VALU ops of one affine-gap local-alignment cell
PEAK_TCUPS = PEAK_LANE_OPS / OPS_PER_CELL / 1e12
HBM_PEAK_GBS = 8000.0


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


def cpu_baseline(batch, loci, budget_s=20.0):
    """Reference CPU path on a bounded sample: every alignment by the reference's ssw.c (oracle/_ref,
    one ssw_init + ssw_align per pair exactly as ssw_wrap.py does) + numpy likelihood oracle."""
    from oracle import lik_oracle as lo
    from oracle import pyoracle as po
    from tredparse_amd import synth
    kind = "reference" if po.have_ref() else "port"
    classify = po.ref_classify if kind == "reference" else po.classify
    ls = po.LocusSet(batch.ladders)
    n_samples = batch.n_units // len(batch.ladders)
    # sample units round-robin over loci so the period mix matches the batch
    order = [li * n_samples + s for s in range(n_samples) for li in range(len(batch.ladders))]
    done, t0 = 0, time.perf_counter()
    for u in order:
        r0, r1 = int(batch.unit_read_off[u]), int(batch.unit_read_off[u + 1])
        reads = [synth.decode(r) for r in batch.codes[r0:r1]]
        lad = int(batch.unit_ladder[u])
        cls = classify(reads, np.full(len(reads), lad, np.int32), ls, threads=1)
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
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "genotypes/s", "cores": 1, "kind": kind,
            "sample": "{} units (round-robin over the 30 loci) of the same batch, {:.1f} s; SW by the reference's "
                      "ssw.c via oracle/_ref (C driver, no Python per alignment), likelihood by the numpy oracle"
                      .format(done, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=1000, help="synthetic samples per GPU (x 30 loci)")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    from tredparse_amd import synth
    loci = bench_loci(synth.load_loci())
    p = synth.SynthParams(coverage=args.coverage, readlen=150)
    # synthetic data first (process pool), before this process touches the GPU
    workers = max(1, min(len(loci), (os.cpu_count() or 8) // max(1, world)))
    batch = synth.build_batch(args.seed + rank, loci, args.samples, p, workers=workers)

    import torch
    from oracle import lik_oracle as lo   # model constants only (data file parser)
    from tredparse_amd import _lib
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ   # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ctx = _lib.Context(local_rank)
    ctx.set_ladders(batch.ladders)
    step, w = lo.load_model()
    ctx.set_model(np.array([step[k] for k in range(1, 7)]), np.array(w))

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
    params = _lib.default_sw_params(max_read_len=150)
    torch.cuda.synchronize()

    def one_step():
        ctx.genotype_batch(_lib.MEM_DEVICE, d_packed, d_roff, d_rlen, n, d_uoff, d_ulad, d_units, g, params, None,
                           d_gl, len(batch.global_lens), d_tl, len(batch.target_lens), d_tag, d_h, d_sc, hs,
                           d_full, d_pref, d_rept, d_calls)

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if use_dist:
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
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    sw_n, sw_ms = ctx.get_timing(_lib.KERNEL_SW)
    gr_n, gr_ms = ctx.get_timing(_lib.KERNEL_GRID)
    ta_n, ta_ms = ctx.get_timing(_lib.KERNEL_TALLY)
    calls = np.frombuffer(d_calls.cpu().numpy().tobytes(), _lib.CALL_DTYPE)
    ok = int((calls["status"] == 0).sum())
    # sanity: the genotypes are real (most simulated alleles recovered exactly on the short allele)
    short_ok = float(np.mean((calls["h1"] // batch.units["period"]) == batch.h_true[:, 0]))

    if rank == 0:
        units_total = g * world * args.steps
        value = units_total / elapsed
        alg = algorithmic_cells(batch)
        sw_s = sw_ms / 1e3 / max(sw_n, 1)
        cnt = ctx.get_sw_counters()
        cols_per_launch = (cnt["trunk_cols"] + cnt["continuation_cols"]) / max(sw_n, 1)
        swept = cols_per_launch * 4 * 160          # a column sweep = 4 reads x 16 lanes x 10 rows
        alg_bytes = int(batch.packed.nbytes + n * 12 + n * 5)  # packed reads + offsets/lengths in, tag/h/score out
        out = {
            "metric": "sample x TRED genotypes/sec at 30x 150bp",
            "value": value, "unit": "genotypes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32 (SW) + f64 (likelihood)", "data": "synthetic",
            "config": {"workload": "{} synthetic 30x 150bp samples x 30 TRED loci per GPU (BASELINE configs[2]); "
                                   "fused SW+tagging -> histograms -> (h1,h2) grid, inputs resident in HBM"
                                   .format(args.samples),
                       "units_per_step_per_gpu": g, "reads_per_step_per_gpu": n, "coverage": args.coverage,
                       "readlen": 150, "maxinsert": 300, "alleles": "uniform 5..60 units (SURVEY 8d)",
                       "parallelism": "sample-sharded x{} (no collective)".format(world)},
            "roofline": {"kernel": "sw_cont_kernel<10,4>", "bound": "valu", "achieved": alg / sw_s / 1e12,
                         "peak": PEAK_TCUPS, "unit": "TCUPS", "frac": alg / sw_s / 1e12 / PEAK_TCUPS,
                         "traffic": None,
                         "note": "achieved = brute-force forward cells of SURVEY 8(d) per launch / HIP-event launch "
                                 "time; peak = int32 VALU lane-ops/s / 10 ops per cell. frac > 1 is the effect of the "
                                 "exact shortcuts (shared-prefix ladder, suffix continuation vectors, strand filter, "
                                 "score-bound pruning): the kernel sweeps {:.1f}x fewer cells than the brute-force "
                                 "count (trunk + continuation-pass columns; a combined template costs about one "
                                 "more column); see swept_*"
                                 .format(alg / max(swept, 1)),
                         "avg_launch_ms": sw_s * 1e3, "algorithmic_cells_per_launch": alg,
                         "ladder_cells_per_launch": ladder_cells(batch),
                         "swept_cells_per_launch": swept, "swept_TCUPS": swept / sw_s / 1e12,
                         "swept_frac_of_peak": swept / sw_s / 1e12 / PEAK_TCUPS,
                         "sw_counters": cnt,
                         "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                                 "achieved_GBps": alg_bytes / sw_s / 1e9, "peak_GBps": HBM_PEAK_GBS,
                                 "frac": alg_bytes / sw_s / 1e9 / HBM_PEAK_GBS}},
            "kernels_ms_per_step": {"sw_ladder": sw_ms / max(sw_n, 1), "tally": ta_ms / max(ta_n, 1),
                                    "grid": gr_ms / max(gr_n, 1)},
            "check": {"units_ok": ok, "units": g, "short_allele_exact_frac": short_ok,
                      "mean_grid_pairs": float(calls["n_pairs"].mean()), "max_grid_pairs": int(calls["n_pairs"].max()),
                      "run_pe_frac": float(calls["run_pe"].mean())},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch, loci, args.cpu_budget)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
