#!/usr/bin/env python3
"""Per-phase cycle shares of grid_prepare_kernel / grid_reduce_kernel (how DESIGN section 7's percentages were measured).

  python tools/grid_phase_timers.py build kde|prepare|reduce
      writes tredparse_amd/csrc/_exp_grid_prof.hip -- grid.hip with a clock64() mark in thread 0 at every phase boundary
      of the chosen kernel, each mark adding the cycles since the previous one to a device counter -- and builds
      tredparse_amd/libtredgpu_prof.so from it (run here, no GPU needed).
  TREDGPU_LIB=$PWD/tredparse_amd/libtredgpu_prof.so python tools/grid_phase_timers.py run kde|prepare|reduce
      on the GPU box: one rank of bench.py (4 steps), then the counters as shares.

The marks perturb what they measure (12-13 global atomics per unit on shared counters), so the shares are a guide to
where the latency sits, not a time split; stripped builds (a phase removed, grid ms compared) gave the absolute numbers.
Both generated files are scratch (git-ignored).  The anchors are source lines of grid.hip: the script fails loudly when
one of them has changed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tredparse_amd", "csrc")

PHASES = {
    "kde": ["loop wrap", "ticket", "params + clear", "histogram maxima", "lengths: loads + bins", "lengths: reductions",
            "run_pe", "mean, variance", "sigma, kernel table", "bins copy", "convolution", "normalise + store", "outcome"],
    "prepare": ["loop wrap", "ticket", "params + lists", "pe outcome + pair lengths", "axes", "descriptor", "pool alloc", "obs copy",
                "far rows + scan", "near columns", "rept table", "roll tables", "descriptor write"],
    "reduce": ["loop wrap", "ticket", "descriptor load", "arg-max", "sweep", "P_h2", "block sums", "P_h1", "prefix sums",
               "CI marks", "call", "marginals out"],
}
ANCHORS = {
    # (the KDE's phases sit in the helpers kde_collect / kde_finish, in front of the kernel: its range starts there)
    "kde": ("struct KdeLens {", "// ---- kernel 1: per-unit preparation", [
        "        const tredgpu_unit_params u = uniform_unit(a.units + g);\n        if (!(u.n_global >= 100",
        "        const int32_t* fc = a.full_cnt + (size_t)g * a.hist_stride;\n        const int32_t* pc = a.pref_cnt + (size_t)g * a.hist_stride;\n        int hf = 0",
        "    double s = 0;\n    int vlo = INT_MAX",
        "    for (int o = 32; o > 0; o >>= 1) {\n        vlo = min(vlo",
        "        const int max_full = __builtin_amdgcn_readfirstlane(sh[1]) * u.period;\n",
        "    const double mean = in.total / n;\n",
        "    const double factor = pow((double)n, -1. / 5);\n",
        "    for (int i = tid; i < SPAN + 2; i += NT) khist[kswz(i)]",
        "    const int x0 = tid * XPER;\n",
        "    double part = 0;\n#pragma unroll\n    for (int qx = 0; qx < XPER; ++qx)\n        if (x0 + qx < SPAN) part += acc[qx];",
        "        if (tid == 0) a.unit_kde_rc[g] = rc;\n    }\n}"]),
    "prepare": ("__global__ __launch_bounds__(NT, PREP_WAVES) void grid_prepare_kernel", "// ---- kernel 2: every pair of every unit", [
        "        const tredgpu_unit_params u = uniform_unit(a.units + g);\n",
        "        // ---- paired-end model (models.py:131-132, 428-439): built by grid_kde_kernel ----\n",
        "        // ---- grid axes (models.py:239-257) ----\n",
        "        UnitDesc d;\n        d.status = status;",
        "        // ---- room in the pool and a run of work items;",
        "        // ---- hand the unit's lists to the slot ----\n",
        "        // ---- rows: count of valid h2 per h1 (h1 <= h2), dump offsets;",
        "        d.n_pairs = __builtin_amdgcn_readfirstlane(S.row_off[nrow]);\n",
        "        // ---- the repeat-only table\n",
        "        // ---- the paired-end tables: .5 * roll(h)[x_t] per row and (two alleles) per column\n",
        "        if (tid == 0) descs[g] = d;\n    }\n}"]),
    "reduce": ("template <bool JOINT>   // JOINT: also the sparse joint distribution", "hipError_t launch_pe_kde", [
        "        if (descs[g].status == UNIT_SKIP) continue;",
        "        // ---- arg-max over the items' winners ----\n",
        "        // ---- one pass over the grid: exp(ml - max)",
        "        // marginal P_h2 by distinct h2 value.",
        "        all = block_sum_r(all, S.red);",
        "        // marginal P_h1 by distinct h1 value, rows merged",
        "        // ---- calc_CI, models.py:319-340 on each marginal",
        "        for (int which = 0; which < 2; ++which) {\n            const double* P = which ? S.ph2 : S.ph1;",
        "        if (tid == 0) {\n            for (int which = 0; which < 2; ++which) {\n                call.ci[2 * which]",
        "        if (a.marg != nullptr) {\n            for (int m = tid; m < a.marg_stride; m += NR) {"]),
}


def mark(k):
    # (the previous mark's time lives in LDS: the KDE's phases are in helper functions, out of the kernel's scope)
    return ("        if (threadIdx.x == 0) { const long long t_ = clock64(); atomicAdd(&g_prof[%d], (unsigned long long)(t_ - g_tprev));"
            " g_tprev = t_; }\n" % k)


def build(which):
    s = open(os.path.join(SRC, "grid.hip")).read()
    head = "namespace tredgpu {\nnamespace {"
    assert head in s
    s = s.replace(head, "namespace tredgpu {\n__device__ unsigned long long g_prof[16];\n__shared__ long long g_tprev;\nnamespace {", 1)
    first, last, anchors = ANCHORS[which]
    a, b = s.index(first), s.index(last)
    body = s[a:b]
    top = "    while (true) {\n        __syncthreads();\n"
    assert top in body
    body = body.replace(top, "    if (threadIdx.x == 0) g_tprev = clock64();\n" + top + mark(0), 1)
    for k, anchor in enumerate(anchors, 1):
        assert anchor in body, anchor
        body = body.replace(anchor, mark(k) + anchor, 1)
    s = s[:a] + body + s[b:]
    s += ('\nextern "C" int tredgpu_debug_prof(unsigned long long* out) {\n'
          "    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tredgpu::g_prof), sizeof(unsigned long long) * 16);\n}\n")
    open(os.path.join(SRC, "_exp_grid_prof.hip"), "w").write(s)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-Wno-unused-function", "-shared", "-o", os.path.join(ROOT, "tredparse_amd", "libtredgpu_prof.so"),
                           "capi.hip", "sw_ladder.hip", "inflate_decode.hip", "walk.hip", "inflater_api.hip", "_exp_grid_prof.hip"], cwd=SRC)


def run(which):
    import ctypes as C
    import numpy as np
    sys.path.insert(0, ROOT)
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511")):
        os.environ.setdefault(k, v)
    sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"]
    import bench
    bench.main()
    from tredparse_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    out = np.zeros(16, np.uint64)
    lib.tredgpu_debug_prof.argtypes = [C.c_void_p]
    rc = lib.tredgpu_debug_prof(out.ctypes.data)
    assert rc == 0, rc
    total = float(out.sum())
    for k, name in enumerate(PHASES[which]):
        print("%2d %-18s %14d  %5.1f %%" % (k, name, int(out[k]), 100 * out[k] / total), file=sys.stderr)


if __name__ == "__main__":
    if len(sys.argv) != 3 or sys.argv[1] not in ("build", "run") or sys.argv[2] not in PHASES:
        sys.exit(__doc__)
    (build if sys.argv[1] == "build" else run)(sys.argv[2])
