#!/usr/bin/env python3
"""Instruction census of sw_cont_kernel's column loop, per instantiation, from the compiler's own assembly (build time:
tredparse_amd/csrc/Makefile writes tredparse_amd/data/sw_isa_census.json next to the library it describes).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only csrc/sw_ladder.hip

For every sw_cont_kernel<R, W, GENERIC> the basic blocks are cut at labels; a COLUMN block is one that holds the recurrences
of R cells -- two v_max3_i32 per cell (H without and with the vertical-gap term: csrc/sw_ladder.hip sweep_column) -- i.e. at
least 2 R of them.  Per such block the census counts the vector-ALU instructions by issue class on gfx950 (measured by
tools/ubench_valu.hip, profiles/r04_ubench_valu.txt: v_max / v_max3 / v_add3 / v_cmp / v_and_or / any DPP-modified ALU op
issue at 4.1 cycles per wave64, v_add / v_sub / v_mov / v_and at 2.1; v_cndmask separately), the scalar instructions and
the s_nop wait states.  bench.py turns the census and the kernel's own work counters (tredgpu_get_sw_counters: columns swept
by how many reads) into roofline.mix_ceiling_frac: what the column loop's instruction mix allows of the 10-op-per-cell peak
when nothing else costs anything -- a number per instantiation, computed, not typed.

usage: python tools/isa_census.py [--asm file.s] [--out tredparse_amd/data/sw_isa_census.json]
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tredparse_amd", "csrc", "sw_ladder.hip")
# issue cycles per wave64 instruction on gfx950 (profiles/r04_ubench_valu.txt, re-taken by tools/profile_all.sh)
CYCLES = {"valu2": 2.16, "valu4": 4.12, "cndmask": 4.12, "salu": 0.0, "nop": 2.0}
FOUR = ("v_max", "v_min", "v_max3", "v_min3", "v_add3", "v_cmp", "v_and_or", "v_lshl_or", "v_lshl_add", "v_add_lshl", "v_med3", "v_bfe",
        "v_mul", "v_mad", "v_perm", "v_xad", "v_or3", "v_cmpx")


def classify(op, text):
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if not op.startswith("v_"):
        return "other"                     # ds_ / global_ / buffer_ / flat_ ...
    if op.startswith("v_cndmask"):
        return "cndmask"
    if "row_shr" in text or "row_bcast" in text or "quad_perm" in text or "row_shl" in text or "wave_shr" in text or "row_mirror" in text:
        return "valu4"                     # DPP-modified: issued at the slow rate whatever the operation
    if any(op.startswith(p) for p in FOUR) or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "valu4"
    return "valu2"


def census(asm):
    """{"R,W,generic": {...}} from the assembly text."""
    out = {}
    kern = re.compile(r"^_ZN7tredgpu.*sw_cont_kernelILi(\d+)ELi(\d+)ELb([01])E.*:\s*;")
    lines = asm.split("\n")
    starts = [(i, m) for i, m in ((i, kern.match(l)) for i, l in enumerate(lines)) if m]
    for i0, m in starts:
        R, W, G = int(m.group(1)), int(m.group(2)), m.group(3) == "1"
        i1 = next((j for j in range(i0 + 1, len(lines)) if lines[j].startswith(".Lfunc_end")), len(lines))
        blocks, cur, name = [], [], "entry"
        for l in lines[i0 + 1:i1]:
            if re.match(r"^\.LBB\d+_\d+:", l):
                if cur:
                    blocks.append((name, cur))
                cur, name = [], l.split(":")[0].strip()
                continue
            t = l.strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue
            cur.append(t)
        if cur:
            blocks.append((name, cur))
        cols = []
        for name, body in blocks:
            ops = [t.split()[0] for t in body]
            if sum(1 for o in ops if o.startswith("v_max3_i32")) < 2 * R:
                continue
            c = {"valu2": 0, "valu4": 0, "cndmask": 0, "salu": 0, "nop": 0, "other": 0}
            for o, t in zip(ops, body):
                c[classify(o, t)] += 1
            c["max3"] = sum(1 for o in ops if o.startswith("v_max3_i32"))
            c["cycles"] = round(sum(CYCLES.get(k, 0.0) * v for k, v in c.items() if k in CYCLES), 1)
            c["label"] = name
            cols.append(c)
        if not cols:
            continue
        # Two families of column blocks (csrc/sw_ladder.hip): sweep_column_free<R, letter> -- the continuation passes: exactly
        # two v_max3 per cell, nothing else --, and sweep_column -- the exact trunk sweep, whose best-cell tracking adds v_max3s
        # and LDS notes.  Each family's column costs the mean over its blocks (the period-3 loop goes through its blocks in turn).
        lo = min(c["max3"] for c in cols)
        free_blocks = [c for c in cols if c["max3"] == lo]
        trunk_blocks = [c for c in cols if c["max3"] > lo] or free_blocks

        def mean(blocks):
            m = {k: round(sum(b[k] for b in blocks) / len(blocks), 2) for k in ("valu2", "valu4", "cndmask", "salu", "nop", "other", "max3", "cycles")}
            m["blocks"] = [b["label"] for b in blocks]
            return m
        trunk, free = mean(trunk_blocks), mean(free_blocks)
        vgpr = None
        for l in lines[i1:i1 + 400]:
            mm = re.search(r"; NumVgprs: (\d+)", l)
            if mm:
                vgpr = int(mm.group(1))
                break
        out["{},{},{}".format(R, W, int(G))] = {
            "rows_per_lane": R, "waves_per_simd": W, "generic": G, "column_blocks": len(cols), "vgprs": vgpr,
            "trunk_column": trunk, "free_column": free,
            "trunk_cycles_per_column": trunk["cycles"], "free_cycles_per_column": free["cycles"],
            "trunk_valu_per_cell": round((trunk["valu2"] + trunk["valu4"] + trunk["cndmask"]) / R, 3)}
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", help="assembly already made (else hipcc -S is run on csrc/sw_ladder.hip)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tredparse_amd", "data", "sw_isa_census.json"))
    ap.add_argument("--arch", default="gfx950")
    a = ap.parse_args(argv)
    if a.asm:
        asm = open(a.asm).read()
    else:
        with tempfile.TemporaryDirectory() as tmp:
            s = os.path.join(tmp, "sw_ladder.s")
            subprocess.check_call([os.environ.get("HIPCC", "hipcc"), "--offload-arch=" + a.arch, "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                                   "--cuda-device-only", "-o", s, SRC])
            asm = open(s).read()
    h = hashlib.sha256()
    for name in ("sw_ladder.hip", "tredgpu_internal.h"):
        with open(os.path.join(ROOT, "tredparse_amd", "csrc", name), "rb") as fp:
            h.update(fp.read())
    rec = {"what": "VALU census of sw_cont_kernel's column block per instantiation <R, W, generic> (tools/isa_census.py)",
           "source_sha16": h.hexdigest()[:16], "cycles_per_instruction": CYCLES, "kernels": census(asm)}
    with open(a.out, "w") as fp:
        json.dump(rec, fp, indent=1, sort_keys=True)
        fp.write("\n")
    print("{}: {} instantiations".format(a.out, len(rec["kernels"])))
    return 0 if rec["kernels"] else 1


if __name__ == "__main__":
    sys.exit(main())
