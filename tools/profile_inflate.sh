#!/bin/bash
# tools/profile_inflate.sh TAG [samples per call] -- run ON THE GPU BOX (through gpurun): the rocprofv3 evidence for
# inflate_kernel.  The profiled program is tools/inflate_prof.hip, built here into /tmp (a plain binary,
# directly after `--`), decoding the BGZF blocks of a synthetic 30x BAM of the bench (made here as well, by a plain process).
#   gpurun_out/prof_TAG/kernel_stats.csv   --kernel-trace --stats
#   gpurun_out/prof_TAG/pmc_*              one counter pass each (never combined with a trace option other than --kernel-trace)
#   gpurun_out/prof_TAG/pmc_summary.json   tools/pmc_to_json.py
set -u
TAG=${1:-r04_inflate}
M=${2:-8}     # (counter passes serialise the dispatches: 56 samples per call did not finish in 25 minutes)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
WORK=/tmp/profile_inflate_$$
mkdir -p $WORK
BIN=$WORK/inflate_prof
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $BIN $ROOT/tools/inflate_prof.hip 2> $OUT/build.err || { tail -5 $OUT/build.err; exit 1; }
python3 $ROOT/tools/walk_prof.py make $WORK > /dev/null 2>&1
BAM=$(ls $WORK/*.bam | head -1)
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BIN $BAM $M 6 1 > "$OUT/run_stats.json" 2> "$OUT/stats.err"
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
pass() {
    local name=$1; shift
    timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $BIN $BAM $M 2 1 > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass sq3 SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_MISC
python3 "$ROOT/tools/pmc_to_json.py" "$OUT" > "$OUT/pmc_summary.json" 2> "$OUT/pmc_summary.err"
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
find "$OUT" -name '*counter_collection.csv' -size +4M -delete
ls "$OUT"; tail -3 "$OUT"/*.err | head -40
