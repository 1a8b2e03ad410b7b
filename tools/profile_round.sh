#!/bin/bash
# tools/profile_round.sh TAG -- run ON THE GPU BOX (through gpurun): the rocprofv3 evidence of one build.
#   gpurun_out/prof_TAG/stats   --kernel-trace --stats of one rank of bench.py (per-kernel durations)
#   gpurun_out/prof_TAG/pmc_*   one counter pass each (FETCH_SIZE and WRITE_SIZE do not fit one pass; PMC passes
#                               are never combined with any trace option other than --kernel-trace)
#   gpurun_out/prof_TAG/pmc_summary.json (carries tredgpu_version(): the build the counters belong to) + kernel_stats.csv +
#                               bench_under_rocprof.json + ubench_valu.txt
# The profiled program is the rank body itself (RANK=0 WORLD_SIZE=1 in the environment): `python3 bench.py` directly
# after `--`, no launcher hop.  Copy what should be judged into profiles/.
set -u
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
export TRED_BENCH_WORKERS=1     # no forked batch builders out of a process the profiler has put on the GPU
# BENCH_ARGS: another configuration of the same rank body (e.g. "--readlen 250 --samples 500", "--workload config5
# --samples 200"); PASSES: which counter passes to take (default: all)
BENCH="python3 $ROOT/bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
PASSES=${PASSES:-fetch write sq sq2}

timeout -k 5 ${PASS_TIMEOUT:-900} rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;

pass() {   # pass NAME counters...
    local name=$1; shift
    timeout -k 5 ${PASS_TIMEOUT:-900} rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err"
}
case " $PASSES " in *" fetch "*) pass fetch FETCH_SIZE;; esac
case " $PASSES " in *" write "*) pass write WRITE_SIZE;; esac
case " $PASSES " in *" sq "*) pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY;; esac
case " $PASSES " in *" sq2 "*) pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE;; esac

python3 "$ROOT/tools/pmc_to_json.py" "$OUT" > "$OUT/pmc_summary.json" 2> "$OUT/pmc_summary.err"

if [ -z "${BENCH_ARGS:-}" ]; then
hipcc --offload-arch=gfx950 -O3 -o "$OUT/ubench_valu" "$ROOT/tools/ubench_valu.hip" 2> "$OUT/ubench.err" && "$OUT/ubench_valu" > "$OUT/ubench_valu.txt" 2>&1
rm -f "$OUT/ubench_valu"
fi
# keep the merged-back directory small: drop the per-dispatch traces, keep the summaries
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
ls -la "$OUT"
