#!/bin/bash
# tools/profile_all.sh [ROUND = r05] -- run ON THE GPU BOX (through gpurun): every rocprofv3 summary of the round, taken on
# ONE build (the tree's libtredgpu.so), in one go:
#   headline (config3, 150 bp, 1 000 samples), 100 bp, 250 bp, configs[4]     tools/profile_round.sh: kernel stats + PMC passes
#   front end: inflate + pair walk + alternative-locus walk at 16 and 48 samples per call; inflate_kernel alone
#                                                                              rocprofv3 ... -- python3 tools/walk_prof.py
# Results land in gpurun_out/prof_<ROUND>_*/; tools/collect_profiles.py copies the summaries into profiles/ under the
# round's names and checks that all of them carry the same library_version (tests/test_profiles.py checks it again on CPU).
set -u
R=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
STEPS=3 bash tools/profile_round.sh ${R}_headline > gpurun_out/prof_${R}_headline.log 2>&1
PASSES="fetch write sq" BENCH_ARGS="--readlen 100 --samples 500" bash tools/profile_round.sh ${R}_len100 > gpurun_out/prof_${R}_len100.log 2>&1
PASSES="fetch write sq" BENCH_ARGS="--readlen 250 --samples 500" bash tools/profile_round.sh ${R}_len250 > gpurun_out/prof_${R}_len250.log 2>&1
PASSES="fetch write sq" BENCH_ARGS="--workload config5 --samples 200" bash tools/profile_round.sh ${R}_config5 > gpurun_out/prof_${R}_config5.log 2>&1
# ---- the front-end kernels (a process under the profiler must not fork: the BAMs are made first, by a plain process)
BAMS=/tmp/prof_bams_$$
python3 tools/walk_prof.py make $BAMS > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for M in 16 48; do
    OUT=$ROOT/gpurun_out/prof_${R}_walk$M
    mkdir -p "$OUT"
    W="python3 $ROOT/tools/walk_prof.py $M $ROOT/tredparse_amd/libtredgpu.so $BAMS"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $W > "$OUT/run.json" 2> "$OUT/stats.err"
    find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
    if [ "$M" = 16 ]; then      # (counter passes serialise the dispatches: the small call only)
        pass() { local name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $W > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err"; }
        pass fetch FETCH_SIZE
        pass write WRITE_SIZE
        pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
        pass sq3 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_IFETCH
    fi
    python3 "$ROOT/tools/pmc_to_json.py" "$OUT" > "$OUT/pmc_summary.json" 2> "$OUT/pmc_summary.err"
    find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
    find "$OUT" -name '*counter_collection.csv' -size +4M -delete
done
cd "$ROOT"
# ---- the decoder alone (tools/inflate_prof.hip, built on the box)
bash tools/profile_inflate.sh ${R}_inflate 8 > gpurun_out/prof_${R}_inflate.log 2>&1
python3 tools/collect_profiles.py $R
