#!/bin/bash
# tools/inflate_capacity.sh [TAG] -- run ON THE GPU BOX: inflate_kernel's rate against the blocks per call (tools/inflate_prof.hip
# over the BGZF blocks of one synthetic 30x BAM of the bench, m copies per call, no profiler): what the decoder gives at 0.6, 1.2,
# 2.5, 5 ... wavefronts per resident slot (7 per SIMD x 1 024 SIMDs = 7 168), i.e. whether a from-BAM leg at N samples/s has it
# saturated.  One JSON line per m into gpurun_out/TAG_inflate_capacity.txt.
set -u
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WORK=/tmp/inflate_capacity_$$
mkdir -p $WORK "$ROOT/gpurun_out"
OUT=$ROOT/gpurun_out/${TAG}_inflate_capacity.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $WORK/inflate_prof $ROOT/tools/inflate_prof.hip 2> $WORK/build.err || { tail -5 $WORK/build.err; exit 1; }
python3 $ROOT/tools/walk_prof.py make $WORK > /dev/null 2>&1
BAM=$(ls $WORK/*.bam | head -1)
: > $OUT
for M in 1 2 4 8 13 16 26 32 48 64; do
    echo -n "{\"samples_per_call\": $M, \"run\": " >> $OUT
    timeout 120 $WORK/inflate_prof $BAM $M 8 1 | tr -d '\n' >> $OUT
    echo "}" >> $OUT
done
cat $OUT
rm -rf $WORK
