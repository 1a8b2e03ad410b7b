#!/bin/bash
# tools/sw_waves_ab.sh -- GPU box: the 100 bp and 250 bp legs with sw_cont_kernel<7,.> at 5 instead of 4 waves per SIMD (96 VGPRs,
# 3 spilled) and <16,.> at 3 instead of 2 (168 VGPRs, 50 spilled), against the tree's build.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_sw_waves_ab.txt
echo "# ms per step of sw_cont_kernel and roofline.frac, legs config3:100:500 and config3:250:500 (bench.py --legs); build A = tree, B = -DSW_WAVES_R7=5 -DSW_WAVES_R16=3" > $O
run() {
  timeout 400 python bench.py --steps 5 --warmup 2 --samples 200 --no-cpu-baseline --e2e-samples 0 --no-sweep --legs config3:100:500,config3:250:500 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', ' | '.join('{} value {:.0f} ms/step {:.3f} frac {:.4f}'.format(l['leg'], l['value'], l['ms_per_step'], l['frac']) for l in d['legs']))" >> $O
}
run A; run A
touch tredparse_amd/csrc/sw_ladder.hip; make -C tredparse_amd/csrc -s -j6 EXTRA="-DSW_WAVES_R7=5 -DSW_WAVES_R16=3" ../libtredgpu.so > /dev/null 2>&1
run B; run B
cat $O
