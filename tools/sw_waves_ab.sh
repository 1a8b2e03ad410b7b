#!/bin/bash
# tools/sw_waves_ab.sh -- GPU box: sw_cont_kernel's instantiations compiled for more waves per SIMD than their registers admit
# without spills, against the tree's build: ms per step and roofline.frac of the headline and the legs.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_sw_waves_ab2.txt
echo "# bench.py --steps 5 --warmup 2 --samples 1000 --legs config3:100:500,config3:250:500 (headline = 150 bp); per build: value, ms/step, frac" > $O
run() {
  timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --e2e-samples 0 --no-sweep --legs config3:100:500,config3:250:500 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', '150bp value {:.0f} ms/step {:.3f} sw {:.3f} frac {:.4f} |'.format(d['value'], d['ms_per_step'], d['kernels_ms_per_step']['sw_ladder'], d['roofline']['frac']), ' | '.join('{} value {:.0f} ms/step {:.3f} frac {:.4f}'.format(l['leg'], l['value'], l['ms_per_step'], l['frac']) for l in d['legs'] if 'value' in l))" >> $O
}
build() { touch tredparse_amd/csrc/sw_ladder.hip; make -C tredparse_amd/csrc -s -j6 EXTRA="$1" ../libtredgpu.so > /dev/null 2>&1; }
run "tree"
build "-DSW_WAVES_R10=5"; run "R10=5"
build "-DSW_WAVES_R10=6"; run "R10=6"
build "-DSW_WAVES_R7=6 -DSW_WAVES_R16=4 -DSW_WAVES_R4=8"; run "R7=6,R16=4,R4=8"
build "-DSW_WAVES_R7=5 -DSW_WAVES_R16=3 -DSW_WAVES_R4=7"; run "R7=5,R16=3,R4=7"
cat $O
