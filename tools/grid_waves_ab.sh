#!/bin/bash
# tools/grid_waves_ab.sh -- GPU box: the four grid kernels compiled for other register budgets (launch bounds), against the tree's build:
# kernels_ms_per_step of the headline (150 bp) and of the 100 bp leg, where the grid is half the step.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_grid_waves_ab.txt
echo "# bench.py --steps 5 --warmup 2 --legs config3:100:500; per build: grid kernels' ms per step at 150 bp (30 000 units) | at 100 bp (15 000 units)" > $O
run() {
  timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --e2e-samples 0 --no-sweep --legs config3:100:500 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels_ms_per_step']
det = json.load(open('bench_detail.json'))
l = [x for x in det['legs'] if x.get('leg') == 'config3:100:500'][0]['kernels_ms_per_step']
f = lambda k: 'grid {:.3f} = kde {:.3f} + prepare {:.3f} + pairs {:.3f} + reduce {:.3f}'.format(k['grid'], k['grid_kde'], k['grid_prepare'], k['grid_pairs'], k['grid_reduce'])
print('$1', f(k), '|', f(l))" >> $O
}
build() { touch tredparse_amd/csrc/grid.hip; make -C tredparse_amd/csrc -s -j6 EXTRA="$1" ../libtredgpu.so > /dev/null 2>&1; }
run "tree"
build "-DGRID_PAIRS_WAVES=4"; run "PAIRS=4"
build "-DGRID_KDE_WAVES=4"; run "KDE=4"
build "-DGRID_PREP_WAVES=5"; run "PREP=5"
build "-DGRID_REDUCE_WAVES=5"; run "REDUCE=5"
build "-DGRID_PREP_WAVES=3 -DGRID_REDUCE_WAVES=3 -DGRID_PAIRS_WAVES=2 -DGRID_KDE_WAVES=2"; run "PREP=3,REDUCE=3,PAIRS=2,KDE=2"
cat $O
