#!/bin/bash
# tools/grid_fuse_probe.sh -- GPU box: what the first pass of a fused pairs+reduce kernel would cost, MEASURED on
# the kernel that exists.  grid_pairs_kernel is patched HERE (the tree's file is put back and rebuilt at the end) so that its last
# pass computes every ml and keeps the item's arg-max but never stores the rectangle -- exactly "pass A" of the fused design (pass B
# would compute every pair again and add exp / the marginal sums on top).  The results of such a build are wrong on purpose
# (grid_reduce reads a rectangle nobody wrote); only kernels_ms_per_step is read.  gpurun_out/r06_grid_fuse_probe.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_grid_fuse_probe.txt
SRC=tredparse_amd/csrc/grid.hip
cp $SRC /tmp/grid_tree.hip
run() {
  timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --e2e-samples 0 --no-sweep --legs config3:100:500 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels_ms_per_step']
det = json.load(open('bench_detail.json'))
l = [x for x in det['legs'] if x.get('leg') == 'config3:100:500'][0]['kernels_ms_per_step']
f = lambda k: 'pairs {:.3f} + reduce {:.3f} (kde {:.3f}, prepare {:.3f})'.format(k['grid_pairs'], k['grid_reduce'], k['grid_kde'], k['grid_prepare'])
print('$1', f(k), '|', f(l))" >> $O
}
build() { touch $SRC; make -C tredparse_amd/csrc -s -j6 ../libtredgpu.so > /dev/null 2>&1; }
echo "# bench.py --steps 5 --warmup 2 --legs config3:100:500: grid kernels' ms per step at 150 bp (30 000 units) | at 100 bp (15 000 units)" > $O
run "tree:                         "
python - <<'P'
p = "tredparse_amd/csrc/grid.hip"
s = open(p).read()
old = "                            mlbuf[pos] = ml;\n"
assert s.count(old) == 1
# (a store the compiler cannot prove dead and the data never takes: ml is a log-likelihood, <= 0)
s = s.replace(old, "                            if (ml == 1.2345e300) mlbuf[pos] = ml;\n", 1)
open(p, "w").write(s)
P
build
run "pairs without the ml store:   "
cp /tmp/grid_tree.hip $SRC
build
run "tree again:                   "
cat $O
