cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05_run3.txt; : > $O
python tools/conc_probe.py make /tmp/cp_bams >> $O 2>&1
timeout 600 python tools/conc_probe.py sweep /tmp/cp_bams > gpurun_out/r05_conc_probe.jsonl 2> gpurun_out/conc_err.txt; tail -3 gpurun_out/conc_err.txt >> $O
cat gpurun_out/r05_conc_probe.jsonl >> $O
for lib in tools/_ab/libtredgpu_base.so tools/_ab/libtredgpu_padonly.so tredparse_amd/libtredgpu.so; do
  echo "## grid A/B $lib" >> $O
  TREDGPU_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-sweep --e2e-samples 0 --legs '' --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'library': d['library'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'kernels': d['kernels_ms_per_step']}))" >> $O
done
cd /tmp && export TMPDIR=/tmp
for m in 16 32 48; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp$m -- python3 $GRAFT_REPO_ROOT/tools/walk_prof.py $m $GRAFT_REPO_ROOT/tredparse_amd/libtredgpu.so /tmp/cp_bams > /tmp/wp$m.json 2> /tmp/wp$m.err
  find /tmp/wp$m -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r05_walk${m}_kernel_stats.csv \;
  echo "## walk_prof $m" >> $GRAFT_REPO_ROOT/$O; cut -d, -f1-4,6-8 $GRAFT_REPO_ROOT/gpurun_out/r05_walk${m}_kernel_stats.csv | head -12 >> $GRAFT_REPO_ROOT/$O
done
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_c.log 2>&1; tail -3 gpurun_out/r05_gputest_c.log >> $O
cat $O | cut -c1-400
