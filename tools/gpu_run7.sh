cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05_run7.txt; : > $O
timeout 900 python -m pytest tests/test_pairwalk_gpu.py tests/test_inflate_gpu.py tests/test_edge_gpu.py -x -q > gpurun_out/r05_gputest_walk3.log 2>&1; tail -8 gpurun_out/r05_gputest_walk3.log >> $O
python tools/conc_probe.py make /tmp/cp_bams >> $O 2>&1
cd /tmp && export TMPDIR=/tmp
for m in 16 48; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp$m -- python3 $GRAFT_REPO_ROOT/tools/walk_prof.py $m $GRAFT_REPO_ROOT/tredparse_amd/libtredgpu.so /tmp/cp_bams > /tmp/wp$m.json 2> /tmp/wp$m.err
  find /tmp/wp$m -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r05d_walk${m}_kernel_stats.csv \;
  echo "## walk_prof $m" >> $GRAFT_REPO_ROOT/$O; cat /tmp/wp$m.json >> $GRAFT_REPO_ROOT/$O; python3 - >> $GRAFT_REPO_ROOT/$O <<P
import csv
for r in csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/r05d_walk${m}_kernel_stats.csv')):
    print(r['Name'][22:60], r['Calls'], 'avg_ms', round(float(r['AverageNs'])/1e6,3), 'min', round(float(r['MinNs'])/1e6,3), 'max', round(float(r['MaxNs'])/1e6,3))
P
done
cd $GRAFT_REPO_ROOT
timeout 600 python tools/fuzz_walk.py 40 3 > gpurun_out/r05_fuzz_walk_b.json 2> gpurun_out/fuzz_walk_err.txt; tail -c 700 gpurun_out/r05_fuzz_walk_b.json >> $O; tail -3 gpurun_out/fuzz_walk_err.txt >> $O
run() { echo "## e2e $*" >> $O; timeout 300 python bench.py --e2e-only --e2e-seconds 6 --e2e-samples 1024 "$@" 2>gpurun_out/e2e_err.txt | python -c "
import sys, json
for line in sys.stdin:
    line=line.rstrip()
    if line.startswith('{') and 'role' in line:
        d=json.loads(line); print('  ', d['role'][:9], 'drv',d['drivers'],'thr',d['host_threads_per_driver'],'batch',d['samples_per_gpu_batch'],'value',d.get('value'),'first',d.get('first_pass_value'),'whole',d.get('whole_run_value'),'startup',d.get('startup_s'),'warm',d.get('warmup_s'), d.get('error',''))
    elif line.startswith('    {'):
        d=json.loads(line); print('      drv', {k:round(v,2) for k,v in d.items() if k in('seconds','scan_wait','gpu','write','inflate','inflate_gpu','walk_declined','walk_call','walk_fetch')})
    else: print(line[:200])
" >> $O; tail -2 gpurun_out/e2e_err.txt | cut -c1-300 >> $O; }
run
run --e2e-inflate-batch 32
run --e2e-drivers 4 --e2e-threads 4
run --e2e-drivers 2 --e2e-threads 8 --e2e-inflate-batch 32
cat $O | cut -c1-330
