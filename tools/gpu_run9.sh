cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05_run9.txt; : > $O
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_f.log 2>&1; tail -3 gpurun_out/r05_gputest_f.log >> $O
run() { echo "## e2e $*" >> $O; timeout 300 python bench.py --e2e-only --e2e-seconds 6 --e2e-samples 1024 "$@" 2>gpurun_out/e2e_err.txt | python -c "
import sys, json
for line in sys.stdin:
    line=line.rstrip()
    if line.startswith('{') and 'role' in line:
        d=json.loads(line); print('  ', d['role'][:9], 'drv',d['drivers'],'thr',d['host_threads_per_driver'],'batch',d['samples_per_gpu_batch'],'value',d.get('value'),'first',d.get('first_pass_value'),'whole',d.get('whole_run_value'),'startup',d.get('startup_s'),'pinned',d.get('pinned_MB_per_gpu'), d.get('error',''))
    elif line.startswith('    {'):
        d=json.loads(line); print('      drv', {k:round(v,2) for k,v in d.items() if k in('seconds','scan_wait','gpu','pack','write','inflate','inflate_gpu','walk_declined','walk_call','walk_fetch')})
    else: print(line[:200])
" >> $O; tail -2 gpurun_out/e2e_err.txt | cut -c1-300 >> $O; }
run --e2e-inflate-batch 12
run --e2e-inflate-batch 16
run --e2e-inflate-batch 12
run --e2e-inflate-batch 16
cat $O | cut -c1-400
