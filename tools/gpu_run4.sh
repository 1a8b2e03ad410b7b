cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05_run4.txt; : > $O
timeout 900 python -m pytest tests/test_pairwalk_gpu.py tests/test_inflate_gpu.py -x -q > gpurun_out/r05_gputest_walk.log 2>&1; tail -15 gpurun_out/r05_gputest_walk.log >> $O
python tools/conc_probe.py make /tmp/cp_bams >> $O 2>&1
cd /tmp && export TMPDIR=/tmp
for m in 16 48; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp$m -- python3 $GRAFT_REPO_ROOT/tools/walk_prof.py $m $GRAFT_REPO_ROOT/tredparse_amd/libtredgpu.so /tmp/cp_bams > /tmp/wp$m.json 2> /tmp/wp$m.err
  find /tmp/wp$m -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r05b_walk${m}_kernel_stats.csv \;
  echo "## walk_prof $m" >> $GRAFT_REPO_ROOT/$O; python3 - >> $GRAFT_REPO_ROOT/$O <<P
import csv
for r in csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/r05b_walk${m}_kernel_stats.csv')):
    print(r['Name'][22:60], r['Calls'], 'avg_ms', round(float(r['AverageNs'])/1e6,3), 'min', round(float(r['MinNs'])/1e6,3), 'max', round(float(r['MaxNs'])/1e6,3))
P
  tail -2 /tmp/wp$m.err >> $GRAFT_REPO_ROOT/$O
done
cd $GRAFT_REPO_ROOT
for cfg in "16 1 1" "16 1 3" "16 1 6" "32 1 3"; do set -- $cfg
  echo "## conc $cfg" >> $O
  for p in $(seq 1 $3); do timeout 120 python tools/conc_probe.py run /tmp/cp_bams $1 $2 4 > /tmp/conc_$p.json 2>/dev/null & done; wait
  cat /tmp/conc_*.json | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin if l.strip()]
print(len(rows), 'procs', round(sum(r['samples_per_s'] for r in rows),1), 'samples/s', round(sum(r['ms_per_call'] for r in rows)/len(rows),2), 'ms/call')" >> $O
  rm -f /tmp/conc_*.json
done
timeout 600 python tools/fuzz_walk.py 40 > gpurun_out/r05_fuzz_walk_a.json 2> gpurun_out/fuzz_walk_err.txt; tail -c 600 gpurun_out/r05_fuzz_walk_a.json >> $O; tail -3 gpurun_out/fuzz_walk_err.txt >> $O
echo "## e2e" >> $O
timeout 600 python bench.py --e2e-only --e2e-seconds 8 --e2e-samples 1024 >> $O 2>gpurun_out/e2e_err.txt || tail -5 gpurun_out/e2e_err.txt >> $O
timeout 600 python bench.py --e2e-only --e2e-seconds 8 --e2e-samples 1024 --e2e-inflate-batch 32 >> $O 2>gpurun_out/e2e_err.txt || tail -5 gpurun_out/e2e_err.txt >> $O
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_d.log 2>&1; tail -3 gpurun_out/r05_gputest_d.log >> $O
cat $O | cut -c1-500
