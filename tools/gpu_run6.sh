cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05_run6.txt; : > $O
timeout 900 python -m pytest tests/test_pairwalk_gpu.py tests/test_inflate_gpu.py -x -q > gpurun_out/r05_gputest_walk2.log 2>&1; tail -5 gpurun_out/r05_gputest_walk2.log >> $O
python tools/conc_probe.py make /tmp/cp_bams >> $O 2>&1
cd /tmp && export TMPDIR=/tmp
for m in 16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp$m -- python3 $GRAFT_REPO_ROOT/tools/walk_prof.py $m $GRAFT_REPO_ROOT/tredparse_amd/libtredgpu.so /tmp/cp_bams > /tmp/wp$m.json 2> /tmp/wp$m.err
  find /tmp/wp$m -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r05c_walk${m}_kernel_stats.csv \;
  echo "## walk_prof $m" >> $GRAFT_REPO_ROOT/$O; python3 - >> $GRAFT_REPO_ROOT/$O <<P
import csv
for r in csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/r05c_walk${m}_kernel_stats.csv')):
    print(r['Name'][22:60], r['Calls'], 'avg_ms', round(float(r['AverageNs'])/1e6,3), 'min', round(float(r['MinNs'])/1e6,3), 'max', round(float(r['MaxNs'])/1e6,3))
P
done
cd $GRAFT_REPO_ROOT
echo "## e2e b16 gather fetch" >> $O
TREDGPU_TRACE=1 timeout 300 python bench.py --e2e-only --e2e-seconds 6 --e2e-samples 1024 >> $O 2>gpurun_out/e2e_trace16.txt
grep "tredgpu" gpurun_out/e2e_trace16.txt | awk 'NR%40==0' | head -30 >> $O
echo "## e2e b16 dma fetch" >> $O
TREDGPU_FETCH_DMA=1 timeout 300 python bench.py --e2e-only --e2e-seconds 6 --e2e-samples 1024 >> $O 2>gpurun_out/e2e_err.txt
echo "## e2e b32 gather fetch" >> $O
timeout 300 python bench.py --e2e-only --e2e-seconds 6 --e2e-samples 1024 --e2e-inflate-batch 32 >> $O 2>gpurun_out/e2e_err.txt
echo "## e2e b16 4 drivers" >> $O
timeout 300 python bench.py --e2e-only --e2e-seconds 6 --e2e-samples 1024 --e2e-drivers 4 --e2e-threads 4 >> $O 2>gpurun_out/e2e_err.txt
cat $O | cut -c1-420
