cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05_run5.txt; : > $O
echo "## e2e b16 trace" >> $O
TREDGPU_TRACE=1 timeout 300 python bench.py --e2e-only --e2e-seconds 4 --e2e-samples 512 >> $O 2>gpurun_out/e2e_trace16.txt
grep "tredgpu" gpurun_out/e2e_trace16.txt | awk 'NR%7==0' | head -60 >> $O
echo "## e2e b32 trace" >> $O
TREDGPU_TRACE=1 timeout 300 python bench.py --e2e-only --e2e-seconds 4 --e2e-samples 512 --e2e-inflate-batch 32 >> $O 2>gpurun_out/e2e_trace32.txt
grep "tredgpu" gpurun_out/e2e_trace32.txt | awk 'NR%5==0' | head -60 >> $O
cat $O | cut -c1-400
