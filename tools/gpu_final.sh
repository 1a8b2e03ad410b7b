# tools/gpu_final.sh -- the round's evidence in one GPU call, all on the tree's build:
#   the GPU test suite, the default bench (as the driver runs it), the CLI's rate, the randomised campaigns (SW, grid, decoder,
#   walks, read selection), the from-BAM leg at 8 CPUs, every rocprofv3 summary (tools/profile_all.sh).  Everything lands under
#   gpurun_out/ and profiles/.  The A/B records of the round (cu_mask, sw_waves, grid_waves, virtual8, grid_fuse_probe, inflate_capacity, h2d_probe) have scripts of their own.
cd $GRAFT_REPO_ROOT
R=${1:-r06}
mkdir -p gpurun_out
O=gpurun_out/${R}_final.txt; : > $O
python -c "from tredparse_amd import _lib; print(_lib.version())" >> $O
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${R}_gputest.log 2>&1; tail -3 gpurun_out/${R}_gputest.log >> $O
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err; cp bench_detail.json gpurun_out/${R}_bench_detail.json; cat gpurun_out/${R}_bench.json >> $O
timeout 600 python tools/cli_rate.py 12288 512 > gpurun_out/${R}_cli_rate.json 2> gpurun_out/cli_rate.err; cat gpurun_out/${R}_cli_rate.json >> $O; tail -2 gpurun_out/cli_rate.err >> $O
(timeout 400 taskset -c 0-7 python bench.py --e2e-only --e2e-seconds 8 --e2e-wgs-samples 0 2>/dev/null | grep "\"plan\"\|identical" | cut -c1-900) > gpurun_out/${R}_e2e_8cpus.txt; cat gpurun_out/${R}_e2e_8cpus.txt >> $O
timeout 400 python tools/fuzz_select.py 150 20261005 > gpurun_out/${R}_fuzz_select.json 2>> $O
timeout 400 python tools/fuzz_walk.py 250 5 > gpurun_out/${R}_fuzz_walk.json 2>> $O
timeout 200 python tools/fuzz_inflate.py 120 20281001 > gpurun_out/${R}_fuzz_inflate.json 2>> $O
timeout 400 python tools/fuzz_parity.py 120 20291001 > gpurun_out/${R}_fuzz_parity.json 2>> $O
timeout 300 python tools/fuzz_selfcheck.py 100 3 > gpurun_out/${R}_fuzz_selfcheck.json 2>> $O
timeout 300 python tools/fuzz_grid.py 20 3 > gpurun_out/${R}_fuzz_grid.json 2>> $O
timeout 300 python tools/fuzz_hist.py 1000 3 > gpurun_out/${R}_fuzz_hist.json 2>> $O
for f in select walk inflate parity selfcheck grid hist; do echo "fuzz_$f: $(head -c 400 gpurun_out/${R}_fuzz_$f.json)" >> $O; done
timeout 2400 bash tools/profile_all.sh $R >> $O 2>&1
ls profiles | grep "^${R}_" | head -80 >> $O
tail -c 6000 $O
