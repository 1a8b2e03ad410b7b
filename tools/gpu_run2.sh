cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_b.log 2>&1; tail -3 gpurun_out/r05_gputest_b.log
O=gpurun_out/r05_e2e_grid.txt; : > $O
run() { echo "## $*" >> $O; timeout 600 python bench.py --e2e-only --e2e-seconds 8 --e2e-samples 1024 "$@" >> $O 2>gpurun_out/e2e_err.txt || tail -5 gpurun_out/e2e_err.txt >> $O; }
run --e2e-drivers 6 --e2e-threads 3 --e2e-inflate-batch 16
run --e2e-drivers 6 --e2e-threads 3 --e2e-inflate-batch 16 --e2e-python-writer
run --e2e-drivers 2 --e2e-threads 8 --e2e-inflate-batch 32
run --e2e-drivers 2 --e2e-threads 8 --e2e-inflate-batch 48
run --e2e-drivers 3 --e2e-threads 5 --e2e-inflate-batch 32
run --e2e-drivers 3 --e2e-threads 5 --e2e-inflate-batch 16
run --e2e-drivers 4 --e2e-threads 4 --e2e-inflate-batch 24
run --e2e-drivers 1 --e2e-threads 14 --e2e-inflate-batch 48
cat $O | cut -c1-600
