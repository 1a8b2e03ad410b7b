cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
nproc; python -c "from tredparse_amd import shard; print('usable', shard.usable_cpus())"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_a.log 2>&1; tail -3 gpurun_out/r05_gputest_a.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err; tail -c 3000 gpurun_out/r05_bench_a.json; grep -v "bench detail" gpurun_out/r05_bench_a.err | tail -20
