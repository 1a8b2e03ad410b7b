# tools/gpu_long_fuzz.sh -- the long SW campaigns on the round's final library (>= 2 M reads against the compiled
# ssw.c, >= 8 M reads production launch vs unpruned dump), and the long selection campaign
cd $GRAFT_REPO_ROOT
R=${1:-r06}
mkdir -p gpurun_out
timeout 1500 python tools/fuzz_parity.py 480 20291101 > gpurun_out/${R}_fuzz_parity_long.json 2> gpurun_out/fp_err.txt; cut -c1-400 gpurun_out/${R}_fuzz_parity_long.json
timeout 1500 python tools/fuzz_selfcheck.py 400 20291102 > gpurun_out/${R}_fuzz_selfcheck_long.json 2> gpurun_out/fs_err.txt; cut -c1-300 gpurun_out/${R}_fuzz_selfcheck_long.json
TREDGPU_FUZZ_READLENS=100,250 timeout 600 python tools/fuzz_parity.py 120 20291103 > gpurun_out/${R}_fuzz_parity_long_100_250.json 2>> gpurun_out/fp_err.txt; cut -c1-400 gpurun_out/${R}_fuzz_parity_long_100_250.json
TREDGPU_FUZZ_READLENS=100,250 timeout 600 python tools/fuzz_selfcheck.py 100 20291104 > gpurun_out/${R}_fuzz_selfcheck_long_100_250.json 2>> gpurun_out/fs_err.txt; cut -c1-300 gpurun_out/${R}_fuzz_selfcheck_long_100_250.json
timeout 900 python tools/fuzz_select.py 600 20261006 > gpurun_out/${R}_fuzz_select_long.json 2> gpurun_out/fsel_err.txt; cut -c1-500 gpurun_out/${R}_fuzz_select_long.json
