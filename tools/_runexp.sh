for v in "$@"; do
  lib=$PWD/tredparse_amd/libtredgpu_exp$v.so
  [ "$v" = main ] && lib=$PWD/tredparse_amd/libtredgpu.so
  TREDGPU_LIB=$lib timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['kernels_ms_per_step']['sw_ladder'], d['roofline']['sw_counters']['trunk_cols'], d['roofline']['sw_counters']['waves'])"
done
