#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
timeout 900 python -m pytest tests/test_pairwalk_gpu.py tests/test_inflate_gpu.py tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -5
python tools/fuzz_walk.py 40 777 2>&1 | tail -1 | cut -c1-420
