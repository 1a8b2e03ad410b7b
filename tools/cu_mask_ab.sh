#!/bin/bash
# tools/cu_mask_ab.sh [seconds = 8] -- GPU box: the planned from-BAM leg with the device's CUs split between the front end's
# streams and the genotyping context's (TREDGPU_CTX_CUS = CUs reserved for the context) against no split.
# Per setting: genotypes/s, and per driver the genotyping calls' wall (seconds in the call / calls) and the decode thread's.
cd $GRAFT_REPO_ROOT
S=${1:-8}
O=gpurun_out/r06_cu_mask_ab.txt
echo "# planned from-BAM leg (3 drivers x 36 samples per call, selection on the device), ${S} s per setting; library $(python -c 'from tredparse_amd import _lib; print(_lib.version())')" > $O
for cus in 0 16 32 64 0; do
  export TREDGPU_CTX_CUS=$cus
  timeout 300 python bench.py --e2e-only --e2e-seconds $S --e2e-repeats 1 --e2e-wgs-samples 0 2>/dev/null | python -c "
import sys, json
plan = None; drivers = []
for l in sys.stdin:
    l = l.strip()
    if not l.startswith('{'): continue
    d = json.loads(l)
    if d.get('role') == 'plan': plan = d; drivers = []
    elif plan is not None and 'gpu' in d and len(drivers) < 2: drivers.append(d)
if plan:
    print('TREDGPU_CTX_CUS=$cus', 'value', round(plan['value']), 'first_pass', round(plan['first_pass_value']),
          ' | '.join('call wall {:.1f} ms x {} calls, decode thread {:.2f} s of {:.2f}'.format(1e3 * d['gpu'] / max(d['gpu_calls'], 1), int(d['gpu_calls']), d['inflate_gpu'], d['seconds']) for d in drivers))
" >> $O
done
unset TREDGPU_CTX_CUS
cat $O
