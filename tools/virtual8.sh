#!/bin/bash
# tools/virtual8.sh [TAG] -- GPU box with ONE GPU: the 8-GPU commands rehearsed on eight counted devices that are
# one physical one (TRED_VIRTUAL_GPUS=8: shard.virtual_gpus), every rank and driver process real:
#   gpurun_out/TAG_virtual8_torchrun.json  the command the driver launches for N = 8 (torch.distributed.run, one rank per "GPU"; gloo
#                                          instead of RCCL, because the ranks share a device): the kernel-path line of rank 0
#   gpurun_out/TAG_virtual8.json           python bench.py --gpus 8 (this script's own launcher: the sweep n = 1, 2, 4, 8, and for
#                                          every n the host-only leg and three repeats of the planned leg over n x 4 096 files)
cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
mkdir -p gpurun_out
record() {   # name, seconds, exit code, stdout file, description
python - "$@" <<'P'
import json, sys
name, secs, rc, path, what = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
line = lines[-1] if lines else ""
out = {"what": what, "exit_code": rc, "seconds_whole_command": round(secs), "line_bytes": len(line), "json_lines_on_stdout": len(lines),
       "line": json.loads(line) if line else None}
json.dump(out, open(name, "w"), indent=1)
print(name, {k: out[k] for k in ("exit_code", "seconds_whole_command", "line_bytes", "json_lines_on_stdout")})
if out["line"]:
    l = out["line"]
    print("  value", l.get("value"), "n_gpus", l.get("n_gpus"), "oversubscribed", l.get("oversubscribed"), "sweep", json.dumps(l.get("scaling_sweep"))[:700])
P
}
S0=$SECONDS
TRED_VIRTUAL_GPUS=8 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 20 --warmup 5 > gpurun_out/${TAG}_virtual8_torchrun.out 2> gpurun_out/${TAG}_virtual8_torchrun.err
RC=$?
record gpurun_out/${TAG}_virtual8_torchrun.json $((SECONDS - S0)) $RC gpurun_out/${TAG}_virtual8_torchrun.out "TRED_VIRTUAL_GPUS=8 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps 20 --warmup 5 on a 1-GPU box: the driver's N = 8 command, eight real ranks sharing one device (gloo; oversubscribed: no scaling point)"
S0=$SECONDS
TRED_VIRTUAL_GPUS=8 timeout 1500 python bench.py --gpus 8 --steps 20 --warmup 5 > gpurun_out/${TAG}_virtual8_full.out 2> gpurun_out/${TAG}_virtual8_full.err
RC=$?
D=$((SECONDS - S0))
cp bench_detail.json gpurun_out/${TAG}_virtual8_full_detail.json 2>/dev/null
record gpurun_out/${TAG}_virtual8.json $D $RC gpurun_out/${TAG}_virtual8_full.out "TRED_VIRTUAL_GPUS=8 python bench.py --gpus 8 --steps 20 --warmup 5 on a 1-GPU box (16 usable CPUs): eight real ranks, for n = 1, 2, 4, 8 the host-only leg and three repeats of the planned leg over n x 4096 hard-linked files; every n > 1 is oversubscribed (no scaling point)"
