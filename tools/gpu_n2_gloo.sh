#!/bin/bash
# bench.py's N = 2 control flow on a ONE-GPU box (test hooks: both ranks on device 0, gloo instead of RCCL): tools/gpu_n2_gloo.sh
set -u
export TMPDIR=/tmp
VGPU_BENCH_BACKEND=gloo VGPU_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 6 --warmup 2 > gpurun_out/bench_n2_gloo.json 2> gpurun_out/bench_n2_gloo.err
echo rc=$?
tail -c 400 gpurun_out/bench_n2_gloo.err
python - <<'P'
import json
l = [x for x in open("gpurun_out/bench_n2_gloo.json").read().splitlines() if x.startswith("{")]
d = json.loads(l[-1])
print(d["n_gpus"], round(d["value"], 2), round(d["ms_per_step"], 2), d["scaling"], d["config"]["parallelism"][:90], d["shader_clock"])
P
