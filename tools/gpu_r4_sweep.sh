#!/bin/bash
# GPU_MAX_HW_QUEUES x proofs in flight on the round-4 kernels (one session): tools/gpu_r4_sweep.sh <outdir>
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r4sweep}; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "# queues in-flight proofs/s ms/step" > "$OUT/sweep.txt"
for cfg in "4 3" "4 4" "4 5" "4 2" "6 4" "5 4" "4 6" "4 3"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 python bench.py --no-cpu-baseline --no-extra-legs --inflight $2 --steps 24 --warmup 6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($1, $2, round(d['value'],2), round(d['ms_per_step'],2))" >> "$OUT/sweep.txt"
done
cat "$OUT/sweep.txt"
