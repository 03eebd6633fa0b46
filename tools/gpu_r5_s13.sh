#!/bin/bash
# Round 5, session 13: which streams share a hardware queue (VGPU_QUEUE_MAP, runtime.hpp), three proofs in flight unless noted; alternating repetitions.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5s13; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "# name queues in-flight map proofs/s ms/step" > "$OUT/sweep.txt"
run() { # name Q inflight map
  local line
  line=$(GPU_MAX_HW_QUEUES=$2 VGPU_QUEUE_MAP=$4 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --inflight $3 --steps 24 --warmup 6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))")
  echo "$1 $2 $3 $4 $line" >> "$OUT/sweep.txt"
}
for rep in 1 2 3; do
  GPU_MAX_HW_QUEUES=4 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --inflight 3 --steps 24 --warmup 6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default 4 3 -', round(d['value'],2), round(d['ms_per_step'],2))" >> "$OUT/sweep.txt"
  run A 4 3 0,1,2,3,0,1
  run F 4 3 0,1,2,3,2,3
  run G 4 3 0,1,2,3,0,3
  run E 4 3 0,3,1,3,2,3
  run K 4 3 0,1,2,1,0,3
  run L 4 3 0,1,0,2,0,3
  run N 5 3 0,1,2,3,4,1
  run B 6 3 0,1,2,3,4,5
  run P 3 3 0,1,2,1,0,1
  run R4 4 4 0,1,2,3,0,1,2,3
  run S4 4 4 0,3,1,3,2,3,0,3
  run T4 5 4 0,4,1,4,2,4,3,4
done
cat "$OUT/sweep.txt"
