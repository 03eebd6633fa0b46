#!/bin/bash
# Round 5, session 12: GPU_MAX_HW_QUEUES x in-flight corners that round 4's sweep left out, final code, alternating (first run: 4x3 6x3 8x3 8x4 3x3 2x3 into r5s12; this list: the second run, r5s12b).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5s12b; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "# queues in-flight proofs/s ms/step" > "$OUT/sweep.txt"
for cfg in "4 3" "5 3" "7 3" "4 3" "5 3" "7 3" "4 3" "5 3"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --inflight $2 --steps 24 --warmup 6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($1, $2, round(d['value'],2), round(d['ms_per_step'],2))" >> "$OUT/sweep.txt"
done
cat "$OUT/sweep.txt"
