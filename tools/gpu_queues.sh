#!/bin/bash
# hardware-queue count x proofs in flight, one session; plus the host CPU a multi-rank-style run (blocking waits) uses
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-queues}; mkdir -p "$OUT"; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra-legs"
run() { env $1 $B --inflight $2 > "$OUT/$3.json" 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/$3.json').read().strip().splitlines()[-1]); print('$3', round(d['value'],2), round(d['ms_per_step'],2))"; }
run A=1 3 base_m3
run GPU_MAX_HW_QUEUES=8 3 q8_m3
run GPU_MAX_HW_QUEUES=8 4 q8_m4
run GPU_MAX_HW_QUEUES=8 6 q8_m6
run GPU_MAX_HW_QUEUES=2 3 q2_m3
run A=1 4 base_m4
run A=1 3 base_m3b
echo "--- CPU time of a blocking-wait run (what each rank of a multi-GPU job costs the host)"
( time env VGPU_SPIN_WAIT=0 $B --steps 60 --warmup 6 > "$OUT/block.json" 2>/dev/null ) 2>&1 | tail -3
python -c "
import json; d=json.loads(open('$OUT/block.json').read().strip().splitlines()[-1]); print('blocking', round(d['value'],2), round(d['ms_per_step'],2))"
