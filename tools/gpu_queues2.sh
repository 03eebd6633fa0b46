#!/bin/bash
# hardware-queue count x proofs in flight on the round-3 kernels (24 steps each), one session
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/queues_r03; mkdir -p "$OUT"; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6"
run() { env $1 $B --inflight $2 > "$OUT/$3.json" 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/$3.json').read().strip().splitlines()[-1]); print('$3', round(d['value'],2), round(d['ms_per_step'],2))"; }
run A=1 3 q4_m3
run GPU_MAX_HW_QUEUES=8 3 q8_m3
run GPU_MAX_HW_QUEUES=8 4 q8_m4
run GPU_MAX_HW_QUEUES=6 3 q6_m3
run A=1 3 q4_m3b
