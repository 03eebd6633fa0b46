#!/bin/bash
# Stream priorities (VGPU_STREAM_PRIO="<main>,<aux>") on the round-4 kernels, one session: tools/gpu_r4_prio.sh <outdir>
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r4prio}; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "# main,aux proofs/s ms/step (three in flight) | lone ms" > "$OUT/prio.txt"
for rep in 1 2; do
for cfg in "0,0" "0,-1" "-1,0" "-1,-1" "0,1" "1,0"; do
  a=$(VGPU_STREAM_PRIO=$cfg python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))")
  b=$(VGPU_STREAM_PRIO=$cfg python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 12 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))")
  echo "$cfg $a | $b" >> "$OUT/prio.txt"
done
done
cat "$OUT/prio.txt"
