#!/bin/bash
# clocks / power of the GPU while the default bench runs (is the full-load regime power-limited?)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-pwr}; mkdir -p "$OUT"; export TMPDIR=/tmp
python bench.py --no-cpu-baseline --no-extra-legs --steps 700 --warmup 6 > "$OUT/bench.json" 2>/dev/null &
BP=$!
for i in $(seq 1 40); do
  echo "t=$i $(rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E 'sclk|Power \(W\)|GPU use' | sed 's/.*: //' | tr '\n' ' ')"
  sleep 0.6
  kill -0 $BP 2>/dev/null || break
done > "$OUT/smi_load3.txt"
wait $BP
cat "$OUT/smi_load3.txt"
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
