#!/bin/bash
# the shader clock the chip sustains under the bench loop: one probing wave (build/microbench clock-probe) beside `bench.py` in another process
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4clock; mkdir -p "$OUT"; export TMPDIR=/tmp
{
echo "== idle GPU"; $ROOT/build/microbench clock-probe 4 200
python bench.py --no-cpu-baseline --no-extra-legs --steps 1200 --warmup 6 > "$OUT/bench.json" 2>/dev/null &
BP=$!
sleep 9
echo "== beside bench.py (three proofs in flight)"; $ROOT/build/microbench clock-probe 24 400
wait $BP
python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 600 --warmup 3 > "$OUT/bench1.json" 2>/dev/null &
BP=$!
sleep 9
echo "== beside bench.py --inflight 1"; $ROOT/build/microbench clock-probe 16 400
wait $BP
} > "$OUT/clock.txt" 2>&1
python -c "
import json
for f in ('bench','bench1'):
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, 'ms_per_step', round(d['ms_per_step'],3), 'steps', d['steps'])
" >> "$OUT/clock.txt"
cat "$OUT/clock.txt"
