#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-infl}; mkdir -p "$OUT"; export TMPDIR=/tmp
for rep in 1 2; do for m in 3 4 2; do
  python bench.py --no-cpu-baseline --no-extra-legs --inflight $m > "$OUT/m${m}_$rep.json" 2>/dev/null
done; done
python -c "
import json
for f in ('m3_1','m4_1','m2_1','m3_2','m4_2','m2_2'):
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['value'],2), round(d['ms_per_step'],2))
"
