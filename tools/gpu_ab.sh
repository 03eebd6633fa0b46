#!/bin/bash
# A/B of two library builds in ONE session (boxes differ by a few per cent): A = valida_amd/libvgpu.so, B = $1
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${2:-ab}; mkdir -p "$OUT"; export TMPDIR=/tmp
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-extra-legs > "$OUT/a_$i.json" 2>/dev/null
  VGPU_LIB_PATH=$ROOT/$1 python bench.py --no-cpu-baseline --no-extra-legs > "$OUT/b_$i.json" 2>/dev/null
done
python -c "
import json
for f in ('a_1','b_1','a_2','b_2','a_3','b_3'):
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['value'],2), round(d['ms_per_step'],2))
"
