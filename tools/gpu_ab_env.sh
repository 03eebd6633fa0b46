#!/bin/bash
# A/B of an environment switch of the library in ONE session: A = unset, B = "$1" (e.g. VGPU_KECCAK_ASM=1); parity tests run under B first
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${2:-abenv}; mkdir -p "$OUT"; export TMPDIR=/tmp
env "$1" timeout 600 python -m pytest tests -m gpu -x -q -k "${3:-fib25_proof or mixed_height or golden_fixture or full_size_c2 or poseidon_mmcs_commit}" 2>&1 | tail -3
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-extra-legs > "$OUT/a_$i.json" 2>/dev/null
  env "$1" python bench.py --no-cpu-baseline --no-extra-legs > "$OUT/b_$i.json" 2>/dev/null
done
python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 > "$OUT/a_single.json" 2>/dev/null
env "$1" python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 > "$OUT/b_single.json" 2>/dev/null
python -c "
import json
for f in ('a_1','b_1','a_2','b_2','a_3','b_3','a_single','b_single'):
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['value'],2), round(d['ms_per_step'],2), d.get('roofline',{}).get('kernel'), d.get('dominant_kernel_us'))
"
