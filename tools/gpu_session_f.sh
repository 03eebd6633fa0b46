#!/bin/bash
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-r02f}
mkdir -p "$OUT"
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > "$OUT/pytest.log" 2>&1
tail -14 "$OUT/pytest.log"
for i in 1 2; do python bench.py --no-cpu-baseline > "$OUT/bench_default_$i.json" 2> "$OUT/bench_default_$i.err"; done
python bench.py --no-cpu-baseline --inflight 1 --no-extra-legs > "$OUT/bench_inflight1.json" 2> "$OUT/bench_inflight1.err"
python -c "
import json
for f in ('bench_default_1','bench_default_2','bench_inflight1'):
    try:
        d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['value'],2), round(d['ms_per_step'],2), d.get('prover_ms_single_proof_in_flight'))
        k=d['kernel_ms_per_step']; print('   ', {a: round(b,2) for a,b in list(k.items())[:12]})
    except Exception as e: print(f, 'ERR', e)
"
