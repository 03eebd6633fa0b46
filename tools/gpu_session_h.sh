#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r02h}; mkdir -p "$OUT"; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fib25_proof or two_provers or async or fine_grained or full_size_c2 or handles" ) > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log"
for i in 1 2; do python bench.py --no-cpu-baseline > "$OUT/bench_default_$i.json" 2>/dev/null; done
python bench.py --no-cpu-baseline --no-kernel-events > "$OUT/bench_noevents.json" 2>/dev/null
python bench.py --no-cpu-baseline --inflight 1 --no-extra-legs > "$OUT/bench_inflight1.json" 2>/dev/null
python -c "
import json
for f in ('bench_default_1','bench_default_2','bench_noevents','bench_inflight1'):
    try:
        d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['value'],2), round(d['ms_per_step'],2), d.get('prover_ms_single_proof_in_flight'))
    except Exception as e: print(f,'ERR',e)
"
