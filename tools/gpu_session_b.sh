#!/bin/bash
# Round-2 GPU session B: Keccak permutation microbench, the whole GPU test suite (incl. new full-size word-for-word cases), baseline bench.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r02b
mkdir -p "$OUT"
export TMPDIR=/tmp
python tools/microbench.py "$OUT/microbench.txt" > /dev/null 2> "$OUT/microbench.err"
grep -i keccak "$OUT/microbench.txt"
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > "$OUT/pytest.log" 2>&1
tail -4 "$OUT/pytest.log"
python bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python bench.py --no-cpu-baseline --inflight 1 --no-extra-legs > "$OUT/bench_inflight1.json" 2> "$OUT/bench_inflight1.err"
python -c "
import json
for f in ('bench_default','bench_inflight1'):
    try:
        d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('prover_ms_single_proof_in_flight'))
    except Exception as e: print(f, 'ERR', e)
"
