#!/bin/bash
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r02e
mkdir -p "$OUT"
export TMPDIR=/tmp
( nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "from oracle import pyoracle as po; print('usable', po.usable_cores())" ) > "$OUT/cpus.txt" 2>&1
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q --durations=10 -k "sharded or rccl or two_rank" ) > "$OUT/pytest_new.log" 2>&1
tail -22 "$OUT/pytest_new.log"
( time python bench.py ) > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
tail -3 "$OUT/bench_full.err"
python -c "
import json
d=json.loads(open('$OUT/bench_full.json').read().strip().splitlines()[0]); print(d['value'], d['ms_per_step'], d.get('prover_ms_single_proof_in_flight')); print(d['cpu_baseline']); print(json.dumps(d['valu_roofline'])); print(json.dumps(d['roofline']))
"
