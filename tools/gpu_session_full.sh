#!/bin/bash
# Round-end rehearsal ON the MI355X box: the whole -m gpu suite, smoke(), the default bench line (with the CPU baseline leg)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r02full}; mkdir -p "$OUT"; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > "$OUT/pytest_gpu.log" 2>&1
tail -6 "$OUT/pytest_gpu.log"
( time python -c "import __graft_entry__ as g; g.smoke()" ) > "$OUT/smoke.log" 2>&1
tail -4 "$OUT/smoke.log"
( time python bench.py ) > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
tail -4 "$OUT/bench_full.err"
python - <<P
import json
try:
    d=json.loads(open("$OUT/bench_full.json").read().strip().splitlines()[-1]); print("bench", round(d["value"],2), "proofs/s", round(d["ms_per_step"],2), "ms/step single", d["prover_ms_single_proof_in_flight"], "cpu", d.get("cpu_baseline",{}).get("seconds_per_proof"), "roofline", d["roofline"]["frac"], d["valu_roofline"]["frac"])
except Exception as e: print("bench ERR", e)
P
