#!/bin/bash
# Round-3 session ON the MI355X box: the whole -m gpu suite (all failures listed, not only the first), smoke(), the bench line without the CPU leg.
#   tools/gpu_session_r3.sh <outdir-name> [extra pytest args]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r03a}; shift || true; mkdir -p "$OUT"; export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 "$@" ) > "$OUT/pytest_gpu.log" 2>&1
grep -E "passed|failed|error" "$OUT/pytest_gpu.log" | tail -5
grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu.log" | head -40
( time python -c "import __graft_entry__ as g; g.smoke()" ) > "$OUT/smoke.log" 2>&1
tail -2 "$OUT/smoke.log"
( time python bench.py --no-cpu-baseline ) > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -3 "$OUT/bench.err"
python - <<P
import json
try:
    d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("bench", round(d["value"],2), "proofs/s", round(d["ms_per_step"],2), "ms/step single", d["prover_ms_single_proof_in_flight"], "roofline", d["roofline"]["bound"], d["roofline"]["frac"], d["roofline"]["hbm"]["frac"])
    print({k: round(v,2) for k,v in list(d["kernel_ms_per_step"].items())[:14]})
except Exception as e: print("bench ERR", e)
P
