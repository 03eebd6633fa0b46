#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r02u}; mkdir -p "$OUT"; export TMPDIR=/tmp
( timeout 420 python -m pytest tests/test_sharded_prove_gpu.py -q --tb=short -p no:cacheprovider -k "headline" ) > "$OUT/sharded_full.log" 2>&1
tail -15 "$OUT/sharded_full.log"
( timeout 400 python bench.py --no-cpu-baseline ) > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -3 "$OUT/bench.err"
python - <<P
import json
try:
    d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("bench", round(d["value"],2), "proofs/s", round(d["ms_per_step"],2), "ms/step single", d["prover_ms_single_proof_in_flight"]); print(json.dumps(d["one_proof_over_w_ranks_on_this_gpu"]))
except Exception as e: print("bench ERR", e)
P
