#!/bin/bash
# Run ON the MI355X box (through gpurun): the sharded prover's parity tests, the parity tests that exercise the quotient kernel, a short bench.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r02s}; mkdir -p "$OUT"; export TMPDIR=/tmp
( timeout 420 python -m pytest tests/test_sharded_prove_gpu.py -q --tb=short -p no:cacheprovider ) > "$OUT/sharded.log" 2>&1
tail -60 "$OUT/sharded.log"
( timeout 300 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -k "fib25_proof or ffi_captured or general_log_quotient or check_constraints or fine_grained or sharded_commit or alu" ) > "$OUT/parity_subset.log" 2>&1
tail -8 "$OUT/parity_subset.log"
( timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 12 --warmup 3 ) > "$OUT/bench.json" 2> "$OUT/bench.err"
python - <<P
import json
try:
    d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("bench", round(d["value"],2), "proofs/s", round(d["ms_per_step"],2), "ms/step", {k: round(v,3) for k,v in list(d["kernel_ms_per_step"].items())[:6]})
except Exception as e: print("bench ERR", e)
P
