#!/bin/bash
# Round 4, session 1 (run ON the MI355X box through gpurun): the multi-process sharded prover — its test file, then the world-4 proof in a loop
# on 16 cores (the driver box's count), then the whole GPU suite with per-test durations.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
nproc > "$OUT/r04_nproc.txt"
timeout 900 python -m pytest tests/test_zz_sharded_multiprocess_gpu.py -m gpu -q --durations=0 > "$OUT/r04_pytest_mp.log" 2>&1
echo "mp tests rc $?" >> "$OUT/r04_pytest_mp.log"
timeout 900 taskset -c 0-15 python tools/mp_loop.py ${1:-30} 4 > "$OUT/r04_mp_loop.log" 2>&1
echo "loop rc $?" >> "$OUT/r04_mp_loop.log"
timeout 1500 python -m pytest tests -m gpu -q -x --durations=25 > "$OUT/r04_pytest_gpu.log" 2>&1
echo "suite rc $?" >> "$OUT/r04_pytest_gpu.log"
tail -n 5 "$OUT/r04_pytest_mp.log" "$OUT/r04_mp_loop.log" "$OUT/r04_pytest_gpu.log"
